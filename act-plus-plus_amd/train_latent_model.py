"""Training of the VQ-ACT latent prior with the reference's names and CLI (reference train_latent_model.py: main :26-143,
forward_pass :323-343, train_bc :346-436, CLI :455-470): a frozen VQ-ACT policy encodes every batch's action chunk into one-hot
codes (``policy.vq_encode``), and ``Latent_Model_Transformer`` learns to predict code t from codes < t.  Everything that computes
runs in libactmi: the policy's encoder through the ACT engine, the prior's forward / backward / AdamW through
``actmi.latent_model`` (hand-derived backward on the library's kernels, no autograd).

What is kept from the reference, quirks included: the cross entropy is ``F.cross_entropy(logits [B,T,V], labels [B,T,V])`` --
class axis = dim 1; validation before training in every epoch; ``latent_model_last.ckpt`` / ``latent_model_epoch_<e>_seed_<s>.ckpt``
names (eval_bc loads ``latent_model_last.ckpt``, imitate_episodes.py:252-262); torch.optim.AdamW defaults.  Dropped: the
matplotlib curves (``plot_history``) and the commented-out eval_bc copy."""
import argparse
import os
import pickle
import sys
from copy import deepcopy

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from actmi.constants import SIM_TASK_CONFIGS  # noqa: E402
from actmi.sim_utils import compute_dict_mean, set_seed  # noqa: E402
from imitate_episodes import make_policy  # noqa: E402


def detach_dict(d):
    """reference utils.py:382-386"""
    return {k: v.detach() for k, v in d.items()}


def forward_pass(data, policy, latent_model):
    """reference train_latent_model.py:323-343"""
    image_data, qpos_data, action_data, is_pad = data
    dev = policy.model.device
    qpos_data, action_data, is_pad = (t.to(dev, non_blocking=True) for t in (qpos_data, action_data, is_pad))
    gt_labels = policy.vq_encode(qpos_data, action_data, is_pad)                                   # [B, vq_class, vq_dim] one-hot
    inputs = torch.cat([torch.zeros_like(gt_labels)[:, [0]], gt_labels[:, :-1]], dim=1)           # (host-side plumbing)
    output_logits = latent_model(inputs)
    ce_loss = latent_model.cross_entropy(output_logits, gt_labels)
    return {"loss": ce_loss, "l1_error": ce_loss.l1_error}


def _mean_of(dicts):
    return compute_dict_mean([{k: (v.value if hasattr(v, "value") else v) for k, v in d.items()} for d in dicts])


def train_bc(train_dataloader, val_dataloader, config, ckpt_name):
    """reference train_latent_model.py:346-436"""
    from actmi.latent_model import LatentModelTransformer, latent_model_spec
    from actmi.weights import generate_latent_model_state_dict
    num_epochs, ckpt_dir, seed = config["num_epochs"], config["ckpt_dir"], config["seed"]
    policy_class, policy_config = config["policy_class"], config["policy_config"]
    set_seed(seed)
    vq_dim, vq_class = policy_config["vq_dim"], policy_config["vq_class"]
    policy = make_policy(policy_class, policy_config)
    ckpt_path = os.path.join(ckpt_dir, ckpt_name)
    print(policy.deserialize(torch.load(ckpt_path, weights_only=True)))
    policy.eval()
    latent_model = LatentModelTransformer(vq_dim, vq_dim, vq_class, device=str(policy.model.device))
    # the reference starts from nn.Module's default initialisers; the seeded generator of the package stands in for them
    latent_model.load_state_dict(generate_latent_model_state_dict(latent_model_spec(vq_dim, vq_dim, vq_class), seed))
    latent_model.dropout_seed = seed
    optimizer = latent_model.configure_optimizer(config["lr"])

    train_history, validation_history = [], []
    min_val_loss, best_ckpt_info = np.inf, None
    for epoch in range(num_epochs):
        print(f"\nEpoch {epoch}")
        latent_model.eval()
        epoch_dicts = [forward_pass(data, policy, latent_model) for data in val_dataloader]
        epoch_summary = _mean_of(epoch_dicts)
        validation_history.append(epoch_summary)
        epoch_val_loss = float(epoch_summary["loss"])
        if epoch_val_loss < min_val_loss:
            min_val_loss = epoch_val_loss
            best_ckpt_info = (epoch, min_val_loss, deepcopy(latent_model.state_dict()))
        print(f"Val loss:   {epoch_val_loss:.5f}")
        print(" ".join(f"{k}: {float(v):.3f}" for k, v in epoch_summary.items()))

        latent_model.train()
        optimizer.zero_grad()
        n_batches = 0
        for data in train_dataloader:
            forward_dict = forward_pass(data, policy, latent_model)
            forward_dict["loss"].backward()
            optimizer.step()
            optimizer.zero_grad()
            train_history.append({k: (v.value if hasattr(v, "value") else v).detach().clone() for k, v in forward_dict.items()})
            n_batches += 1
        epoch_summary = compute_dict_mean(train_history[n_batches * epoch: n_batches * (epoch + 1)])
        print(f"Train loss: {float(epoch_summary['loss']):.5f}")
        print(" ".join(f"{k}: {float(v):.3f}" for k, v in epoch_summary.items()))
        if epoch % 100 == 0:
            torch.save(latent_model.state_dict(), os.path.join(ckpt_dir, f"latent_model_epoch_{epoch}_seed_{seed}.ckpt"))

    torch.save(latent_model.state_dict(), os.path.join(ckpt_dir, "latent_model_last.ckpt"))
    best_epoch, min_val_loss, best_state_dict = best_ckpt_info
    torch.save(best_state_dict, os.path.join(ckpt_dir, f"latent_model_epoch_{best_epoch}_seed_{seed}.ckpt"))
    print(f"Training finished:\nSeed {seed}, val loss {min_val_loss:.6f} at epoch {best_epoch}")
    return best_ckpt_info, train_history, validation_history


def main(args):
    """reference train_latent_model.py:26-143"""
    set_seed(1)
    task_config = SIM_TASK_CONFIGS[args["task_name"]]
    camera_names = task_config["camera_names"]
    if args["policy_class"] != "ACT":
        raise NotImplementedError("the latent prior belongs to VQ-ACT")
    policy_config = {"lr": args["lr"], "num_queries": args["chunk_size"], "kl_weight": args["kl_weight"],
                     "hidden_dim": args["hidden_dim"], "dim_feedforward": args["dim_feedforward"], "lr_backbone": 1e-5,
                     "backbone": "resnet18", "enc_layers": 4, "dec_layers": 7, "nheads": 8, "camera_names": camera_names,
                     "vq": True, "vq_class": args["vq_class"], "vq_dim": args["vq_dim"], "action_dim": 16, "state_dim": 14,
                     "max_batch": args.get("max_batch") or args["batch_size"]}
    config = {"num_epochs": args["num_epochs"], "ckpt_dir": args["ckpt_dir"], "episode_len": task_config["episode_len"],
              "state_dim": 14, "lr": args["lr"], "policy_class": args["policy_class"], "policy_config": policy_config,
              "task_name": args["task_name"], "seed": args["seed"], "temporal_agg": args["temporal_agg"],
              "camera_names": camera_names, "real_robot": False}
    dataset_dir = args.get("dataset_dir") or task_config.get("dataset_dir")
    if dataset_dir and os.path.isdir(dataset_dir):
        from actmi.data import load_data
        name_filter = task_config.get("name_filter", lambda n: True)
        train_dl, val_dl, stats, _ = load_data(dataset_dir, name_filter, camera_names, args["batch_size"], args["batch_size"],
                                               args.get("chunk_size") or 100, policy_class="ACT")
        # the reference iterates its DataLoader once per epoch, but utils.py's BatchSampler never ends; here one epoch = as many
        # batches as the transitions on disk give
        n_train = max(1, len(train_dl.dataset) // args["batch_size"])
        n_val = min(20, max(1, len(val_dl.dataset) // args["batch_size"]))
        train_dl, val_dl = _Epoch(train_dl, n_train), _Epoch(val_dl, n_val)
    else:
        from actmi.config import ACTConfig
        from actmi.envs import SyntheticDataset
        cfg = ACTConfig.from_policy_config(policy_config)
        train_dl = SyntheticDataset(cfg, args["batch_size"], 8, seed=args["seed"])
        val_dl = SyntheticDataset(cfg, args["batch_size"], 2, seed=args["seed"] + 100003)
        stats = None
    os.makedirs(config["ckpt_dir"], exist_ok=True)
    if stats is not None:
        with open(os.path.join(config["ckpt_dir"], "dataset_stats.pkl"), "wb") as f:
            pickle.dump(stats, f)
    ckpt_name = "policy_last.ckpt"                                   # train_latent_model.py:145
    best_ckpt_info, _, _ = train_bc(train_dl, val_dl, config, ckpt_name)
    best_epoch, min_val_loss, best_state_dict = best_ckpt_info
    torch.save(best_state_dict, os.path.join(config["ckpt_dir"], "latent_model_best.ckpt"))
    print(f"Best ckpt, val loss {min_val_loss:.6f} @ epoch{best_epoch}")
    return best_ckpt_info


class _Epoch:
    """`n` batches of an endless loader per iteration"""

    def __init__(self, loader, n):
        self.it, self.n = iter(loader), n

    def __iter__(self):
        for _ in range(self.n):
            yield next(self.it)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("--onscreen_render", action="store_true")
    parser.add_argument("--ckpt_dir", action="store", type=str, required=True)
    parser.add_argument("--policy_class", action="store", type=str, required=True)
    parser.add_argument("--task_name", action="store", type=str, required=True)
    parser.add_argument("--batch_size", action="store", type=int, required=True)
    parser.add_argument("--seed", action="store", type=int, required=True)
    parser.add_argument("--num_epochs", action="store", type=int, required=True)
    parser.add_argument("--lr", action="store", type=float, required=True)
    parser.add_argument("--kl_weight", action="store", type=int)
    parser.add_argument("--chunk_size", action="store", type=int)
    parser.add_argument("--hidden_dim", action="store", type=int)
    parser.add_argument("--dim_feedforward", action="store", type=int)
    parser.add_argument("--temporal_agg", action="store_true")
    parser.add_argument("--use_vq", action="store_true")
    parser.add_argument("--vq_class", action="store", type=int)
    parser.add_argument("--vq_dim", action="store", type=int)
    # additions of this build (same meaning as in imitate_episodes.py)
    parser.add_argument("--max_batch", action="store", type=int, default=None)
    parser.add_argument("--dataset_dir", action="store", type=str, default=None)
    main(vars(parser.parse_args()))
