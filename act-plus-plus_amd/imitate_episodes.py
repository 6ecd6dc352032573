"""Train / eval entry points of the ACT path with the reference's names and CLI
(reference imitate_episodes.py: main :37, make_policy :182, make_optimizer :194, get_image :206, eval_bc :228,
forward_pass :529, train_bc :535, repeater :624, CLI :633-666), re-designed for MI355X:

* eval rollouts are BATCHED: E episodes step in lock-step, one policy query of batch E per timestep through
  libactmi, one temporal-ensemble kernel for all episodes, envs stepped by host threads (the reference runs
  episodes one after another with batch 1, :319-353);
* episodes SHARD across ranks (one process per GPU): poses are pre-drawn in the reference's RNG order, every rank
  takes a contiguous slice, a single all-gather of (episode_return, highest_reward) ends the run;
* the 10 warm-up queries per rollout and the real-time sleep (:377-382, :467-470) are off by default.
"""
import argparse
import os
import pickle
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy
from itertools import repeat

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from actmi.constants import FPS, SIM_TASK_CONFIGS  # noqa: E402
from actmi.sim_utils import compute_dict_mean, draw_episode_poses, set_seed, shard_range  # noqa: E402
from actmi import dist_utils  # noqa: E402


def make_policy(policy_class, policy_config, device=None):
    """reference imitate_episodes.py:182-191 (ACT only on this path).  ``device``: the GPU this process owns
    (cuda:LOCAL_RANK under a one-process-per-GPU launcher); None = the process's current device."""
    if policy_class == "ACT":
        from policy import ACTPolicy
        return ACTPolicy(policy_config, device=device)
    if policy_class == "Diffusion":
        from policy import DiffusionPolicy
        return DiffusionPolicy(policy_config, device=device)
    raise NotImplementedError(f"policy_class {policy_class} is outside the accelerated path (SURVEY §2)")


def make_optimizer(policy_class, policy):
    """reference imitate_episodes.py:194-203."""
    if policy_class == "ACT":
        return policy.configure_optimizers()
    raise NotImplementedError


def center_crop_resize(image, ratio=0.95):
    """The Diffusion policy's eval-time image transform (reference imitate_episodes.py:214-224): the centre `ratio` crop of
    f32 [..., H, W] images, resized back to (H, W).  torchvision's ``transforms.Resize(size, antialias=True)`` on a tensor is
    ``F.interpolate(mode='bilinear', align_corners=False, antialias=True)``, which is what runs here (torchvision itself is
    not importable offline)."""
    H, W = image.shape[-2:]
    crop = image[..., int(H * (1 - ratio) / 2): int(H * (1 + ratio) / 2), int(W * (1 - ratio) / 2): int(W * (1 + ratio) / 2)]
    lead = crop.shape[:-3]
    out = torch.nn.functional.interpolate(crop.reshape(-1, *crop.shape[-3:]), size=(H, W), mode="bilinear", align_corners=False,
                                          antialias=True)
    return out.reshape(*lead, *out.shape[-3:])


def get_image(ts, camera_names, rand_crop_resize=False):
    """reference imitate_episodes.py:206-225: one timestep -> f32 [1,C,3,H,W] in [0,1] on the GPU."""
    imgs = np.stack([np.moveaxis(ts.observation["images"][c], -1, 0) for c in camera_names], axis=0)
    curr_image = torch.from_numpy(imgs / 255.0).float().cuda().unsqueeze(0)
    if rand_crop_resize:
        curr_image = center_crop_resize(curr_image)
    return curr_image


def make_post_process(policy_class, stats):
    """reference imitate_episodes.py:290-293: Diffusion policies emit actions in [-1, 1] of the dataset's min / max range,
    the others z-scored actions."""
    if policy_class == "Diffusion":
        return lambda a: ((a + 1) / 2) * (stats["action_max"] - stats["action_min"]) + stats["action_min"]
    return lambda a: a * stats["action_std"] + stats["action_mean"]


def get_image_batch_u8(ts_list, camera_names, pinned=None):
    """Fast path of the same contract: E timesteps -> u8 [E,C,H,W,3] (4x fewer H2D bytes; /255 and the ImageNet
    normalisation are fused into the conv1 loader with the reference's float arithmetic)."""
    E, C = len(ts_list), len(camera_names)
    h, w, _ = ts_list[0].observation["images"][camera_names[0]].shape
    if pinned is None or tuple(pinned.shape) != (E, C, h, w, 3):
        pinned = torch.empty((E, C, h, w, 3), dtype=torch.uint8)
        if torch.cuda.is_available():
            pinned = pinned.pin_memory()
    buf = pinned.numpy()
    for e, ts in enumerate(ts_list):
        for c, name in enumerate(camera_names):
            buf[e, c] = ts.observation["images"][name]
    return pinned


def _default_stats(state_dim, action_dim=16):
    return {"qpos_mean": np.zeros(state_dim), "qpos_std": np.ones(state_dim),
            "action_mean": np.zeros(action_dim), "action_std": np.ones(action_dim),
            "action_min": -np.ones(action_dim), "action_max": np.ones(action_dim)}


def eval_bc(config, ckpt_name, save_episode=True, num_rollouts=50, policy=None, ensemble_factory=None,
            env_factory=None, stats=None, max_parallel=None, warmup_queries=0, realtime=False, verbose=True,
            trace=None, vq_sampler=None, pipeline_groups=None):
    """reference imitate_episodes.py:228-526, batched + sharded.  Returns (success_rate, avg_return).

    VQ-ACT: the reference samples the latent code from its prior model each query
    (``latent_model.generate(1, temperature=1)``, imitate_episodes.py:252-262,393-394): ``actmi.latent_model`` mirrors it
    on the library's kernels; ``vq_sampler(n) -> [n, vq_class, vq_dim]`` overrides the source of the codes."""
    set_seed(1000)
    ckpt_dir = config["ckpt_dir"]
    state_dim = config["state_dim"]
    policy_class = config["policy_class"]
    policy_config = config["policy_config"]
    camera_names = config["camera_names"]
    max_timesteps = int(config["episode_len"])
    task_name = config["task_name"]
    temporal_agg = config["temporal_agg"]
    rank, world, local_rank = dist_utils.init_from_env()
    on_gpu = torch.cuda.is_available()
    # one process per GPU: everything of this rank (engine arena, streams, frames, ensemble ring) lives on cuda:LOCAL_RANK
    dev = torch.device("cuda", local_rank if world > 1 else torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    if policy is None:
        # inference-only handle (no grad buffers), built on this rank's device
        policy = make_policy(policy_class, dict(policy_config, training=False), device=str(dev))
        ckpt_path = os.path.join(ckpt_dir, ckpt_name)
        if os.path.isfile(ckpt_path):
            loading_status = policy.deserialize(torch.load(ckpt_path, weights_only=True))
            if verbose:
                print(loading_status)
        elif verbose:
            print(f"no checkpoint at {ckpt_path}: evaluating the random-init policy")
        policy.cuda()
        policy.eval()
    if stats is None:
        stats_path = os.path.join(ckpt_dir, "dataset_stats.pkl")
        if os.path.isfile(stats_path):
            with open(stats_path, "rb") as f:
                stats = pickle.load(f)               # written by this program's own main()
        else:
            stats = _default_stats(state_dim)
    pre_process = lambda s_qpos: (s_qpos - stats["qpos_mean"]) / stats["qpos_std"]          # noqa: E731
    post_process = make_post_process(policy_class, stats)
    is_diffusion = policy_class == "Diffusion"

    use_vq = bool(policy_config.get("vq", False))
    if use_vq and vq_sampler is None:
        # reference imitate_episodes.py:252-262: the latent prior, weights from latent_model_last.ckpt next to the policy
        from actmi.latent_model import LatentModelTransformer, latent_model_spec
        from actmi.weights import generate_latent_model_state_dict
        vq_dim, vq_class = policy_config["vq_dim"], policy_config["vq_class"]
        latent_model = LatentModelTransformer(vq_dim, vq_dim, vq_class, device=str(dev))
        lm_path = os.path.join(ckpt_dir, "latent_model_last.ckpt")
        if os.path.isfile(lm_path):
            latent_model.load_state_dict(torch.load(lm_path, weights_only=True))
        else:
            if verbose:
                print(f"no latent model at {lm_path}: sampling codes from a random-init prior")
            latent_model.load_state_dict(generate_latent_model_state_dict(latent_model_spec(vq_dim, vq_dim, vq_class), 0))
        _draws = [0]

        def vq_sampler(n):                       # imitate_episodes.py:393: latent_model.generate(1, temperature=1, x=None)
            _draws[0] += 1
            return latent_model.generate(n, temperature=1, x=None, seed=1000 * rank + _draws[0])
    num_queries = policy_config["num_queries"]
    query_frequency = 1 if temporal_agg else num_queries
    action_dim = policy_config.get("action_dim", 16)
    if ensemble_factory is None:
        from actmi.ops import TemporalEnsemble
        ensemble_factory = lambda E: TemporalEnsemble(E, num_queries, action_dim, 0.01, dev)   # noqa: E731
    if env_factory is None:
        from actmi.envs import make_sim_env
        env_factory = lambda pose, idx: make_sim_env(task_name, camera_names, pose, seed=idx,   # noqa: E731
                                                     synthetic=True if config.get("synthetic_env") else None)

    # poses in the reference's draw order, then this rank's contiguous shard
    poses = draw_episode_poses(task_name, num_rollouts, seed=1000)
    lo, hi = shard_range(num_rollouts, rank, world)
    counts = [shard_range(num_rollouts, r, world)[1] - shard_range(num_rollouts, r, world)[0] for r in range(world)]
    if max_parallel is None:
        max_parallel = getattr(getattr(policy, "model", None), "max_batch", None) or (hi - lo) or 1
    pdev = getattr(getattr(policy, "model", None), "device", None)
    if on_gpu and pdev is not None and torch.device(pdev) != dev:
        raise RuntimeError(f"rank {rank}: policy engine is on {pdev} but this rank's tensors are on {dev}")
    # on-screen rendering and per-episode video dumps (imitate_episodes.py:284-287, 342-346, 516-517) need matplotlib / cv2
    # windows and files per episode: outside the accelerated path -- refused or announced, never silently dropped
    if config.get("onscreen_render"):
        raise NotImplementedError("onscreen_render is not available on the batched MI355X eval path (episodes step in "
                                  "lock-step on the GPU; there is no per-episode matplotlib window)")
    if save_episode and verbose and rank == 0:
        print("note: save_episode=True writes the result_*.txt summary only; per-episode videos are not produced here")

    local_results = []
    env_max_reward = None
    DT = 1 / FPS
    t_policy, n_queries = 0.0, 0
    pool = ThreadPoolExecutor(max_workers=min(16, max(1, max_parallel)))
    engine = getattr(policy, "model", None)
    flags_dev = engine.flags_tensor() if (on_gpu and hasattr(engine, "flags_tensor")) else None

    def _copy_frames(buf, e, ts):
        for c, name in enumerate(camera_names):
            buf[e, c] = ts.observation["images"][name]
        return ts

    for w0 in range(lo, hi, max_parallel):
        ids = list(range(w0, min(hi, w0 + max_parallel)))
        E = len(ids)
        envs = [env_factory(poses[i], i) for i in ids]
        env_max_reward = envs[0].task.max_reward
        # Two groups of episodes in ping-pong (GPU only): while the GPU runs the policy query of one group, the host steps the
        # other group's simulators, gathers their frames into pinned memory and queues that group's next query -- the H2D copy,
        # the forward and the D2H of the actions of a group are queued on the stream back to back and waited for with an event,
        # never with a device-wide synchronisation.  Episodes are independent, so each one sees exactly the steps it would see
        # alone (tests/test_gpu_rollout.py).  pipeline_groups=1 (or a CPU stand-in policy) keeps the single lock-step group.
        G = 2 if (on_gpu and E >= 2 and pipeline_groups != 1 and not realtime) else 1
        cuts = [0, E] if G == 1 else [0, (E + 1) // 2, E]
        h, w, _ = None, None, None
        groups = []
        for gi in range(G):
            sl = slice(cuts[gi], cuts[gi + 1])
            g_envs = envs[sl]
            g = {"ids": ids[sl], "envs": g_envs, "E": len(g_envs), "ens": None, "rewards": [[] for _ in g_envs], "pinned": None,
                 "all_actions": None, "ev": torch.cuda.Event() if on_gpu else None, "raw_host": None, "flag_host": None, "tq": 0.0}
            g["ens"] = ensemble_factory(g["E"]) if (temporal_agg and not is_diffusion) else None
            g["ts"] = list(pool.map(lambda e: e.reset(), g_envs))
            groups.append(g)

        def enqueue(g, t):
            """queue step t of a group: (H2D of its frames +) policy query + ensemble + D2H of the actions, then an event"""
            nonlocal n_queries
            ts_list = g["ts"]
            qpos_numpy = np.stack([np.array(ts.observation["qpos"]) for ts in ts_list])
            qpos = torch.from_numpy(pre_process(qpos_numpy)).float().to(dev, non_blocking=True)
            g["tq"] = time.time()
            if t % query_frequency == 0:
                if g["pinned"] is None or t == 0:
                    g["pinned"] = get_image_batch_u8(ts_list, camera_names, g["pinned"])
                curr_image = g["pinned"].to(dev, non_blocking=True)
                if is_diffusion:
                    # get_image(..., rand_crop_resize=True) of the reference (:214-224, :374): f32 in [0, 1], centre 0.95 crop,
                    # resized back -- on the device, for all episodes of the group at once
                    curr_image = center_crop_resize(curr_image.permute(0, 1, 4, 2, 3).float().div(255.0))
                if t == 0:
                    for _ in range(warmup_queries):
                        policy(qpos, curr_image, vq_sample=vq_sampler(qpos.shape[0])) if use_vq else policy(qpos, curr_image)
                g["all_actions"] = (policy(qpos, curr_image, vq_sample=vq_sampler(qpos.shape[0])) if use_vq
                                    else policy(qpos, curr_image))            # [E_g,Q,A]
                n_queries += g["E"]
            if g["ens"] is not None:
                raw = g["ens"].step(g["all_actions"])                        # [E_g,A] float64, like the reference
            else:
                raw = g["all_actions"][:, t % query_frequency]
            if on_gpu:
                if g["raw_host"] is None or g["raw_host"].dtype != raw.dtype:
                    g["raw_host"] = torch.empty(tuple(raw.shape), dtype=raw.dtype).pin_memory()
                    g["flag_host"] = torch.zeros(1, dtype=torch.int32).pin_memory()
                g["raw_host"].copy_(raw, non_blocking=True)
                if flags_dev is not None:
                    g["flag_host"].copy_(flags_dev, non_blocking=True)       # the range guard's word rides along: no extra sync
                g["ev"].record()
            else:
                g["raw_host"] = raw

        def finish(g, t):
            """wait for step t of a group, step its simulators (frames of the next query straight into pinned memory)"""
            nonlocal t_policy
            if on_gpu:
                g["ev"].synchronize()                                        # this group's actions are on the host
                if flags_dev is not None and int(g["flag_host"][0]) != 0:
                    engine.check_flags()                                     # raises with the reason (and clears the word)
            raw_action = g["raw_host"].numpy().copy() if on_gpu else g["raw_host"].cpu().numpy()
            if t % query_frequency == 0:
                t_policy += time.time() - g["tq"]
            if trace is not None:
                trace.append((g["ids"], t, raw_action.copy()))
            action = post_process(raw_action)
            target_qpos = action[:, :-2]
            need_frames = (t + 1) % query_frequency == 0 and g["pinned"] is not None and t + 1 < max_timesteps
            buf = g["pinned"].numpy() if need_frames else None

            def step_one(p):
                e, env, a = p
                ts = env.step(a)
                return _copy_frames(buf, e, ts) if buf is not None else ts
            g["ts"] = list(pool.map(step_one, zip(range(g["E"]), g["envs"], target_qpos)))
            for e in range(g["E"]):
                g["rewards"][e].append(g["ts"][e].reward)

        time0 = time.time()
        for g in groups:
            enqueue(g, 0)
        for t in range(max_timesteps):
            time1 = time.time()
            for g in groups:
                finish(g, t)
                if t + 1 < max_timesteps:
                    enqueue(g, t + 1)
            if realtime:
                time.sleep(max(0, DT - (time.time() - time1)))
        if verbose:
            print(f"rank {rank}: episodes {ids[0]}..{ids[-1]} avg fps {max_timesteps / (time.time() - time0):.1f} "
                  f"({E} parallel episodes{', two ping-pong groups' if G == 2 else ''})")
        for g in groups:
            for e, i in enumerate(g["ids"]):
                r = np.array(g["rewards"][e])
                episode_return = float(np.sum(r[r != None]))                  # noqa: E711  (reference :499)
                local_results.append([episode_return, float(np.max(r))])
    pool.shutdown()

    local = torch.tensor(local_results, dtype=torch.float32).reshape(-1, 2)
    allr = dist_utils.all_gather_rows(local, counts).numpy()                 # the ONE collective of the path
    episode_returns, highest_rewards = allr[:, 0], allr[:, 1]
    success_rate = float(np.mean(highest_rewards == env_max_reward))
    avg_return = float(np.mean(episode_returns))
    summary_str = f"\nSuccess rate: {success_rate}\nAverage return: {avg_return}\n\n"
    for r in range(env_max_reward + 1):
        more_or_equal_r = int((highest_rewards >= r).sum())
        summary_str += f"Reward >= {r}: {more_or_equal_r}/{num_rollouts} = {more_or_equal_r / num_rollouts * 100}%\n"
    if rank == 0:
        if verbose:
            print(summary_str)
            if n_queries:
                # issue-to-host latency summed over the queries: with two groups in flight these intervals overlap each other and the
                # simulators, so this is a latency figure, not wall time ("avg fps" above is the throughput)
                print(f"policy queries: {n_queries} on rank 0, mean issue-to-host latency {1e3 * t_policy / n_queries:.2f} ms per query group")
        os.makedirs(ckpt_dir, exist_ok=True)
        result_file_name = "result_" + ckpt_name.split(".")[0] + ".txt"
        with open(os.path.join(ckpt_dir, result_file_name), "w") as f:
            f.write(summary_str)
            f.write(repr(episode_returns.tolist()))
            f.write("\n\n")
            f.write(repr(highest_rewards.tolist()))
    return success_rate, avg_return


def forward_pass(data, policy):
    """reference imitate_episodes.py:529-532."""
    image_data, qpos_data, action_data, is_pad = data
    dev = getattr(getattr(policy, "model", None), "device", None) or "cuda"
    image_data, qpos_data, action_data, is_pad = (t.to(dev, non_blocking=True) for t in (image_data, qpos_data, action_data, is_pad))
    return policy(qpos_data, image_data, action_data, is_pad)


def train_bc(train_dataloader, val_dataloader, config, log=None):
    """reference imitate_episodes.py:535-622 (wandb replaced by an optional ``log(dict, step)`` callable)."""
    num_steps = config["num_steps"]
    ckpt_dir = config["ckpt_dir"]
    seed = config["seed"]
    policy_class = config["policy_class"]
    policy_config = config["policy_config"]
    eval_every = config.get("eval_every") or 0
    validate_every = config["validate_every"]
    save_every = config["save_every"]
    # data parallel (one process per GPU under torch.distributed.run): every rank builds the same initial policy on its own
    # device, draws ITS OWN batches and dropout masks (seed + rank), averages gradients over RCCL in loss.backward(); only
    # rank 0 writes checkpoints
    rank, world, local_rank = dist_utils.init_from_env()
    set_seed(seed + rank)
    dev = f"cuda:{local_rank}" if world > 1 else None
    policy = make_policy(policy_class, dict(policy_config, seed=policy_config.get("seed", seed) * world + rank), device=dev)
    if config.get("load_pretrain"):
        # reference :548-550 reads a path in its author's home directory; here: config["pretrain_ckpt_path"], else the
        # environment's ACTMI_PRETRAIN_CKPT, else that same path -- and a missing file is an error, not a silent skip
        pre = (config.get("pretrain_ckpt_path") or os.environ.get("ACTMI_PRETRAIN_CKPT") or
               os.path.join("/home/zfu/interbotix_ws/src/act/ckpts/pretrain_all", "policy_step_50000_seed_0.ckpt"))
        if not os.path.isfile(pre):
            raise FileNotFoundError(f"--load_pretrain: no checkpoint at {pre} (set --pretrain_ckpt_path or ACTMI_PRETRAIN_CKPT)")
        loading_status = policy.deserialize(torch.load(pre, weights_only=True))
        print(f"loaded! {loading_status}")
    if config.get("resume_ckpt_path"):
        loading_status = policy.deserialize(torch.load(config["resume_ckpt_path"], weights_only=True))
        print(f'Resume policy from: {config["resume_ckpt_path"]}, Status: {loading_status}')
    policy.cuda()
    optimizer = make_optimizer(policy_class, policy)
    min_val_loss = np.inf
    best_ckpt_info = None
    os.makedirs(ckpt_dir, exist_ok=True)
    from actmi.data import DevicePrefetcher
    train_dataloader = DevicePrefetcher(repeater(train_dataloader))      # next batch's H2D copy overlaps this step
    for step in range(num_steps + 1):
        if step % validate_every == 0:
            policy.eval()
            validation_dicts = []
            for batch_idx, data in enumerate(val_dataloader):
                validation_dicts.append({k: v.detach().float().cpu() for k, v in forward_pass(data, policy).items()})
                if batch_idx > 50:
                    break
            validation_summary = compute_dict_mean(validation_dicts)
            epoch_val_loss = float(validation_summary["loss"])
            if epoch_val_loss < min_val_loss:
                min_val_loss = epoch_val_loss
                best_ckpt_info = (step, min_val_loss, deepcopy(policy.serialize()))
            if log:
                log({f"val_{k}": float(v) for k, v in validation_summary.items()}, step)
            print(f"Val loss:   {epoch_val_loss:.5f}")
        if eval_every and step > 0 and step % eval_every == 0:
            # reference :590-596: first save, then evaluate that checkpoint with 10 rollouts (eval_bc builds its own
            # inference-only handle from the file; under data-parallel training the rollouts shard over the ranks)
            ckpt_name = f"policy_step_{step}_seed_{seed}.ckpt"
            if rank == 0:
                torch.save(policy.serialize(), os.path.join(ckpt_dir, ckpt_name))
            dist_utils.barrier()
            success, _ = eval_bc(config, ckpt_name, save_episode=True, num_rollouts=config.get("eval_rollouts", 10),
                                 verbose=(rank == 0))
            if log:
                log({"success": success}, step)
        policy.train()
        optimizer.zero_grad()
        data = next(train_dataloader)
        forward_dict = forward_pass(data, policy)
        loss = forward_dict["loss"]
        loss.backward()
        optimizer.step()
        if log:
            log({k: float(v) for k, v in forward_dict.items()}, step)
        if step % save_every == 0 or step % validate_every == 0:
            policy.model.check_flags()                   # non-finite loss / weight beyond its split scale: fail loudly
        if step % save_every == 0 and rank == 0:
            torch.save(policy.serialize(), os.path.join(ckpt_dir, f"policy_step_{step}_seed_{seed}.ckpt"))
    best_step, min_val_loss, best_state_dict = best_ckpt_info
    if rank == 0:
        torch.save(policy.serialize(), os.path.join(ckpt_dir, "policy_last.ckpt"))
        torch.save(best_state_dict, os.path.join(ckpt_dir, f"policy_step_{best_step}_seed_{seed}.ckpt"))
        print(f"Training finished:\nSeed {seed}, val loss {min_val_loss:.6f} at step {best_step}")
    dist_utils.barrier()                # checkpoints are on disk before any rank goes on to read them
    return best_ckpt_info


def repeater(data_loader):
    """reference imitate_episodes.py:624-630."""
    epoch = 0
    for loader in repeat(data_loader):
        for data in loader:
            yield data
        print(f"Epoch {epoch} done")
        epoch += 1


def build_config(args):
    """The config dict of reference main() (:37-143)."""
    task_name = args["task_name"]
    task_config = SIM_TASK_CONFIGS[task_name]
    camera_names = task_config["camera_names"]
    policy_class = args["policy_class"]
    if policy_class == "ACT":                        # reference :74-94
        policy_config = {"lr": args["lr"], "num_queries": args["chunk_size"], "kl_weight": args["kl_weight"],
                         "hidden_dim": args["hidden_dim"], "dim_feedforward": args["dim_feedforward"], "lr_backbone": 1e-5,
                         "backbone": "resnet18", "enc_layers": 4, "dec_layers": 7, "nheads": 8,
                         "camera_names": camera_names, "vq": args.get("use_vq", False), "vq_class": args.get("vq_class"),
                         "vq_dim": args.get("vq_dim"), "action_dim": 16, "no_encoder": args.get("no_encoder", False),
                         "state_dim": 14, "max_batch": args.get("max_batch") or args["batch_size"]}
    elif policy_class == "Diffusion":                # reference :95-106
        policy_config = {"lr": args["lr"], "camera_names": camera_names, "action_dim": 16, "observation_horizon": 1,
                         "action_horizon": 8, "prediction_horizon": args["chunk_size"], "num_queries": args["chunk_size"],
                         "num_inference_timesteps": 10, "ema_power": 0.75, "vq": False}
    else:
        raise NotImplementedError(f"policy_class {policy_class} is outside the accelerated path (SURVEY §2)")
    return {"num_steps": args["num_steps"], "eval_every": args["eval_every"], "validate_every": args["validate_every"],
            "save_every": args["save_every"], "ckpt_dir": args["ckpt_dir"], "resume_ckpt_path": args.get("resume_ckpt_path"),
            "episode_len": task_config["episode_len"], "state_dim": 14, "lr": args["lr"],
            "policy_class": args["policy_class"], "onscreen_render": args.get("onscreen_render", False),
            "policy_config": policy_config, "task_name": task_name, "seed": args["seed"],
            "temporal_agg": args["temporal_agg"], "camera_names": camera_names, "real_robot": False,
            "load_pretrain": bool(args.get("load_pretrain", False)), "pretrain_ckpt_path": args.get("pretrain_ckpt_path"),
            "synthetic_env": bool(args.get("synthetic_env", False))}


def main(args):
    set_seed(1)
    config = build_config(args)
    rank, world, _ = dist_utils.init_from_env()          # one process per GPU when a launcher set WORLD_SIZE
    os.makedirs(config["ckpt_dir"], exist_ok=True)
    if rank == 0:
        with open(os.path.join(config["ckpt_dir"], "config.pkl"), "wb") as f:
            pickle.dump(config, f)
    if args["eval"]:
        results = []
        for ckpt_name in ["policy_last.ckpt"]:
            success_rate, avg_return = eval_bc(config, ckpt_name, save_episode=True, num_rollouts=args["num_rollouts"])
            results.append([ckpt_name, success_rate, avg_return])
        for ckpt_name, success_rate, avg_return in results:
            print(f"{ckpt_name}: {success_rate=} {avg_return=}")
        return results
    from actmi.config import ACTConfig
    from actmi.envs import SyntheticDataset
    cfg = ACTConfig.from_policy_config(config["policy_config"])
    task_config = SIM_TASK_CONFIGS[args["task_name"]]
    camera_names, policy_class = config["camera_names"], config["policy_class"]
    dataset_dir = args.get("dataset_dir") or task_config.get("dataset_dir")
    if dataset_dir and os.path.isdir(dataset_dir):
        # reference imitate_episodes.py:141-147: episodes on disk (HDF5 via h5py, or .npz with the same keys), z-scored
        # qpos / actions, u8 images; batches reach the device through pinned staging on a side stream
        from actmi.data import load_data
        name_filter = task_config.get("name_filter", lambda n: True)
        train_dl, val_dl, stats, _ = load_data(dataset_dir, name_filter, camera_names, args["batch_size"], args["batch_size"],
                                               args.get("chunk_size") or 100, args.get("skip_mirrored_data", False),
                                               policy_class=policy_class, stats_dir_l=task_config.get("stats_dir"),
                                               sample_weights=task_config.get("sample_weights"),
                                               train_ratio=task_config.get("train_ratio", 0.99))
    else:
        train_dl = SyntheticDataset(cfg, args["batch_size"], 8, seed=args["seed"] * world + rank)     # disjoint per rank
        val_dl = SyntheticDataset(cfg, args["batch_size"], 2, seed=args["seed"] * world + rank + 100003)
        stats = _default_stats(14)
    if rank == 0:
        with open(os.path.join(config["ckpt_dir"], "dataset_stats.pkl"), "wb") as f:
            pickle.dump(stats, f)
    best_step, min_val_loss, best_state_dict = train_bc(train_dl, val_dl, config)
    if rank == 0:
        torch.save(best_state_dict, os.path.join(config["ckpt_dir"], "policy_best.ckpt"))
        print(f"Best ckpt, val loss {min_val_loss:.6f} @ step{best_step}")
    dist_utils.barrier()


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("--onscreen_render", action="store_true")
    parser.add_argument("--ckpt_dir", action="store", type=str, required=True)
    parser.add_argument("--policy_class", action="store", type=str, required=True)
    parser.add_argument("--task_name", action="store", type=str, required=True)
    parser.add_argument("--batch_size", action="store", type=int, required=True)
    parser.add_argument("--seed", action="store", type=int, required=True)
    parser.add_argument("--num_steps", action="store", type=int, required=True)
    parser.add_argument("--lr", action="store", type=float, required=True)
    parser.add_argument("--load_pretrain", action="store_true", default=False)
    parser.add_argument("--pretrain_ckpt_path", action="store", type=str, default=None,
                        help="checkpoint --load_pretrain reads (the reference hard-codes a path in its author's home directory)")
    parser.add_argument("--eval_every", action="store", type=int, default=500)
    parser.add_argument("--validate_every", action="store", type=int, default=500)
    parser.add_argument("--save_every", action="store", type=int, default=500)
    parser.add_argument("--resume_ckpt_path", action="store", type=str)
    parser.add_argument("--skip_mirrored_data", action="store_true")
    parser.add_argument("--kl_weight", action="store", type=int)
    parser.add_argument("--chunk_size", action="store", type=int)
    parser.add_argument("--hidden_dim", action="store", type=int)
    parser.add_argument("--dim_feedforward", action="store", type=int)
    parser.add_argument("--temporal_agg", action="store_true")
    parser.add_argument("--use_vq", action="store_true")
    parser.add_argument("--vq_class", action="store", type=int)
    parser.add_argument("--vq_dim", action="store", type=int)
    parser.add_argument("--no_encoder", action="store_true")
    # additions (rollouts count hard-coded to 10 in the reference, :156)
    parser.add_argument("--num_rollouts", action="store", type=int, default=50)
    parser.add_argument("--max_batch", action="store", type=int, default=None)
    parser.add_argument("--synthetic_env", action="store_true",
                        help="evaluate on the SyntheticEnv stand-in even when a sim_env module is importable "
                             "(throughput / plumbing runs; its success rates are not task results)")
    parser.add_argument("--dataset_dir", action="store", type=str, default=None,
                        help="episode files (reference HDF5 layout, or .npz with the same keys); default: the task's dataset_dir")
    main(vars(parser.parse_args()))
