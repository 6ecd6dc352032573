#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Authoring-container only: imports the upstream python files from /root/reference (read-only,
never copied, never shipped) with third-party shims for packages that are absent offline:

  * IPython                -> stub exposing ``embed`` (the reference files do ``e = IPython.embed``)
  * torchvision            -> ``__version__``, ``models._utils.IntermediateLayerGetter`` and
                              ``transforms.Normalize`` restated (third-party code, not reference code)
  * robomimic / diffusers  -> empty name holders (only DiffusionPolicy uses them; not on this path)

The torchvision ``resnet18`` network itself is restated here with torchvision attribute names so that
the reference's ``BackboneBase`` / ``Joiner`` / ``FrozenBatchNorm2d`` wrap it and the resulting state_dict
keys equal real checkpoints'.  ``build_backbone`` / ``build`` are NOT called: they request ImageNet
weights over the network (reference backbone.py:121-124).

What is executed from the reference: DETRVAE, Transformer*, build_transformer, build_encoder,
reparametrize, get_sinusoid_encoding_table, FrozenBatchNorm2d, BackboneBase, Joiner,
PositionEmbeddingSine, and ``ACTPolicy.__call__`` + ``kl_divergence`` from policy.py (the policy object
is created with ``__new__`` because ``ACTPolicy.__init__`` parses sys.argv and calls ``.cuda()``).

Usage:  python tools/gen_golden.py [--only tiny|full4|full3] [--check]
"""
import argparse
import hashlib
import importlib
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------
# reference import
# ---------------------------------------------------------------------------------------------

def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("IPython", embed=lambda *a, **k: None)

    class IntermediateLayerGetter(nn.ModuleDict):
        def __init__(self, model, return_layers):
            orig, rl, layers = dict(return_layers), dict(return_layers), OrderedDict()
            for name, m in model.named_children():
                layers[name] = m
                rl.pop(name, None)
                if not rl:
                    break
            super().__init__(layers)
            self.return_layers = orig

        def forward(self, x):
            out = OrderedDict()
            for name, m in self.items():
                x = m(x)
                if name in self.return_layers:
                    out[self.return_layers[name]] = x
            return out

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            return (t - mean) / std

    tv = stub("torchvision", __version__="0.15.0")
    tvm = stub("torchvision.models")
    tvu = stub("torchvision.models._utils", IntermediateLayerGetter=IntermediateLayerGetter)
    tvt = stub("torchvision.transforms", Normalize=Normalize)
    tv.models, tvm._utils, tv.transforms = tvm, tvu, tvt
    for n, names in [("robomimic", []), ("robomimic.models", []),
                     ("robomimic.models.base_nets", ["ResNet18Conv", "SpatialSoftmax"]),
                     ("robomimic.algo", []),
                     ("robomimic.algo.diffusion_policy", ["replace_bn_with_gn", "ConditionalUnet1D"]),
                     ("diffusers", []), ("diffusers.schedulers", []),
                     ("diffusers.schedulers.scheduling_ddpm", ["DDPMScheduler"]),
                     ("diffusers.schedulers.scheduling_ddim", ["DDIMScheduler"]),
                     ("diffusers.training_utils", ["EMAModel"])]:
        stub(n, **{k: None for k in names})
    sys.path.insert(0, os.path.join(REF, "detr"))
    sys.path.insert(0, REF)
    ns = types.SimpleNamespace()
    ns.policy = importlib.import_module("policy")
    ns.dv = importlib.import_module("detr.models.detr_vae")
    ns.tr = importlib.import_module("detr.models.transformer")
    ns.bb = importlib.import_module("detr.models.backbone")
    ns.pe = importlib.import_module("detr.models.position_encoding")
    ns.lm = importlib.import_module("detr.models.latent_model")
    return ns


# torchvision-named resnet18 trunk (third-party architecture, restated)
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = norm(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = norm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), norm(cout))

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _ResNet18(nn.Module):
    def __init__(self, norm, w=64):
        super().__init__()
        self.conv1 = nn.Conv2d(3, w, 7, 2, 3, bias=False)
        self.bn1 = norm(w)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_BasicBlock(w, w, 1, norm), _BasicBlock(w, w, 1, norm))
        self.layer2 = nn.Sequential(_BasicBlock(w, 2 * w, 2, norm), _BasicBlock(2 * w, 2 * w, 1, norm))
        self.layer3 = nn.Sequential(_BasicBlock(2 * w, 4 * w, 2, norm), _BasicBlock(4 * w, 4 * w, 1, norm))
        self.layer4 = nn.Sequential(_BasicBlock(4 * w, 8 * w, 2, norm), _BasicBlock(8 * w, 8 * w, 1, norm))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(8 * w, 1000)


def build_reference_policy(ref, cfg):
    args = types.SimpleNamespace(hidden_dim=cfg.hidden_dim, position_embedding=cfg.position_embedding,
                                 dropout=cfg.dropout, nheads=cfg.nheads, dim_feedforward=cfg.dim_feedforward,
                                 enc_layers=cfg.enc_layers, dec_layers=cfg.dec_layers, pre_norm=cfg.pre_norm)
    backbones = []
    for _ in cfg.camera_names:
        body = ref.bb.BackboneBase(_ResNet18(ref.bb.FrozenBatchNorm2d, cfg.base_width), True, 8 * cfg.base_width, False)
        j = ref.bb.Joiner(body, ref.pe.build_position_encoding(args))
        j.num_channels = body.num_channels
        backbones.append(j)
    model = ref.dv.DETRVAE(backbones, ref.tr.build_transformer(args), ref.dv.build_encoder(args),
                           state_dim=cfg.state_dim, num_queries=cfg.num_queries, camera_names=cfg.camera_names,
                           vq=cfg.vq, vq_class=cfg.vq_class, vq_dim=cfg.vq_dim, action_dim=cfg.action_dim,
                           pcl_backbone=None, depth_backbones=None)
    pol = ref.policy.ACTPolicy.__new__(ref.policy.ACTPolicy)
    nn.Module.__init__(pol)
    pol.model = model
    pol.kl_weight = cfg.kl_weight
    pol.vq = cfg.vq
    pol.use_depth = False
    pol.use_pcd = False
    return pol


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sub(t: torch.Tensor, max_elems=None):
    """float32 copy; when max_elems is given, the strided flat sample of actmi.weights.fixture_sample."""
    from actmi.weights import fixture_sample
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
    return fixture_sample(a, max_elems) if max_elems else a


def make_fixture(ref, name, cfg, batch, seed_w, seed_in, train=True, store_all_grads=False, stage_step=None):
    from actmi import weights as W
    spec = W.act_state_dict_spec(cfg)
    pol = build_reference_policy(ref, cfg)
    ref_sd = pol.model.state_dict()
    assert list(ref_sd.keys()) == list(spec.keys()), "state_dict key order differs from reference"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    assert list(pol.state_dict().keys()) == ["model." + k for k in spec], "policy prefix"
    sd_np = W.generate_state_dict(cfg, seed_w)
    assert np.array_equal(sd_np["pos_table"], ref_sd["pos_table"].numpy()), "pos_table restatement"
    pol.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    pol.eval()
    inp = W.generate_inputs(cfg, batch, seed_in, with_actions=True)
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    qpos = torch.from_numpy(inp["qpos"])
    out = {"config_json": np.array(__import__("json").dumps(cfg.to_dict())), "batch": np.array(batch),
           "seed_w": np.array(seed_w), "seed_in": np.array(seed_in),
           "n_params": np.array(sum(int(np.prod(s)) for k, s in spec.items() if not W.is_buffer(k)))}
    # hashes of a few generated tensors so the GPU box can verify regeneration
    for k in ["transformer.encoder.layers.0.self_attn.in_proj_weight", "backbones.0.0.body.conv1.weight",
              "action_head.weight", "query_embed.weight"]:
        out["sha:" + k] = np.array(sha(sd_np[k]))
    out["sha:image_u8"] = np.array(sha(inp["image_u8"]))
    out["sha:qpos"] = np.array(sha(inp["qpos"]))

    # ---- stage capture through hooks
    stages = {}

    def keep(d, k, v):           # forward hooks must return None or they replace the module output
        d.setdefault(k, v.detach().clone())

    body = pol.model.backbones[0][0].body
    hooks = [body.relu.register_forward_hook(lambda m, i, o: keep(stages, "cam0_conv1", o)),
             body.maxpool.register_forward_hook(lambda m, i, o: keep(stages, "cam0_maxpool", o))]
    for li in (1, 2, 3, 4):
        hooks.append(getattr(body, f"layer{li}").register_forward_hook(
            lambda m, i, o, li=li: keep(stages, f"cam0_layer{li}", o)))
    hooks.append(pol.model.transformer.encoder.register_forward_pre_hook(
        lambda m, a: keep(stages, "src", a[0])))
    hooks.append(pol.model.transformer.encoder.register_forward_hook(
        lambda m, i, o: keep(stages, "memory", o)))
    hooks.append(pol.model.transformer.register_forward_hook(
        lambda m, i, o: keep(stages, "hs_all", o)))
    vq_sample = torch.from_numpy(inp["vq_sample"]) if cfg.vq else None
    with torch.no_grad():
        a_hat = pol(qpos, image, vq_sample=vq_sample) if cfg.vq else pol(qpos, image)
    for h in hooks:
        h.remove()
    out["infer.a_hat"] = sub(a_hat)
    hs_all = stages.pop("hs_all")                                  # [L,B,Q,D]
    out["infer.hs"] = sub(hs_all[0], stage_step)
    out["infer.is_pad_hat"] = sub(pol.model.is_pad_head(hs_all[0]))
    for k, v in stages.items():
        out["stage." + k] = sub(v, stage_step)
    out["sample_max_elems"] = np.array(stage_step or 0)

    if train:
        actions = torch.from_numpy(inp["actions"])
        is_pad = torch.from_numpy(inp["is_pad"])
        # reference draws eps inside reparametrize (detr_vae.py:19-22) from the global torch RNG; in eval
        # mode nothing else consumes the RNG before it, so the same seed replays the same draw.
        torch.manual_seed(4321)
        eps = torch.empty(batch, cfg.latent_dim).normal_()
        pol.zero_grad()
        cap = {}
        h = pol.model.latent_proj.register_forward_hook(lambda m, i, o: keep(cap, "latent_info", o))
        h2 = pol.model.latent_out_proj.register_forward_hook(lambda m, i, o: keep(cap, "z", i[0]))
        h3 = pol.model.action_head.register_forward_hook(lambda m, i, o: keep(cap, "a_hat", o))
        torch.manual_seed(4321)
        # torch 2.10 CPU autograd with >1 thread returns run-to-run different (up to 10 % off) weight
        # gradients for the 1x1 stride-2 downsample convs; single-threaded fp32 agrees with fp64 to 1e-7.
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)
        loss_dict = pol(qpos, image, actions, is_pad)
        h.remove(); h2.remove(); h3.remove()
        if cfg.vq:
            # the reference drew the code with torch.multinomial (detr_vae.py:140); the straight-through input of
            # latent_out_proj is binaries - probs + probs = binaries up to rounding: recover the exact one-hot code
            code = (cap["z"] > 0.5).float().view(batch, cfg.vq_class, cfg.vq_dim)
            assert torch.equal(code.sum(-1), torch.ones(batch, cfg.vq_class))
            probs = torch.softmax(cap["latent_info"].view(batch, cfg.vq_class, cfg.vq_dim), -1)
            out["train.vq_code"], out["train.vq_probs"] = sub(code), sub(probs)
            out["train.vq_discrepancy"] = np.array(loss_dict["vq_discrepancy"].detach().numpy(), dtype=np.float32).reshape(-1)
            mu = logvar = None
        else:
            mu, logvar = cap["latent_info"][:, :cfg.latent_dim], cap["latent_info"][:, cfg.latent_dim:]
            z_expect = mu + (logvar / 2).exp() * eps
            assert torch.allclose(cap["z"], z_expect, atol=0, rtol=0), "eps replay mismatch"
        loss_dict["loss"].backward()
        torch.set_num_threads(nthreads)
        if not cfg.vq:
            out["train.eps"] = sub(eps)
            out["train.mu"], out["train.logvar"] = sub(mu), sub(logvar)
        out["train.a_hat"] = sub(cap["a_hat"])
        for k in ("l1", "kl", "loss"):
            out["train." + k] = np.array(torch.as_tensor(loss_dict[k]).detach().numpy(), dtype=np.float32).reshape(-1)
        gnames, gnorm, gsum, gnone = [], [], [], []
        sel = ["action_head.weight", "action_head.bias", "latent_proj.weight", "latent_out_proj.weight",
               "encoder.layers.0.linear1.weight", "encoder_action_proj.weight", "input_proj.weight",
               "transformer.encoder.layers.0.self_attn.in_proj_bias", "transformer.decoder.layers.0.norm2.weight",
               "transformer.decoder.layers.0.multihead_attn.out_proj.weight", "query_embed.weight",
               "additional_pos_embed.weight", "input_proj_robot_state.weight",
               "backbones.0.0.body.conv1.weight", "backbones.1.0.body.layer4.1.conv2.weight",
               "backbones.0.0.body.layer2.0.downsample.0.weight"]
        for k, p in pol.model.named_parameters():
            gnames.append(k)
            if p.grad is None:
                gnone.append(k); gnorm.append(-1.0); gsum.append(0.0)
                continue
            g = p.grad.detach().double()
            gnorm.append(float(g.norm())); gsum.append(float(g.sum()))
            if store_all_grads or k in sel:
                out["grad." + k] = sub(p.grad, stage_step)
        out["grad_names"] = np.array(gnames)
        out["grad_l2"] = np.array(gnorm, dtype=np.float64)
        out["grad_sum"] = np.array(gsum, dtype=np.float64)
        out["grad_none"] = np.array(gnone)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KB)  a_hat mean|.|={float(a_hat.abs().mean()):.4f}")
    return out, pol, sd_np, inp


def cross_check_oracle(cfg, out, sd_np, inp, tol=2e-5):
    """The CPU restatement must reproduce the reference outputs (it is the thing that travels)."""
    sys.path.insert(0, ROOT)
    from oracle import act_ref as R
    from actmi import weights as W
    sd = {"model." + k: torch.from_numpy(v) for k, v in sd_np.items()}
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    qpos = torch.from_numpy(inp["qpos"])
    with torch.no_grad():
        vq_sample = torch.from_numpy(inp["vq_sample"]) if cfg.vq else None
        a = R.policy_call(sd, cfg, qpos, image, vq_sample=vq_sample)
        a_live = R.policy_call(sd, cfg, qpos, image, live_only=True, vq_sample=vq_sample)
    d = float((a - torch.from_numpy(out["infer.a_hat"])).abs().max())
    d2 = float((a_live - a).abs().max())
    print(f"  oracle vs reference: max|a_hat diff| = {d:.3e}; live_only vs as-written = {d2:.3e}")
    assert d < tol and d2 == 0.0
    if "train.loss" in out:
        with torch.no_grad():
            r = R.policy_call(sd, cfg, qpos, image, torch.from_numpy(inp["actions"]), torch.from_numpy(inp["is_pad"]),
                              None if cfg.vq else torch.from_numpy(out["train.eps"]),
                              vq_sample=torch.from_numpy(out["train.vq_code"]).view(-1, cfg.vq_class, cfg.vq_dim) if cfg.vq else None)
        for k in ("l1", "kl", "loss"):
            dd = abs(float(r[k]) - float(out["train." + k][0]))
            print(f"  oracle {k}: {float(r[k]):.6f} ref {float(out['train.'+k][0]):.6f} diff {dd:.2e}")
            assert dd < 5e-5 * max(1.0, abs(float(r[k])))


def make_latent_prior_train_fixture(ref):
    """The prior's TRAINING step from the reference's own module and torch's autograd / AdamW (train_latent_model.py:323-343,
    395-404), dropout off (module in eval mode; gradients enabled, so nn.MultiheadAttention takes its regular path).  Default
    width / depth (256, 8 heads, 3 blocks); vq_class = vq_dim = 8 because the reference's cross entropy needs T == V.  The
    weights are the seeded synthetic ones (not stored); gradients and updated parameters are stored in full for the small
    tensors and as norm + first 256 elements for the matrices."""
    from actmi import weights as W
    from actmi.latent_model import latent_model_spec
    vq, n, lr, steps = 8, 5, 1e-3, 3
    m = ref.lm.Latent_Model_Transformer(vq, vq, vq)
    spec = latent_model_spec(vq, vq, vq)
    sd_np = W.generate_latent_model_state_dict(spec, 41)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m.eval()
    pick = (W.uniform01(42, "lm:labels", n * vq) * vq).astype(np.int64).reshape(n, vq).clip(0, vq - 1)
    labels = np.zeros((n, vq, vq), dtype=np.float32)
    for b in range(n):
        for t in range(vq):
            labels[b, t, pick[b, t]] = 1.0
    gt = torch.from_numpy(labels)
    inputs = torch.cat([torch.zeros_like(gt)[:, [0]], gt[:, :-1]], dim=1)            # train_latent_model.py:327
    opt = torch.optim.AdamW(m.parameters(), lr=lr)                                   # :361
    out = {"vq": np.array(vq), "seed_w": np.array(41), "lr": np.array(lr), "steps": np.array(steps), "labels": labels,
           "inputs": inputs.numpy()}
    losses = []
    for it in range(steps):
        opt.zero_grad()
        logits = m(inputs)
        loss = torch.nn.functional.cross_entropy(logits, gt)                         # :329
        loss.backward()
        if it == 0:
            with torch.no_grad():
                onehot = torch.nn.functional.one_hot(torch.argmax(logits, dim=-1), num_classes=vq).float()
                out["l1_error"] = torch.nn.functional.l1_loss(onehot, gt, reduction="mean").numpy()
            out["logits"] = logits.detach().numpy()
            for k, prm in m.named_parameters():
                g = prm.grad.detach().numpy()
                out["gnorm:" + k] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
                out["g:" + k] = g.copy() if g.size <= 4096 else g.reshape(-1)[:256].copy()
        opt.step()
        losses.append(float(loss.detach()))
        if it in (0, steps - 1):
            for k, prm in m.named_parameters():
                v = prm.detach().numpy()
                out[f"p{it + 1}:" + k] = v.copy() if v.size <= 4096 else v.reshape(-1)[:256].copy()
    out["losses"] = np.array(losses)
    path = os.path.join(GOLD, "latent_prior_train.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KB); losses {losses}")
    sys.path.insert(0, ROOT)
    from oracle import act_ref as R
    r = R.latent_model_train_step({k: torch.from_numpy(v) for k, v in sd_np.items()}, inputs, gt, lr=lr, steps=steps)
    print(f"  oracle loss {float(r['loss']):.7f} ref {losses[0]:.7f}; losses after steps {r['losses']}")
    assert abs(float(r["loss"]) - losses[0]) < 1e-5 and abs(r["losses"][-1] - losses[-1]) < 1e-5


def make_latent_prior_fixture(ref):
    """VQ-ACT prior: the reference's Latent_Model_Transformer (latent_model.py:35-56) in eval mode on one-hot prefixes."""
    from actmi import weights as W
    from actmi.latent_model import latent_model_spec
    vq_dim, vq_class, n = 8, 4, 3
    m = ref.lm.Latent_Model_Transformer(vq_dim, vq_dim, vq_class)
    spec = latent_model_spec(vq_dim, vq_dim, vq_class)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(spec.keys()), "latent model key order differs from reference"
    for k, shp in spec.items():
        assert tuple(ref_sd[k].shape) == tuple(shp), k
    sd_np = W.generate_latent_model_state_dict(spec, 31)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m.eval()
    pick = (W.uniform01(32, "lm:x", n * vq_class) * vq_dim).astype(np.int64).reshape(n, vq_class)
    x = np.zeros((n, vq_class, vq_dim), dtype=np.float32)          # row 0 = the all-zero start token of generate()
    for b in range(n):
        for t in range(1, vq_class):
            x[b, t, min(pick[b, t], vq_dim - 1)] = 1.0
    with torch.no_grad():
        logits = m(torch.from_numpy(x))
    out = {"vq_dim": np.array(vq_dim), "vq_class": np.array(vq_class), "seed_w": np.array(31), "x": x, "logits": logits.numpy(),
           "sha:output_layer.weight": np.array(sha(sd_np["output_layer.weight"])),
           "sha:attention_blocks.2.mlp.0.weight": np.array(sha(sd_np["attention_blocks.2.mlp.0.weight"]))}
    path = os.path.join(GOLD, "latent_prior.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KB)")
    sys.path.insert(0, ROOT)
    from oracle import act_ref as R
    with torch.no_grad():
        lo = R.latent_model_forward({k: torch.from_numpy(v) for k, v in sd_np.items()}, torch.from_numpy(x))
    d = float((lo - logits).abs().max())
    print(f"  oracle vs reference latent prior: max|logit diff| = {d:.3e}")
    assert d < 2e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    from actmi.config import ACTConfig, tiny_config
    ref = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    jobs = {
        "tiny": dict(cfg=tiny_config(), batch=2, seed_w=0, seed_in=1234, train=True, store_all_grads=True, stage_step=None),
        "tiny_c3": dict(cfg=tiny_config(camera_names=["a", "b", "c"], num_queries=10, image_h=96, image_w=64),
                        batch=3, seed_w=5, seed_in=77, train=True, store_all_grads=False, stage_step=None),
        "full4": dict(cfg=ACTConfig(), batch=2, seed_w=0, seed_in=1234, train=True, store_all_grads=False, stage_step=8192),
        "full3": dict(cfg=ACTConfig(camera_names=["top", "left_wrist", "right_wrist"]), batch=1, seed_w=3, seed_in=99,
                      train=False, store_all_grads=False, stage_step=8192),
        # VQ-ACT inference (detr_vae.py:155-156): latent = latent_out_proj(given one-hot code)
        "tiny_vq": dict(cfg=tiny_config(vq=True, vq_class=4, vq_dim=8), batch=3, seed_w=9, seed_in=21, train=True,
                        store_all_grads=True, stage_step=None),
    }
    if not args.only or args.only == "latent_prior":
        print("== latent_prior")
        make_latent_prior_fixture(ref)
    if not args.only or args.only == "latent_prior_train":
        print("== latent_prior_train")
        make_latent_prior_train_fixture(ref)
    for name, j in jobs.items():
        if args.only and name != args.only:
            continue
        print(f"== {name}")
        out, pol, sd_np, inp = make_fixture(ref, name, **j)
        cross_check_oracle(j["cfg"], out, sd_np, inp)


if __name__ == "__main__":
    main()
