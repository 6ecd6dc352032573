"""The binding printed in INTEGRATION.md section 1 is executed VERBATIM (extracted from the markdown file):

* CPU: its ``Cfg`` structure has the size and field offsets `gcc` gives `struct actmi_config` of include/actmi.h, and a stale
  binding (the round-2 text, which lacked the trailing vq fields) is rejected by ``actmi_create`` through ``struct_size``
  instead of making the library read past the struct;
* GPU: one policy query through the stub's ``ACTPolicy`` equals the CPU oracle at the 1e-4 bar of BASELINE.json.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER_DIR = os.path.join(ROOT, "include")


def _stub_namespace():
    from actmi import lib as L
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(.*?)```", txt, re.S)
    assert m, "INTEGRATION.md lost its binding block"
    os.environ["ACTMI_LIB"] = L.LIB_PATH
    ns = {}
    exec(compile(m.group(1), "INTEGRATION.md#binding", "exec"), ns)
    return ns


def _c_layout(tmp_path):
    """sizeof / offsetof of struct actmi_config as the C compiler lays it out."""
    src = tmp_path / "layout.c"
    fields = ["struct_size", "num_cams", "image_h", "image_w", "base_width", "hidden_dim", "nheads", "dim_feedforward",
              "enc_layers", "dec_layers", "num_queries", "state_dim", "action_dim", "latent_dim", "has_cvae_encoder",
              "max_batch", "enable_training", "kl_weight", "vq", "vq_class", "vq_dim"]
    body = "".join(f'  printf("{f} %zu\\n", offsetof(actmi_config, {f}));\n' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "actmi.h"\nint main(void) {\n'
                   '  printf("sizeof %zu\\n", sizeof(actmi_config));\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", HEADER_DIR, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return dict(zip(out[0::2], map(int, out[1::2])))


def test_stub_struct_matches_the_header(tmp_path):
    from actmi import lib as L
    ns = _stub_namespace()
    lay = _c_layout(tmp_path)
    for cls in (ns["Cfg"], L.ActmiConfig):
        assert C.sizeof(cls) == lay["sizeof"], (cls, C.sizeof(cls), lay["sizeof"])
        for name, _ in cls._fields_:
            assert getattr(cls, name).offset == lay[name], (cls, name)
        assert [n for n, _ in cls._fields_] == [k for k in lay if k != "sizeof"]


def test_stale_binding_is_rejected_by_struct_size():
    """actmi_create checks struct_size before anything else (no HIP call yet): a binding built against an older header fails
    with ACTMI_E_INVALID and a message naming both sizes."""
    ns = _stub_namespace()
    lib, Cfg = ns["lib"], ns["Cfg"]              # (the stub itself declares actmi_last_error's return type)

    class OldCfg(C.Structure):                       # round-2 INTEGRATION.md: no struct_size, ends at kl_weight
        _fields_ = [(n, C.c_int32) for n in ("num_cams", "image_h", "image_w", "base_width", "hidden_dim", "nheads",
                    "dim_feedforward", "enc_layers", "dec_layers", "num_queries", "state_dim", "action_dim", "latent_dim",
                    "has_cvae_encoder", "max_batch", "enable_training")] + [("kl_weight", C.c_float)]
    old = OldCfg(4, 480, 640, 64, 512, 8, 3200, 4, 7, 100, 14, 16, 32, 1, 8, 0, 10.0)
    h = C.c_void_p()
    assert lib.actmi_create(C.byref(old), C.byref(h)) == -1          # ACTMI_E_INVALID; first field read as struct_size = 4
    assert b"struct_size" in lib.actmi_last_error(None)
    assert not h.value
    # the right struct with a wrong size field is rejected the same way
    cfg = Cfg(C.sizeof(Cfg) - 12, 4, 480, 640, 64, 512, 8, 3200, 4, 7, 100, 14, 16, 32, 1, 8, 0, 10.0, 0, 0, 0)
    assert lib.actmi_create(C.byref(cfg), C.byref(h)) == -1
    assert str(C.sizeof(Cfg)).encode() in lib.actmi_last_error(None)


def test_stub_error_path_raises_with_the_library_message():
    """a config the library refuses (here: 7 heads on a width of 512; on a box without a GPU the missing device is reported
    first) must surface as the RuntimeError the stub promises, text included -- not as a ctypes conversion error"""
    ns = _stub_namespace()
    a = dict(camera_names=["a"], hidden_dim=512, nheads=7, dim_feedforward=3200, enc_layers=4, dec_layers=7, num_queries=100,
             action_dim=16, no_encoder=False, kl_weight=10)
    with pytest.raises(RuntimeError) as e:
        ns["ACTPolicy"](a)
    assert len(str(e.value)) > 8, str(e.value)


@pytest.mark.gpu
def test_stub_policy_query_matches_the_oracle():
    import torch
    from actmi import weights as W
    from actmi.config import tiny_config
    from oracle import act_ref as R
    ns = _stub_namespace()
    cfg = tiny_config()
    a = dict(camera_names=cfg.camera_names, image_h=cfg.image_h, image_w=cfg.image_w, base_width=cfg.base_width,
             hidden_dim=cfg.hidden_dim, nheads=cfg.nheads, dim_feedforward=cfg.dim_feedforward, enc_layers=cfg.enc_layers,
             dec_layers=cfg.dec_layers, num_queries=cfg.num_queries, state_dim=cfg.state_dim, action_dim=cfg.action_dim,
             no_encoder=False, max_batch=2, kl_weight=cfg.kl_weight)
    pol = ns["ACTPolicy"](a)
    sd_np = W.generate_state_dict(cfg, seed=0)
    sd = {"model." + k: torch.from_numpy(v) for k, v in sd_np.items()}
    pol.deserialize(sd)
    inp = W.generate_inputs(cfg, 2, seed=77)
    qpos = torch.from_numpy(inp["qpos"]).cuda()
    img = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    got = pol(qpos, img.cuda()).cpu().numpy()
    with torch.no_grad():
        exp = R.policy_call(sd, cfg, torch.from_numpy(inp["qpos"]), img).numpy()
    assert np.abs(got - exp).max() <= 1e-4
    ns["lib"].actmi_destroy(pol.h)


@pytest.mark.gpu
def test_stub_training_steps_match_the_package_host_side():
    """ACTTrainer of the INTEGRATION.md binding (ctypes only): two forward / backward / AdamW steps give the losses the package's
    own host side (actmi.engine.ACTEngine) gives on the same weights, inputs, eps and dropout 0"""
    import torch
    from actmi import weights as W
    from actmi.config import tiny_config
    from actmi.engine import ACTEngine
    ns = _stub_namespace()
    cfg = tiny_config()
    B = 2
    a = dict(camera_names=cfg.camera_names, image_h=cfg.image_h, image_w=cfg.image_w, base_width=cfg.base_width,
             hidden_dim=cfg.hidden_dim, nheads=cfg.nheads, dim_feedforward=cfg.dim_feedforward, enc_layers=cfg.enc_layers,
             dec_layers=cfg.dec_layers, num_queries=cfg.num_queries, state_dim=cfg.state_dim, action_dim=cfg.action_dim,
             no_encoder=False, max_batch=B, kl_weight=cfg.kl_weight, lr=1e-4)
    sd = W.generate_state_dict(cfg, seed=3)
    inp = W.generate_inputs(cfg, B, seed=5, with_actions=True)
    d = "cuda:0"
    t = {k: torch.from_numpy(v).to(d) for k, v in inp.items()}
    img_f32 = t["image_u8"].permute(0, 1, 4, 2, 3).float().div(255.0).contiguous()        # the stub feeds f32 NCHW frames
    tr = ns["ACTTrainer"](a)
    tr.deserialize({"model." + k: torch.from_numpy(v) for k, v in sd.items()})
    eng = ACTEngine(cfg, max_batch=B, device=d, training=True)
    eng.load_state_dict(sd)
    eng.finalize()
    for step in range(2):
        got = tr.loss(t["qpos"], img_f32, t["actions"], t["is_pad"], t["eps"], dropout_p=0.0)
        tr.backward()
        tr.step()
        eng.zero_grad()
        ref = eng.forward_train(t["qpos"], img_f32, t["actions"], t["is_pad"], eps=t["eps"])
        eng.backward()
        eng.adamw_step(1e-4, 1e-5, 1e-4, step=step + 1)
        for k, kk in (("l1", "l1"), ("kl", "kl"), ("loss", "loss")):
            assert abs(float(got[k]) - float(ref[kk])) <= 1e-6 * max(1.0, abs(float(ref[kk]))), (step, k, float(got[k]), float(ref[kk]))
    assert float(got["loss"]) != 0.0

