"""Full-size parity at the batch sizes the benchmark and the eval rollouts actually run:
* the two full4 golden samples (reference DETRVAE outputs, B=2) placed in slots of a B=8 batch must reproduce the golden rows
  -- a DIRECT comparison of the benchmark configuration (C=4, 480x640, B=8) with outputs of the reference;
* B=50, C=3, 480x640 (BASELINE config 2's batch): finite, batch-independent, hipGraph replay == eager launches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_fixture, regenerate  # noqa: E402
from actmi import weights as W  # noqa: E402
from actmi import ops  # noqa: E402
from actmi.engine import ACTEngine  # noqa: E402

ATOL = 1e-4


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_full4_golden_rows_inside_a_batch_of_8(prec):
    z, cfg = load_fixture("full4")
    sd_np, inp = regenerate(z, cfg, with_actions=True)
    assert int(z["batch"]) == 2 and cfg.num_cams == 4 and (cfg.image_h, cfg.image_w) == (480, 640)
    eng = ACTEngine(cfg, max_batch=8, gemm_prec=prec)
    eng.load_state_dict(sd_np)
    eng.finalize()
    filler = W.generate_inputs(cfg, 8, seed=77)
    qpos, img = filler["qpos"].copy(), filler["image_u8"].copy()
    slots = (2, 7)                                   # golden sample i sits in batch slot slots[i]
    for i, s in enumerate(slots):
        qpos[s], img[s] = inp["qpos"][i], inp["image_u8"][i]
    a = eng.forward_infer(torch.from_numpy(qpos).cuda(), torch.from_numpy(img).cuda()).cpu().numpy()
    for i, s in enumerate(slots):
        err = np.abs(a[s] - z["infer.a_hat"][i]).max()
        print(f"full4 sample {i} in slot {s} of B=8 [{prec}]: max|a_hat - ref| = {err:.3e}")
        assert err <= ATOL
    assert np.isfinite(a).all()


def test_rollout_batch_50_three_cameras_properties():
    from actmi.config import ACTConfig
    cfg = ACTConfig(camera_names=["top", "left_wrist", "right_wrist"])
    B = 50
    eng = ACTEngine(cfg, max_batch=B)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
    eng.finalize()
    inp = W.generate_inputs(cfg, B, seed=31)
    q, im = torch.from_numpy(inp["qpos"]).cuda(), torch.from_numpy(inp["image_u8"]).cuda()
    a = eng.forward_infer(q, im).clone()
    assert torch.isfinite(a).all()
    assert torch.equal(a, eng.forward_infer(q, im))                                  # run-to-run identical
    # a sample's output does not depend on its batch neighbours (FrozenBN: no batch statistics); small batches take
    # different tile shapes / split contractions, so agreement is to summation-order noise
    for s in (0, 23, 49):
        a1 = eng.forward_infer(q[s:s + 1].contiguous(), im[s:s + 1].contiguous())
        assert float((a1[0] - a[s]).abs().max()) <= 3e-5
    # the captured graph of the whole step (forward + ensemble) replays to the same bits as eager launches
    ens_g = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, eng.device)
    ens_e = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, eng.device)
    replay = eng.capture_infer(B, with_ensemble=ens_g)
    a_g, raw_g = replay(q, im)
    raw_e = ens_e.step(a)
    assert torch.equal(a_g, a) and torch.equal(raw_g, raw_e)


def test_failed_call_leaves_no_stale_error_message():
    """C ABI: a failing call sets the handle's message, the next successful call clears it (robustness #13)."""
    import ctypes as C
    from actmi.config import tiny_config
    from actmi import lib as L
    cfg = tiny_config()
    eng = ACTEngine(cfg, max_batch=2)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=2))
    eng.finalize()
    lib = L.load()
    inp = W.generate_inputs(cfg, 2, seed=1)
    q, im = torch.from_numpy(inp["qpos"]).cuda(), torch.from_numpy(inp["image_u8"]).cuda()
    out = torch.empty((2, cfg.num_queries, cfg.action_dim), device="cuda")
    rc = lib.actmi_forward_infer(eng.h, C.c_void_p(q.data_ptr()), C.c_void_p(im.data_ptr()), 0, 3,       # B > max_batch
                                 C.c_void_p(out.data_ptr()), eng._sp())
    assert rc != 0 and b"max_batch" in lib.actmi_last_error(eng.h)
    rc = lib.actmi_forward_infer(eng.h, C.c_void_p(q.data_ptr()), C.c_void_p(im.data_ptr()), 0, 2,
                                 C.c_void_p(out.data_ptr()), eng._sp())
    assert rc == 0 and lib.actmi_last_error(eng.h) == b""
    rc = lib.actmi_set_param(eng.h, b"no.such.key", C.c_void_p(q.data_ptr()), None, 0, 1)
    assert rc != 0 and b"unknown" in lib.actmi_last_error(eng.h)
    t = torch.empty(cfg.action_dim)
    rc = lib.actmi_get_param(eng.h, b"action_head.bias", C.c_void_p(t.data_ptr()), t.numel() * 4, 0)
    assert rc == 0 and lib.actmi_last_error(eng.h) == b""


def test_engine_binds_to_its_device_not_the_process_default():
    """the engine lives on the device it was given and launches there whatever device is current; tensors
    from another device are refused before any kernel launch."""
    from actmi.config import tiny_config
    cfg = tiny_config()
    eng = ACTEngine(cfg, max_batch=2, device="cuda:0")
    assert eng.device == torch.device("cuda", 0)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=2))
    inp = W.generate_inputs(cfg, 2, seed=1)
    with pytest.raises(ValueError):
        eng.forward_infer(torch.from_numpy(inp["qpos"]), torch.from_numpy(inp["image_u8"]))          # CPU tensors
    from policy import ACTPolicy
    pol = ACTPolicy({"lr": 1e-5, "num_queries": cfg.num_queries, "kl_weight": 10, "hidden_dim": cfg.hidden_dim,
                     "dim_feedforward": cfg.dim_feedforward, "enc_layers": cfg.enc_layers, "dec_layers": cfg.dec_layers,
                     "nheads": cfg.nheads, "camera_names": cfg.camera_names, "image_h": cfg.image_h, "image_w": cfg.image_w,
                     "base_width": cfg.base_width, "training": False, "max_batch": 2})
    assert pol.model.device == torch.device("cuda", torch.cuda.current_device())
    ens = ops.TemporalEnsemble(2, cfg.num_queries, cfg.action_dim, 0.01, pol.model.device)
    a = pol(torch.from_numpy(inp["qpos"]).cuda(), torch.from_numpy(inp["image_u8"]).cuda())
    assert a.device == pol.model.device and ens.step(a).device == pol.model.device
