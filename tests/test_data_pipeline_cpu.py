"""Input pipeline (actmi/data.py) against an independent restatement of utils.py:41-301 written here.
The reference module itself cannot be imported (h5py, cv2, torchvision absent): parity unpinned, checked by
construction on fabricated episodes with the reference's key layout."""
import os

import numpy as np
import pytest
import torch

from actmi import data as D


def _write_episode(path, T, cams, rng, sim, base=False, A=14):
    ep = {"/observations/qpos": rng.standard_normal((T, 14)).astype(np.float32) * 2 + 1,
          "/observations/qvel": rng.standard_normal((T, 14)).astype(np.float32),
          "/action": rng.standard_normal((T, A)).astype(np.float32) * 3 - 0.5}
    for c in cams:
        ep[f"/observations/images/{c}"] = rng.integers(0, 256, (T, 12, 16, 3), dtype=np.uint8)
    if base:
        ep["/base_action"] = rng.standard_normal((T, 2)).astype(np.float32)
    if sim is not None:
        ep["attrs_sim"] = np.array(sim)
    np.savez(path, **ep)
    return ep


@pytest.fixture()
def episodes(tmp_path):
    rng = np.random.default_rng(0)
    cams = ["top", "left_wrist"]
    eps, paths = [], []
    for i, (T, sim, base) in enumerate([(40, True, False), (25, False, False), (33, None, False)]):
        p = str(tmp_path / f"episode_{i}.npz")
        eps.append(_write_episode(p, T, cams, rng, sim, base))
        paths.append(p)
    return paths, eps, cams


def test_norm_stats_match_the_reference_formulas(episodes):
    paths, eps, _ = episodes
    stats, lens = D.get_norm_stats(paths)
    qpos = np.concatenate([e["/observations/qpos"] for e in eps]); act = np.concatenate([e["/action"] for e in eps])
    assert lens == [40, 25, 33]
    assert np.allclose(stats["qpos_mean"], qpos.mean(0), atol=1e-6)
    assert np.allclose(stats["qpos_std"], np.clip(qpos.std(0, ddof=1), 1e-2, np.inf), atol=1e-6)      # torch.std is unbiased
    assert np.allclose(stats["action_std"], np.clip(act.std(0, ddof=1), 1e-2, np.inf), atol=1e-6)
    assert np.allclose(stats["action_min"], act.min(0) - 1e-4) and np.allclose(stats["action_max"], act.max(0) + 1e-4)
    assert np.array_equal(stats["example_qpos"], eps[-1]["/observations/qpos"])


def test_getitem_semantics(episodes):
    paths, eps, cams = episodes
    stats, lens = D.get_norm_stats(paths)
    ids = [2, 0, 1]                                         # shuffled episode ids, as load_data produces
    ds = D.EpisodicDataset(paths, cams, stats, ids, [lens[i] for i in ids], chunk_size=30, policy_class="ACT")
    assert len(ds) == sum(lens) and ds.max_episode_len == 40
    for index in [0, 5, 32, 33, 60, 72, 73, 80, 97]:
        # restatement of _locate_transition / __getitem__
        cum = np.cumsum([lens[i] for i in ids])
        e = int(np.argmax(cum > index)); start = index - (cum[e] - lens[ids[e]]); ep = eps[ids[e]]
        sim = bool(ep["attrs_sim"]) if "attrs_sim" in ep else False
        a = ep["/action"]
        a_tail = a[start:] if sim else a[max(0, start - 1):]
        pad = np.zeros((40, a.shape[1]), np.float32); pad[:len(a_tail)] = a_tail
        ip = np.zeros(40); ip[len(a_tail):] = 1
        img, qpos, act, is_pad = ds[index]
        assert img.dtype == torch.uint8 and tuple(img.shape) == (2, 12, 16, 3)
        assert np.array_equal(img.numpy(), np.stack([ep[f"/observations/images/{c}"][start] for c in cams]))
        assert np.allclose(qpos.numpy(), (ep["/observations/qpos"][start] - stats["qpos_mean"]) / stats["qpos_std"], atol=1e-6)
        assert np.allclose(act.numpy(), ((pad - stats["action_mean"]) / stats["action_std"])[:30], atol=1e-5)
        assert np.array_equal(is_pad.numpy(), ip[:30].astype(bool)) and is_pad.dtype == torch.bool
    # the reference's float contract on request
    dsf = D.EpisodicDataset(paths, cams, stats, ids, [lens[i] for i in ids], 30, "ACT", f32_images=True)
    imgf = dsf[5][0]
    assert imgf.dtype == torch.float32 and tuple(imgf.shape) == (2, 3, 12, 16)
    assert torch.equal(imgf, torch.einsum("k h w c -> k c h w", ds[5][0]) / 255.0)


def test_base_action_is_smoothed_and_appended(tmp_path):
    rng = np.random.default_rng(1)
    p = str(tmp_path / "episode_0.npz")
    ep = _write_episode(p, 20, ["top"], rng, True, base=True)
    stats, lens = D.get_norm_stats([p])
    assert stats["action_mean"].shape == (16,)
    smooth = np.stack([np.convolve(ep["/base_action"][:, i], np.ones(5) / 5, mode="same") for i in range(2)], -1)
    full = np.concatenate([ep["/action"], smooth.astype(np.float32)], -1)
    assert np.allclose(stats["action_mean"], full.mean(0), atol=1e-6)
    ds = D.EpisodicDataset([p], ["top"], stats, [0], lens, 10, "ACT")
    assert tuple(ds[3][2].shape) == (10, 16)


def test_batch_sampler_ranges_and_weights():
    lens_l = [[10, 20], [5]]                                # two dataset dirs
    it = D.BatchSampler(64, lens_l, [0.75, 0.25], rng=np.random.default_rng(3))
    idx = np.array([next(it) for _ in range(200)]).reshape(-1)
    assert idx.min() >= 0 and idx.max() < 35
    frac_second = float((idx >= 30).mean())
    assert abs(frac_second - 0.25) < 0.02
    it2 = D.BatchSampler(8, lens_l, None, rng=np.random.default_rng(3))
    assert len(next(it2)) == 8


def test_load_data_and_prefetcher_end_to_end(tmp_path):
    rng = np.random.default_rng(5)
    for i in range(6):
        _write_episode(str(tmp_path / f"episode_{i}.npz"), 15 + i, ["top", "left_wrist"], rng, True)
    _write_episode(str(tmp_path / "episode_features_0.npz"), 9, ["top", "left_wrist"], rng, True)     # skipped by name
    tr, va, stats, is_sim = D.load_data(str(tmp_path), lambda n: True, ["top", "left_wrist"], 4, 2, 12, num_workers=0,
                                        train_ratio=0.7, rng=np.random.default_rng(11))
    assert is_sim is False and stats["qpos_mean"].shape == (14,)
    pf = D.DevicePrefetcher((b for _, b in zip(range(5), tr)), device=None)
    n = 0
    for img, qpos, act, is_pad in pf:
        assert tuple(img.shape) == (4, 2, 12, 16, 3) and img.dtype == torch.uint8
        assert tuple(qpos.shape) == (4, 14) and tuple(act.shape) == (4, 12, 14) and tuple(is_pad.shape) == (4, 12)
        n += 1
    assert n == 5
    vb = next(iter(va))
    assert tuple(vb[0].shape) == (2, 2, 12, 16, 3)


def test_compressed_episode_frames_are_decoded_like_cv2_imdecode(tmp_path):
    """utils.py:104-107: attr `compress` -> every stored frame is a zero-padded JPEG byte string; the loader returns what
    ``cv2.imdecode(buf, 1)`` would: the decoded picture in B, G, R order.  cv2 is absent: the expected pixels come from PIL's
    decoder on the unpadded bytes (parity with cv2 itself unpinned)."""
    import io
    from PIL import Image
    rng = np.random.default_rng(5)
    T, cams = 6, ["top"]
    ep = _write_episode(str(tmp_path / "raw.npz"), T, cams, rng, True)
    os.remove(str(tmp_path / "raw.npz"))
    frames = np.kron(rng.integers(0, 256, (T, 6, 8, 3), dtype=np.uint8), np.ones((1, 4, 4, 1), dtype=np.uint8))   # smooth blocks
    bufs, expect = [], []
    for t in range(T):
        bio = io.BytesIO()
        Image.fromarray(frames[t]).save(bio, format="JPEG", quality=90)
        raw = np.frombuffer(bio.getvalue(), dtype=np.uint8)
        bufs.append(raw)
        expect.append(np.asarray(Image.open(io.BytesIO(raw.tobytes())).convert("RGB"))[..., ::-1])
    L = max(len(b) for b in bufs) + 7
    padded = np.zeros((T, L), dtype=np.uint8)
    for t, b in enumerate(bufs):
        padded[t, :len(b)] = b
    ep["/observations/images/top"] = padded
    ep["attrs_compress"] = np.array(True)
    path = str(tmp_path / "episode_0.npz")
    np.savez(path, **ep)
    stats, lens = D.get_norm_stats([path])
    ds = D.EpisodicDataset([path], cams, stats, [0], lens, 10, "ACT")
    for idx in (0, 3, T - 1):
        img, _, _, _ = ds[idx]
        assert img.dtype == torch.uint8 and tuple(img.shape) == (1, 24, 32, 3)
        assert np.array_equal(img[0].numpy(), expect[idx])
        got = img[0].numpy().astype(int)                                   # it IS the picture, channel-flipped (JPEG is lossy)
        assert np.abs(got[..., ::-1] - frames[idx]).mean() < 0.5 * np.abs(got - frames[idx]).mean()
