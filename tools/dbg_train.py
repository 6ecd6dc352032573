import sys, os, tempfile
sys.path.insert(0, "act-plus-plus_amd"); sys.path.insert(0, ".")
import torch, numpy as np
import imitate_episodes as IE
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "data"); os.makedirs(data)
cams = ["top", "left_wrist", "right_wrist"]
for i in range(4):
    T = 5 + i
    ep = {"/observations/qpos": rng.standard_normal((T, 14)).astype(np.float32),
          "/observations/qvel": rng.standard_normal((T, 14)).astype(np.float32),
          "/action": rng.standard_normal((T, 16)).astype(np.float32), "attrs_sim": np.array(True)}
    for c in cams:
        ep[f"/observations/images/{c}"] = rng.integers(0, 256, (T, 480, 640, 3), dtype=np.uint8)
    np.savez(os.path.join(data, f"episode_{i}.npz"), **ep)
orig = IE.forward_pass
def fp(data, policy):
    image_data, qpos_data, action_data, is_pad = data
    out = orig(data, policy)
    print("train" if policy.training else "val", {k: float(v) for k, v in out.items()},
          "qpos", float(qpos_data.abs().max()), "act", float(action_data.abs().max()), "finite", bool(torch.isfinite(action_data).all()),
          "pad", is_pad.sum(1).tolist(), "flags", policy.model.read_flags(clear=False))
    return out
IE.forward_pass = fp
args = dict(eval=False, ckpt_dir=os.path.join(tmp, "ck"), policy_class="ACT", task_name="sim_transfer_cube_scripted", batch_size=2,
            seed=0, num_steps=2, lr=1e-5, kl_weight=10, chunk_size=100, hidden_dim=512, dim_feedforward=3200,
            temporal_agg=False, eval_every=1000, validate_every=2, save_every=1000, dataset_dir=data, num_rollouts=1, max_batch=2)
IE.main(args)
