"""CPU restatement (torch fp32 functional ops) of the DiffusionPolicy INFERENCE path -- TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path never does.

PARITY UNPINNED.  The reference (policy.py:20-241) delegates this arithmetic to two packages that are not vendored and not
importable offline: robomimic (`ResNet18Conv`, `SpatialSoftmax`, `replace_bn_with_gn`, `ConditionalUnet1D`; README.md:46
installs the r2d2 branch) and diffusers (`DDIMScheduler`; unpinned).  What follows restates their published definitions
(robomimic base_nets.py / algo/diffusion_policy.py, diffusers scheduling_ddim.py, torchvision resnet18) and is anchored on the
reference's own call sites only:
  policy.py:42-49   per camera ResNet18Conv(input_channel 3, pretrained False) / SpatialSoftmax(input [512,15,20], num_kp 32,
                    temperature 1.0) / Linear(64, 64)
  policy.py:66      replace_bn_with_gn: every BatchNorm2d -> GroupNorm(num_groups = features // 16, num_channels = features)
  policy.py:70-73   ConditionalUnet1D(input_dim = action_dim, global_cond_dim = obs_dim * observation_horizon)
  policy.py:102-109 DDIMScheduler(num_train_timesteps 50, squaredcos_cap_v2, clip_sample, set_alpha_to_one, steps_offset 0, epsilon)
  policy.py:177-223 inference: features per camera, obs_cond = cat(features, qpos), Gaussian start, num_inference_timesteps
                    scheduler steps on noise_pred_net(sample, timestep, global_cond)
No fixture of the reference covers it (robomimic / diffusers outputs cannot be generated here)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _gn(x, w, b, groups):
    return F.group_norm(x, groups, w, b, 1e-5)


def backbone(sd, i, x):
    """torchvision resnet18 children [:-2] with GroupNorm(C // 16) in place of BatchNorm (robomimic ResNet18Conv +
    replace_bn_with_gn); x [B,3,H,W] in [0,1] -> [B,512,H/32,W/32]."""
    P = f"policy.backbones.{i}.nets."
    x = F.conv2d(x, sd[P + "0.weight"], stride=2, padding=3)
    x = F.relu(_gn(x, sd[P + "1.weight"], sd[P + "1.bias"], x.shape[1] // 16))
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            q = f"{P}{3 + li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            idt = x
            y = F.conv2d(x, sd[q + "conv1.weight"], stride=stride, padding=1)
            y = F.relu(_gn(y, sd[q + "bn1.weight"], sd[q + "bn1.bias"], y.shape[1] // 16))
            y = F.conv2d(y, sd[q + "conv2.weight"], padding=1)
            y = _gn(y, sd[q + "bn2.weight"], sd[q + "bn2.bias"], y.shape[1] // 16)
            if q + "downsample.0.weight" in sd:
                idt = F.conv2d(x, sd[q + "downsample.0.weight"], stride=stride)
                idt = _gn(idt, sd[q + "downsample.1.weight"], sd[q + "downsample.1.bias"], idt.shape[1] // 16)
            x = F.relu(y + idt)
    return x


def spatial_softmax(sd, i, feat, temperature=1.0):
    """robomimic SpatialSoftmax.forward: 1x1 conv to num_kp maps, softmax over H*W, expected coordinates on linspace(-1,1)."""
    P = f"policy.pools.{i}."
    f = F.conv2d(feat, sd[P + "nets.weight"], sd[P + "nets.bias"])
    B, K, H, W = f.shape
    px, py = np.meshgrid(np.linspace(-1.0, 1.0, W), np.linspace(-1.0, 1.0, H))
    px = torch.from_numpy(px.reshape(1, H * W)).float()
    py = torch.from_numpy(py.reshape(1, H * W)).float()
    att = F.softmax(f.reshape(-1, H * W) / temperature, dim=-1)
    ex = torch.sum(px * att, dim=1, keepdim=True)
    ey = torch.sum(py * att, dim=1, keepdim=True)
    return torch.cat([ex, ey], 1).view(B, K, 2)


def obs_features(sd, cams, qpos, image):
    feats = []
    for i in range(cams):
        f = backbone(sd, i, image[:, i])
        kp = spatial_softmax(sd, i, f)
        feats.append(F.linear(torch.flatten(kp, 1), sd[f"policy.linears.{i}.weight"], sd[f"policy.linears.{i}.bias"]))
    return torch.cat(feats + [qpos], dim=1)


def sinusoidal_pos_emb(t, dim):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    e = t[:, None].float() * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _conv_block(sd, p, x, groups=8):
    w = sd[p + "block.0.weight"]
    x = F.conv1d(x, w, sd[p + "block.0.bias"], padding=w.shape[-1] // 2)
    return F.mish(_gn(x, sd[p + "block.1.weight"], sd[p + "block.1.bias"], groups))


def _crb(sd, p, x, cond):
    """ConditionalResidualBlock1D with FiLM (scale and bias predicted from the conditioning)."""
    out = _conv_block(sd, p + "blocks.0.", x)
    emb = F.linear(F.mish(cond), sd[p + "cond_encoder.1.weight"], sd[p + "cond_encoder.1.bias"])
    oc = out.shape[1]
    emb = emb.reshape(emb.shape[0], 2, oc, 1)
    out = emb[:, 0] * out + emb[:, 1]
    out = _conv_block(sd, p + "blocks.1.", out)
    res = F.conv1d(x, sd[p + "residual_conv.weight"], sd[p + "residual_conv.bias"]) if p + "residual_conv.weight" in sd else x
    return out + res


def unet(sd, sample, timestep, global_cond, down_dims=(256, 512, 1024)):
    """ConditionalUnet1D.forward: sample [B,T,A] -> noise prediction [B,T,A]."""
    P = "policy.noise_pred_net."
    B = sample.shape[0]
    x = sample.moveaxis(-1, -2)
    t = torch.full((B,), int(timestep), dtype=torch.long)
    dsed = sd[P + "diffusion_step_encoder.3.weight"].shape[0]
    g = sinusoidal_pos_emb(t, dsed)
    g = F.linear(g, sd[P + "diffusion_step_encoder.1.weight"], sd[P + "diffusion_step_encoder.1.bias"])
    g = F.linear(F.mish(g), sd[P + "diffusion_step_encoder.3.weight"], sd[P + "diffusion_step_encoder.3.bias"])
    g = torch.cat([g, global_cond], dim=-1)
    h = []
    n = len(down_dims)
    for i in range(n):
        x = _crb(sd, f"{P}down_modules.{i}.0.", x, g)
        x = _crb(sd, f"{P}down_modules.{i}.1.", x, g)
        h.append(x)
        if i < n - 1:
            x = F.conv1d(x, sd[f"{P}down_modules.{i}.2.conv.weight"], sd[f"{P}down_modules.{i}.2.conv.bias"], stride=2, padding=1)
    for i in range(2):
        x = _crb(sd, f"{P}mid_modules.{i}.", x, g)
    for i in range(n - 1):
        x = torch.cat((x, h.pop()), dim=1)
        x = _crb(sd, f"{P}up_modules.{i}.0.", x, g)
        x = _crb(sd, f"{P}up_modules.{i}.1.", x, g)
        x = F.conv_transpose1d(x, sd[f"{P}up_modules.{i}.2.conv.weight"], sd[f"{P}up_modules.{i}.2.conv.bias"], stride=2, padding=1)
    x = _conv_block(sd, P + "final_conv.0.", x)
    x = F.conv1d(x, sd[P + "final_conv.1.weight"], sd[P + "final_conv.1.bias"])
    return x.moveaxis(-1, -2)


def ddim_alphas_cumprod(num_train_timesteps=50):
    """diffusers betas_for_alpha_bar (squaredcos_cap_v2), float32 like DDIMScheduler."""
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = [min(1 - alpha_bar((i + 1) / num_train_timesteps) / alpha_bar(i / num_train_timesteps), 0.999)
             for i in range(num_train_timesteps)]
    return torch.cumprod(1.0 - torch.tensor(betas, dtype=torch.float32), dim=0)


def ddim_timesteps(num_inference_steps, num_train_timesteps=50):
    ratio = num_train_timesteps // num_inference_steps
    return (np.arange(0, num_inference_steps) * ratio).round()[::-1].astype(np.int64)       # steps_offset 0


def policy_call(sd, cams, qpos, image, noise, num_inference_timesteps=10, num_train_timesteps=50):
    """DiffusionPolicy.__call__(qpos, image) (policy.py:177-223) with the Gaussian start passed in (`noise` [B,Tp,A])."""
    cond = obs_features(sd, cams, qpos, image)
    ac = ddim_alphas_cumprod(num_train_timesteps)
    ratio = num_train_timesteps // num_inference_timesteps
    x = noise.clone()
    for k in ddim_timesteps(num_inference_timesteps, num_train_timesteps):
        eps = unet(sd, x, int(k), cond)
        a_t = ac[k]
        a_prev = ac[k - ratio] if k - ratio >= 0 else torch.tensor(1.0)            # set_alpha_to_one
        x0 = ((x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5).clamp(-1, 1)               # clip_sample
        x = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps                          # eta = 0
    return x
