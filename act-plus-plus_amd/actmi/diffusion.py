"""DiffusionPolicy inference on libactmi (reference policy.py:20-241; SURVEY 8 f2, BASELINE config 4).

Per camera: ResNet18 with GroupNorm (robomimic ResNet18Conv + replace_bn_with_gn) -> SpatialSoftmax(32 keypoints) ->
Linear(64, 64); obs_cond = cat(camera features, qpos); ConditionalUnet1D noise prediction under a DDIM schedule
(num_inference_timesteps steps, policy.py:102-109,209-223).  Every contraction runs on the library's MFMA GEMM /
implicit-GEMM convolution (actmi_op_gemm), GroupNorm / Mish / FiLM / SpatialSoftmax / im2col rows / the DDIM update on the
kernels of csrc/diffusion.hip; torch only holds the tensors (channel-last layouts: maps [cam][B][H][W][C], sequences
[B][T][C]).  The arithmetic of robomimic / diffusers is restated from their published definitions: PARITY UNPINNED
(oracle/diffusion_ref.py; neither package is importable offline).  Training of this policy is outside the accelerated path."""
from collections import OrderedDict
import math

import numpy as np
import torch

from . import ops
from . import weights as W


def diffusion_state_dict_spec(camera_names, action_dim=16, state_dim=14, down_dims=(256, 512, 1024), kernel_size=5, dsed=256,
                              num_kp=32, feature_dim=64):
    """key -> shape of `nets.state_dict()` of the reference module (policy.py:75-83), robomimic / torchvision attribute names."""
    out = OrderedDict()
    ncam = len(camera_names)
    for i in range(ncam):
        P = f"policy.backbones.{i}.nets."
        out[P + "0.weight"] = (64, 3, 7, 7)
        out[P + "1.weight"] = (64,)
        out[P + "1.bias"] = (64,)
        cin = 64
        for li in range(1, 5):
            cout = 64 << (li - 1)
            for bi in range(2):
                q = f"{P}{3 + li}.{bi}."
                out[q + "conv1.weight"] = (cout, cin, 3, 3)
                out[q + "bn1.weight"] = (cout,); out[q + "bn1.bias"] = (cout,)
                out[q + "conv2.weight"] = (cout, cout, 3, 3)
                out[q + "bn2.weight"] = (cout,); out[q + "bn2.bias"] = (cout,)
                if bi == 0 and li > 1:
                    out[q + "downsample.0.weight"] = (cout, cin, 1, 1)
                    out[q + "downsample.1.weight"] = (cout,); out[q + "downsample.1.bias"] = (cout,)
                cin = cout
    for i in range(ncam):
        out[f"policy.pools.{i}.nets.weight"] = (num_kp, 512, 1, 1)
        out[f"policy.pools.{i}.nets.bias"] = (num_kp,)
    for i in range(ncam):
        out[f"policy.linears.{i}.weight"] = (feature_dim, num_kp * 2)
        out[f"policy.linears.{i}.bias"] = (feature_dim,)
    P = "policy.noise_pred_net."
    cond_dim = dsed + feature_dim * ncam + state_dim
    out[P + "diffusion_step_encoder.1.weight"] = (dsed * 4, dsed); out[P + "diffusion_step_encoder.1.bias"] = (dsed * 4,)
    out[P + "diffusion_step_encoder.3.weight"] = (dsed, dsed * 4); out[P + "diffusion_step_encoder.3.bias"] = (dsed,)

    def crb(p, cin, cout):
        for bi, ci in ((0, cin), (1, cout)):
            out[f"{p}blocks.{bi}.block.0.weight"] = (cout, ci, kernel_size); out[f"{p}blocks.{bi}.block.0.bias"] = (cout,)
            out[f"{p}blocks.{bi}.block.1.weight"] = (cout,); out[f"{p}blocks.{bi}.block.1.bias"] = (cout,)
        out[p + "cond_encoder.1.weight"] = (2 * cout, cond_dim); out[p + "cond_encoder.1.bias"] = (2 * cout,)
        if cin != cout:
            out[p + "residual_conv.weight"] = (cout, cin, 1); out[p + "residual_conv.bias"] = (cout,)
    dims = [action_dim] + list(down_dims)
    n = len(down_dims)
    for i in range(n):
        crb(f"{P}down_modules.{i}.0.", dims[i], dims[i + 1])
        crb(f"{P}down_modules.{i}.1.", dims[i + 1], dims[i + 1])
        if i < n - 1:
            out[f"{P}down_modules.{i}.2.conv.weight"] = (dims[i + 1], dims[i + 1], 3); out[f"{P}down_modules.{i}.2.conv.bias"] = (dims[i + 1],)
    for i in range(2):
        crb(f"{P}mid_modules.{i}.", dims[-1], dims[-1])
    for i, (din, dout) in enumerate(reversed(list(zip(dims[1:-1], dims[2:])))):
        crb(f"{P}up_modules.{i}.0.", dout * 2, din)
        crb(f"{P}up_modules.{i}.1.", din, din)
        out[f"{P}up_modules.{i}.2.conv.weight"] = (din, din, 4); out[f"{P}up_modules.{i}.2.conv.bias"] = (din,)
    out[P + "final_conv.0.block.0.weight"] = (dims[1], dims[1], kernel_size); out[P + "final_conv.0.block.0.bias"] = (dims[1],)
    out[P + "final_conv.0.block.1.weight"] = (dims[1],); out[P + "final_conv.0.block.1.bias"] = (dims[1],)
    out[P + "final_conv.1.weight"] = (action_dim, dims[1], 1); out[P + "final_conv.1.bias"] = (action_dim,)
    return out


def generate_diffusion_state_dict(spec, seed=0):
    """Deterministic random init with the magnitudes of the torch defaults (kaiming-uniform-like for weights, GroupNorm
    gains near 1): counter based like weights.generate_state_dict, so the GPU box regenerates the same bytes."""
    sd = OrderedDict()
    for k, shape in spec.items():
        n = int(np.prod(shape))
        if k.endswith("weight") and len(shape) == 1:                           # GroupNorm gain
            v = 1.0 + 0.1 * W.normal(seed, k, n)
        elif k.endswith("bias"):
            v = 0.05 * W.normal(seed, k, n)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = W.normal(seed, k, n) * (1.0 / math.sqrt(fan_in))
        sd[k] = v.reshape(shape).astype(np.float32)
    return sd


def _alphas_cumprod(num_train_timesteps):
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = np.array([min(1 - alpha_bar((i + 1) / num_train_timesteps) / alpha_bar(i / num_train_timesteps), 0.999)
                      for i in range(num_train_timesteps)], dtype=np.float32)          # diffusers keeps float32
    return np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)


def _on_own_device(fn):
    """Run a method with the net's device current: every ops.* wrapper launches on torch's CURRENT stream, which must be a
    stream of the device the tensors live on (a process may hold nets on several GPUs)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with torch.cuda.device(self.dev):
            for t in list(a) + list(k.values()):
                if torch.is_tensor(t) and t.is_cuda and t.device != self.dev:
                    raise ValueError(f"tensor on {t.device} given to a DiffusionNet bound to {self.dev}")
            return fn(self, *a, **k)
    return wrapped


class DiffusionNet:
    """Weights prepared once (OHWI convolution weights, unfolded Conv1d matrices, per-camera stacks) + the inference graph."""

    def __init__(self, camera_names, action_dim=16, state_dim=14, prediction_horizon=32, num_inference_timesteps=10,
                 num_train_timesteps=50, device="cuda:0", gemm_prec="f16x3"):
        if prediction_horizon % 4:
            raise ValueError("prediction_horizon must be a multiple of 4 (two stride-2 stages of the UNet)")
        if not torch.cuda.is_available():
            raise RuntimeError("DiffusionNet needs an MI355X; no CPU fallback exists")
        self.cams, self.A, self.S, self.T = list(camera_names), action_dim, state_dim, prediction_horizon
        self.steps, self.train_steps = num_inference_timesteps, num_train_timesteps
        if num_train_timesteps % num_inference_timesteps:
            raise ValueError("num_inference_timesteps must divide the scheduler's num_train_timesteps")
        self.dev, self.prec = torch.device(device), gemm_prec
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        # f16x3 (fp32 products from exactly split fp16 pieces): weights of 1e-2 magnitude get a 2^8 pre-scale so that
        # their lo pieces stay normal fp16 numbers; gemm_prec="f32" runs the native fp32 matrix instruction
        self.bs = 256.0 if gemm_prec == "f16x3" else 0.0
        self.spec = diffusion_state_dict_spec(self.cams, action_dim, state_dim)
        self.ac = _alphas_cumprod(num_train_timesteps)
        self.w = None

    # ---- weights ----------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.spec if k not in sd]
        extra = [k for k in sd if k not in self.spec and not k.split(".")[-1] in ("pos_x", "pos_y", "temperature")]
        if strict and (missing or extra):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:4]} unexpected {extra[:4]}")
        t = {}
        for k, shape in self.spec.items():
            if k in sd:
                v = torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k]).to(torch.float32)
                if tuple(v.shape) != tuple(shape):
                    raise RuntimeError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
                t[k] = v
        self.raw = t
        d = self.dev
        ncam = len(self.cams)
        w = {}

        def stack_conv(suffix):
            # OIHW per camera -> [cam][O][H][W][I] (I padded to a multiple of 4 for the stem)
            ws = []
            for i in range(ncam):
                x = t[f"policy.backbones.{i}.nets.{suffix}"].permute(0, 2, 3, 1)
                if x.shape[-1] % 4:
                    x = torch.nn.functional.pad(x, (0, 4 - x.shape[-1] % 4))
                ws.append(x)
            return torch.stack(ws).contiguous().to(d)

        def stack_vec(suffix):
            return torch.stack([t[f"policy.backbones.{i}.nets.{suffix}"] for i in range(ncam)]).contiguous().to(d)
        w["stem"] = stack_conv("0.weight")
        if self.prec == "f16x3":
            # u8 frames: the stem on the ACT path's 7x7 kernel (its lookup table holds v / 255 here, no ImageNet normalisation,
            # and no ReLU: GroupNorm comes first) instead of a K = 196 implicit GEMM with 64 output channels
            w["stem_ws"] = ops.conv1_prepare(torch.stack([t[f"policy.backbones.{i}.nets.0.weight"] for i in range(ncam)]).to(d), lut_mode=1)
        w["stem_gn"] = (stack_vec("1.weight"), stack_vec("1.bias"))
        for li in range(1, 5):
            for bi in range(2):
                q = f"{3 + li}.{bi}."
                blk = {"c1": stack_conv(q + "conv1.weight"), "g1": (stack_vec(q + "bn1.weight"), stack_vec(q + "bn1.bias")),
                       "c2": stack_conv(q + "conv2.weight"), "g2": (stack_vec(q + "bn2.weight"), stack_vec(q + "bn2.bias"))}
                if bi == 0 and li > 1:
                    blk["ds"] = stack_conv(q + "downsample.0.weight")
                    blk["gd"] = (stack_vec(q + "downsample.1.weight"), stack_vec(q + "downsample.1.bias"))
                if li == 1 and self.prec == "f16x3":
                    # layer1 (64 -> 64 channels) runs on the direct convolution kernel of the ACT trunk (conv3.hip: the
                    # implicit GEMM is L2-traffic bound at 64 output channels); its split weight images are built once here
                    blk["c1_16"], blk["c2_16"] = ops.split16(blk["c1"], 256.0), ops.split16(blk["c2"], 256.0)
                    blk["ones"] = torch.ones(ncam, 64, device=d)
                    blk["zeros"] = torch.zeros(ncam, 64, device=d)
                w[f"l{li}b{bi}"] = blk
        w["kp"] = torch.stack([t[f"policy.pools.{i}.nets.weight"].permute(0, 2, 3, 1) for i in range(ncam)]).contiguous().to(d)
        w["kp_b"] = torch.stack([t[f"policy.pools.{i}.nets.bias"] for i in range(ncam)]).contiguous().to(d)
        w["lin"] = [(t[f"policy.linears.{i}.weight"].contiguous().to(d), t[f"policy.linears.{i}.bias"].contiguous().to(d))
                    for i in range(ncam)]
        P = "policy.noise_pred_net."
        for k in self.spec:
            if not k.startswith(P):
                continue
            v = t[k]
            if v.dim() == 3:
                if ".2.conv.weight" in k and "up_modules" in k:
                    v = v.permute(1, 2, 0).reshape(v.shape[1], -1)           # ConvTranspose1d [in][out][k] -> [out][(j, in)]
                else:
                    v = v.permute(0, 2, 1).reshape(v.shape[0], -1)           # Conv1d [out][in][k] -> [out][(j, in)]
            if k.endswith("cond_encoder.1.weight") and v.shape[1] % 4:
                v = torch.nn.functional.pad(v, (0, 4 - v.shape[1] % 4))          # the GEMM wants K % 4 == 0: zero columns
            w[k[len(P):]] = v.contiguous().to(d)
        self.w = w
        return missing, extra

    def state_dict(self):
        """The reference module's `nets.state_dict()`: the learned tensors plus the three buffers robomimic's SpatialSoftmax
        registers per camera (temperature [1], pos_x / pos_y [1, 15*20] on the [-1, 1] grid of its fixed input_shape
        [512, 15, 20], policy.py:46) -- the reference's strict load_state_dict wants them."""
        out = OrderedDict()
        px, py = np.meshgrid(np.linspace(-1.0, 1.0, 20), np.linspace(-1.0, 1.0, 15))
        for k, v in self.raw.items():
            out[k] = v.clone()
            if k.startswith("policy.pools.") and k.endswith(".nets.bias"):
                base = k[:-len("nets.bias")]
                out[base + "temperature"] = torch.ones(1)
                out[base + "pos_x"] = torch.from_numpy(px.reshape(1, -1)).float()
                out[base + "pos_y"] = torch.from_numpy(py.reshape(1, -1)).float()
        return out

    # ---- pieces -----------------------------------------------------------------------------------------------------
    def _gn_maps(self, x, gn, res=None, relu=True):
        """GroupNorm(C // 16) per camera (each camera has its own gain / bias) on [cam][B][H][W][C]."""
        out = torch.empty_like(x)
        for c in range(x.shape[0]):
            ops.groupnorm(x[c], gn[0][c], gn[1][c], x.shape[-1] // 16, act="relu" if relu else None,
                          res=None if res is None else res[c], out=out[c])
        return out

    def _conv(self, x, w, stride, pad):
        return ops.conv2d_nhwc(x, w, stride=stride, pad=pad, prec=self.prec, b_scale=self.bs)

    @_on_own_device
    def obs_cond(self, qpos, image_u8):
        w = self.w
        if image_u8.dtype == torch.uint8 and "stem_ws" in w:
            x = ops.conv1_prepared(image_u8, w["stem_ws"], 64, relu=False)
        else:
            if image_u8.dtype == torch.uint8:
                x = ops.u8_to_nhwc4(image_u8)                             # [cam][B][H][W][4] in [0,1]: no ImageNet normalisation here
            else:                                                         # f32 [B][cam][3][H][W] in [0,1] (the reference's contract)
                x = torch.nn.functional.pad(image_u8.to(torch.float32).permute(1, 0, 3, 4, 2), (0, 1)).contiguous()
            x = self._conv(x, w["stem"], 2, 3)
        x = self._gn_maps(x, w["stem_gn"])
        ncam_, B_ = x.shape[0], x.shape[1]
        x = ops.maxpool3x3s2(x.reshape(ncam_ * B_, *x.shape[2:]))
        x = x.reshape(ncam_, B_, *x.shape[1:])
        for li in range(1, 5):
            for bi in range(2):
                blk = w[f"l{li}b{bi}"]
                s = 2 if (li > 1 and bi == 0) else 1
                if "c1_16" in blk:
                    y = self._gn_maps(ops.conv3x3_c64(x, blk["c1"], blk["ones"], blk["zeros"], w16=blk["c1_16"]), blk["g1"])
                    y = ops.conv3x3_c64(y, blk["c2"], blk["ones"], blk["zeros"], w16=blk["c2_16"])
                else:
                    y = self._gn_maps(self._conv(x, blk["c1"], s, 1), blk["g1"])
                    y = self._conv(y, blk["c2"], 1, 1)
                idt = x
                if "ds" in blk:
                    idt = self._gn_maps(self._conv(x, blk["ds"], s, 0), blk["gd"], relu=False)
                x = self._gn_maps(y, blk["g2"], res=idt)                   # relu(gn(conv2) + identity)
        ncam, B, H, Wd, _ = x.shape
        lg = ops.conv2d_nhwc(x, w["kp"], bias=w["kp_b"], stride=1, pad=0, prec=self.prec, b_scale=self.bs)          # [cam][B][H][W][K]
        feats = []
        for c in range(ncam):
            kp = ops.spatial_softmax(lg[c].reshape(B, H * Wd, -1), H, Wd)                           # [B][K][2]
            feats.append(ops.gemm(kp.reshape(B, -1), w["lin"][c][0], bias=w["lin"][c][1], prec=self.prec, b_scale=self.bs))
        return torch.cat(feats + [qpos.to(torch.float32)], dim=1).contiguous()

    def _conv1d(self, x, wk, bk, k, stride=1, pad=None, transposed=False):
        B, T, Cc = x.shape
        pad = k // 2 if pad is None else pad
        cols = ops.unfold1d(x, k, stride, pad, transposed)
        To = cols.shape[1]
        # (small grids with long contractions -- 256..1024 rows x K up to 5120 -- are split over K: the UNet pass is these)
        y = ops.gemm(cols.reshape(B * To, k * Cc), self.w[wk], bias=self.w[bk], prec=self.prec, b_scale=self.bs, splitk="auto")
        return y.reshape(B, To, -1)

    def _crb(self, p, x, gm, k=5):
        w = self.w
        B = x.shape[0]
        y = self._conv1d(x, p + "blocks.0.block.0.weight", p + "blocks.0.block.0.bias", k)
        emb = ops.gemm(gm, w[p + "cond_encoder.1.weight"], bias=w[p + "cond_encoder.1.bias"], prec=self.prec, b_scale=self.bs,
                       splitk="auto")     # [B][2*out]
        oc = y.shape[-1]
        y = ops.groupnorm(y, w[p + "blocks.0.block.1.weight"], w[p + "blocks.0.block.1.bias"], 8, act="mish",
                          film=(emb[:, :oc], emb[:, oc:]))
        z = self._conv1d(y, p + "blocks.1.block.0.weight", p + "blocks.1.block.0.bias", k)
        res = self._conv1d(x, p + "residual_conv.weight", p + "residual_conv.bias", 1) if (p + "residual_conv.weight") in w else x
        return ops.groupnorm(z, w[p + "blocks.1.block.1.weight"], w[p + "blocks.1.block.1.bias"], 8, act="mish", res=res,
                             res_after=True)

    def _step_embedding(self, timestep, half):
        """SinusoidalPosEmb of a diffusion step as a device row [1, 2 * half]; cached per step, so a step seen before costs no
        host-to-device copy (none at all inside a captured graph: capture_infer warms every step of the schedule first)."""
        cache = self.__dict__.setdefault("_temb", {})
        key = (timestep, half)
        if key not in cache:
            e = np.exp(np.arange(half, dtype=np.float32) * np.float32(-(math.log(10000) / (half - 1))))
            arg = np.float32(timestep) * e
            cache[key] = torch.from_numpy(np.concatenate([np.sin(arg), np.cos(arg)]).astype(np.float32)).view(1, -1).to(self.dev)
        return cache[key]

    @_on_own_device
    def unet(self, sample, timestep, cond):
        """ConditionalUnet1D.forward on channel-last sequences: sample [B][T][A] -> noise prediction [B][T][A]."""
        w = self.w
        B = sample.shape[0]
        dsed = w["diffusion_step_encoder.3.weight"].shape[0]
        half = dsed // 2
        emb = self._step_embedding(int(timestep), half).repeat(B, 1)
        g = ops.gemm(emb, w["diffusion_step_encoder.1.weight"], bias=w["diffusion_step_encoder.1.bias"], prec=self.prec, b_scale=self.bs)
        g = ops.gemm(ops.mish(g), w["diffusion_step_encoder.3.weight"], bias=w["diffusion_step_encoder.3.bias"], prec=self.prec, b_scale=self.bs)
        gm = ops.mish(torch.cat([g, cond], dim=1).contiguous())            # every cond_encoder starts with the same Mish
        if gm.shape[1] % 4:
            gm = torch.nn.functional.pad(gm, (0, 4 - gm.shape[1] % 4)).contiguous()      # matches the zero-padded weight columns
        x, h = sample, []
        for i in range(3):
            x = self._crb(f"down_modules.{i}.0.", x, gm)
            x = self._crb(f"down_modules.{i}.1.", x, gm)
            h.append(x)
            if i < 2:
                x = self._conv1d(x, f"down_modules.{i}.2.conv.weight", f"down_modules.{i}.2.conv.bias", 3, stride=2, pad=1)
        for i in range(2):
            x = self._crb(f"mid_modules.{i}.", x, gm)
        for i in range(2):
            x = torch.cat((x, h.pop()), dim=2).contiguous()
            x = self._crb(f"up_modules.{i}.0.", x, gm)
            x = self._crb(f"up_modules.{i}.1.", x, gm)
            x = self._conv1d(x, f"up_modules.{i}.2.conv.weight", f"up_modules.{i}.2.conv.bias", 4, stride=2, pad=1, transposed=True)
        y = self._conv1d(x, "final_conv.0.block.0.weight", "final_conv.0.block.0.bias", 5)
        y = ops.groupnorm(y, w["final_conv.0.block.1.weight"], w["final_conv.0.block.1.bias"], 8, act="mish")
        return self._conv1d(y, "final_conv.1.weight", "final_conv.1.bias", 1)

    @_on_own_device
    def forward_infer(self, qpos, image_u8, noise=None):
        """DiffusionPolicy.__call__(qpos, image) (policy.py:177-223): [B][Tp][A] action sequence.  `noise` replaces the
        torch.randn start (policy.py:203-205) for reproducible runs."""
        if self.w is None:
            raise RuntimeError("load_state_dict first")
        B = qpos.shape[0]
        cond = self.obs_cond(qpos, image_u8)
        x = (torch.randn((B, self.T, self.A), device=self.dev) if noise is None else noise.to(self.dev, torch.float32)).contiguous().clone()
        ratio = self.train_steps // self.steps
        for k in (np.arange(0, self.steps) * ratio)[::-1]:
            eps = self.unet(x, int(k), cond).contiguous()
            a_t = float(self.ac[k])
            a_prev = float(self.ac[k - ratio]) if k - ratio >= 0 else 1.0
            ops.ddim_step(x, eps, a_t, a_prev, clip=True)
        return x

    @_on_own_device
    def capture_infer(self, batch, image_like):
        """Capture one whole policy query (observation features + every DDIM step) into a hipGraph and return
        ``replay(qpos, image, noise=None) -> actions``.  Eager, the query is ~1.3k op-level launches issued from Python (the
        ten U-Net passes alone ~100 small launches each): the launch overhead, not the kernels, sets its time.  `image_like`
        fixes the image format of the captured graph (u8 [B, cams, H, W, 3] or f32 [B, cams, 3, H, W])."""
        if self.w is None:
            raise RuntimeError("load_state_dict first")
        dev = self.dev
        s_qpos = torch.zeros((batch, self.S), dtype=torch.float32, device=dev)
        s_img = torch.zeros((batch,) + tuple(image_like.shape[1:]), dtype=image_like.dtype, device=dev)
        s_noise = torch.zeros((batch, self.T, self.A), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                     # warm-up: function attributes, step embeddings, workspaces
            self.forward_infer(s_qpos, s_img, noise=s_noise)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            s_out = self.forward_infer(s_qpos, s_img, noise=s_noise)

        def replay(qpos, image, noise=None):
            s_qpos.copy_(qpos, non_blocking=True)
            s_img.copy_(image, non_blocking=True)
            if noise is None:
                s_noise.normal_()                         # the Gaussian start of policy.py:203-205, drawn outside the graph
            else:
                s_noise.copy_(noise, non_blocking=True)
            graph.replay()
            return s_out

        replay.graph = graph
        replay.static = (s_qpos, s_img, s_noise, s_out)
        return replay
