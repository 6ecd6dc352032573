"""Python handle of one libactmi context: owns nothing but the C handle; tensors stay torch-owned.

Mirrors what ``build_ACT_model_and_optimizer`` returns in the reference (detr/main.py:92-112): a model object with
``num_queries`` / ``encoder`` attributes that ``imitate_episodes.py`` reads, loadable from / dumpable to the
reference's state_dict.
"""
from collections import OrderedDict
import ctypes as C
import os

import numpy as np
import time

import torch

from . import lib as L
from .config import ACTConfig
from .weights import act_state_dict_spec, is_buffer


# ACTMI_CHECK_FINITE=1: verify every inference output (one host sync per call; off by default)
_CHECK_FINITE = os.environ.get("ACTMI_CHECK_FINITE") == "1"


class ACTEngine:
    def __init__(self, cfg: ACTConfig, max_batch: int = 8, device: str = "cuda:0", training: bool = False,
                 gemm_prec: str = None, train_prec: str = None):
        """gemm_prec: "f16x3" (default; fp32 products formed from three fp16 MFMA products of exactly split operands,
        fp32-grade results) or "f32" (native fp32 MFMA); None = environment ACTMI_GEMM_PREC or the default."""
        if not torch.cuda.is_available():
            raise RuntimeError("ACTEngine needs an MI355X (torch.cuda.is_available() is False); no CPU fallback exists")
        self.cfg = cfg.validate()
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        if self.device.type != "cuda":
            raise ValueError(f"ACTEngine device must be a cuda device, got {device!r}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = L.load()
        c = L.ActmiConfig(struct_size=C.sizeof(L.ActmiConfig), num_cams=cfg.num_cams, image_h=cfg.image_h, image_w=cfg.image_w, base_width=cfg.base_width,
                          hidden_dim=cfg.hidden_dim, nheads=cfg.nheads, dim_feedforward=cfg.dim_feedforward,
                          enc_layers=cfg.enc_layers, dec_layers=cfg.dec_layers, num_queries=cfg.num_queries,
                          state_dim=cfg.state_dim, action_dim=cfg.action_dim, latent_dim=cfg.latent_dim,
                          has_cvae_encoder=0 if cfg.no_encoder else 1, max_batch=self.max_batch,
                          enable_training=1 if training else 0, kl_weight=float(cfg.kl_weight),
                          vq=1 if cfg.vq else 0, vq_class=int(cfg.vq_class or 0), vq_dim=int(cfg.vq_dim or 0))
        h = C.c_void_p()
        # the handle binds to the device that is current at create time; the process-wide current device is left alone
        # (one process may hold engines on several GPUs; every later call switches to the handle's device by itself)
        with torch.cuda.device(self.device):
            rc = self.lib.actmi_create(C.byref(c), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"actmi_create failed ({rc}): {self.lib.actmi_last_error(None).decode()}")
        self.h = h
        if gemm_prec is not None:
            L.check(self.lib.actmi_set_gemm_prec(self.h, {"f32": 1, "f16x3": 2}[gemm_prec]), self.h, "set_gemm_prec")
        if train_prec is not None:
            # "bf16": the GEMMs of the training step form ONE bf16 product per fp32 product (fp32 accumulate, fp32 master weights
            # and optimizer state) -- BASELINE config 3's bf16, an opt-in speed mode (~1e-2 relative); None / "f16x3" = default
            L.check(self.lib.actmi_set_train_prec(self.h, {"f32": 1, "f16x3": 2, "bf16": 3}[train_prec]), self.h, "set_train_prec")
        self.spec = act_state_dict_spec(cfg)
        self._finalized = False
        # attributes imitate_episodes.py touches on policy.model
        self.num_queries = cfg.num_queries
        self.encoder = None if cfg.no_encoder else True

    def _sp(self):
        """The caller's current stream ON THE ENGINE'S DEVICE (never another device's stream)."""
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_dev(self, **tensors):
        for name, t in tensors.items():
            if t is not None and t.device != self.device:
                raise ValueError(f"{name} lives on {t.device} but this engine is bound to {self.device}")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.actmi_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------------
    def load_state_dict(self, sd, prefix: str = "", strict: bool = True):
        """sd: {key: numpy array | torch tensor}. Returns (missing_keys, unexpected_keys) like nn.Module."""
        missing, unexpected = [], []
        for k in self.spec:
            if prefix + k not in sd:
                missing.append(prefix + k)
        for k in sd:
            kk = k[len(prefix):] if k.startswith(prefix) else None
            if kk is None or kk not in self.spec:
                if not (kk and kk.endswith("num_batches_tracked")):      # dropped by FrozenBatchNorm2d, backbone.py:37-41
                    unexpected.append(k)
        if strict and (missing or unexpected):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        # actmi_set_param / actmi_get_param copy with blocking hipMemcpy on the null stream, which is not ordered with work in
        # flight on torch's (non-blocking) streams or the library's branch streams: drain the device first
        torch.cuda.synchronize(self.device)
        for k, shape in self.spec.items():
            if prefix + k not in sd:
                continue
            v = sd[prefix + k]
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(np.ascontiguousarray(v))
            v = v.detach().to(torch.float32).contiguous()
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
            shp = (C.c_int64 * len(shape))(*shape)
            is_dev = 1 if v.is_cuda else 0
            L.check(self.lib.actmi_set_param(self.h, k.encode(), C.c_void_p(v.data_ptr()), shp, len(shape), is_dev),
                    self.h, f"set_param({k})")
        self._finalized = False
        return missing, unexpected

    def state_dict(self, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        torch.cuda.synchronize(self.device)          # (see load_state_dict: an optimizer step may still be running)
        for k, shape in self.spec.items():
            t = torch.empty(shape, dtype=torch.float32)
            L.check(self.lib.actmi_get_param(self.h, k.encode(), C.c_void_p(t.data_ptr()), t.numel() * 4, 0), self.h,
                    f"get_param({k})")
            out[prefix + k] = t
        return out

    def parameters(self):
        """nn.Module.parameters() of the reference model: one read-only float32 VIEW (no copy) per learnable state_dict entry,
        in registration order (FrozenBatchNorm2d statistics are buffers, backbone.py:30-35, and are not listed)."""
        for k, shape in self.spec.items():
            if is_buffer(k):
                continue
            p, n = C.c_void_p(), C.c_int64()
            L.check(self.lib.actmi_param_ptr(self.h, k.encode(), C.byref(p), C.byref(n)), self.h, f"param_ptr({k})")
            yield _from_ptr(p.value, n.value, self.device).view(shape)

    def finalize(self):
        L.check(self.lib.actmi_finalize(self.h, self._sp()), self.h, "finalize")
        self._finalized = True

    # ---- forward ------------------------------------------------------------------------------
    def _image_fmt(self, image: torch.Tensor, B: int):
        cfg = self.cfg
        if image.dtype == torch.uint8:
            want = (B, cfg.num_cams, cfg.image_h, cfg.image_w, 3)
            fmt = L.IMG_U8_NHWC
        elif image.dtype == torch.float32:
            want = (B, cfg.num_cams, 3, cfg.image_h, cfg.image_w)
            fmt = L.IMG_F32_NCHW
        else:
            raise TypeError(f"image dtype {image.dtype} not supported (uint8 NHWC or float32 NCHW)")
        if tuple(image.shape) != want:
            raise ValueError(f"image shape {tuple(image.shape)} != {want}")
        return fmt

    def forward_infer(self, qpos: torch.Tensor, image: torch.Tensor, out: torch.Tensor = None,
                      vq_sample: torch.Tensor = None) -> torch.Tensor:
        if not self._finalized:
            self.finalize()
        cfg = self.cfg
        B = qpos.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        if not (qpos.is_cuda and image.is_cuda):
            raise ValueError("qpos and image must be CUDA tensors on the engine's device")
        self._check_dev(qpos=qpos, image=image, out=out)
        qpos = qpos.to(torch.float32).contiguous()
        image = image.contiguous()
        fmt = self._image_fmt(image, B)
        if tuple(qpos.shape) != (B, cfg.state_dim):
            raise ValueError(f"qpos shape {tuple(qpos.shape)} != {(B, cfg.state_dim)}")
        if out is None:
            out = torch.empty((B, cfg.num_queries, cfg.action_dim), dtype=torch.float32, device=qpos.device)
        if cfg.vq:
            # VQ-ACT: the latent code comes from the caller (the latent prior model's sample, imitate_episodes.py:393-394)
            if vq_sample is None:
                raise ValueError("a vq policy needs vq_sample [B, vq_class, vq_dim] (reference policy.py:322-332)")
            code = vq_sample.to(device=qpos.device, dtype=torch.float32).reshape(B, cfg.vq_class * cfg.vq_dim).contiguous()
            L.check(self.lib.actmi_forward_infer_vq(self.h, C.c_void_p(qpos.data_ptr()), C.c_void_p(image.data_ptr()), fmt,
                                                    B, C.c_void_p(code.data_ptr()), C.c_void_p(out.data_ptr()),
                                                    self._sp()), self.h, "forward_infer_vq")
        else:
            L.check(self.lib.actmi_forward_infer(self.h, C.c_void_p(qpos.data_ptr()), C.c_void_p(image.data_ptr()), fmt, B,
                                                 C.c_void_p(out.data_ptr()), self._sp()), self.h,
                    "forward_infer")
        if _CHECK_FINITE and not bool(torch.isfinite(out).all()):
            # f16x3 needs finite operands with |x| < 65504: an activation beyond that shows up here
            raise FloatingPointError("non-finite a_hat: an operand left the fp16-split range; rerun with gemm_prec='f32' "
                                     "(ACTMI_GEMM_PREC=f32)")
        return out

    def set_forward_phase(self, phase: int):
        """0: forward_infer runs the whole step; 1: trunk + token assembly only; 2: transformer only (actmi_set_forward_phase)"""
        L.check(self.lib.actmi_set_forward_phase(self.h, int(phase)), self.h, "set_forward_phase")

    def capture_infer(self, batch: int, image_dtype=torch.uint8, with_ensemble=None, statics=None, phase: int = 0):
        """Capture one forward (optionally + the temporal-ensemble kernel) into a hipGraph and return
        ``replay(qpos, image) -> a_hat`` that copies into static inputs and replays.  The forward path allocates
        nothing and never synchronises, so the whole step is one graph launch (removes ~60 kernel-launch gaps; matters
        at small batch where the step is launch-bound)."""
        if not self._finalized:
            self.finalize()
        cfg, dev = self.cfg, self.device
        shape = (batch, cfg.num_cams, cfg.image_h, cfg.image_w, 3) if image_dtype == torch.uint8 else \
                (batch, cfg.num_cams, 3, cfg.image_h, cfg.image_w)
        if statics is not None:
            s_qpos, s_img, s_out = statics
        else:
            s_qpos = torch.zeros((batch, cfg.state_dim), dtype=torch.float32, device=dev)
            s_img = torch.zeros(shape, dtype=image_dtype, device=dev)
            s_out = torch.empty((batch, cfg.num_queries, cfg.action_dim), dtype=torch.float32, device=dev)
        # warm-up on a side stream (first launches set function attributes; not allowed during capture)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self.forward_infer(s_qpos, s_img, out=s_out)
                if with_ensemble is not None:
                    with_ensemble.step(s_out)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if with_ensemble is not None:
            with_ensemble.reset()
        graph = torch.cuda.CUDAGraph()
        ens_out = None
        self.set_forward_phase(phase)          # (InferPipeline: the trunk and the transformer as graphs of their own)
        try:
            # thread_local: calls made by other threads (e.g. the RCCL watchdog of a multi-rank bench) must not void the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self.forward_infer(s_qpos, s_img, out=s_out)
                if with_ensemble is not None and phase != 1:
                    ens_out = with_ensemble.step(s_out)
        finally:
            self.set_forward_phase(0)
        if with_ensemble is not None:
            with_ensemble.reset()               # the capture itself does not execute, but keep the state explicit

        def replay(qpos, image):
            # a caller that owns the step's inputs writes them straight into replay.static (H2D copies land there): no copy
            if qpos.data_ptr() != s_qpos.data_ptr():
                s_qpos.copy_(qpos, non_blocking=True)
            if image.data_ptr() != s_img.data_ptr():
                s_img.copy_(image, non_blocking=True)
            graph.replay()
            return (s_out, ens_out) if with_ensemble is not None else s_out

        replay.graph = graph
        replay.static = (s_qpos, s_img, s_out)
        return replay

    # ---- training -----------------------------------------------------------------------------
    def forward_train(self, qpos, image, actions, is_pad, eps=None, dropout_p: float = 0.0, dropout_seed: int = 0,
                      vq_code=None):
        """ACTPolicy.__call__ training branch (policy.py:288-320). Returns dict(l1, kl, loss, a_hat, mu, logvar) of
        device tensors.  ``eps`` replaces the normal_() draw of reparametrize (detr_vae.py:19-22).
        VQ-ACT (cfg.vq): ``vq_code`` [B, vq_class, vq_dim] replaces the multinomial draw of detr_vae.py:140 (None: drawn
        on the device from ``dropout_seed``); the dict then carries probs, binaries and vq_discrepancy instead of
        mu / logvar, and kl = 0 (policy.py:307-312)."""
        if not self._finalized:
            self.finalize()
        cfg = self.cfg
        B = qpos.shape[0]
        Q, A, Lz = cfg.num_queries, cfg.action_dim, cfg.latent_in_dim
        self._check_dev(qpos=qpos, image=image, actions=actions, is_pad=is_pad)
        qpos = qpos.to(torch.float32).contiguous()
        image = image.contiguous()
        fmt = self._image_fmt(image, B)
        actions = actions[:, :Q].to(torch.float32)                        # policy.py:289-290
        is_pad_u8 = is_pad[:, :Q].to(torch.uint8)
        if tuple(actions.shape[:2]) != tuple(is_pad_u8.shape[:2]) or actions.shape[0] != B or actions.shape[-1] != A:
            raise ValueError(f"actions {tuple(actions.shape)} / is_pad {tuple(is_pad_u8.shape)} do not match [B={B}, T, A={A}]")
        if actions.shape[1] < Q:
            # episodes shorter than the chunk (the reference's padded_action is max_episode_len long, utils.py:120-133, and
            # its l1 loss would fail on the shape): continue the dataset's own padding -- zero actions, is_pad = True -- up to
            # the Q steps the library reads (it used to read past the end of the shorter tensors)
            T_ = actions.shape[1]
            actions = torch.cat([actions, actions.new_zeros((B, Q - T_, A))], dim=1)
            is_pad_u8 = torch.cat([is_pad_u8, is_pad_u8.new_ones((B, Q - T_))], dim=1)
        actions, is_pad_u8 = actions.contiguous(), is_pad_u8.contiguous()
        dev = qpos.device
        if cfg.vq:
            eps = None if vq_code is None else vq_code.to(device=dev, dtype=torch.float32).reshape(B, Lz).contiguous()
        else:
            if eps is None:
                eps = torch.randn((B, Lz), dtype=torch.float32, device=dev)
            eps = eps.to(torch.float32).contiguous()
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        a_hat = torch.empty((B, Q, A), dtype=torch.float32, device=dev)
        mu = torch.empty((B, Lz), dtype=torch.float32, device=dev)
        logvar = torch.empty((B, Lz), dtype=torch.float32, device=dev)
        self._keep = (qpos, image, actions, is_pad_u8, eps)               # the library reads qpos again in backward
        L.check(self.lib.actmi_forward_train(
            self.h, C.c_void_p(qpos.data_ptr()), C.c_void_p(image.data_ptr()), fmt, C.c_void_p(actions.data_ptr()),
            C.c_void_p(is_pad_u8.data_ptr()), C.c_void_p(eps.data_ptr() if eps is not None else 0), C.c_uint64(dropout_seed),
            float(dropout_p), B, C.c_void_p(losses.data_ptr()), C.c_void_p(a_hat.data_ptr()), C.c_void_p(mu.data_ptr()),
            C.c_void_p(logvar.data_ptr()), self._sp()), self.h, "forward_train")
        self._last_losses = losses
        if os.environ.get("ACTMI_DEBUG_LOSS") == "1" and not bool(torch.isfinite(losses).all()):
            print("[actmi debug] non-finite losses", losses.tolist(), "B", B, "a_hat finite", bool(torch.isfinite(a_hat).all()),
                  "mu finite", bool(torch.isfinite(mu).all()), "logvar finite", bool(torch.isfinite(logvar).all()),
                  "logvar max", float(logvar.max()), "dropout", dropout_p, "actions finite", bool(torch.isfinite(actions).all()),
                  "qpos finite", bool(torch.isfinite(qpos).all()), "is_pad sum", is_pad_u8.sum(1).tolist(), flush=True)
        out = {"l1": losses[0], "kl": losses[1], "loss": losses[2], "a_hat": a_hat}
        if cfg.vq:
            probs, binaries = mu.view(B, cfg.vq_class, cfg.vq_dim), logvar.view(B, cfg.vq_class, cfg.vq_dim)
            out.update(probs=probs, binaries=binaries, mu=None, logvar=None,
                       vq_discrepancy=(probs - binaries).abs().mean())      # F.l1_loss(probs, binaries), policy.py:311-312
        else:
            out.update(mu=mu, logvar=logvar)
        return out

    def backward(self, loss_scale: float = 1.0):
        L.check(self.lib.actmi_backward(self.h, float(loss_scale), self._sp()), self.h, "backward")

    def zero_grad(self):
        L.check(self.lib.actmi_zero_grad(self.h, self._sp()), self.h, "zero_grad")

    def adamw_step(self, lr, lr_backbone, weight_decay=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
        L.check(self.lib.actmi_adamw_step(self.h, lr, lr_backbone, weight_decay, beta1, beta2, eps, int(step),
                                          self._sp()), self.h, "adamw_step")

    def grad_arena(self) -> torch.Tensor:
        """Flat float32 view (no copy) of the whole gradient arena, for data-parallel all-reduce."""
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.actmi_grad_arena(self.h, C.byref(p), C.byref(n)), self.h, "grad_arena")
        return _from_ptr(p.value, n.value, self.device)

    def allreduce_grads(self, group=None, bucket_mb: int = 64):
        """Sum the gradient arena over the data-parallel ranks in buckets (RCCL when the backend is nccl).  Call
        ``backward(loss_scale=1/world)`` first so that the sum is the mean over the global batch (the reference's loss
        is a per-sample mean, policy.py:314-318,386-387)."""
        from .dist_utils import allreduce_buckets
        allreduce_buckets(self.grad_arena(), bucket_mb * (1 << 20) // 4, group)
        self.sync_flags(group)

    def grad_phase_range(self, phase: int):
        off, cnt = C.c_int64(), C.c_int64()
        L.check(self.lib.actmi_grad_phase_range(self.h, int(phase), C.byref(off), C.byref(cnt)), self.h, "grad_phase_range")
        return off.value, cnt.value

    def backward_allreduce(self, loss_scale: float = 1.0, group=None, bucket_mb: int = 64):
        """loss.backward() + data-parallel gradient averaging with the collective OVERLAPPED with the backward: the library
        records an event once the transformer.* gradients (the head of the arena, ~45 % of it) are final; a side stream waits
        for it and all-reduces that range in buckets (RCCL when the backend is nccl) while the backbone / CVAE-encoder backward
        is still running on the compute stream; the remainder is reduced when the backward has drained.  Pass
        loss_scale = 1 / world so that the sum is the mean over the global batch."""
        import torch.distributed as dist
        from .dist_utils import allreduce_buckets
        self.backward(loss_scale)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        arena = self.grad_arena()
        lo, n = self.grad_phase_range(1)
        cur = torch.cuda.current_stream(self.device)
        if not hasattr(self, "_comm_stream"):
            self._comm_stream = torch.cuda.Stream(device=self.device)
        side = self._comm_stream
        L.check(self.lib.actmi_wait_grad_phase(self.h, 1, C.c_void_p(side.cuda_stream)), self.h, "wait_grad_phase")
        with torch.cuda.stream(side):
            allreduce_buckets(arena[lo:lo + n], bucket_mb * (1 << 20) // 4, group)       # under the rest of the backward
        allreduce_buckets(arena[lo + n:], bucket_mb * (1 << 20) // 4, group)             # after the backward has drained
        cur.wait_stream(side)
        self.sync_flags(group)

    # ---- sharded optimizer (reduce-scatter -> sharded fused AdamW -> all-gather of the parameters) ---------
    def param_arena(self) -> torch.Tensor:
        """Flat float32 view (no copy) of the fp32 parameter arena (same layout as grad_arena)."""
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.actmi_param_arena(self.h, C.byref(p), C.byref(n)), self.h, "param_arena")
        return _from_ptr(p.value, n.value, self.device)

    def _shard_plan(self, world: int, bucket_elems: int):
        """Buckets [lo, hi) of the arena whose size is a multiple of 64 * world floats (the optimizer works on 64-float slots):
        rank r owns the r-th of `world` equal slices of every bucket.  The arena's tail that does not fill such a unit
        (< 64 * world floats) is all-reduced and updated by every rank.  Buckets never straddle the phase-1 boundary (the
        transformer gradients, final before the backbone backward runs)."""
        n = self.grad_arena().numel()
        lo1, n1 = self.grad_phase_range(1)
        unit = 64 * world
        bsz = max(unit, bucket_elems // unit * unit)
        buckets, tails = [], []
        for a, b in ((lo1, lo1 + n1), (lo1 + n1, n)):
            main = a + (b - a) // unit * unit
            x = a
            while x < main:
                y = min(main, x + bsz)
                buckets.append((x, y, 1 if b <= lo1 + n1 else 2))
                x = y
            if main < b:
                tails.append((main, b))
        return buckets, tails

    def backward_reduce_scatter(self, loss_scale: float = 1.0, group=None, bucket_mb: int = 64, comm_dtype=None):
        """loss.backward() + the gradient half of the sharded data-parallel step: every bucket of the gradient arena is
        REDUCE-SCATTERED (RCCL over xGMI when the backend is nccl) so that each rank ends up with the summed gradients of
        the slices it owns -- half the bytes of an all-reduce; the buckets of the transformer range go out from a side stream
        while the backbone backward still runs (actmi_wait_grad_phase), the rest when the backward has drained.
        comm_dtype=torch.bfloat16 sends the buckets as bf16 (half the link bytes again; the sum is formed in bf16, so this is
        an opt-in speed mode, never the default).  After this call only the owned slices of grad_arena() hold gradients of
        the global batch.  Follow with adamw_step_sharded()."""
        import torch.distributed as dist
        from .dist_utils import reduce_scatter_flat
        self.backward(loss_scale)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            self._shard = None
            return
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        buckets, tails = self._shard_plan(world, bucket_mb * (1 << 20) // 4)
        arena = self.grad_arena()
        on_gpu = arena.is_cuda                 # (a CPU stand-in of the engine drives the same plan in the gloo tests)
        if on_gpu:
            cur = torch.cuda.current_stream(self.device)
            if not hasattr(self, "_comm_stream"):
                self._comm_stream = torch.cuda.Stream(device=self.device)
            side = self._comm_stream

        def rs(lo, hi):
            k = (hi - lo) // world
            own = arena[lo + rank * k: lo + (rank + 1) * k]
            if comm_dtype is None:
                return reduce_scatter_flat(own, arena[lo:hi], group), None
            src = arena[lo:hi].to(comm_dtype)
            dst = torch.empty(k, dtype=comm_dtype, device=arena.device)
            return reduce_scatter_flat(dst, src, group), (own, dst, src)

        def finish(works):
            for w, cast in works:
                w.wait()
                if cast is not None:
                    cast[0].copy_(cast[1])
        if on_gpu:
            L.check(self.lib.actmi_wait_grad_phase(self.h, 1, C.c_void_p(side.cuda_stream)), self.h, "wait_grad_phase")
            with torch.cuda.stream(side):
                finish([rs(lo, hi) for lo, hi, ph in buckets if ph == 1])        # under the rest of the backward
        else:
            finish([rs(lo, hi) for lo, hi, ph in buckets if ph == 1])
        finish([rs(lo, hi) for lo, hi, ph in buckets if ph == 2])
        for lo, hi in tails:
            dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM, group=group)
        if on_gpu:
            cur.wait_stream(side)
        self.sync_flags(group)
        self._shard = (world, rank, buckets, tails, group)

    def adamw_step_sharded(self, lr, lr_backbone, weight_decay=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
        """The optimizer half: fused AdamW on the slices this rank owns (1/world of the arena: its Adam moments are the only
        ones ever touched), then the updated parameter slices are ALL-GATHERED bucket by bucket and the derived weights
        rebuilt.  Without a process group (or before backward_reduce_scatter) this is adamw_step."""
        from .dist_utils import all_gather_flat
        sh = getattr(self, "_shard", None)
        if sh is None:
            return self.adamw_step(lr, lr_backbone, weight_decay, beta1, beta2, eps, step)
        world, rank, buckets, tails, group = sh
        params = self.param_arena()
        sp = self._sp()

        def upd(lo, cnt):
            L.check(self.lib.actmi_adamw_step_range(self.h, lr, lr_backbone, weight_decay, beta1, beta2, eps, int(step), int(lo),
                                                    int(cnt), sp), self.h, "adamw_step_range")
        works = []
        for lo, hi, _ in buckets:
            k = (hi - lo) // world
            upd(lo + rank * k, k)
            works.append(all_gather_flat(params[lo:hi], params[lo + rank * k: lo + (rank + 1) * k], group))
        for lo, hi in tails:
            upd(lo, hi - lo)                           # all-reduced gradients: every rank computes the same update
        for w in works:
            w.wait()
        L.check(self.lib.actmi_refresh_weights(self.h, sp), self.h, "refresh_weights")
        self._shard = None

    def flags_tensor(self) -> torch.Tensor:
        """int32 view (no copy) of the handle's device flag word."""
        p = C.c_void_p()
        L.check(self.lib.actmi_flags_ptr(self.h, C.byref(p)), self.h, "flags_ptr")
        return _from_ptr(p.value, 1, self.device, typestr="<i4")

    def sync_flags(self, group=None):
        """OR the flag word over the data-parallel ranks (no host sync): the AdamW update is gated on it on the device, and
        every rank must skip the same steps or the replicas diverge.  NCCL has no bitwise reduction: the three bits travel as
        three integers under MAX."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        f = self.flags_tensor()
        shifts = torch.arange(3, dtype=torch.int32, device=f.device)
        bits = (f >> shifts) & 1
        dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=group)
        f.copy_((bits << shifts).sum(dtype=torch.int32).reshape(1) | f)

    def grad(self, key: str) -> torch.Tensor:
        """Copy of the gradient of one state_dict entry (shape of the parameter)."""
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.actmi_grad_ptr(self.h, key.encode(), C.byref(p), C.byref(n)), self.h, f"grad_ptr({key})")
        torch.cuda.synchronize(self.device)
        return _from_ptr(p.value, n.value, self.device).clone().view(self.spec[key])

    # ---- range / finiteness guard -----------------------------------------------------------------
    FLAG_OUTPUT, FLAG_WEIGHT, FLAG_LOSS = 1, 2, 4

    def read_flags(self, clear: bool = True) -> int:
        """The handle's device flag word (synchronises the current stream: call where the host waits anyway, e.g. right
        after the actions of a step were copied to the host)."""
        f = C.c_uint32(0)
        L.check(self.lib.actmi_get_flags(self.h, C.byref(f), 1 if clear else 0, self._sp()), self.h, "get_flags")
        return int(f.value)

    def check_flags(self):
        """Raise FloatingPointError when a kernel of this handle reported a non-finite output / loss or a weight beyond the
        range of its split image since the last check (default-on guard of the f16x3 arithmetic)."""
        f = self.read_flags(clear=True)
        if f:
            what = [n for b, n in ((1, "an inference output was not finite (an operand left the fp16-split range |x| < 65504)"),
                                   (2, "a weight has outgrown the scale of its split image: call finalize() again"),
                                   (4, "a training loss was not finite")) if f & b]
            last = getattr(self, "_last_losses", None)
            if f & 4 and last is not None:
                what.append(f"last [l1, kl, loss] = {last.tolist()}")
            raise FloatingPointError("libactmi range guard: " + "; ".join(what) + " -- gemm_prec='f32' (ACTMI_GEMM_PREC=f32) "
                                     "runs the same step on the native fp32 MFMA")

    # ---- debug --------------------------------------------------------------------------------
    def debug_stop_after(self, stage: str):
        L.check(self.lib.actmi_debug_stop_after(self.h, (stage or "").encode()), self.h, "debug_stop_after")

    def debug_tensor(self, name: str) -> torch.Tensor:
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.actmi_debug_tensor(self.h, name.encode(), C.byref(p), C.byref(n)), self.h, f"debug_tensor({name})")
        t = torch.empty(n.value, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        src = _from_ptr(p.value, n.value, self.device)     # wrap the raw pointer, then D2D copy through torch
        t.copy_(src)
        torch.cuda.synchronize(self.device)
        return t


class InferPipeline:
    """Double-buffered policy queries fed from the HOST: the copy of frame t + 1 runs beside the TRANSFORMER of step t.

    Two sets of device input buffers ("slots"), each with two captured graphs: the trunk (the only reader of the frame, and the
    HBM-bound part of the step) and the transformer + ensemble (actmi_set_forward_phase).  Between the two graphs of step t the
    stream records an ordinary event; the copy stream waits for it, copies the next frame into the OTHER slot and records
    `ev_copy`, which the trunk graph of step t + 1 waits for.  The 29.5 MB copy (0.53 ms alone) thereby overlaps the compute-bound
    transformer: released beside the START of the step -- the stem -- it slowed the step by 0.33-0.37 ms (tools/h2d_probe.py),
    and as a memcpy node inside one whole-step graph it did not overlap at all (+0.52 ms, profiles/r03_h2d_probe.json).
    A rollout loop that knows its next frame one step early (cameras running ahead of the policy, replayed episodes, the
    benchmark's with_h2d leg) pays ~max(copy, step) instead of copy + step.

        pipe = InferPipeline(engine, batch, with_ensemble=ens)
        pipe.feed(qpos_host[0], frames_host[0])            # pinned host tensors
        for t in range(T):
            a_hat, raw = pipe.step(next_inputs=(qpos_host[t + 1], frames_host[t + 1]) if t + 1 < T else None)

    No host synchronisation anywhere; the pinned host tensors of a feed must stay untouched until the step AFTER the one they were
    passed to has been issued and `pipe.copied(k)` has completed (or simply use one host buffer per step in flight).  Outputs of
    step t stay valid until step t + 2 is issued."""

    def __init__(self, engine: "ACTEngine", batch: int, with_ensemble=None, image_dtype=torch.uint8, copy_stream_candidates: int = 8):
        self.engine, self.dev = engine, engine.device
        cfg, dev = engine.cfg, engine.device
        shape = (batch, cfg.num_cams, cfg.image_h, cfg.image_w, 3) if image_dtype == torch.uint8 else \
                (batch, cfg.num_cams, 3, cfg.image_h, cfg.image_w)
        self.slots = []
        for _ in range(2):
            st = (torch.zeros((batch, cfg.state_dim), dtype=torch.float32, device=dev), torch.zeros(shape, dtype=image_dtype, device=dev),
                  torch.empty((batch, cfg.num_queries, cfg.action_dim), dtype=torch.float32, device=dev))
            trunk = engine.capture_infer(batch, image_dtype=image_dtype, statics=st, phase=1)
            rest = engine.capture_infer(batch, image_dtype=image_dtype, with_ensemble=with_ensemble, statics=st, phase=2)
            self.slots.append((st, trunk, rest))
        self.ev_copy = [torch.cuda.Event() for _ in range(2)]       # slot k's inputs have landed
        self.ev_trunk = [torch.cuda.Event() for _ in range(2)]      # slot k's trunk has run: its inputs are dead
        self.fed = [False, False]
        self.ran = [False, False]
        self.k_run = 0
        self.copy_stream, self.copy_stream_trials = self._pick_copy_stream(max(1, int(copy_stream_candidates)), with_ensemble)

    def _pick_copy_stream(self, n_cand, ens):
        """HIP maps its streams onto a few hardware queues (4 by default), in order of creation.  The barrier packet behind an
        SDMA copy (the event record that publishes it) holds up whatever shares the copy stream's queue: measured, one of the
        transformer's two branches sat out the whole 0.56 ms copy (profiles/r03_h2d_timeline.txt), and a high-priority copy
        stream slowed every kernel of the step instead, and so did a stream with a hardware queue of its own
        (hipExtStreamCreateWithCUMask: 6.3 ms per step against 4.4) and GPU_MAX_HW_QUEUES=8 -- more than four active queues are
        time-sliced on this part.  Which queue a stream lands on is not visible through the API, so the
        pipeline times a few steps over each of `n_cand` consecutive streams and keeps the one that does not collide."""
        dev = self.dev
        if n_cand == 1:
            return torch.cuda.Stream(device=dev), []
        (q0, im0, _), _, _ = self.slots[0]
        hq, him = torch.zeros(q0.shape, dtype=q0.dtype).pin_memory(), torch.zeros(im0.shape, dtype=im0.dtype).pin_memory()
        trials = []
        for _ in range(n_cand):
            self.copy_stream = cs = torch.cuda.Stream(device=dev)
            dt = []
            for rep in range(2):                                    # (the first round also warms the graphs up)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                self.feed(hq, him)
                for i in range(12):
                    self.step(next_inputs=(hq, him) if i < 11 else None)
                torch.cuda.synchronize(dev)
                dt.append((time.perf_counter() - t0) / 12)
            trials.append((dt[1], cs))
        self.fed, self.ran, self.k_run = [False, False], [False, False], 0
        if ens is not None:
            ens.reset()
        best = min(trials, key=lambda t: t[0])
        return best[1], [round(t[0] * 1e3, 4) for t in trials]

    def feed(self, qpos_host, image_host, slot=None):
        """enqueue the copy of one step's inputs (pinned host tensors) into slot `slot` (default: the slot the next step() runs
        from); it starts once the trunk that read the slot last has run AND the step in flight has left its own trunk"""
        k = self.k_run if slot is None else slot
        s_qpos, s_img, _ = self.slots[k][0]
        cs = self.copy_stream
        if self.ran[k]:
            cs.wait_event(self.ev_trunk[k])
        if self.ran[k ^ 1]:
            cs.wait_event(self.ev_trunk[k ^ 1])                     # the step in flight: copy beside its transformer, not its stem
        with torch.cuda.stream(cs):
            s_qpos.copy_(qpos_host, non_blocking=True)
            s_img.copy_(image_host, non_blocking=True)
            self.ev_copy[k].record(cs)
        self.fed[k] = True

    def copied(self, k: int):
        """block until the last feed into slot k has landed (its host tensors may be rewritten)"""
        self.ev_copy[k].synchronize()

    def step(self, next_inputs=None):
        """run one step from the current slot; `next_inputs` = (qpos_host, image_host) of the FOLLOWING step, copied into the
        other slot beside this step's transformer"""
        k = self.k_run
        if not self.fed[k]:
            raise RuntimeError("InferPipeline.step: no inputs were fed for this step")
        (s_qpos, s_img, _), trunk, rest = self.slots[k]
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.ev_copy[k])
        trunk(s_qpos, s_img)
        self.ev_trunk[k].record(cur)
        self.ran[k] = True
        self.fed[k] = False
        if next_inputs is not None:
            self.feed(next_inputs[0], next_inputs[1], slot=k ^ 1)
        out = rest(s_qpos, s_img)
        self.k_run ^= 1
        return out


def _from_ptr(ptr: int, numel: int, device, typestr: str = "<f4") -> torch.Tensor:
    """View raw device memory as a float32 (or `typestr`) torch tensor (no ownership)."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (numel,), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=device)
