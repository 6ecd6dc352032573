"""VQ-ACT latent prior: host-side mirror of the reference's ``Latent_Model_Transformer`` (detr/models/latent_model.py:35-72)
over the library's kernels — every matrix product, LayerNorm, causal attention, GELU and the categorical draw run in
``libactmi`` through the C ABI; this file only sequences them, as the reference's module does in Python.  ``forward`` in eval mode and
``generate`` serve the rollout; in train mode ``forward`` keeps the activations and ``cross_entropy(...).backward()`` /
``configure_optimizer(lr).step()`` are the training step of train_latent_model.py:323-343,395-404 (autograd restated by hand:
linear, LayerNorm, causal-attention, GELU and dropout backward on the library's kernels, torch.optim.AdamW as one fused launch).

The block is the reference's, quirks included (latent_model.py:24-31): ``x = ln_1(x); x = x + attn(x, x, x, causal)``,
``x = ln_2(x); x = x + mlp(x)`` — the residuals branch off the NORMALISED activations."""
from collections import OrderedDict

import torch

from . import ops

DROPOUT_RATE = 0.1      # latent_model.py:5


def latent_model_spec(input_dim, output_dim, seq_len, latent_dim=256, num_layer=3):
    """state_dict keys and shapes in the reference's registration order (nn.Sequential: 0 = Dropout, 1..L = blocks,
    L+1 = LayerNorm)."""
    o = OrderedDict()
    o["input_layer.weight"] = (latent_dim, input_dim); o["input_layer.bias"] = (latent_dim,)
    o["weight_pos_embed.weight"] = (seq_len, latent_dim)
    for i in range(1, num_layer + 1):
        p = f"attention_blocks.{i}."
        o[p + "ln_1.weight"] = (latent_dim,); o[p + "ln_1.bias"] = (latent_dim,)
        o[p + "attn.in_proj_weight"] = (3 * latent_dim, latent_dim); o[p + "attn.in_proj_bias"] = (3 * latent_dim,)
        o[p + "attn.out_proj.weight"] = (latent_dim, latent_dim); o[p + "attn.out_proj.bias"] = (latent_dim,)
        o[p + "ln_2.weight"] = (latent_dim,); o[p + "ln_2.bias"] = (latent_dim,)
        o[p + "mlp.0.weight"] = (4 * latent_dim, latent_dim); o[p + "mlp.0.bias"] = (4 * latent_dim,)
        o[p + "mlp.2.weight"] = (latent_dim, 4 * latent_dim); o[p + "mlp.2.bias"] = (latent_dim,)
    p = f"attention_blocks.{num_layer + 1}."
    o[p + "weight"] = (latent_dim,); o[p + "bias"] = (latent_dim,)
    o["output_layer.weight"] = (output_dim, latent_dim); o["output_layer.bias"] = (output_dim,)
    return o


class LatentModelTransformer:
    """reference latent_model.py:35-72; constructor signature and ``generate`` semantics kept."""

    def __init__(self, input_dim, output_dim, seq_len, latent_dim=256, num_head=8, num_layer=3, device="cuda:0",
                 gemm_prec=None):
        if not torch.cuda.is_available():
            raise RuntimeError("LatentModelTransformer needs an MI355X; no CPU fallback exists")
        if latent_dim % num_head or latent_dim // num_head not in (16, 32, 64):
            raise ValueError("latent_dim / num_head must be 16, 32 or 64")
        if input_dim % 4:
            raise ValueError("input_dim must be a multiple of 4")
        self.input_dim, self.output_dim, self.seq_len = input_dim, output_dim, seq_len
        self.latent_dim, self.num_head, self.num_layer = latent_dim, num_head, num_layer
        self.device = torch.device(device)
        self.prec = gemm_prec
        self.spec = latent_model_spec(input_dim, output_dim, seq_len, latent_dim, num_layer)
        self.sd = None
        self.training = False                 # the rollout loads a checkpoint and calls generate(); train() switches dropout on
        self.dropout_seed = 0                 # keys the counter-based dropout masks of train mode (torch uses its global RNG)
        self._fwd_count = 0
        self._saved = None

    # ---- nn.Module-like surface
    def _alloc(self):
        """parameters, gradients and AdamW moments as flat arenas (one optimizer launch); ``self.sd`` / ``self.grad`` are views"""
        n = 0
        self._off = {}
        for k, shp in self.spec.items():
            numel = 1
            for d in shp:
                numel *= d
            self._off[k] = (n, numel, tuple(shp))
            n += (numel + 3) & ~3                     # 16-byte aligned starts (the GEMM and LayerNorm loads are float4)
        dev = self.device
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.sd = {k: self.params[o:o + m].view(shp) for k, (o, m, shp) in self._off.items()}
        self.grad = {k: self.grads[o:o + m].view(shp) for k, (o, m, shp) in self._off.items()}
        self._ws = torch.empty(2 * self.latent_dim * 1024, dtype=torch.float32, device=dev)      # ordered-sum scratch
        self._step = 0

    def load_state_dict(self, sd):
        missing = [k for k in self.spec if k not in sd]
        unexpected = [k for k in sd if k not in self.spec]
        if missing or unexpected:
            raise KeyError(f"latent model state_dict mismatch: missing {missing[:3]}, unexpected {unexpected[:3]}")
        if self.sd is None:
            self._alloc()
        for k, shp in self.spec.items():
            t = torch.as_tensor(sd[k]).to(device=self.device, dtype=torch.float32)
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(t.shape)} != {tuple(shp)}")
            self.sd[k].copy_(t)
        return self

    def state_dict(self):
        return OrderedDict((k, self.sd[k].clone()) for k in self.spec)

    def parameters(self):
        return [self.sd[k] for k in self.spec]

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def cuda(self):
        return self

    # ---- latent_model.py:50-56.  eval mode: the Dropout modules are identities and nothing is kept; train mode: dropout
    # (p = DROPOUT_RATE) at the three sites of the reference and the activations the backward needs
    def forward(self, x):
        sd, D, H = self.sd, self.latent_dim, self.num_head
        n, T, _ = x.shape
        if T > self.seq_len:
            raise ValueError("sequence longer than seq_len")
        train = self.training
        p_drop = DROPOUT_RATE if train else 0.0
        if train:
            self._fwd_count += 1
        seed = lambda site: (self.dropout_seed << 20) + (self._fwd_count << 6) + site      # noqa: E731
        x2 = x.to(device=self.device, dtype=torch.float32).reshape(n * T, self.input_dim).contiguous()
        # input_layer + position embedding (row m of the [n*T] matrix is position m % T)
        h = ops.gemm(x2, sd["input_layer.weight"], bias=sd["input_layer.bias"], res=sd["weight_pos_embed.weight"][:T],
                     res_mod=T, prec=self.prec)
        if p_drop:
            h = ops.dropout(h, p_drop, seed(0))
        saved = {"x2": x2, "n": n, "T": T, "blocks": [], "p": p_drop, "seeds": seed} if train else None
        for i in range(1, self.num_layer + 1):
            p = f"attention_blocks.{i}."
            a_in = h
            n1 = ops.layernorm(a_in, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
            qkv = ops.gemm(n1, sd[p + "attn.in_proj_weight"], bias=sd[p + "attn.in_proj_bias"], prec=self.prec).view(n, T, 3 * D)
            if train:
                att = ops.small_attention(qkv, H, causal=True, drop_p=p_drop, seed=seed(3 * i + 1))
            else:
                att = ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, causal=True, split=False, prec=self.prec)
            r1 = ops.gemm(att.view(n * T, D), sd[p + "attn.out_proj.weight"], bias=sd[p + "attn.out_proj.bias"], res=n1,
                          prec=self.prec)
            n2 = ops.layernorm(r1, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
            if train:
                u = ops.gemm(n2, sd[p + "mlp.0.weight"], bias=sd[p + "mlp.0.bias"], prec=self.prec)
                m = ops.gelu(u)
                # nn.Dropout on the second Linear's output, then the residual: the GEMM epilogue's own order (its mask is
                # keep(seed, element index), which ops.dropout regenerates in the backward)
                h = ops.gemm(m, sd[p + "mlp.2.weight"], bias=sd[p + "mlp.2.bias"], res=n2, drop_p=p_drop,
                             drop_seed=seed(3 * i + 2), prec=self.prec)
                saved["blocks"].append({"a_in": a_in, "n1": n1, "qkv": qkv, "att": att, "r1": r1, "n2": n2, "u": u, "m": m})
            else:
                m = ops.gemm(n2, sd[p + "mlp.0.weight"], bias=sd[p + "mlp.0.bias"], relu="gelu", prec=self.prec)
                h = ops.gemm(m, sd[p + "mlp.2.weight"], bias=sd[p + "mlp.2.bias"], res=n2, prec=self.prec)
        p = f"attention_blocks.{self.num_layer + 1}."
        hf = ops.layernorm(h, sd[p + "weight"], sd[p + "bias"])
        logits = ops.gemm(hf, sd["output_layer.weight"], bias=sd["output_layer.bias"], prec=self.prec).view(n, T, self.output_dim)
        if train:
            saved["o_last"], saved["hf"] = h, hf
            self._saved = saved
        return logits

    __call__ = forward

    # ---- latent_model.py:58-72
    @torch.no_grad()
    def generate(self, n, temperature=0.1, x=None, seed=0):
        """autoregressive draw of ``seq_len`` one-hot codes; ``seed`` keys the device-side categorical draws (the
        reference uses torch.multinomial on the global RNG)."""
        if x is None:
            x = torch.zeros((n, 1, self.input_dim), device=self.device)
        for i in range(self.seq_len):
            logits = self.forward(x)[:, -1]
            onehot = ops.sample_onehot(logits, temperature=temperature, seed=(int(seed) << 8) + i)
            x = torch.cat([x, onehot[:, None, :]], dim=1)
        return x[:, 1:, :]

    # ---- training step (train_latent_model.py:323-343: forward_pass; :395-404: backward, AdamW) -----------------------------
    def cross_entropy(self, logits, labels):
        """``F.cross_entropy(output_logits, gt_labels)`` as train_latent_model.py:329 calls it -- [B, T, V] logits against [B, T, V]
        probabilities, so the class axis is dim 1 (the sequence axis; vq_class == vq_dim keeps the shapes legal).  Returns a
        ``PriorLoss``: a 0-dim view of the device loss with ``.backward()``; ``.l1_error`` is the metric of lines 331-334."""
        labels = labels.to(device=self.device, dtype=torch.float32).contiguous()
        logits = logits.contiguous()
        loss, dlogits = ops.soft_ce_dim1(logits, labels, want_grad=self.training)
        l1 = ops.argmax_l1(logits, labels)
        return PriorLoss(self, loss, dlogits, l1)

    def zero_grad(self):
        self.grads.zero_()

    def _backward(self, dlogits):
        """gradients of every parameter into ``self.grads`` (accumulating, as autograd does); native fp32 products"""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward without a train-mode forward before it")
        self._saved = None
        sd, g, ws, H = self.sd, self.grad, self._ws, self.num_head
        n, T, D, p_drop, seed = sv["n"], sv["T"], self.latent_dim, sv["p"], sv["seeds"]
        M = n * T

        def linear_bwd(dy, x, wkey, bkey, need_dx=True, dx_res=None):
            # dW += dY^T X ; db += colsum(dY) ; dX = dY W (+ dx_res)
            ops.gemm_t(dy, x, ta=True, tb=True, res=g[wkey], out=g[wkey])
            ops.colsum(dy, g[bkey], ws)
            return ops.gemm_t(dy, sd[wkey], tb=True, res=dx_res) if need_dx else None

        dl = dlogits.view(M, self.output_dim)
        dhf = linear_bwd(dl, sv["hf"], "output_layer.weight", "output_layer.bias")
        p = f"attention_blocks.{self.num_layer + 1}."
        do = ops.layernorm_bwd(sv["o_last"], sd[p + "weight"], dhf, g[p + "weight"], g[p + "bias"], ws)
        for i in range(self.num_layer, 0, -1):
            p = f"attention_blocks.{i}."
            b = sv["blocks"][i - 1]
            # o = n2 + drop(mlp.2(gelu(mlp.0(n2))))
            dz = ops.dropout(do, p_drop, seed(3 * i + 2)) if p_drop else do
            dm = linear_bwd(dz, b["m"], p + "mlp.2.weight", p + "mlp.2.bias")
            du = ops.gelu_bwd(b["u"], dm)
            dn2 = linear_bwd(du, b["n2"], p + "mlp.0.weight", p + "mlp.0.bias", dx_res=do)
            dr1 = ops.layernorm_bwd(b["r1"], sd[p + "ln_2.weight"], dn2, g[p + "ln_2.weight"], g[p + "ln_2.bias"], ws)
            # r1 = n1 + out_proj(attention(in_proj(n1)))
            datt = linear_bwd(dr1, b["att"].view(M, D), p + "attn.out_proj.weight", p + "attn.out_proj.bias")
            dqkv = ops.small_attention_bwd(b["qkv"], datt.view(n, T, D), H, causal=True, drop_p=p_drop, seed=seed(3 * i + 1))
            dn1 = linear_bwd(dqkv.view(M, 3 * D), b["n1"], p + "attn.in_proj_weight", p + "attn.in_proj_bias", dx_res=dr1)
            do = ops.layernorm_bwd(b["a_in"], sd[p + "ln_1.weight"], dn1, g[p + "ln_1.weight"], g[p + "ln_1.bias"], ws)
        dh0 = ops.dropout(do, p_drop, seed(0)) if p_drop else do
        linear_bwd(dh0, sv["x2"], "input_layer.weight", "input_layer.bias", need_dx=False)
        ops.sum_batch(dh0.view(n, T, D), g["weight_pos_embed.weight"][:T], accumulate=True)

    def configure_optimizer(self, lr, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8):
        """``torch.optim.AdamW(latent_model.parameters(), lr=lr)`` (train_latent_model.py:361) with torch's defaults"""
        return PriorAdamW(self, lr, weight_decay, betas, eps)


class PriorLoss:
    """what ``forward_dict['loss']`` is in the reference: ``.backward()``, ``.item()``, ``float()``; the device value in ``.value``"""

    def __init__(self, model, loss, dlogits, l1):
        self.model, self.value, self.dlogits, self.l1_error = model, loss.view(()), dlogits, l1.view(())

    def backward(self):
        if self.dlogits is None:
            raise RuntimeError("the loss was computed in eval mode: nothing to differentiate")
        self.model._backward(self.dlogits)
        self.dlogits = None

    def item(self):
        return float(self.value.item())

    __float__ = item

    def detach(self):
        return self.value

    def cpu(self):
        return self.value.cpu()


class PriorAdamW:
    def __init__(self, model, lr, weight_decay, betas, eps):
        self.model, self.lr, self.wd, self.betas, self.eps = model, float(lr), float(weight_decay), betas, float(eps)

    def step(self):
        m = self.model
        m._step += 1
        ops.adamw(m.params, m.grads, m.exp_avg, m.exp_avg_sq, self.lr, self.wd, m._step, self.betas, self.eps)

    def zero_grad(self):
        self.model.zero_grad()

    def state_dict(self):
        m = self.model
        return {"step": m._step, "exp_avg": m.exp_avg.clone(), "exp_avg_sq": m.exp_avg_sq.clone(), "lr": self.lr}

    def load_state_dict(self, st):
        m = self.model
        m._step = int(st["step"])
        m.exp_avg.copy_(st["exp_avg"]); m.exp_avg_sq.copy_(st["exp_avg_sq"])

