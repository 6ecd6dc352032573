"""VQ-ACT latent prior: host-side mirror of the reference's ``Latent_Model_Transformer`` (detr/models/latent_model.py:35-72)
over the library's kernels — every matrix product, LayerNorm, causal attention, GELU and the categorical draw run in
``libactmi`` through the C ABI; this file only sequences them, as the reference's module does in Python.  Inference only
(``forward`` in eval mode and ``generate``); training the prior (train_latent_model.py) is outside this path.

The block is the reference's, quirks included (latent_model.py:24-31): ``x = ln_1(x); x = x + attn(x, x, x, causal)``,
``x = ln_2(x); x = x + mlp(x)`` — the residuals branch off the NORMALISED activations."""
from collections import OrderedDict

import torch

from . import ops


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

    # ---- nn.Module-like surface
    def load_state_dict(self, sd):
        missing = [k for k in self.spec if k not in sd]
        unexpected = [k for k in sd if k not in self.spec]
        if missing or unexpected:
            raise KeyError(f"latent model state_dict mismatch: missing {missing[:3]}, unexpected {unexpected[:3]}")
        out = {}
        for k, shp in self.spec.items():
            t = torch.as_tensor(sd[k]).to(device=self.device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(t.shape)} != {tuple(shp)}")
            out[k] = t
        self.sd = out
        return self

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self.sd.items())

    def eval(self):
        return self

    def cuda(self):
        return self

    # ---- latent_model.py:50-56 (eval mode: the Dropout modules are identities)
    def forward(self, x):
        sd, D, H = self.sd, self.latent_dim, self.num_head
        n, T, _ = x.shape
        if T > self.seq_len:
            raise ValueError("sequence longer than seq_len")
        x2 = x.to(device=self.device, dtype=torch.float32).reshape(n * T, self.input_dim).contiguous()
        # input_layer + position embedding (row m of the [n*T] matrix is position m % T)
        h = ops.gemm(x2, sd["input_layer.weight"], bias=sd["input_layer.bias"], res=sd["weight_pos_embed.weight"][:T],
                     res_mod=T, prec=self.prec)
        for i in range(1, self.num_layer + 1):
            p = f"attention_blocks.{i}."
            h = ops.layernorm(h, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
            qkv = ops.gemm(h, sd[p + "attn.in_proj_weight"], bias=sd[p + "attn.in_proj_bias"], prec=self.prec).view(n, T, 3 * D)
            a = ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, causal=True, split=False, prec=self.prec)
            h = ops.gemm(a.view(n * T, D), sd[p + "attn.out_proj.weight"], bias=sd[p + "attn.out_proj.bias"], res=h,
                         prec=self.prec)
            h = ops.layernorm(h, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
            m = ops.gemm(h, sd[p + "mlp.0.weight"], bias=sd[p + "mlp.0.bias"], relu="gelu", prec=self.prec)
            h = ops.gemm(m, sd[p + "mlp.2.weight"], bias=sd[p + "mlp.2.bias"], res=h, prec=self.prec)
        p = f"attention_blocks.{self.num_layer + 1}."
        h = ops.layernorm(h, sd[p + "weight"], sd[p + "bias"])
        return ops.gemm(h, sd["output_layer.weight"], bias=sd["output_layer.bias"], prec=self.prec).view(n, T, self.output_dim)

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
