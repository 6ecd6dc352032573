"""CPU ORACLE for the ACT policy hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path (``act-plus-plus_amd``)
never imports it and fails loudly when its HIP library is missing.

What it is: a functional restatement (plain ``torch`` fp32 CPU ops on a state_dict) of the arithmetic
the reference performs on this path.  Each function cites the reference file:line it follows
(paths relative to the upstream repo jie0530/act-plus-plus).

Pinning: the reference holds no tests or golden vectors for this path.  The oracle is
pinned against *outputs of the reference itself*: ``tools/gen_golden.py`` imports the reference's own
``DETRVAE`` / ``Transformer`` / ``FrozenBatchNorm2d`` / ``Joiner`` / ``PositionEmbeddingSine`` modules in
the authoring container, runs them on generated weights/inputs and commits the outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against those fixtures.
The torchvision ``resnet18`` topology is third-party (torchvision 0.15.0 pinned by the reference's
conda_env.yaml:9-10, absent here) and restated from its published architecture: parity for that
piece is pinned only by the well-known BasicBlock definition ("parity unpinned" by reference tests).
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# ------------------------------------------------------------------------------------------------
# policy.py
# ------------------------------------------------------------------------------------------------

def normalize_image(image: torch.Tensor) -> torch.Tensor:
    """torchvision ``transforms.Normalize(mean, std)`` on [...,3,H,W]  (reference policy.py:268-272)."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=image.dtype).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=image.dtype).view(3, 1, 1)
    return (image - mean) / std


def kl_divergence(mu: torch.Tensor, logvar: torch.Tensor):
    """reference policy.py:378-391."""
    klds = -0.5 * (1 + logvar - mu.pow(2) - logvar.exp())
    total_kld = klds.sum(1).mean(0, True)
    dimension_wise_kld = klds.mean(0)
    mean_kld = klds.mean(1).mean(0, True)
    return total_kld, dimension_wise_kld, mean_kld


# ------------------------------------------------------------------------------------------------
# backbone.py + torchvision resnet18 (restated) + position_encoding.py
# ------------------------------------------------------------------------------------------------

def frozen_bn(x, sd, p):
    """FrozenBatchNorm2d.forward, reference backbone.py:47-57 (eps inside rsqrt)."""
    w = sd[p + "weight"].reshape(1, -1, 1, 1)
    b = sd[p + "bias"].reshape(1, -1, 1, 1)
    rv = sd[p + "running_var"].reshape(1, -1, 1, 1)
    rm = sd[p + "running_mean"].reshape(1, -1, 1, 1)
    scale = w * (rv + 1e-5).rsqrt()
    bias = b - rm * scale
    return x * scale + bias


def basic_block(x, sd, p, stride):
    """torchvision BasicBlock: relu(bn2(conv2(relu(bn1(conv1 x)))) + downsample(x))."""
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)
    out = F.relu(frozen_bn(out, sd, p + "bn1."))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1)
    out = frozen_bn(out, sd, p + "bn2.")
    if (p + "downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride, 0)
        idt = frozen_bn(idt, sd, p + "downsample.1.")
    else:
        idt = x
    return F.relu(out + idt)


def resnet18_layer4(x, sd, p, stages: Optional[dict] = None):
    """torchvision resnet18 trunk up to layer4 (IntermediateLayerGetter, reference backbone.py:66-71)."""
    x = F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)
    x = F.relu(frozen_bn(x, sd, p + "bn1."))
    if stages is not None:
        stages["conv1"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    if stages is not None:
        stages["maxpool"] = x
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = basic_block(x, sd, f"{p}layer{li}.0.", stride)
        x = basic_block(x, sd, f"{p}layer{li}.1.", 1)
        if stages is not None:
            stages[f"layer{li}"] = x
    return x


def position_embedding_sine(h: int, w: int, num_pos_feats: int, temperature=10000.0) -> torch.Tensor:
    """PositionEmbeddingSine(normalize=True), reference position_encoding.py:30-52 -> [1, 2*npf, h, w]."""
    not_mask = torch.ones(1, h, w)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------
# transformer.py (post-norm path only: pre_norm is never set by the reference CLI)
# ------------------------------------------------------------------------------------------------

def mha(q_in, k_in, v_in, sd, p, nheads, key_padding_mask=None, dropout_p=0.0):
    """nn.MultiheadAttention forward (seq-first [L,B,D]) with packed in_proj, as used at
    reference transformer.py:217-218, 282-289.  Returns [L,B,D]."""
    L, B, D = q_in.shape
    S = k_in.shape[0]
    hd = D // nheads
    W, bias = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(q_in, W[:D], bias[:D])
    k = F.linear(k_in, W[D:2 * D], bias[D:2 * D])
    v = F.linear(v_in, W[2 * D:], bias[2 * D:])
    q = q.reshape(L, B * nheads, hd).transpose(0, 1)
    k = k.reshape(S, B * nheads, hd).transpose(0, 1)
    v = v.reshape(S, B * nheads, hd).transpose(0, 1)
    attn = torch.bmm(q * (1.0 / math.sqrt(hd)), k.transpose(1, 2))          # [B*h, L, S]
    if key_padding_mask is not None:                                          # [B,S] True = ignore
        m = key_padding_mask.view(B, 1, 1, S).expand(B, nheads, L, S).reshape(B * nheads, L, S)
        attn = attn.masked_fill(m, float("-inf"))
    attn = F.softmax(attn, dim=-1)
    if dropout_p > 0.0:
        attn = F.dropout(attn, dropout_p)
    out = torch.bmm(attn, v).transpose(0, 1).reshape(L, B, D)
    return F.linear(out, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def encoder_layer(src, pos, sd, p, nheads, key_padding_mask=None, dropout_p=0.0):
    """TransformerEncoderLayer.forward_post, reference transformer.py:211-224."""
    drop = (lambda t: F.dropout(t, dropout_p)) if dropout_p > 0 else (lambda t: t)
    q = k = src + pos
    src2 = mha(q, k, src, sd, p + "self_attn.", nheads, key_padding_mask, dropout_p)
    src = _ln(src + drop(src2), sd, p + "norm1.")
    src2 = F.linear(drop(F.relu(F.linear(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"]))),
                    sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return _ln(src + drop(src2), sd, p + "norm2.")


def decoder_layer(tgt, memory, pos, query_pos, sd, p, nheads, dropout_p=0.0):
    """TransformerDecoderLayer.forward_post, reference transformer.py:274-295."""
    drop = (lambda t: F.dropout(t, dropout_p)) if dropout_p > 0 else (lambda t: t)
    q = k = tgt + query_pos
    tgt2 = mha(q, k, tgt, sd, p + "self_attn.", nheads, None, dropout_p)
    tgt = _ln(tgt + drop(tgt2), sd, p + "norm1.")
    tgt2 = mha(tgt + query_pos, memory + pos, memory, sd, p + "multihead_attn.", nheads, None, dropout_p)
    tgt = _ln(tgt + drop(tgt2), sd, p + "norm2.")
    tgt2 = F.linear(drop(F.relu(F.linear(tgt, sd[p + "linear1.weight"], sd[p + "linear1.bias"]))),
                    sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return _ln(tgt + drop(tgt2), sd, p + "norm3.")


# ------------------------------------------------------------------------------------------------
# detr_vae.py
# ------------------------------------------------------------------------------------------------

def cvae_encode(sd, cfg, qpos, actions, is_pad, eps, dropout_p=0.0, p="", return_latent_info=False):
    """DETRVAE.encode, training branch, reference detr_vae.py:117-151.  ``eps`` replaces the
    ``normal_()`` draw of ``reparametrize`` (detr_vae.py:19-22) so that results are reproducible."""
    B = qpos.shape[0]
    action_embed = F.linear(actions, sd[p + "encoder_action_proj.weight"], sd[p + "encoder_action_proj.bias"])
    qpos_embed = F.linear(qpos, sd[p + "encoder_joint_proj.weight"], sd[p + "encoder_joint_proj.bias"]).unsqueeze(1)
    cls_embed = sd[p + "cls_embed.weight"].unsqueeze(0).repeat(B, 1, 1)
    x = torch.cat([cls_embed, qpos_embed, action_embed], dim=1).permute(1, 0, 2)      # [Q+2,B,D]
    mask = torch.cat([torch.zeros(B, 2, dtype=torch.bool), is_pad], dim=1)
    pos = sd[p + "pos_table"].permute(1, 0, 2)                                         # [Q+2,1,D]
    for i in range(cfg.enc_layers):
        x = encoder_layer(x, pos, sd, f"{p}encoder.layers.{i}.", cfg.nheads, mask, dropout_p)
    latent_info = F.linear(x[0], sd[p + "latent_proj.weight"], sd[p + "latent_proj.bias"])
    if return_latent_info:
        return latent_info
    mu, logvar = latent_info[:, :cfg.latent_dim], latent_info[:, cfg.latent_dim:]
    z = mu + logvar.div(2).exp() * eps
    latent_input = F.linear(z, sd[p + "latent_out_proj.weight"], sd[p + "latent_out_proj.bias"])
    return latent_input, mu, logvar


def detrvae_forward(sd: Dict[str, torch.Tensor], cfg, qpos, image_norm, actions=None, is_pad=None, eps=None,
                    dropout_p=0.0, live_only=False, p="", stages: Optional[dict] = None, vq_sample=None):
    """DETRVAE.forward (reference detr_vae.py:163-254) + Transformer.forward (transformer.py:49-122).

    image_norm: [B,C,3,H,W] already ImageNet-normalised.  ``live_only`` skips decoder layers 1.. whose
    outputs are discarded by ``[0]`` at detr_vae.py:245; the default runs all of
    them as the reference does.  Returns a_hat [B,Q,A], is_pad_hat [B,Q,1], mu, logvar.
    """
    B = qpos.shape[0]
    D, H = cfg.hidden_dim, cfg.nheads
    if actions is not None and getattr(cfg, "vq", False):
        # VQ training, detr_vae.py:137-145.  `vq_sample` = the one-hot code the reference draws with torch.multinomial
        # (an explicit input here, like eps); straight-through estimator: value = code, gradient -> probs.
        logits = cvae_encode(sd, cfg, qpos, actions, is_pad, None, dropout_p, p, return_latent_info=True)
        probs = torch.softmax(logits.reshape(B, cfg.vq_class, cfg.vq_dim), dim=-1)
        binaries_flat = vq_sample.reshape(B, cfg.vq_class * cfg.vq_dim)
        probs_flat = probs.reshape(B, cfg.vq_class * cfg.vq_dim)
        straight_through = binaries_flat - probs_flat.detach() + probs_flat
        latent_input = F.linear(straight_through, sd[p + "latent_out_proj.weight"], sd[p + "latent_out_proj.bias"])
        mu, logvar = probs, vq_sample.reshape(B, cfg.vq_class, cfg.vq_dim)        # returned in the (mu, logvar) slots
    elif actions is not None:
        latent_input, mu, logvar = cvae_encode(sd, cfg, qpos, actions, is_pad, eps, dropout_p, p)
    elif getattr(cfg, "vq", False):
        # VQ-ACT inference, detr_vae.py:155-156: the latent is the given code (from the latent prior model)
        mu = logvar = None
        latent_input = F.linear(vq_sample.reshape(-1, cfg.vq_class * cfg.vq_dim), sd[p + "latent_out_proj.weight"],
                                sd[p + "latent_out_proj.bias"])
    else:
        mu = logvar = None
        z = torch.zeros(B, cfg.latent_dim)
        latent_input = F.linear(z, sd[p + "latent_out_proj.weight"], sd[p + "latent_out_proj.bias"])
    feats, poss = [], []
    for c in range(cfg.num_cams):
        st = {} if (stages is not None and c == 0) else None
        f = resnet18_layer4(image_norm[:, c], sd, f"{p}backbones.{c}.0.body.", st)
        if st is not None:
            stages.update({f"cam0_{k}": v for k, v in st.items()})
        feats.append(F.conv2d(f, sd[p + "input_proj.weight"], sd[p + "input_proj.bias"]))
        poss.append(position_embedding_sine(f.shape[2], f.shape[3], D // 2))
    proprio = F.linear(qpos, sd[p + "input_proj_robot_state.weight"], sd[p + "input_proj_robot_state.bias"])
    src = torch.cat(feats, dim=3)                                      # concat along width, detr_vae.py:216
    pos = torch.cat(poss, dim=3)
    # Transformer.forward
    src = src.flatten(2).permute(2, 0, 1)                              # [hw,B,D]
    pos = pos.flatten(2).permute(2, 0, 1).repeat(1, B, 1)
    query_pos = sd[p + "query_embed.weight"].unsqueeze(1).repeat(1, B, 1)
    add_pos = sd[p + "additional_pos_embed.weight"].unsqueeze(1).repeat(1, B, 1)
    pos = torch.cat([add_pos, pos], dim=0)
    src = torch.cat([torch.stack([latent_input, proprio], dim=0), src], dim=0)
    if stages is not None:
        stages["src"] = src
    tgt = torch.zeros_like(query_pos)
    memory = src
    for i in range(cfg.enc_layers):
        memory = encoder_layer(memory, pos, sd, f"{p}transformer.encoder.layers.{i}.", H, None, dropout_p)
    if stages is not None:
        stages["memory"] = memory
    out = tgt
    inter = []
    nl = 1 if live_only else cfg.dec_layers
    for i in range(nl):
        out = decoder_layer(out, memory, pos, query_pos, sd, f"{p}transformer.decoder.layers.{i}.", H, dropout_p)
        inter.append(_ln(out, sd, p + "transformer.decoder.norm."))
    hs = torch.stack(inter).transpose(1, 2)[0]                         # transformer.py:184,120; detr_vae.py:245
    if stages is not None:
        stages["hs"] = hs
    a_hat = F.linear(hs, sd[p + "action_head.weight"], sd[p + "action_head.bias"])
    is_pad_hat = F.linear(hs, sd[p + "is_pad_head.weight"], sd[p + "is_pad_head.bias"])
    return a_hat, is_pad_hat, mu, logvar


def policy_call(sd, cfg, qpos, image, actions=None, is_pad=None, eps=None, dropout_p=0.0, live_only=False, p="model.",
                vq_sample=None):
    """ACTPolicy.__call__, reference policy.py:264-332.  ``image`` is the reference's contract:
    f32 [B,C,3,H,W] in [0,1].  Training returns {'l1','kl','loss'} (+ a_hat, mu, logvar for tests)."""
    image = normalize_image(image)
    if actions is not None:
        Q = cfg.num_queries
        actions, is_pad = actions[:, :Q], is_pad[:, :Q]
        a_hat, _, mu, logvar = detrvae_forward(sd, cfg, qpos, image, actions, is_pad, eps, dropout_p, live_only, p,
                                               vq_sample=vq_sample)
        all_l1 = F.l1_loss(actions, a_hat, reduction="none")
        l1 = (all_l1 * ~is_pad.unsqueeze(-1)).mean()
        if getattr(cfg, "vq", False):                      # policy.py:307-312: no KL; the discrepancy is only logged
            out = {"l1": l1, "kl": torch.tensor(0.0), "vq_discrepancy": F.l1_loss(mu, logvar, reduction="mean")}
        else:
            total_kld, _, _ = kl_divergence(mu, logvar)
            out = {"l1": l1, "kl": total_kld[0]}
        out["loss"] = out["l1"] + out["kl"] * cfg.kl_weight
        out["a_hat"], out["mu"], out["logvar"] = a_hat, mu, logvar
        return out
    a_hat, _, _, _ = detrvae_forward(sd, cfg, qpos, image, live_only=live_only, p=p, vq_sample=vq_sample)
    return a_hat


# ------------------------------------------------------------------------------------------------
# detr/models/latent_model.py
# ------------------------------------------------------------------------------------------------

def latent_model_forward(sd, x, num_head=8, num_layer=3):
    """Latent_Model_Transformer.forward in eval mode (latent_model.py:50-56 over the block of :24-31): residuals branch
    off the normalised activations, causal self-attention, exact GELU."""
    T = x.shape[1]
    h = F.linear(x, sd["input_layer.weight"], sd["input_layer.bias"]) + sd["weight_pos_embed.weight"][:T]
    D = h.shape[-1]
    mask = torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)
    for i in range(1, num_layer + 1):
        p = f"attention_blocks.{i}."
        h = F.layer_norm(h, (D,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        a, _ = F.multi_head_attention_forward(
            h.transpose(0, 1), h.transpose(0, 1), h.transpose(0, 1), D, num_head, sd[p + "attn.in_proj_weight"],
            sd[p + "attn.in_proj_bias"], None, None, False, 0.0, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"],
            training=False, attn_mask=mask, need_weights=False)
        h = h + a.transpose(0, 1)
        h = F.layer_norm(h, (D,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        m = F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
        h = h + F.linear(m, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    p = f"attention_blocks.{num_layer + 1}."
    h = F.layer_norm(h, (D,), sd[p + "weight"], sd[p + "bias"])
    return F.linear(h, sd["output_layer.weight"], sd["output_layer.bias"])


def latent_model_train_step(sd, x, labels, lr=None, steps=0, num_head=8, num_layer=3, weight_decay=0.01):
    """One forward_pass of train_latent_model.py:323-343 with the dropout modules off (the parity mode: torch's dropout stream
    cannot be reproduced): logits, ``F.cross_entropy(output_logits, gt_labels)`` -- [B, T, V] probabilities as targets, so the
    class axis is dim 1 --, the one-hot L1 metric of :331-334, the gradients of every parameter, and (lr given) the parameters
    after ``steps`` iterations of ``torch.optim.AdamW(parameters, lr=lr)`` on the same batch (:395-404).  float64 throughout."""
    prm = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    x, labels = x.double(), labels.double()

    def fwd():
        logits = latent_model_forward(prm, x, num_head, num_layer)
        return logits, F.cross_entropy(logits, labels)

    logits, loss = fwd()
    grads = dict(zip(prm.keys(), torch.autograd.grad(loss, list(prm.values()))))
    with torch.no_grad():
        onehot = F.one_hot(torch.argmax(logits, dim=-1), num_classes=labels.shape[-1]).double()
        l1 = F.l1_loss(onehot, labels, reduction="mean")
    out = {"logits": logits.detach(), "loss": loss.detach(), "l1_error": l1, "grads": grads}
    if lr is not None and steps > 0:
        opt = torch.optim.AdamW(list(prm.values()), lr=lr, weight_decay=weight_decay)
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            _, l = fwd()
            l.backward()
            opt.step()
            losses.append(float(l))
        out["params_after"] = {k: v.detach().clone() for k, v in prm.items()}
        out["losses"] = losses
    return out


# ------------------------------------------------------------------------------------------------
# imitate_episodes.py
# ------------------------------------------------------------------------------------------------

def get_image_from_u8(img_u8_nhwc: np.ndarray) -> torch.Tensor:
    """get_image, reference imitate_episodes.py:206-212: HWC u8 -> CHW, /255.0 in float64, .float()."""
    x = np.moveaxis(img_u8_nhwc, -1, -3)
    return torch.from_numpy(x / 255.0).float()


class TemporalEnsembleRef:
    """Temporal ensembling exactly as reference imitate_episodes.py:338-339, 402-411 (one episode).

    Keeps the full [T, T+Q, A] buffer like the reference, float32 storage, float64 weights: the product
    ``actions_for_curr_step * exp_weights`` promotes to float64 and ``raw_action`` is float64.
    """

    def __init__(self, max_timesteps: int, num_queries: int, action_dim: int = 16, k: float = 0.01):
        self.buf = torch.zeros([max_timesteps, max_timesteps + num_queries, action_dim])
        self.Q, self.k = num_queries, k

    def step(self, t: int, all_actions: torch.Tensor):
        """all_actions [1,Q,A] f32 -> (raw_action [1,A] f64, populated mask [T] bool)."""
        self.buf[[t], t:t + self.Q] = all_actions
        actions_for_curr_step = self.buf[:, t]
        actions_populated = torch.all(actions_for_curr_step != 0, axis=1)
        actions_for_curr_step = actions_for_curr_step[actions_populated]
        exp_weights = np.exp(-self.k * np.arange(len(actions_for_curr_step)))
        exp_weights = exp_weights / exp_weights.sum()
        exp_weights = torch.from_numpy(exp_weights).unsqueeze(dim=1)
        raw_action = (actions_for_curr_step * exp_weights).sum(dim=0, keepdim=True)
        return raw_action, actions_populated


def adamw_reference_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-4):
    """torch.optim.AdamW single-tensor update (defaults used at reference detr/main.py:109-110)."""
    params = params * (1 - lr * wd)
    exp_avg = exp_avg * beta1 + grads * (1 - beta1)
    exp_avg_sq = exp_avg_sq * beta2 + grads * grads * (1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = exp_avg_sq.sqrt() / math.sqrt(bc2) + eps
    params = params - (lr / bc1) * exp_avg / denom
    return params, exp_avg, exp_avg_sq


def to_torch_sd(np_sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_sd.items()}
