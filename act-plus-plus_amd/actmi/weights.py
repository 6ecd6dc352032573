"""state_dict layout of the ACT network and a deterministic weight / input generator.

The key names and shapes are the checkpoint wire format of the reference
(``ACTPolicy.serialize`` = ``nn.Module.state_dict()``, reference policy.py:344-348, keys prefixed
``model.``): DETRVAE registration order detr_vae.py:49-105, Transformer transformer.py:28-37,
torchvision resnet18 attribute names behind ``IntermediateLayerGetter`` (backbone.py:70) and
``FrozenBatchNorm2d`` buffers (backbone.py:30-35).  ``tools/gen_golden.py`` asserts that this spec
equals the state_dict of the reference classes instantiated in the authoring container.

The generator is counter based (splitmix64 over ``hash(seed, key) + element index``) so that tensors
do not depend on torch / numpy RNG versions and can be regenerated identically on the GPU box.
"""
from collections import OrderedDict
import hashlib
import math

import numpy as np

from .config import ACTConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key_seed(seed: int, key: str) -> np.uint64:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def uniform01(seed: int, key: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (seed, key, stream, index)."""
    with np.errstate(over="ignore"):
        base = _key_seed(seed, key) + np.uint64(stream) * np.uint64(0xD1342543DE82EF95)
        idx = np.arange(n, dtype=np.uint64) + base
        bits = _splitmix64(idx)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, key: str, n: int) -> np.ndarray:
    """Box-Muller on two independent streams; float64."""
    u1 = uniform01(seed, key, n, stream=1)
    u2 = uniform01(seed, key, n, stream=2)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def rand_u8(seed: int, key: str, shape) -> np.ndarray:
    n = int(np.prod(shape))
    base = _key_seed(seed, key)
    with np.errstate(over="ignore"):
        bits = _splitmix64(np.arange((n + 7) // 8, dtype=np.uint64) + base)
    return bits.view(np.uint8)[:n].reshape(shape).copy()


# ----------------------------------------------------------------------------------------------
# state_dict spec
# ----------------------------------------------------------------------------------------------

def _mha(prefix, D, out):
    out[prefix + "in_proj_weight"] = (3 * D, D)
    out[prefix + "in_proj_bias"] = (3 * D,)
    out[prefix + "out_proj.weight"] = (D, D)
    out[prefix + "out_proj.bias"] = (D,)


def _enc_layer(prefix, D, F, out):
    _mha(prefix + "self_attn.", D, out)
    out[prefix + "linear1.weight"] = (F, D); out[prefix + "linear1.bias"] = (F,)
    out[prefix + "linear2.weight"] = (D, F); out[prefix + "linear2.bias"] = (D,)
    for n in ("norm1", "norm2"):
        out[prefix + n + ".weight"] = (D,); out[prefix + n + ".bias"] = (D,)


def _dec_layer(prefix, D, F, out):
    _mha(prefix + "self_attn.", D, out)
    _mha(prefix + "multihead_attn.", D, out)
    out[prefix + "linear1.weight"] = (F, D); out[prefix + "linear1.bias"] = (F,)
    out[prefix + "linear2.weight"] = (D, F); out[prefix + "linear2.bias"] = (D,)
    for n in ("norm1", "norm2", "norm3"):
        out[prefix + n + ".weight"] = (D,); out[prefix + n + ".bias"] = (D,)


def _fbn(prefix, c, out):
    for n in ("weight", "bias", "running_mean", "running_var"):
        out[prefix + n] = (c,)


def resnet18_layout(base_width=64):
    """(name, cin, cout, ksize, stride, pad) for every conv of the resnet18 trunk, in execution order."""
    w = base_width
    convs = [("conv1", 3, w, 7, 2, 3)]
    cin = w
    for li, (cout, stride) in enumerate([(w, 1), (2 * w, 2), (4 * w, 2), (8 * w, 2)], start=1):
        for bi in range(2):
            s = stride if bi == 0 else 1
            convs.append((f"layer{li}.{bi}.conv1", cin, cout, 3, s, 1))
            convs.append((f"layer{li}.{bi}.conv2", cout, cout, 3, 1, 1))
            if bi == 0 and (s != 1 or cin != cout):
                convs.append((f"layer{li}.{bi}.downsample.0", cin, cout, 1, s, 0))
            cin = cout
    return convs


def _backbone(prefix, base_width, out):
    for name, cin, cout, k, _, _ in resnet18_layout(base_width):
        if name.endswith("downsample.0"):
            continue  # emitted after bn2 of the block, below
        out[prefix + name + ".weight"] = (cout, cin, k, k)
        bn = name.replace("conv", "bn") if name != "conv1" else "bn1"
        _fbn(prefix + bn + ".", cout, out)
        if name.endswith(".0.conv2"):
            blk = name[: -len("conv2")]
            li = int(name[5])
            if li > 1:
                cin_blk = cout // 2
                out[prefix + blk + "downsample.0.weight"] = (cout, cin_blk, 1, 1)
                _fbn(prefix + blk + "downsample.1.", cout, out)


def act_state_dict_spec(cfg: ACTConfig, prefix: str = "") -> "OrderedDict[str, tuple]":
    """Ordered {key: shape} equal to ``DETRVAE(...).state_dict()`` of the reference for this config."""
    D, F, Q, S, A, L = cfg.hidden_dim, cfg.dim_feedforward, cfg.num_queries, cfg.state_dim, cfg.action_dim, cfg.latent_dim
    o = OrderedDict()
    o["pos_table"] = (1, Q + 2, D)                                  # buffer, detr_vae.py:91
    for i in range(cfg.enc_layers):
        _enc_layer(f"transformer.encoder.layers.{i}.", D, F, o)
    for i in range(cfg.dec_layers):
        _dec_layer(f"transformer.decoder.layers.{i}.", D, F, o)
    o["transformer.decoder.norm.weight"] = (D,); o["transformer.decoder.norm.bias"] = (D,)
    if not cfg.no_encoder:
        for i in range(cfg.enc_layers):
            _enc_layer(f"encoder.layers.{i}.", D, F, o)
    o["action_head.weight"] = (A, D); o["action_head.bias"] = (A,)
    o["is_pad_head.weight"] = (1, D); o["is_pad_head.bias"] = (1,)
    o["query_embed.weight"] = (Q, D)
    C4 = 8 * cfg.base_width
    o["input_proj.weight"] = (D, C4, 1, 1); o["input_proj.bias"] = (D,)
    for c in range(cfg.num_cams):
        _backbone(f"backbones.{c}.0.body.", cfg.base_width, o)
    o["input_proj_robot_state.weight"] = (D, S); o["input_proj_robot_state.bias"] = (D,)
    o["cls_embed.weight"] = (1, D)
    o["encoder_action_proj.weight"] = (D, A); o["encoder_action_proj.bias"] = (D,)
    o["encoder_joint_proj.weight"] = (D, S); o["encoder_joint_proj.bias"] = (D,)
    o["latent_proj.weight"] = (cfg.latent_proj_dim, D); o["latent_proj.bias"] = (cfg.latent_proj_dim,)
    o["latent_out_proj.weight"] = (D, cfg.latent_in_dim); o["latent_out_proj.bias"] = (D,)
    o["additional_pos_embed.weight"] = (2, D)
    if prefix:
        o = OrderedDict((prefix + k, v) for k, v in o.items())
    return o


BUFFER_SUFFIXES = ("running_mean", "running_var", "pos_table")


def is_buffer(key: str) -> bool:
    """FrozenBatchNorm2d holds *buffers* only (backbone.py:30-35): weight/bias there are not parameters."""
    if key.endswith(BUFFER_SUFFIXES):
        return True
    if "backbones." in key and (".bn" in key or "downsample.1." in key):
        return True
    return False


def is_backbone_param(key: str) -> bool:
    """AdamW group rule of the reference: '"backbone" in name' (detr/main.py:102-108)."""
    return "backbone" in key


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """1-D sinusoid table, float64 -> float32 exactly as detr_vae.py:25-33."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return ang.astype(np.float32)[None]


def generate_state_dict(cfg: ACTConfig, seed: int = 0, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 numpy state_dict with activation-preserving scales.

    Random-init weights of the reference architecture (there is no network for checkpoints,
    and the reference's ImageNet fetch at backbone.py:121-124 cannot run offline).
    """
    spec = act_state_dict_spec(cfg)
    sd = OrderedDict()
    for k, shape in spec.items():
        n = int(np.prod(shape))
        if k == "pos_table":
            v = sinusoid_table(cfg.num_queries + 2, cfg.hidden_dim)
        elif k.endswith("running_var"):
            v = 0.5 + uniform01(seed, k, n)
        elif k.endswith("running_mean"):
            v = 0.1 * normal(seed, k, n)
        elif "backbones." in k and (".bn" in k or "downsample.1." in k):
            if k.endswith("weight"):
                # second BN of a block is damped so that the residual sum keeps O(1) variance
                v = (0.3 + 0.4 * uniform01(seed, k, n)) if ".bn2." in k else (0.7 + 0.6 * uniform01(seed, k, n))
            else:
                v = 0.1 * normal(seed, k, n)
        elif len(shape) == 4:                                      # conv weight, kaiming-normal fan_in
            fan_in = shape[1] * shape[2] * shape[3]
            v = normal(seed, k, n) * math.sqrt(2.0 / fan_in)
        elif "norm" in k:                                           # LayerNorm
            v = (1.0 + 0.1 * normal(seed, k, n)) if k.endswith("weight") else 0.05 * normal(seed, k, n)
        elif "embed" in k:                                          # nn.Embedding ~ N(0,1)
            v = normal(seed, k, n)
        elif len(shape) == 2:                                       # Linear / in_proj: xavier-uniform
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            v = (2.0 * uniform01(seed, k, n) - 1.0) * a
        else:                                                       # biases
            v = 0.05 * normal(seed, k, n)
        sd[prefix + k] = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(shape))
    return sd


def generate_inputs(cfg: ACTConfig, batch: int, seed: int = 1234, with_actions: bool = False):
    """Synthetic inputs of SURVEY §8(d): u8 NHWC images, N(0,1) qpos, optional actions / is_pad / eps."""
    C = cfg.num_cams
    img = rand_u8(seed, "image", (batch, C, cfg.image_h, cfg.image_w, 3))
    qpos = normal(seed, "qpos", batch * cfg.state_dim).astype(np.float32).reshape(batch, cfg.state_dim)
    out = {"image_u8": img, "qpos": qpos}
    if cfg.vq:
        # one-hot code per class, as Latent_Model_Transformer.generate emits (latent_model.py:60-72)
        pick = (uniform01(seed, "vq_pick", batch * cfg.vq_class) * cfg.vq_dim).astype(np.int64).reshape(batch, cfg.vq_class)
        code = np.zeros((batch, cfg.vq_class, cfg.vq_dim), dtype=np.float32)
        for b in range(batch):
            code[b, np.arange(cfg.vq_class), np.minimum(pick[b], cfg.vq_dim - 1)] = 1.0
        out["vq_sample"] = code
    if with_actions:
        Q, A = cfg.num_queries, cfg.action_dim
        out["actions"] = normal(seed, "actions", batch * Q * A).astype(np.float32).reshape(batch, Q, A)
        npad = (uniform01(seed, "npad", batch) * (Q // 2)).astype(np.int64)
        is_pad = np.zeros((batch, Q), dtype=bool)
        for b in range(batch):
            if npad[b]:
                is_pad[b, Q - npad[b]:] = True
        out["is_pad"] = is_pad
        out["eps"] = normal(seed, "eps", batch * cfg.latent_dim).astype(np.float32).reshape(batch, cfg.latent_dim)
    return out


def u8_nhwc_to_f32_nchw(img_u8: np.ndarray) -> np.ndarray:
    """The reference's image contract: ``(u8 / 255.0 in float64).float()`` as CHW
    (imitate_episodes.py:206-212; utils.py:147-152 does the same in float32 for training)."""
    x = np.moveaxis(img_u8, -1, -3)
    return (x / 255.0).astype(np.float32)


def fixture_sample(a: np.ndarray, max_elems: int) -> np.ndarray:
    """Deterministic strided sample of a flattened array, used to keep full-size golden fixtures small.
    The stride is odd so that it does not alias with power-of-two tensor extents."""
    flat = np.ascontiguousarray(a).reshape(-1)
    if not max_elems or flat.size <= max_elems:
        return flat.copy() if flat is a else flat
    stride = -(-flat.size // max_elems)
    stride += (stride % 2 == 0)
    return flat[::stride].copy()


def generate_latent_model_state_dict(spec, seed: int):
    """Deterministic weights for the VQ latent prior (actmi/latent_model.py:latent_model_spec): LayerNorm gains near 1,
    biases small, matrices ~ N(0, 1/fan_in)."""
    out = {}
    for k, shp in spec.items():
        base = normal(seed, "lm:" + k, int(np.prod(shp))).astype(np.float32).reshape(shp)
        is_gain = k.endswith("ln_1.weight") or k.endswith("ln_2.weight") or (k.startswith("attention_blocks.") and
                                                                               k.count(".") == 2 and k.endswith(".weight"))
        if is_gain:
            v = 1.0 + 0.1 * base
        elif k.endswith("bias"):
            v = 0.1 * base
        else:
            v = base / np.float32(np.sqrt(shp[-1]))
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out
