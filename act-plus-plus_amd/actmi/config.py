"""ACT policy configuration.

One dataclass replaces the reference's two argparse layers
(reference imitate_episodes.py:78-94 builds ``policy_config``; detr/main.py:12-89
supplies model defaults that are merged by ``setattr`` at detr/main.py:96-97).
The reference re-parses ``sys.argv`` inside the policy constructor (detr/main.py:93-94);
this build never touches argv below the CLI.
"""
from dataclasses import dataclass, field, asdict
from typing import List, Optional

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # reference policy.py:268-271
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass
class ACTConfig:
    # keys the reference passes through policy_config (imitate_episodes.py:78-94)
    lr: float = 1e-5
    num_queries: int = 100            # chunk_size
    kl_weight: int = 10
    hidden_dim: int = 512
    dim_feedforward: int = 3200
    lr_backbone: float = 1e-5
    backbone: str = "resnet18"
    enc_layers: int = 4
    dec_layers: int = 7
    nheads: int = 8
    camera_names: List[str] = field(default_factory=lambda: ["top", "left_wrist", "right_wrist", "angle"])
    vq: bool = False
    vq_class: Optional[int] = None
    vq_dim: Optional[int] = None
    action_dim: int = 16
    no_encoder: bool = False
    # detr/main.py defaults
    weight_decay: float = 1e-4        # detr/main.py:17
    dropout: float = 0.1              # detr/main.py:45
    pre_norm: bool = False            # detr/main.py:51 (store_true, never set)
    position_embedding: str = "sine"  # detr/main.py:29
    # reference hard-codes state_dim=7 in detr_vae.py:345 while sim qpos is 14
    # (imitate_episodes.py:71); here it is a field (SURVEY §2.1 fork drift)
    state_dim: int = 14
    latent_dim: int = 32              # detr_vae.py:79
    # image geometry (reference: 480x640 from sim_env.py:110-112)
    image_h: int = 480
    image_w: int = 640
    # resnet stem width; 64 for resnet18. Reduced only by tiny test configs.
    base_width: int = 64

    @property
    def latent_in_dim(self) -> int:
        """in_features of latent_out_proj: the VQ code size when vq, else latent_dim (detr_vae.py:59-62)."""
        return self.vq_class * self.vq_dim if self.vq else self.latent_dim

    @property
    def latent_proj_dim(self) -> int:
        """out_features of latent_proj (detr_vae.py:50-53)."""
        return self.vq_class * self.vq_dim if self.vq else 2 * self.latent_dim

    @property
    def num_cams(self) -> int:
        return len(self.camera_names)

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.nheads

    @property
    def feat_hw(self):
        """Spatial size of the layer4 map (stride 32 with the resnet padding rules)."""
        def down(x, k, s, p):
            return (x + 2 * p - k) // s + 1
        h, w = self.image_h, self.image_w
        h, w = down(h, 7, 2, 3), down(w, 7, 2, 3)      # conv1
        h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)      # maxpool
        for _ in range(3):                              # layer2..4 first blocks
            h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)
        return h, w

    @property
    def num_tokens(self) -> int:
        fh, fw = self.feat_hw
        return 2 + self.num_cams * fh * fw

    def validate(self):
        if self.backbone != "resnet18":
            raise NotImplementedError("only resnet18 is on the accelerated path (reference imitate_episodes.py:73)")
        if self.vq:
            # VQ-ACT (detr_vae.py:50-60): the latent is a [vq_class x vq_dim] one-hot code.  Inference with a given
            # `vq_sample` is on the accelerated path; VQ training and the latent prior model are not.
            if not self.vq_class or not self.vq_dim or self.vq_class <= 0 or self.vq_dim <= 0:
                raise ValueError("vq needs positive vq_class and vq_dim")
            if (self.vq_class * self.vq_dim) % 4:
                raise ValueError("vq_class * vq_dim must be a multiple of 4")
        if self.pre_norm:
            raise NotImplementedError("pre_norm is never enabled by the reference CLI")
        if self.hidden_dim % self.nheads:
            raise ValueError("hidden_dim must be divisible by nheads")
        return self

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def from_policy_config(d: dict) -> "ACTConfig":
        """Build from the dict ``imitate_episodes.main`` hands to ``ACTPolicy`` (reference policy.py:244-257)."""
        known = {f for f in ACTConfig.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in known}
        if "camera_names" in kw:
            kw["camera_names"] = list(kw["camera_names"])
        return ACTConfig(**kw).validate()


def tiny_config(**over) -> ACTConfig:
    """Small configuration used by fast tests and golden fixtures."""
    kw = dict(num_queries=8, hidden_dim=64, dim_feedforward=128, enc_layers=2, dec_layers=2, nheads=4,
              camera_names=["a", "b"], action_dim=16, state_dim=14, image_h=64, image_w=96, base_width=8)
    kw.update(over)
    return ACTConfig(**kw).validate()
