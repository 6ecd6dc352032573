"""Drop-in for the reference's ``policy.py:ACTPolicy`` on the MI355X-native path.

Same constructor argument (the ``policy_config`` dict built at reference imitate_episodes.py:78-94), same call
signature ``policy(qpos, image, actions=None, is_pad=None, vq_sample=None, depth_img=None, pointcloud=None)``
(reference policy.py:264), same ``configure_optimizers / serialize / deserialize / cuda / eval / train``
surface that ``imitate_episodes.py`` uses.  All arithmetic runs in libactmi (hand-written HIP, C ABI); there is no
torch.nn module underneath and no CPU fallback.
"""
from collections import OrderedDict

import torch

from actmi.config import ACTConfig
from actmi.engine import ACTEngine


class _LoadStatus:
    def __init__(self, missing, unexpected):
        self.missing_keys, self.unexpected_keys = missing, unexpected

    def __repr__(self):
        if not self.missing_keys and not self.unexpected_keys:
            return "<All keys matched successfully>"
        return f"_IncompatibleKeys(missing_keys={self.missing_keys}, unexpected_keys={self.unexpected_keys})"


class ACTPolicy:
    """reference policy.py:243-348."""

    def __init__(self, args_override: dict, max_batch: int = None, device: str = "cuda:0", init_seed: int = 0):
        # reference policy.py:247-249: use_depth / use_pcd default False (the fork's depth_camera_names lookup
        # raises KeyError with the stock config; the intent is "absent")
        self.use_depth = args_override.get("use_depth", False)
        self.use_pcd = args_override.get("use_pcd", False)
        self.depth_camera_names = args_override.get("depth_camera_names", None)
        if self.use_depth or self.use_pcd:
            raise NotImplementedError("depth / point-cloud inputs are outside the accelerated ACT path")
        self.cfg = ACTConfig.from_policy_config(args_override)
        self.kl_weight = args_override["kl_weight"]
        self.vq = args_override.get("vq", False)
        mb = max_batch or int(args_override.get("max_batch", 8))
        self.model = ACTEngine(self.cfg, max_batch=mb, device=device)
        # random init of the reference architecture (the ImageNet fetch of backbone.py:121-124 cannot run offline)
        from actmi.weights import generate_state_dict
        self.model.load_state_dict(generate_state_dict(self.cfg, seed=init_seed))
        self.training = True
        self.optimizer = None
        print(f"KL Weight {self.kl_weight}")
        print(f"Use Depth: {self.use_depth}")

    def __call__(self, qpos, image, actions=None, is_pad=None, vq_sample=None, depth_img=None, pointcloud=None):
        if actions is not None:
            raise NotImplementedError("training call (actions given) is not built in this version of libactmi")
        # inference: ImageNet normalisation (policy.py:268-272) is fused into the conv1 loader
        return self.model.forward_infer(qpos, image)

    # ---- nn.Module-like surface used by imitate_episodes.py ---------------------------------------
    def cuda(self):
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def parameters(self):
        return iter(())

    def configure_optimizers(self):
        return self.optimizer

    def serialize(self):
        """state_dict with the reference's key names (``model.`` prefix, policy.py:344-345)."""
        return self.model.state_dict(prefix="model.")

    def deserialize(self, model_dict):
        """policy.py:347-348 (returns the load status that eval_bc prints, imitate_episodes.py:248-249)."""
        missing, unexpected = self.model.load_state_dict(model_dict, prefix="model.", strict=True)
        return _LoadStatus(missing, unexpected)

    @torch.no_grad()
    def vq_encode(self, qpos, actions, is_pad):
        raise NotImplementedError("VQ-ACT is outside the accelerated path (SURVEY §8 f4)")
