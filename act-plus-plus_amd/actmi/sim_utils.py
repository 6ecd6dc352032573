"""Host-side helpers of the eval/train loops (reference utils.py:334-390), restated."""
import numpy as np
import torch


def sample_box_pose():
    """reference utils.py:334-343 (one np.random.uniform call of 3 values)."""
    ranges = np.vstack([[0.0, 0.2], [0.4, 0.6], [0.05, 0.05]])
    cube_position = np.random.uniform(ranges[:, 0], ranges[:, 1])
    return np.concatenate([cube_position, np.array([1, 0, 0, 0])])


def sample_insertion_pose():
    """reference utils.py:345-368 (two np.random.uniform calls: peg then socket)."""
    ranges = np.vstack([[0.1, 0.2], [0.4, 0.6], [0.05, 0.05]])
    peg_position = np.random.uniform(ranges[:, 0], ranges[:, 1])
    peg_pose = np.concatenate([peg_position, np.array([1, 0, 0, 0])])
    ranges = np.vstack([[-0.2, -0.1], [0.4, 0.6], [0.05, 0.05]])
    socket_position = np.random.uniform(ranges[:, 0], ranges[:, 1])
    socket_pose = np.concatenate([socket_position, np.array([1, 0, 0, 0])])
    return peg_pose, socket_pose


def set_seed(seed):
    """reference utils.py:388-390."""
    torch.manual_seed(seed)
    np.random.seed(seed)


def compute_dict_mean(epoch_dicts):
    """reference utils.py:372-380."""
    result = {k: None for k in epoch_dicts[0]}
    n = len(epoch_dicts)
    for k in result:
        s = 0
        for d in epoch_dicts:
            s = s + d[k]
        result[k] = s / n
    return result


def draw_episode_poses(task_name: str, num_rollouts: int, seed: int = 1000):
    """All rollout poses drawn up-front in the reference's order (eval_bc: set_seed(1000) at :229, one
    sample_*_pose() per rollout at :324-327) so that a sharded eval sees exactly the poses the sequential
    reference loop would have drawn, independent of the rank count."""
    state = np.random.get_state()
    np.random.seed(seed)
    poses = []
    for _ in range(num_rollouts):
        if "sim_transfer_cube" in task_name:
            poses.append(sample_box_pose())
        elif "sim_insertion" in task_name:
            poses.append(np.concatenate(sample_insertion_pose()))
        else:
            poses.append(sample_box_pose())
    np.random.set_state(state)
    return poses


def shard_range(n: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of n episodes for this rank; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
