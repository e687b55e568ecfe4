"""Multi-GPU use: one process per GPU (torchrun), envs sharded in contiguous blocks, global env ids so that the RNG
streams -- and therefore every trajectory -- do not depend on how the envs are split.  The ONLY collective on the path is
the all-gather of the per-rank success-metric vector (64 bytes) that the reference's training script computes from the
episode infos of all its sub-process envs (examples/train_rl_controller.py:51-66) to log progress and to drive the
curriculum (:80-85).  backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests."""
import numpy as np

REDUCE_FIELDS = ["episodes", "success_target0", "success_target1", "success_target2", "success_all",
                 "control_variation", "end_error0", "end_error1", "end_error2", "total_error0", "total_error1",
                 "total_error2", "success_time_frac0", "success_time_frac1", "success_time_frac2",
                 "success_time_frac_all"]


def shard(total_envs, rank, world_size):
    """(first global env id, number of envs) of a rank; contiguous blocks, remainder to the low ranks."""
    base, rem = divmod(int(total_envs), int(world_size))
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def local_device(rank=0):
    """GPU of this process: LOCAL_RANK when a launcher set it (multi-node: the global rank exceeds the GPUs of a node),
    otherwise rank modulo the GPUs visible."""
    import os
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return rank % n if n > 0 else rank


def make_sharded_env(config_path=None, total_envs=65536, rank=0, world_size=1, device=None, **kw):
    from .vec_env import FixedWingVecEnv
    first, n = shard(total_envs, rank, world_size)
    return FixedWingVecEnv(config_path, num_envs=n, device=local_device(rank) if device is None else device, env_id_base=first, **kw)


def gather_success(vec, group=None):
    """All-gathers the local sums of the episodes finished since the last call and returns the GLOBAL summary:
    {"episodes": n, "success": {target.., "all"}, "control_variation": mean, ...} (means over episodes)."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    dev = getattr(getattr(vec, "_mem", None), "device", None)
    use_dev = multi and dev is not None and dist.get_backend(group) == "nccl"
    local = None if use_dev else vec.reduce_success()
    if multi:
        # RCCL: the device-resident sums go straight into the all-gather, the only host read is the gathered result
        t = vec.reduce_success_device() if use_dev else torch.as_tensor(local, dtype=torch.float32, device="cpu")
        out = torch.empty(dist.get_world_size(group) * t.numel(), dtype=torch.float32, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        total = out.view(-1, t.numel()).sum(dim=0).cpu().numpy().astype(np.float64)
    else:
        total = np.asarray(local, dtype=np.float64)
    return summarize(total, vec.target_names)


def summarize(total, target_names):
    n = max(total[0], 1.0)
    names = list(target_names) + ["all"]
    res = {"episodes": int(round(total[0]))}
    res["success"] = {nm: total[1 + (k if nm != "all" else 3)] / n for k, nm in enumerate(names)}
    res["control_variation"] = {"all": total[5] / n}
    res["end_error"] = {nm: total[6 + k] / n for k, nm in enumerate(target_names)}
    res["total_error"] = {nm: total[9 + k] / n for k, nm in enumerate(target_names)}
    res["success_time_frac"] = {nm: total[12 + (k if nm != "all" else 3)] / n for k, nm in enumerate(names)}
    return res


class CurriculumSchedule(object):
    """The curriculum rule of the reference's training callback (examples/train_rl_controller.py:80-87): when the
    mean success of recent episodes exceeds the current level, level := min(2 * mean, 1), then a cooldown."""

    def __init__(self, level=0.25, cooldown=15):
        self.level, self.cooldown_steps, self.cooldown = level, cooldown, 0

    def update(self, vec, summary):
        if self.level >= 1 or summary["episodes"] == 0:
            return self.level
        if self.cooldown > 0:
            self.cooldown -= 1
            return self.level
        mean = summary["success"]["all"]
        if mean > self.level:
            self.level = min(mean * 2, 1)
            vec.set_curriculum_level(self.level)   # identical on every rank: summary is global
            self.cooldown = self.cooldown_steps
        return self.level
