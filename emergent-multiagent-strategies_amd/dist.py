"""Multi-GPU layer: env shards + the one exchange the hot path has.

Envs are independent, so rank r simply owns envs [r*E, (r+1)*E) (``env_offset`` of
BatchedFortAttack) and nothing is exchanged during a rollout.  The only cross-rank
quantity is the per-agent advantage mean / unbiased std of JointPPO.update
(rlcore/algo/ppo.py:121-123), which the reference computes over one process's T*P samples
and which must now cover every rank's samples.  What the learner and bench.py call is
``gae_adv_mean_std``: every rank computes its local moments (n, mean, M2) per agent
(fa_gae_moments), ONE all-gather (RCCL over xGMI, backend "nccl") moves N x 3 doubles per rank
-- latency bound, 144 B at 3v3 -- and fa_adv_merge combines the triples exactly
(Chan-Golub-LeVeque, in rank order: every rank gets the same bits).  ``two_pass_mean_std`` is the
two-all-reduce formulation (mean first, then squared deviations), kept as the reference form.
"""
import torch
import torch.distributed as dist


def _all_reduce_sum_(t, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def two_pass_mean_std(pass0, pass1, group=None):
    """pass0() -> (N,3) float64 {n, sum, *}; pass1(mean (N,)) -> (N,) float64 sum of squared
    deviations from `mean`.  Returns global (mean, unbiased std, n), all (N,) float64, on
    the tensors' device; no host synchronisation."""
    s0 = _all_reduce_sum_(pass0().clone(), group)
    n = s0[:, 0]
    mean = (s0[:, 1] / n).contiguous()
    ssd = _all_reduce_sum_(pass1(mean).contiguous().clone(), group)
    std = torch.sqrt(ssd / (n - 1)).contiguous()
    return mean, std, n


def adv_mean_std(eng, group=None):
    """Global per-agent advantage mean / unbiased std for the storage bound to `eng`.
    One rank: fa_adv_mean_std (four launches, no collective).  Several ranks: local two-pass
    moments, ONE all-gather of (N,3) doubles per rank, exact merge kernel (fa_adv_merge)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return eng.adv_mean_std()
    world = dist.get_world_size(group)
    buf = getattr(eng, "_adv_gather", None)
    if buf is None or buf.shape[0] != world:
        buf = eng._adv_gather = torch.zeros((world, eng.N, 3), dtype=torch.float64, device=eng.device)
        eng._adv_local = torch.zeros((eng.N, 3), dtype=torch.float64, device=eng.device)
    eng.adv_moments(out=eng._adv_local)
    dist.all_gather_into_tensor(buf.view(-1), eng._adv_local.view(-1), group=group)   # flat: backend-neutral
    return eng.adv_merge(buf)


def gae_adv_mean_std(eng, gamma=0.99, tau=0.95, group=None):
    """compute_returns (GAE) + the global per-agent advantage mean / unbiased std in one pass over
    the rollout buffers (fa_gae_moments: three launches); with several ranks additionally ONE all-gather
    of the local (N,3) moments and the exact merge kernel."""
    mom, mean, std = eng.gae_moments(gamma, tau)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return mean, std
    world = dist.get_world_size(group)
    buf = getattr(eng, "_adv_gather", None)
    if buf is None or buf.shape[0] != world:
        buf = eng._adv_gather = torch.zeros((world, eng.N, 3), dtype=torch.float64, device=eng.device)
    dist.all_gather_into_tensor(buf.view(-1), mom.view(-1), group=group)
    return eng.adv_merge(buf)


def merge_moments(gathered):
    """Host/torch restatement of fa_adv_merge (tests): gathered (W, N, 3) -> mean, std."""
    W, N, _ = gathered.shape
    n = torch.zeros(N, dtype=torch.float64)
    mean = torch.zeros(N, dtype=torch.float64)
    m2 = torch.zeros(N, dtype=torch.float64)
    for r in range(W):
        nr, mr, m2r = gathered[r, :, 0].cpu(), gathered[r, :, 1].cpu(), gathered[r, :, 2].cpu()
        nn = n + nr
        delta = mr - mean
        mean = mean + delta * (nr / nn)
        m2 = m2 + m2r + delta * delta * (n * nr / nn)
        n = nn
    return mean, torch.sqrt(m2 / (n - 1))


def shard_range(num_envs_total, rank, world):
    """Env index range of `rank` when `num_envs_total` envs are split evenly."""
    if num_envs_total % world:
        raise ValueError("num_envs_total must be divisible by the world size")
    per = num_envs_total // world
    return rank * per, per
