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

Two routes carry the exchange.  The default is torch.distributed (any backend; "nccl" is RCCL).  ``LibraryExchange``
is the in-library route of include/fortattack.h (fa_adv_allreduce / fa_grad_allreduce: RCCL opened by
libfortattack_hip.so itself, the collective enqueued on the caller's stream next to the kernels) -- what a consumer
without torch uses; here torch.distributed only hands the 128-byte communicator id to the other ranks.
``FORCE_COLLECTIVE`` makes a world of ONE rank take the collective branch as well (the all-gather / all-reduce run on
one rank and must change nothing): how the GPU tests execute RCCL on a one-GPU box.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

FORCE_COLLECTIVE = False


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def exchanging(group=None):
    """Whether the torch.distributed route runs its collectives: several ranks, or one rank under FORCE_COLLECTIVE."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVE


class LibraryExchange(object):
    """One RCCL communicator owned by libfortattack_hip.so (fa_rccl_comm_create) for this process's device.
    rank / world come from torch.distributed when it is initialised (its default group moves the id), else (0, 1):
    a single process needs no process group at all to push its exchange through RCCL."""

    def __init__(self, device, group=None):
        from . import _lib
        self._L = _lib
        lib = _lib.load()
        if not lib.fa_rccl_available():
            raise _lib.FaError("RCCL could not be opened by libfortattack_hip.so (set FA_RCCL_LIB to its path)")
        dev = torch.device(device)
        self.device = dev
        self.rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        self.world = world_size(group)
        uid = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(lib.fa_rccl_unique_id(uid), "fa_rccl_unique_id")
        if self.world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        comm = C.c_void_p()
        dev_index = dev.index if dev.index is not None else torch.cuda.current_device()   # an index-less "cuda" = the current device
        _lib.check(lib.fa_rccl_comm_create(C.byref(comm), self.world, uid, self.rank, dev_index), "fa_rccl_comm_create")
        self.comm = comm
        self._gather = None

    def ranks(self):
        return int(self._L.load().fa_rccl_comm_ranks(self.comm))

    def adv_mean_std(self, eng, moments):
        """ppo.py:121-123 over all ranks from this rank's (N,3) moments: fa_adv_allreduce on the current stream."""
        if self._gather is None:
            self._gather = torch.zeros((self.world, eng.N, 3), dtype=torch.float64, device=eng.device)
            self._mean = torch.zeros(eng.N, dtype=torch.float64, device=eng.device)
            self._std = torch.zeros(eng.N, dtype=torch.float64, device=eng.device)
        L = self._L
        L.check(L.load().fa_adv_allreduce(eng._h, moments.data_ptr(), self._gather.data_ptr(), self.comm, self._mean.data_ptr(),
                                          self._std.data_ptr(), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "fa_adv_allreduce")
        return self._mean, self._std

    def gae_allreduce_normalize(self, eng, gamma=0.99, tau=0.95, out=None):
        """The whole several-rank collector tail of one rollout as ONE library call on the current stream
        (fa_gae_allreduce_normalize: GAE scan + this rank's moments, the all-gather, merge + normalisation): what
        bench.py --gpus N enqueues per rollout -- no Python between the four device operations, capturable in a hipGraph.
        Returns (adv (T, E, N, 1) float32, mean (N,), std (N,)) over ALL ranks' samples."""
        if self._gather is None:
            self._gather = torch.zeros((self.world, eng.N, 3), dtype=torch.float64, device=eng.device)
            self._mean = torch.zeros(eng.N, dtype=torch.float64, device=eng.device)
            self._std = torch.zeros(eng.N, dtype=torch.float64, device=eng.device)
        if getattr(self, "_mom", None) is None:
            self._mom = torch.zeros((eng.N, 3), dtype=torch.float64, device=eng.device)
        if out is None:
            out = torch.empty((eng.storage.num_steps, eng.E, eng.N, 1), dtype=torch.float32, device=eng.device)
        L = self._L
        L.check(L.load().fa_gae_allreduce_normalize(eng._h, float(gamma), float(tau), self.comm, self._mom.data_ptr(),
                                                    self._gather.data_ptr(), out.data_ptr(), self._mean.data_ptr(),
                                                    self._std.data_ptr(),
                                                    C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "fa_gae_allreduce_normalize")
        return out, self._mean, self._std

    def all_reduce_(self, flat):
        """Sum `flat` (float32, contiguous) over the ranks in place, on the current stream."""
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        L = self._L
        L.check(L.load().fa_grad_allreduce(flat.data_ptr(), flat.numel(), self.comm,
                                           C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "fa_grad_allreduce")
        return flat

    def close(self):
        if self.comm:
            torch.cuda.synchronize(self.device)
            self._L.load().fa_rccl_comm_destroy(self.comm)
            self.comm = None


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
    if not exchanging(group):
        return eng.adv_mean_std()
    world = dist.get_world_size(group)
    buf = getattr(eng, "_adv_gather", None)
    if buf is None or buf.shape[0] != world:
        buf = eng._adv_gather = torch.zeros((world, eng.N, 3), dtype=torch.float64, device=eng.device)
        eng._adv_local = torch.zeros((eng.N, 3), dtype=torch.float64, device=eng.device)
    eng.adv_moments(out=eng._adv_local)
    dist.all_gather_into_tensor(buf.view(-1), eng._adv_local.view(-1), group=group)   # flat: backend-neutral
    return eng.adv_merge(buf)


def gae_adv_mean_std(eng, gamma=0.99, tau=0.95, group=None, exchange=None):
    """compute_returns (GAE) + the global per-agent advantage mean / unbiased std in one pass over
    the rollout buffers (fa_gae_moments: two launches up to 32 768 columns); with several ranks additionally ONE all-gather
    of the local (N,3) moments and the exact merge kernel -- through torch.distributed, or inside the library
    when `exchange` is a LibraryExchange."""
    mom, mean, std = eng.gae_moments(gamma, tau)
    if exchange is not None:
        return exchange.adv_mean_std(eng, mom)
    if not exchanging(group):
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


# ---- rank -> CPU binding -------------------------------------------------------------------------------------------
# Every rollout ends in a cross-rank exchange, so the slowest rank's host thread paces all GPUs: a rank whose thread
# migrates between sockets, or shares cores with seven other Python ranks, stalls the job.  Each rank therefore pins
# itself to its own slice of the cores that are local to its GPU (the PCI device's NUMA node from sysfs).

def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def choose_cpus(allowed, gpu_local, slot, slots):
    """The CPUs rank `slot` of the `slots` ranks that share `gpu_local` should run on: an even, disjoint split of
    (allowed & gpu_local) -- of `allowed` alone when the GPU's local list is unknown or lies outside the cgroup.
    Pure function (tests/test_dist_cpu.py).  Returns (cpus, source)."""
    allowed = sorted(set(allowed))
    local = sorted(set(allowed) & set(gpu_local or []))
    pool, source = (local, "numa-local") if local else (allowed, "allowed-split")
    slots = max(1, int(slots))
    per = len(pool) // slots
    if per < 1:                                     # more ranks than cores: share the pool
        return pool, source + " (shared: fewer cores than ranks)"
    slot = int(slot) % slots
    return pool[slot * per:(slot + 1) * per], source


def gpu_local_cpulist(device_index):
    """The cpulist sysfs reports as local to the GPU's PCI device ([] when it cannot be told)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            return parse_cpulist(f.read()), bdf
    except Exception:
        return [], None


def pin_rank_to_gpu_local_cpus(device_index, local_rank, local_world, peers_local_lists=None):
    """os.sched_setaffinity for this process (see above).  `peers_local_lists`: every local rank's GPU-local cpulist in
    local-rank order (so that ranks whose GPUs share a NUMA node split it); None = assume all ranks share one list.
    Returns a record for the bench line's `rank_binding`."""
    before = sorted(os.sched_getaffinity(0))
    local, bdf = gpu_local_cpulist(device_index)
    if peers_local_lists:
        same = [r for r, l in enumerate(peers_local_lists) if sorted(l) == sorted(local)]
        slot, slots = (same.index(local_rank), len(same)) if local_rank in same else (local_rank, local_world)
    else:
        slot, slots = local_rank, local_world
    cpus, source = choose_cpus(before, local, slot, slots)
    rec = {"pci": bdf, "gpu_local_cpus": len(local), "source": source, "slot": "%d/%d" % (slot, slots)}
    try:
        os.sched_setaffinity(0, cpus)
        rec["pinned_to"] = len(cpus)
    except OSError as exc:
        rec["pinned_to"] = None
        rec["error"] = str(exc)
    return rec
