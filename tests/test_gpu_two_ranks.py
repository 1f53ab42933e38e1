"""GPU, two processes sharing cuda:0 (gloo rendezvous on 127.0.0.1): the multi-rank path with the REAL
kernels under each rank -- env shards by env_offset, fused rollout + GAE + advantage moments on each
shard, one all-gather of N x 3 doubles, exact merge -- against one process running all the envs; then
the data-parallel BatchedLearner (closed-loop rollout, JointPPO update with the flat all-reduce) and
bench.py's self-launching --gpus 2 path.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(E, N, T):
    g = torch.Generator().manual_seed(77)
    acts = torch.randint(0, 8, (T, E, N, 1), generator=g)
    acts = torch.where(torch.rand(T, E, N, 1, generator=g) < 0.25, torch.full_like(acts, 7), acts)
    vals = torch.randn(T + 1, E, N, 1, generator=g)
    return acts, vals


def _run_shard(fa, E_total, lo, per, G, A, T, max_t, group_world):
    """Rollout + GAE + (all-rank) advantage statistics + normalisation for envs [lo, lo + per)."""
    from emergent_multiagent_strategies_amd.dist import gae_adv_mean_std
    N = G + A
    acts, vals = _inputs(E_total, N, T)
    eng = fa.BatchedFortAttack(per, G, A, max_t, base_seed=31, env_offset=lo)
    st = fa.JointRolloutStorage(T, per, N, device="cuda")
    eng.bind_storage(st)
    eng.collect_reset()
    st.actions.copy_(acts[:, lo:lo + per].cuda())
    st.value_preds.copy_(vals[:, lo:lo + per].cuda())
    eng.collect_rollout(0, T)
    mean, std = gae_adv_mean_std(eng, 0.99, 0.95)
    adv = eng.adv_normalize(mean, std)
    torch.cuda.synchronize()
    out = {k: getattr(st, k).cpu().numpy() for k in ("obs", "rewards", "masks", "done", "returns")}
    out.update(mean=mean.cpu().numpy(), std=std.cpu().numpy(), adv=adv.cpu().numpy())
    return out


def _collector_worker(rank, world, port, cfg, ref_path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)                                    # both ranks on the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emergent_multiagent_strategies_amd as fa
    E, G, A, T, max_t = cfg
    per = E // world
    got = _run_shard(fa, E, rank * per, per, G, A, T, max_t, world)
    ref = np.load(ref_path)
    sl = slice(rank * per, (rank + 1) * per)
    err = []
    for k in ("obs", "rewards", "masks", "done", "returns"):    # rows and GAE: bit for bit
        if not np.array_equal(got[k], ref[k][:, sl]):
            err.append(k)
    if np.abs(got["mean"] - ref["mean"]).max() > 1e-12 or np.abs(got["std"] / ref["std"] - 1).max() > 1e-12:
        err.append("mean/std")
    if np.abs(got["adv"] - ref["adv"][:, sl]).max() > 1e-6:
        err.append("adv")
    q.put((rank, err, got["mean"], got["std"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_kernels_equal_one_process(tmp_path):
    import emergent_multiagent_strategies_amd as fa
    cfg = (512, 3, 3, 48, 15)
    E, G, A, T, max_t = cfg
    ref = _run_shard(fa, E, 0, E, G, A, T, max_t, 1)            # one process, all envs, no collective
    assert int(ref["done"].sum()) > 0
    ref_path = str(tmp_path / "ref.npz")
    np.savez(ref_path, **ref)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_collector_worker, args=(r, 2, port, cfg, ref_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, _, _ in got:
        assert err == [], (rank, err)
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])   # same bits on every rank


def _learner_worker(rank, world, port, use_graph, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emergent_multiagent_strategies_amd as fa
    E, G, A, T = 64, 3, 3, 16
    torch.manual_seed(0)                                        # same initial policies on every rank
    eng = fa.BatchedFortAttack(E, G, A, 10, base_seed=3, env_offset=rank * E)
    L = fa.BatchedLearner(eng, num_steps=T, hidden_dim=32, num_mini_batch=4, ppo_epoch=2, use_graph=use_graph)
    init = [p.detach().clone() for pol in L.policies for p in pol.parameters()]
    torch.manual_seed(100 + rank)                               # different sampling / minibatches per rank
    L.reset()
    for _ in range(2):
        L.collect()
        losses = L.update()
        L.after_update()
    torch.cuda.synchronize()
    params = [p.detach().cpu().numpy() for pol in L.policies for p in pol.parameters()]
    moved = max(float((a - b).abs().max()) for a, b in zip(init, (p for pol in L.policies for p in pol.parameters())))
    q.put((rank, params, losses.cpu().numpy(), moved, L.storage.actions.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_learner_update_keeps_ranks_identical(use_graph):
    """BatchedLearner on two ranks (different env shards, different sampled actions, different minibatch
    permutations): after two collect + update rounds both ranks hold bit-identical parameters."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_learner_worker, args=(r, 2, port, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, p0, l0, moved0, a0), (_, p1, l1, moved1, a1) = got
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)
    assert np.array_equal(l0, l1) and np.isfinite(l0).all()
    assert moved0 > 1e-5 and not np.array_equal(a0, a1)         # they trained, on different data


# ---- f2: the FUSED data-parallel optimizer step (graph 1: fa_ppo_grad into the flat buffer -> all-reduce -> graph 2:
# un-normalise by the all-rank mask mean, clip, Adam; learner.GraphedPPOStep) -- rlcore/algo/ppo.py:116-204, :207-246 ----
_FUSED = dict(clip_param=0.2, ppo_epoch=2, num_mini_batch=4, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5)


def _fused_rows(seed, B, N, dev):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(B, N, 6, generator=g)
    obs[:, :, 0] = (torch.rand(B, N, generator=g) > 0.35).float()            # alive flags: uneven across the shards
    rows = (obs, torch.randint(0, 8, (B, N, 1), generator=g), torch.randn(B, N, 1, generator=g),
            torch.randn(B, N, 1, generator=g), -torch.rand(B, N, 1, generator=g) * 2, torch.randn(B, N, 1, generator=g))
    return tuple(r.to(dev).contiguous() for r in rows)


def _fused_sets(seed, B, epochs, nmb, dev):
    g = torch.Generator().manual_seed(seed)
    mb = B // nmb
    return [[p[k:k + mb].to(dev) for k in range(0, B, mb)] for p in (torch.randperm(B, generator=g) for _ in range(epochs))]


def _fused_policy_and_opt(fa, G, A):
    torch.manual_seed(11)
    pol = fa.MPNN(num_agents=G, num_opp_agents=A, hidden_dim=128, num_actions=8).cuda()
    return pol, torch.optim.Adam(pol.parameters(), lr=1e-4, capturable=True)


def _fused_run(fa, rows, first_set, sets, G, A):
    """One optimizer step on `first_set` (the clipped gradient left in the flat buffer is returned), then the whole
    update on `sets`, all through captured GraphedPPOSteps of the fused kernel."""
    from emergent_multiagent_strategies_amd.learner import joint_ppo_update
    from emergent_multiagent_strategies_amd import mpnn_pack
    pol, opt = _fused_policy_and_opt(fa, G, A)
    graphs = {"fused": True}
    one = dict(_FUSED, ppo_epoch=1)
    joint_ppo_update(pol, opt, slice(0, G), slice(G, G + A), rows, sampler=lambda ep: [first_set], graphs=graphs, **one)
    step = next(v for k, v in graphs.items() if k != "fused")
    assert step.fused
    fp = mpnn_pack.FlatPolicy.of(pol)
    grad1 = fp.gflat[:mpnn_pack.PF_FLOATS].clone()
    losses = joint_ppo_update(pol, opt, slice(0, G), slice(G, G + A), rows, sampler=lambda ep: sets[ep], graphs=graphs, **_FUSED)
    torch.cuda.synchronize()
    return step, grad1.cpu().numpy(), fp.pflat.detach().cpu().numpy().copy(), losses.cpu().numpy()


def _fused_worker(rank, world, port, B, G, A, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emergent_multiagent_strategies_amd as fa
    rows = _fused_rows(100 + rank, B, G + A, "cuda")                        # this rank's env shard
    sets = _fused_sets(200 + rank, B, _FUSED["ppo_epoch"], _FUSED["num_mini_batch"], "cuda")
    step, grad1, params, losses = _fused_run(fa, rows, sets[0][0], sets, G, A)
    q.put((rank, step.g2 is not None and step.world == 2, grad1, params, losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("G,A", [(3, 3), (5, 5)])
def test_two_rank_fused_update_equals_one_rank_on_the_union_minibatches(G, A):
    """The path an N-GPU run takes -- hidden_dim 128, use_graph, the fused kernel: graph 1 (fa_ppo_grad, un-normalised,
    into the flat gradient buffer with the loss sums and the rank's mask mean in its tail) -> all-reduce of the flat
    buffer -> graph 2 (divide by the world size and the ALL-rank mask mean, clip, Adam).  Two ranks with their own
    shards and index sets end with bit-identical parameters, and their first clipped gradient / losses / parameters
    equal one process stepping on the union minibatches (single-rank graph: normalised inside the kernel)."""
    import emergent_multiagent_strategies_amd as fa
    world, B = 2, 512
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, world, port, B, G, A, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, two0, g0, p0, l0), (_, two1, g1, p1, l1) = got
    assert two0 and two1                                                    # both took the g1 / all-reduce / g2 form
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1) and np.array_equal(l0, l1)
    # one process on the union: rows concatenated, minibatch k = rank 0's index set k + rank 1's (offset B)
    parts = [_fused_rows(100 + r, B, G + A, "cuda") for r in range(world)]
    rows = tuple(torch.cat([parts[0][k], parts[1][k]]).contiguous() for k in range(6))
    s0, s1 = [_fused_sets(200 + r, B, _FUSED["ppo_epoch"], _FUSED["num_mini_batch"], "cuda") for r in range(world)]
    union = [[torch.cat([a, b + B]) for a, b in zip(e0, e1)] for e0, e1 in zip(s0, s1)]
    step, gu, pu, lu = _fused_run(fa, rows, union[0][0], union, G, A)
    assert step.g2 is None
    scale = np.abs(gu).max()
    assert scale > 1e-4
    assert np.abs(gu - g0).max() < 2e-5 * scale, np.abs(gu - g0).max() / scale     # the same gradient (fp32 summation order)
    assert np.abs(lu - l0).max() < 2e-5 * max(1.0, np.abs(lu).max())
    # 1 + 8 Adam steps of 1e-4 (the sign of a near-zero gradient entry may differ: at most one step each)
    assert np.abs(pu - p0).max() < 4e-4


def _chains_worker(rank, world, port, together, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emergent_multiagent_strategies_amd as fa
    E, G, A, T = 128, 3, 3, 16
    torch.manual_seed(0)
    eng = fa.BatchedFortAttack(E, G, A, 10, base_seed=3, env_offset=rank * E)
    L = fa.BatchedLearner(eng, num_steps=T, num_mini_batch=4, ppo_epoch=2, use_graph=True, update_backend="fused")
    L._update_graphs["teams_together"] = together
    torch.manual_seed(100 + rank)
    L.reset()
    for _ in range(2):
        L.collect()
        losses = L.update()
        L.after_update()
    torch.cuda.synchronize()
    steps = [v for k, v in L._update_graphs.items() if isinstance(k, tuple)]
    ok = all(s.fused and s.g2 is not None for s in steps) and ("team_streams" in L._update_graphs) == together
    q.put((rank, ok, [fp.pflat.cpu().numpy() for fp in L._flat], losses.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_learner_fused_chains_equal_the_sequential_teams():
    """BatchedLearner at hidden_dim 128 with hipGraphs on two ranks: the two teams' update chains running concurrently,
    each with its own process group for its all-reduces, give bit-identical parameters on both ranks AND the same
    bits as the teams updated one after the other."""
    res = {}
    for together in (False, True):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_chains_worker, args=(r, 2, port, together, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        (_, ok0, p0, l0), (_, ok1, p1, l1) = got
        assert ok0 and ok1
        assert all(np.array_equal(a, b) for a, b in zip(p0, p1)) and np.array_equal(l0, l1)
        assert np.isfinite(l0).all()
        res[together] = (p0, l0)
    assert all(np.array_equal(a, b) for a, b in zip(res[False][0], res[True][0]))
    assert np.array_equal(res[False][1], res[True][1])


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts the ranks itself and prints ONE line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-devices", "--backend",
                          "gloo", "--envs", "512", "--rollout", "32", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--closed-loop-rollouts", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["value"] > 0
    assert r["roofline"]["frac"] > 0 and r["closed_loop"]["rollout_env_steps_per_s"] > 0
    assert r["closed_loop"]["train_env_steps_per_s"] > 0
    assert r["collective"]["ranks"] == 2 and r["collective"]["backend"] == "gloo"


def test_bench_eight_rank_rehearsal_on_one_gpu(tmp_path):
    """The 8-way form of BASELINE configs 4 / 5 as far as one GPU can rehearse it: `bench.py --gpus 8 --share-devices
    --backend gloo` -- eight ranks, eight env shards, the all-gather of the advantage moments over eight ranks, the
    per-team process groups and the fused two-chain update with event-ordered all-reduces -- must rendezvous, finish,
    report eight ranks in the collective and leave every rank with bit-identical parameters."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-devices", "--backend",
                          "gloo", "--envs", "512", "--rollout", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--closed-loop-rollouts", "2", "--closed-loop-updates", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 3 and r["steps_requested"] == 3 and r["value"] > 0
    assert r["collective"]["ranks"] == 8 and len(r["collective"]["rank_binding"]) == 8
    assert sorted(b["rank"] for b in r["collective"]["rank_binding"]) == list(range(8))
    assert r["closed_loop"]["ranks_hold_identical_parameters"] is True
    assert r["closed_loop"]["train_env_steps_per_s"] > 0
    rec = os.environ.get("FA_REHEARSAL_RECORD")
    if rec:
        open(rec, "w").write(lines[0] + "\n")
