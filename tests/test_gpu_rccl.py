"""GPU: RCCL really executes (`ncclAllReduce` on the step stream).  A one-GPU box cannot hold two
RCCL ranks, so both routes run with ONE rank pushed through the collective branch, where the collectives must change
nothing:

  * torch.distributed backend "nccl" (== RCCL), world_size 1, dist.FORCE_COLLECTIVE: librccl is loaded, a communicator
    is created, the f64 all_gather_into_tensor of the advantage moments and the flat gradient all_reduce between the
    two graph replays of every optimizer step run;
  * the library's own route (include/fortattack.h fa_adv_allreduce / fa_grad_allreduce, dist.LibraryExchange): RCCL
    opened by libfortattack_hip.so with dlopen, no process group at all.

Checked against the no-collective path: advantage mean / std bit for bit; the update to the rounding of the one
division that moves (the un-normalised gradients are divided by the mask mean after the exchange instead of inside
the kernel).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _collector(fa, exchange=None):
    """Open-loop rollout + GAE + advantage statistics (bench.py's hot path) -> mean, std as numpy."""
    from emergent_multiagent_strategies_amd.dist import gae_adv_mean_std
    E, G, A, T = 512, 3, 3, 32
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 15, base_seed=5)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    g = torch.Generator(device="cuda").manual_seed(9)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda", generator=g))
    st.value_preds.copy_(torch.randn(st.value_preds.shape, device="cuda", generator=g))
    eng.collect_reset()
    eng.collect_rollout(0, T)
    mean, std = gae_adv_mean_std(eng, 0.99, 0.95, exchange=exchange)
    torch.cuda.synchronize()
    return mean.cpu().numpy().copy(), std.cpu().numpy().copy()


def _one_call_tail(fa, ex):
    """fa_gae_allreduce_normalize (the several-rank collector tail as ONE library call: scan + moments, ncclAllGather,
    merge + normalisation) eagerly and CAPTURED in a hipGraph together with fa_collect_rollout -- what `bench.py --gpus N
    [--graph-hot-path]` enqueues per rollout -- against fa_gae_normalize (the one-rank tail) from the same state, two
    rollouts each: advantages, mean, std."""
    E, G, A, T = 640, 3, 3, 32
    N = G + A
    g = torch.Generator(device="cuda").manual_seed(11)
    acts = torch.randint(0, 8, (T, E, N, 1), device="cuda", generator=g)
    vals = torch.randn((T + 1, E, N, 1), device="cuda", generator=g)

    def fresh():
        eng = fa.BatchedFortAttack(E, G, A, 15, base_seed=6)
        st = fa.JointRolloutStorage(T, E, N, device="cuda")
        eng.bind_storage(st)
        st.actions.copy_(acts)
        st.value_preds.copy_(vals)
        eng.collect_reset()
        return eng, st

    res = {}
    eng, st = fresh()                                           # reference: no collective
    ref = []
    for _ in range(2):
        eng.collect_rollout(0, T)
        adv, _, mean, std = eng.gae_normalize(0.99, 0.95)
        torch.cuda.synchronize()
        ref.append((adv.clone(), mean.clone(), std.clone(), st.returns.clone()))
    eng, st = fresh()                                           # eager: one C call per rollout
    ok = True
    for k in range(2):
        eng.collect_rollout(0, T)
        adv, mean, std = ex.gae_allreduce_normalize(eng, 0.99, 0.95)
        torch.cuda.synchronize()
        ok &= bool(torch.equal(adv, ref[k][0]) and torch.equal(mean, ref[k][1]) and torch.equal(std, ref[k][2])
                   and torch.equal(st.returns, ref[k][3]))
    res["eager"] = ok
    eng, st = fresh()                                           # captured: rollout + tail + the collective in ONE graph
    out = torch.empty((T, E, N, 1), device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        eng.collect_rollout(0, T)
        ex.gae_allreduce_normalize(eng, 0.99, 0.95, out=out)
    ok = True
    for k in range(2):
        graph.replay()
        torch.cuda.synchronize()
        ok &= bool(torch.equal(out, ref[k][0]) and torch.equal(ex._mean, ref[k][1]) and torch.equal(ex._std, ref[k][2])
                   and torch.equal(st.returns, ref[k][3]))
    res["captured"] = ok
    return res


def _learner(fa, exchange="torch"):
    """Two collect + update rounds of the closed-loop learner (hidden_dim 128, hipGraphs, fused update)."""
    torch.manual_seed(0)
    eng = fa.BatchedFortAttack(128, 3, 3, 10, base_seed=3)
    L = fa.BatchedLearner(eng, num_steps=16, num_mini_batch=4, ppo_epoch=2, use_graph=True, update_backend="fused",
                          exchange=exchange)
    torch.manual_seed(100)
    L.reset()
    L.collect()
    acts = L.storage.actions.cpu().numpy().copy()
    losses = L.update().cpu().numpy()                   # compared: one update (8 Adam steps per team) from the same rollout
    params = [fp.pflat.cpu().numpy().copy() for fp in L._flat]
    L.after_update()
    L.collect()                                         # a second round on the captured graphs
    assert bool(torch.isfinite(L.update()).all())
    torch.cuda.synchronize()
    steps = [v for k, v in L._update_graphs.items() if isinstance(k, tuple)]
    return L, params, losses, acts, all(s.g2 is not None for s in steps)


def _loaded_rccl():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})


def _worker(route, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.path.insert(0, ROOT)
        torch.cuda.set_device(0)
        import emergent_multiagent_strategies_amd as fa
        from emergent_multiagent_strategies_amd import dist as fdist
        out = {}
        out["plain_ms"] = _collector(fa)                            # no process group: no collective
        _, out["plain_p"], out["plain_l"], out["plain_a"], two = _learner(fa)
        assert not two
        if route == "torch-nccl":
            import torch.distributed as dist
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            fdist.FORCE_COLLECTIVE = True
            assert fdist.exchanging()
            out["coll_ms"] = _collector(fa)
            _, out["coll_p"], out["coll_l"], out["coll_a"], two = _learner(fa)
            out["ranks"], out["backend"] = dist.get_world_size(), dist.get_backend()
            # a plain f64 all-gather and f32 all-reduce on this communicator, for the record
            x = torch.arange(18, dtype=torch.float64, device="cuda")
            y = torch.empty_like(x)
            dist.all_gather_into_tensor(y, x)
            z = torch.ones(1000, device="cuda")
            dist.all_reduce(z)
            torch.cuda.synchronize()
            assert torch.equal(x, y) and float(z.sum()) == 1000.0
            dist.destroy_process_group()
        else:
            ex = fdist.LibraryExchange("cuda:0")
            out["ranks"], out["backend"] = ex.ranks(), fa._lib.load().fa_rccl_library().decode()
            out["coll_ms"] = _collector(fa, exchange=ex)
            z = torch.full((1000,), 2.0, device="cuda")
            ex.all_reduce_(z)
            torch.cuda.synchronize()
            assert float(z.sum()) == 2000.0
            out["one_call"] = _one_call_tail(fa, ex)
            ex.close()
            L, out["coll_p"], out["coll_l"], out["coll_a"], two = _learner(fa, exchange="rccl")
            assert L._exch.ranks() == 1
        out["two_graphs"] = two
        out["rccl_maps"] = _loaded_rccl()
        q.put(("ok", out))
    except Exception as exc:   # the parent asserts on the message
        import traceback
        q.put(("error", traceback.format_exc()))


@pytest.mark.parametrize("route", ["torch-nccl", "library"])
def test_one_rank_through_rccl_changes_nothing(route):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(route, _free_port(), q))
    p.start()
    status, out = q.get(timeout=900)
    p.join(timeout=120)
    assert status == "ok", out
    assert out["rccl_maps"], "librccl was not mapped into the process"
    assert out["ranks"] == 1 and (out["backend"] == "nccl" if route == "torch-nccl" else "rccl" in out["backend"])
    assert out["two_graphs"]                                        # every optimizer step: graph 1 -> all-reduce -> graph 2
    if route == "library":                                          # the one-call tail, eager and inside a hipGraph
        assert out["one_call"] == {"eager": True, "captured": True}, out["one_call"]
    # the advantage statistics through the all-gather + merge: the same bits
    assert np.array_equal(out["plain_ms"][0], out["coll_ms"][0]) and np.array_equal(out["plain_ms"][1], out["coll_ms"][1])
    # the same rollout (sampling does not depend on the exchange) ...
    assert np.array_equal(out["plain_a"], out["coll_a"])
    # ... and the same update up to the moved division: 8 Adam steps of 1e-4 per team
    assert np.abs(out["plain_l"] - out["coll_l"]).max() < 2e-4 * max(1.0, np.abs(out["plain_l"]).max())
    worst = max(np.abs(a - b).max() for a, b in zip(out["plain_p"], out["coll_p"]))
    assert worst < 4e-4, worst


def test_bench_exchange_on_rccl_with_one_rank():
    """bench.py's several-rank hot path -- GAE, local moments, all-gather, merge, normalisation -- executed on RCCL in a world of
    ONE rank (`--force-collective`), here in its two-stream form (`--two-stream-tail`: everything behind the GAE scan on a
    second stream under the next rollout): the line must report one RCCL rank, and the two-stream form must have left exactly
    the advantages of the one-stream form."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collective", "--two-stream-tail", "--backend", "nccl",
                          "--envs", "1024", "--rollout", "32", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-5v5",
                          "--no-esweep", "--closed-loop-rollouts", "2", "--closed-loop-updates", "1"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = r["collective"]
    assert c["rccl_ranks"] == 1 and c["backend"] == "nccl" and c["forced_on_one_rank"] is True
    assert c["second_stream_exchange_equals_one_stream"] is True
    assert r["steps"] == 10 and r["value"] > 0 and r["closed_loop"]["train_env_steps_per_s"] > 0


@pytest.mark.parametrize("graph", [False, True])
def test_bench_library_exchange_one_call_per_rollout(graph):
    """bench.py's DEFAULT several-rank hot path (`--exchange auto` with backend nccl): the whole collector tail of a rollout --
    scan + moments, ncclAllGather on the library's own communicator, merge + normalisation -- is ONE C call on the launch
    stream (no torch.distributed call per rollout), optionally captured with the rollout in one hipGraph; it must leave the
    advantages the torch route leaves, and the rank pins itself to its slice of the GPU-local cores."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collective", "--backend", "nccl",
                          "--envs", "1024", "--rollout", "32", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-5v5",
                          "--no-esweep", "--no-closed-loop", "--no-live-traffic"] + (["--graph-hot-path"] if graph else []),
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = r["collective"]
    assert c["rccl_ranks"] == 1 and c["library_rccl_ranks"] == 1 and c["exchange_route"].startswith("library")
    assert c["library_exchange_equals_torch_route"] is True and c["hot_path_in_one_graph"] is graph
    b = c["rank_binding"][0]
    assert b["pin"] is not None and b["pin"]["pinned_to"] and b["pin"]["pinned_to"] >= 1
    assert r["steps"] == 10 and r["value"] > 0 and r["roofline"]["avg_launch_us"] > 0
