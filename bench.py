#!/usr/bin/env python
"""bench.py -- env-steps/s of the FortAttack hot path on MI355X (BASELINE.json metric).

One bench "step" = one pass of the hot path over one batch: a T=128-step rollout of
E=4096 envs per GPU (FortAttack 3v3, open-loop uniform-random actions already resident in
the rollout buffers) through the HIP step kernel with its fused collector write
(obs / rewards / masks / done rows of the RolloutStorage layout), followed by the GAE scan, the
one-pass fp64 advantage moments and the advantage normalisation (the only cross-GPU exchange: one
all-gather of N x 3 doubles when --gpus > 1).  Inputs are in HBM before the timed region starts.
That is BASELINE config 2 ("random policy, step-kernel only") plus the collector: `value`.

The same JSON line carries a second record, `closed_loop` = BASELINE config 3 (config 4 when
--gpus > 1): the MPNN actor-critic in the loop (fa_collect_act: the fused forward + sampling kernel; the
PyTorch module is its definition and fallback) -- per env-step one two-team forward, sampling and one
fa_collect_step, then V(obs[T]), GAE and the advantage statistics -- and the PPO update after it.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8                      # spawns 8 ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~4.7 TB/s measured copy


MFMA_BF16_PEAK_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_bf16: 16 x the fp32 MFMA rate (MI355X_MICROARCH.md: ~2.5 PF dense)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32: 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz (MI355X_MICROARCH.md)


def policy_flops_per_row(n, m):
    """Algorithmic flops of the MPNN actor-critic forward (mpnn.py:117-205, folded as DESIGN.md 3.5) per (env, agent)
    row of a team of n facing m opponents: encoders 6x64 (own; the opponents' m rows shared by the n own rows), the
    opponent stage (64x64 projection, m scores + mix over 64, 64x64 output), three rounds of (128x128 projection,
    n-1 scores + mix over 128, 256x128 update), the two 128x128 heads, 128x8 logits + 128x1 value.  2 flop per MAC."""
    mac = 6 * 64 + 6 * 64 * m / n + 64 * 64 + 2 * m * 64 + 64 * 64
    mac += 3 * (128 * 128 + 2 * (n - 1) * 128 + 256 * 128)
    mac += 128 * 256 + 128 * 8 + 128
    return 2.0 * mac


def train_flops_per_row(n, m):
    """fa_train_kernel: forward as above + the backward (two GEMMs per forward GEMM: dX and dW) + the
    recomputed g = h A_m of the three rounds."""
    fwd = policy_flops_per_row(n, m)
    return 3.0 * fwd + 2.0 * 3 * 128 * 128


def measured_stream(dev):
    """What plain streaming kernels sustain on THIS box in THIS run (context for `peak`, the 8 TB/s vendor figure the
    roofline is priced against): torch's elementwise kernels over 1 GiB float32 tensors -- copy (r + w), triad
    (2 r + w), read (a sum) -- best of 5, bytes moved / time.  The round-1 record of the hand-written streaming kernels
    (tools/ubench_hbm.hip) rides along, labelled as what it is."""
    import torch
    out = {"source": "live: torch elementwise kernels over 1 GiB f32 tensors, best of 5, this run"}
    try:
        n = 1 << 28
        a = torch.empty(n, device=dev).normal_()
        b = torch.empty_like(a).normal_()
        c = torch.empty_like(a)

        def best(fn, nbytes):
            fn()
            torch.cuda.synchronize()
            t = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) * 1e-3)
            return nbytes / min(t) / 1e9

        out["copy"] = best(lambda: c.copy_(a), 8.0 * n)
        out["triad"] = best(lambda: torch.add(a, b, alpha=3.0, out=c), 12.0 * n)
        out["read"] = best(lambda: a.sum(), 4.0 * n)
        del a, b, c
        torch.cuda.empty_cache()
    except Exception as exc:
        out["error"] = repr(exc)
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_stream.json")))
        out["committed_round1_record"] = {"source": "profiles/r01_hbm_stream.json (tools/ubench_hbm.hip, round 1; NOT measured in this run)",
                                          "copy": max(v for k, v in d.items() if k.startswith("copy")),
                                          "triad": max(v for k, v in d.items() if k.startswith("triad")),
                                          "read": max(v for k, v in d.items() if k.startswith("read"))}
    except Exception:
        pass
    return out



def git_blob_hash(path):
    """git's blob id of a tracked file (so that a committed PMC record quoted in the line can be pinned), or None."""
    try:
        return subprocess.run(["git", "hash-object", path], capture_output=True, text=True, cwd=ROOT, timeout=20).stdout.strip() or None
    except Exception:
        return None


def live_traffic(kernel_prefix, args, budget_s=150, device=0):
    """HBM bytes per launch of the dominant kernel, measured NOW: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
    share one on gfx950) over `bench.py --pmc-probe` -- this run's workload, 3 + 12 rollout launches, nothing else -- each with
    --kernel-trace only, as MI355X_MICROARCH.md prescribes; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters, the
    gfx950 FETCH_SIZE x2 correction).  Returns (bytes or None, how)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.isfile("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found on this box"
    vals = {}
    t_start = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="fa_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.perf_counter() - t_start)
            if left < 20:
                return None, "PMC passes ran out of their %d s budget" % budget_s
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-f", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-probe", "--envs", str(args.envs), "--rollout", str(args.rollout),
                   "--guards", str(args.guards), "--attackers", str(args.attackers), "--probe-device", str(device)] + (
                       ["--no-counters"] if args.no_counters else [])
            # own session: when the budget runs out the WHOLE group goes (rocprofv3 and the profiled python under it --
            # a surviving grandchild would keep the GPU busy under the sections timed after this one)
            proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp",
                                    env=dict(os.environ, TMPDIR="/tmp"), start_new_session=True)
            try:
                _, err = proc.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.communicate()
                return None, "PMC pass %s ran out of the %d s budget (its process group was killed)" % (ctr, budget_s)
            r = type("R", (), {"returncode": proc.returncode, "stderr": err})
            acc = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr and kernel_prefix in row.get("Kernel_Name", ""):
                        acc.append(float(row["Counter_Value"]))
            if not acc:
                return None, "rocprofv3 --pmc %s produced no rows for %s (rc %d): %s" % (ctr, kernel_prefix, r.returncode, (r.stderr or "")[-200:])
            vals[ctr] = (sum(acc) / len(acc), len(acc))
    except Exception as exc:
        return None, "live PMC passes failed: %r" % (exc,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    b = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return b, ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over bench.py --pmc-probe in this run, "
               "%d / %d launches; (2 * FETCH_SIZE + WRITE_SIZE) * 1024" % (vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]))


def pmc_probe(args):
    """The workload of live_traffic's rocprofv3 passes: the bench's rollout launch, 3 + 12 times, nothing else."""
    import torch
    import emergent_multiagent_strategies_amd as fa
    E, G, A, T = args.envs, args.guards, args.attackers, args.rollout
    d = int(args.probe_device)
    dev = "cuda:%d" % d
    torch.cuda.set_device(d)
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, device=d, track_counters=not args.no_counters)
    st = fa.JointRolloutStorage(T, E, G + A, device=dev)
    eng.bind_storage(st)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
    eng.collect_reset()
    for _ in range(15):
        eng.collect_rollout(0, T)
    torch.cuda.synchronize()


def issue_model(G, A, E, T, launch_s, clock_khz, n_cus, variant=None):
    """roofline.secondary: the issue model of wave 0's step loop generated at build time from the assembly
    (emergent-multiagent-strategies_amd/isa_model.py -> csrc/fa_isa_model.json) next to the cycles per step measured in this run."""
    # this build's model when the objects are here (csrc/_obj: not shipped to a GPU box), else the tracked record of the same build
    path = os.path.join(ROOT, "emergent-multiagent-strategies_amd", "csrc", "_obj", "fa_isa_model.json")
    if not os.path.isfile(path):
        path = os.path.join(ROOT, "emergent-multiagent-strategies_amd", "csrc", "fa_isa_model.json")
    try:
        m = json.load(open(path))
    except Exception as exc:
        return {"error": "no issue model beside the library: %r" % (exc,)}
    epw = 64 // (G + A)
    grid = (E + epw - 1) // epw
    # which build was launched is the LIBRARY's decision (fa_step_variant: "... /3 per CU"), not re-derived from this
    # device's CU count (the dispatch's threshold is a constant of the build)
    three = ("/3 per CU" in variant) if variant is not None else (grid > 2 * n_cus)
    tag = ("%dv%d" % (G, A)) + ("_3percu" if three else "")
    k = m.get(tag)
    if not k or "wave0" not in k.get("loops", {}):
        return {"error": "the issue model has no entry %r" % tag}
    w0 = k["loops"]["wave0"]
    measured = launch_s / T * clock_khz * 1e3
    return {"bound": "instruction issue of the state's wave (wave 0) of fa_step_pipe_kernel: one instruction per 4 cycles, fp64 at full "
                     "rate on gfx950; the other waves' loops are listed for the same build",
            "generated_from": "csrc/_obj/fa_step_pipe.s at build time (isa_model.py), symbol " + k["symbol"],
            "instrs_per_step": w0["instructions_per_step"], "fp64_instrs_per_step": w0["fp64_instructions_per_step"],
            "loop_instructions_static": w0["instructions"], "restage_path_instructions": w0["restage_path_instructions"],
            "model_cycles_per_step": w0["model_cycles_per_step"], "measured_cycles_per_step": measured,
            "frac": w0["model_cycles_per_step"] / measured if measured > 0 else None,
            "shader_clock_khz": clock_khz,
            "other_waves_loop_instructions": {r: v["instructions"] for r, v in k["loops"].items() if r != "wave0"},
            "spill_reloads_in_step_loops": {r: v.get("spill_reloads", 0) for r, v in k["loops"].items()},
            "vgprs": k["vgprs"], "sgpr_spill_count": k["sgpr_spill_count"], "scratch_bytes": k["scratch_bytes"],
            "workgroups": grid, "cus": n_cus,
            "cus_with_two_workgroups": max(0, min(grid, 2 * n_cus) - n_cus) if not three else None,
            "what_the_rest_is": "two workgroup barriers per step (last arrival -> release ~ 200-250 cycles each) and the LDS round trips "
                                "behind them; on CUs that hold two workgroups the wave shares its SIMD with a helper wave of the other"}


def policy_kernel_label(E, G, A):
    """The fa_policy_kernel instantiation fa_collect_act launches for this shape (csrc/fa_policy.hip: policy_rows)."""
    n_max = max(G, A)
    wgs96 = 2 * ((E + 96 // n_max - 1) // (96 // n_max))
    if wgs96 < 192:
        return "fa_policy_kernel<2, 4> (64-row tiles = %d envs of %dv%d, four waves, two workgroups per CU)" % (64 // n_max, G, A)
    return "fa_policy_kernel<3, 8> (96-row tiles = %d envs of %dv%d, eight waves)" % (96 // n_max, G, A)


def algorithmic_bytes_per_env_step(n_agents):
    """SURVEY.md 8(d) / BASELINE.md section 5: per agent 2*(6 f64 + 1 B alive) state r+w
    + 8 B action + 24 B obs + 4 B reward + 4 B mask = 138 B; per env 9 B."""
    return 138 * n_agents + 9


def host_cpu():
    """CPU model, physical cores and hardware threads of the box (for cpu_baseline)."""
    model, cores, threads = "?", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif k == "processor":
                threads += 1
            elif not k and phys is not None:
                cores.add((phys, core))
        if phys is not None:
            cores.add((phys, core))
    except Exception:
        pass
    return model, len(cores) or threads, threads


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), or None when
    unlimited / unknown: a box can expose 256 hardware threads and still be throttled to a fraction."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(E, G, A, T, budget_s=20.0):
    """The CPU oracle (oracle/fa_oracle.c, a C port of the reference's env.step) timed on this box's
    host cores on a bounded sample of the same workload (uniform-random actions, auto-reset).
    Form: fao_rollout -- ONE OpenMP region around the T-step rollout, every thread stepping its own
    static slice of envs (first-touch placed), threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores)
    -- the CPU counterpart of the fused launch.  The thread count is swept (one subprocess each, so
    that libgomp starts with the right settings) and the best is reported with its thread count."""
    avail = len(os.sched_getaffinity(0))
    model, phys_cores, hw_threads = host_cpu()
    quota = cpu_quota()
    cap = avail if quota is None else min(avail, max(1, int(quota * 4)))   # far beyond the quota only throttles
    sweep = sorted({t for t in (1, 8, 16, 32, 64, 128, phys_cores, avail, int(quota or 0)) if 1 <= t <= cap})
    per = max(1.0, budget_s / (len(sweep) + 1))
    runs = []

    def run(t, mode, envs):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND="close", OMP_PLACES="cores",
                   OMP_WAIT_POLICY="active")
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--envs", str(envs), "--guards", str(G),
               "--attackers", str(A), "--rollout", str(T), "--seconds", "%.2f" % per, "--mode", mode]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=per * 8 + 60)
            return json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as exc:  # a reported baseline must not take the bench down
            return {"threads": t, "env_steps_per_s": 0.0, "error": repr(exc), "mode": mode}

    for t in sweep:
        # enough envs that every thread has a slice worth the region's fork/join (>= 16 envs per thread)
        runs.append(run(t, "rollout", max(E, 16 * t)))
    best = max(runs, key=lambda r: r["env_steps_per_s"])
    single = next((r for r in runs if r["threads"] == 1), best)
    per_step = run(best["threads"], "step", best.get("envs", E))   # one parallel region per env-step (round-1 form)
    # BASELINE config 1's counterpart: ONE env on ONE thread, 128-step rollout -- the C oracle env alone,
    # and with this repo's MPNN (h = 128) on the CPU + the numpy collector oracle in the loop (train_fortattack.py:51-110's
    # shape); the reference's own Python on a Xeon thread: 2 580 env-only / 278 rollout env-steps/s
    one_env = run(1, "rollout", 1)
    try:
        c1 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_config1_cpu.py")], capture_output=True, text=True,
                            timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
        config1 = json.loads(c1.stdout.strip().splitlines()[-1])
    except Exception as exc:
        config1 = {"error": repr(exc)}
    return {"value": best["env_steps_per_s"], "unit": "env-steps/s", "cores": best["threads"], "kind": "port",
            "cpu_model": model, "physical_cores": phys_cores, "hardware_threads": hw_threads,
            "cgroup_cpu_quota": quota,   # CPUs' worth of time this container may use (None = unlimited)
            "loadavg": os.getloadavg()[0],
            "sample": "oracle/fa_oracle.c fao_rollout (C port of the reference env.step + auto-reset; one OpenMP "
                      "region per %d-step rollout, static env slices, pinned threads): %d envs x %d steps in %.1f s on "
                      "%d threads; sweep (threads, env-steps/s) = %s over %d usable cpus; 1 thread: %.0f env-steps/s; "
                      "one parallel region per env-step at the best thread count: %.0f env-steps/s" % (
                          T, best.get("envs", 0), best.get("steps", 0), best.get("seconds", 0.0), best["threads"],
                          [(r["threads"], int(r["env_steps_per_s"])) for r in runs], avail,
                          single["env_steps_per_s"], per_step["env_steps_per_s"]),
            "single_thread": single["env_steps_per_s"],
            "single_env_single_thread": {"env_only_c_oracle_env_steps_per_s": one_env["env_steps_per_s"],
                                         "config1_counterpart": config1},
            "reference_python_note": "the reference's own Python env.step, measured in the survey container on "
                                     "1 Xeon 2.1 GHz thread: 2580 env-steps/s at 3v3 (BASELINE.md section 3); "
                                     "it cannot run on the GPU box"}


def _ranges(ids):
    """[0,1,2,5,6] -> '0-2,5-6'"""
    out, k = [], 0
    while k < len(ids):
        j = k
        while j + 1 < len(ids) and ids[j + 1] == ids[j] + 1:
            j += 1
        out.append(str(ids[k]) if j == k else "%d-%d" % (ids[k], ids[j]))
        k = j + 1
    return ",".join(out)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) with the same arguments and pass their output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="rollouts to time: EXACTLY this many when given (default: 20, raised by --min-seconds)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--min-seconds", type=float, default=None,
                    help="keep timing whole rollouts until the timed region is at least this long (the 128-step "
                         "rollout takes ~0.2 ms: 20 of them are a 4 ms sample).  Default: 0.5 when --steps is not "
                         "given, 0 (= exactly --steps) when it is")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--rollout", type=int, default=128, help="env-steps per rollout (T)")
    ap.add_argument("--guards", type=int, default=3)
    ap.add_argument("--attackers", type=int, default=3)
    ap.add_argument("--launch", choices=["fused", "per-step"], default="fused",
                    help="fused: one launch advances all T steps (open-loop actions); "
                         "per-step: T launches replayed from one hipGraph")
    ap.add_argument("--no-counters", action="store_true",
                    help="do not maintain Agent.numHit / numWasHit and the evaluation counters in the timed run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-collector", action="store_true", help="time the step kernel only")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed_loop record (MPNN in the loop)")
    ap.add_argument("--closed-loop-rollouts", type=int, default=20)
    ap.add_argument("--closed-loop-updates", type=int, default=3)
    ap.add_argument("--no-esweep", action="store_true", help="skip the E-sweep record (fused launch at 32 768 ... 1 048 576 envs)")
    ap.add_argument("--no-5v5", action="store_true",
                    help="skip the 5v5 records of a default 3v3 run (fused launch, closed loop, closed loop with the "
                         "five-strategy attacker ensemble = BASELINE config 5's per-GPU shape)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm)")
    ap.add_argument("--two-stream-tail", action="store_true",
                    help="experiment: put the advantage statistics, the exchange and the normalisation of a rollout on a second "
                         "stream, under the next rollout (measured: 0.186 vs 0.188 ms per step on one rank, 0.218 vs 0.203 with the "
                         "forced one-rank collective -- the cross-stream event waits cost what the overlap gains; default off)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: open a process group of ONE rank and take the several-rank path anyway (the "
                         "all-gather, the merge kernel, the second-stream exchange run on one rank and must change nothing): how "
                         "a one-GPU box executes the RCCL path")
    ap.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)   # the workload of the live PMC passes (live_traffic)
    ap.add_argument("--probe-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--exchange", choices=["auto", "library", "torch"], default="auto",
                    help="who carries the per-rollout exchange of the advantage moments when there are several ranks (or "
                         "--force-collective): 'library' = ONE C call per rollout (fa_gae_allreduce_normalize: scan + moments, "
                         "ncclAllGather on the library's own RCCL communicator, merge + normalisation, all on the launch stream); "
                         "'torch' = fa_gae_moments, torch.distributed.all_gather_into_tensor, fa_adv_merge_normalize from Python; "
                         "'auto' = library with --backend nccl when RCCL can be opened, torch otherwise")
    ap.add_argument("--graph-hot-path", action="store_true",
                    help="capture rollout + collector tail (with its collective, when the exchange is the library's) in ONE "
                         "hipGraph and replay it per bench step: the host's share per step is one graph launch.  The dominant "
                         "kernel is then timed with hipEvents over eager launches right behind the timed region")
    ap.add_argument("--no-pin", action="store_true",
                    help="do not pin the rank's process to its slice of the GPU-local cores (collective.rank_binding)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 PMC passes for roofline.traffic (quote the committed record instead)")
    ap.add_argument("--smoke", action="store_true",
                    help="a short self-check of the --gpus N path before a timed run: 2 rollouts + collector tail, 1 closed-loop "
                         "rollout and 1 PPO update, no CPU baseline / E-sweep / 5v5 records (< 20 s after start-up)")
    ap.add_argument("--share-devices", action="store_true",
                    help="map ranks onto the visible GPUs round-robin (smoke-testing the multi-rank "
                         "path on a box with fewer GPUs than ranks; use with --backend gloo)")
    args = ap.parse_args()
    if args.pmc_probe:
        return pmc_probe(args)
    if args.smoke:
        args.steps, args.warmup, args.min_seconds = 2, 1, 0.0
        args.closed_loop_rollouts, args.closed_loop_updates = 1, 1
        args.no_cpu_baseline = args.no_esweep = args.no_5v5 = args.no_live_traffic = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    # The contract is ONE JSON line on stdout: anything a library prints there from C (gloo's connection banner
    # does) goes to stderr instead; the real stdout comes back for the result line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU path)")
    if args.share_devices:
        local_rank = local_rank % torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU (%d visible); --share-devices --backend gloo runs several ranks on one"
                         % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    exchanging = world > 1 or args.force_collective     # the several-rank code path (see --force-collective)
    if exchanging:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import emergent_multiagent_strategies_amd as fa
    from emergent_multiagent_strategies_amd import dist as fa_dist
    from emergent_multiagent_strategies_amd.dist import gae_adv_mean_std
    if args.force_collective:
        fa_dist.FORCE_COLLECTIVE = True

    # which device and which CPUs every rank ended up on (a rank whose CPUs sit on the other socket, or two ranks on one
    # device, show up here and not as an unexplained slow rank)
    orig_affinity = set(os.sched_getaffinity(0))
    pin = None
    if not args.no_pin:
        # every rollout ends in a cross-rank exchange: a rank whose host thread migrates across sockets, or shares its
        # cores with the other ranks' Python, paces all GPUs.  Each rank takes its own slice of its GPU's NUMA-local cores.
        lists = None
        if world > 1:
            box = [None] * world
            dist.all_gather_object(box, (rank, fa_dist.gpu_local_cpulist(local_rank)[0]))   # one node: every rank is local
            lists = [l for _, l in sorted(box)]
        pin = fa_dist.pin_rank_to_gpu_local_cpus(local_rank, rank, world, lists)
    binding = [{"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank),
                "device_index": local_rank, "pid": os.getpid(),
                "cpu_affinity_at_start": "%d cpus: %s" % (len(orig_affinity), _ranges(sorted(orig_affinity))),
                "pin": pin,
                "cpu_affinity": "%d cpus: %s" % (len(os.sched_getaffinity(0)), _ranges(sorted(os.sched_getaffinity(0))))}]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, binding[0])
        binding = box
        print("rank %d -> cuda:%d, %s" % (rank, local_rank, binding[rank]["cpu_affinity"]), file=sys.stderr, flush=True)

    E, G, A, T = args.envs, args.guards, args.attackers, args.rollout
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, env_offset=rank * E, device=local_rank,
                               track_counters=not args.no_counters)
    st = fa.JointRolloutStorage(T, E, N, device=dev)
    eng.bind_storage(st)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device=dev, generator=gen))
    st.value_preds.copy_(torch.randn(st.value_preds.shape, device=dev, generator=gen))
    adv = torch.empty((T, E, N, 1), device=dev)
    eng.collect_reset()

    graph = None
    if args.launch == "per-step":
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for s in range(T):
                eng.collect_step(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if world > 1 else "global"):
            for s in range(T):
                eng.collect_step(s)

    def env_rollout():
        if graph is not None:
            graph.replay()
        else:
            eng.collect_rollout(0, T)

    # The collector tail of rollout k on the launch stream.  One rank: fa_gae_normalize -- the GAE scan that also leaves the
    # one-pass advantage moments, then the fold + normalisation (two launches).  Several ranks: fa_gae_moments (scan + fold),
    # the all-gather of the N x 3 moments, the exact merge, the normalisation.  --two-stream-tail (experiment) moves everything
    # behind the scan + fold to a second stream, under rollout k + 1: those kernels read returns / value_preds only, which no
    # rollout touches, and the next scan, which rewrites `returns`, waits for them.
    main_stream = torch.cuda.current_stream()
    tail_stream = torch.cuda.Stream() if args.two_stream_tail else None
    pending = {"tail_done": None}
    gather_buf = torch.zeros((world, N, 3), dtype=torch.float64, device=dev) if exchanging else None
    # who carries the exchange: the library's own RCCL communicator (ONE C call per rollout, stream-ordered: nothing of
    # the several-rank tail goes through the Python interpreter or torch.distributed) or torch.distributed from Python
    lib_exchange, exchange_route = None, None
    if exchanging:
        want_lib = args.exchange == "library" or (args.exchange == "auto" and args.backend == "nccl"
                                                   and bool(fa._lib.load().fa_rccl_available()))
        lib_fallback = None
        if want_lib:
            # `auto` must not cost the run: a rank that cannot open the library's communicator says so, and ALL ranks fall back to
            # the torch route together (the decision is agreed through the process group that exists anyway)
            try:
                lib_exchange = fa_dist.LibraryExchange(dev)
                ok = lib_exchange.ranks() == world
                err = None if ok else "the library's RCCL communicator saw %d ranks, not %d" % (lib_exchange.ranks(), world)
            except Exception as exc:
                ok, err = False, repr(exc)
            flag = torch.tensor([1 if ok else 0], device=dev if args.backend == "nccl" else "cpu", dtype=torch.int32)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if args.exchange == "library":
                    raise SystemExit("bench.py --exchange library: %s" % (err or "another rank could not open the library's communicator"))
                if lib_exchange is not None:
                    lib_exchange.close()
                lib_exchange, lib_fallback = None, (err or "another rank could not open the library's communicator")
                print("rank %d: library exchange unavailable (%s): torch route" % (rank, lib_fallback), file=sys.stderr, flush=True)
        exchange_route = ("library: fa_gae_allreduce_normalize (scan + moments, ncclAllGather on the library's communicator, merge + "
                          "normalisation; one C call per rollout)" if lib_exchange is not None else
                          "torch: fa_gae_moments, torch.distributed.all_gather_into_tensor (%s), fa_adv_merge_normalize%s" % (
                              args.backend, "" if not lib_fallback else " [fell back from the library route: %s]" % lib_fallback))

    def exchange_and_normalise(mom, mean, std):
        if exchanging:
            dist.all_gather_into_tensor(gather_buf.view(-1), mom.view(-1))   # the path's one collective (RCCL / xGMI)
            eng.adv_merge_normalize(gather_buf, out=adv)           # Chan-Golub-LeVeque in rank order (same bits on every rank)
            return                                                 # + the normalisation, one launch
        eng.adv_normalize(mean, std, out=adv)

    def collector_tail():
        if pending["tail_done"] is not None:
            main_stream.wait_event(pending["tail_done"])           # the previous statistics / normalisation have read `returns`
        if not exchanging and tail_stream is None:
            eng.gae_normalize(0.99, 0.95, out=adv)                 # one rank: scan + moment partials, fold + normalisation
            return
        if lib_exchange is not None and tail_stream is None:
            lib_exchange.gae_allreduce_normalize(eng, 0.99, 0.95, out=adv)   # several ranks: the whole tail, one C call
            return
        mom, mean, std = eng.gae_moments(0.99, 0.95)               # scan + moment partials, fold: this rank's (n, mean, M2)
        if tail_stream is None:
            exchange_and_normalise(mom, mean, std)
            return
        ready = torch.cuda.Event()
        ready.record(main_stream)
        with torch.cuda.stream(tail_stream):
            tail_stream.wait_event(ready)
            exchange_and_normalise(mom, mean, std)
            done = torch.cuda.Event()
            done.record(tail_stream)
        pending["tail_done"] = done

    def drain():
        if tail_stream is not None:
            main_stream.wait_stream(tail_stream)

    def hot_path():
        env_rollout()
        if not args.no_collector:
            collector_tail()

    def barrier():
        if world > 1:
            dist.barrier()

    # number of rollouts to time: --steps, raised so that the timed region lasts >= --min-seconds
    # (decided from a calibration run and agreed across ranks BEFORE the timed region)
    for _ in range(max(args.warmup, 1)):
        hot_path()
    torch.cuda.synchronize()
    hot_graph = None
    if args.graph_hot_path:
        if tail_stream is not None or graph is not None or (exchanging and lib_exchange is None):
            raise SystemExit("--graph-hot-path needs the fused launch, the one-stream tail and (with several ranks) --exchange library")
        eager_hot_path = hot_path
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eager_hot_path()                                       # the side stream's first use of every kernel / of the communicator
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        hot_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hot_graph, capture_error_mode="thread_local"):
            eager_hot_path()
        hot_path = hot_graph.replay
        hot_path()
        torch.cuda.synchronize()
    steps_requested = 20 if args.steps is None else args.steps
    min_seconds = (0.5 if args.steps is None else 0.0) if args.min_seconds is None else args.min_seconds
    steps = steps_requested
    if min_seconds > 0:
        t0 = time.perf_counter()
        for _ in range(5):
            hot_path()
        torch.cuda.synchronize()
        est = (time.perf_counter() - t0) / 5
        steps = max(steps, int(min_seconds / max(est, 1e-6)) + 1)
        drain()
        if world > 1:
            ts = torch.tensor([steps], device=dev, dtype=torch.int64)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            steps = int(ts.item())
    barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream bracket every env rollout launch of the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    # the host's share per step is taken over the first HOST_WINDOW steps: further into a long region the launch queue is full
    # and the "enqueue time" is the host sleeping on it, i.e. the GPU's time again (round 5's 0.125 ms at --steps 2000)
    HOST_WINDOW = 64
    t_host = None
    if hot_graph is not None:
        for k in range(steps):
            if k == HOST_WINDOW:
                t_host = time.perf_counter() - t0
            hot_graph.replay()                    # rollout + tail (+ the collective): one graph launch per bench step
    else:
        for k in range(steps):
            if k == HOST_WINDOW:
                t_host = time.perf_counter() - t0
            ev[k][0].record()
            env_rollout()
            ev[k][1].record()
            if not args.no_collector:
                collector_tail()
    host_steps = min(steps, HOST_WINDOW)
    host_enqueue = (time.perf_counter() - t0) if t_host is None else t_host   # everything above only ENQUEUES work
    drain()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if hot_graph is not None:                     # the dominant kernel's launch time: eager launches right behind the timed region
        for k in range(steps):
            ev[k][0].record()
            env_rollout()
            ev[k][1].record()
        torch.cuda.synchronize()
    rank_ms = [elapsed * 1e3 / steps]
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        rank_ms = [float(x.item()) * 1e3 / steps for x in every]       # every rank's own clock over the same region
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # steady state: the driver's --steps 20 is a ~4 ms region; the same loop kept up for >= 0.5 s rides in the same line
    steady = None
    if elapsed < 0.5 and not args.smoke:
        n2 = max(steps, int(0.55 / max(elapsed / steps, 1e-6)) + 1)
        if world > 1:
            ts = torch.tensor([n2], device=dev, dtype=torch.int64)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            n2 = int(ts.item())
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n2):
            hot_path()
        drain()
        torch.cuda.synchronize()
        barrier()
        e2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([e2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e2 = float(tt.item())
        steady = {"steps": n2, "timed_seconds": e2, "ms_per_step": e2 * 1e3 / n2, "value": world * E * T * n2 / e2,
                  "unit": "env-steps/s", "note": "the same hot path, same buffers, timed right after the exact-%d-step region" % steps}

    # the second-stream exchange must leave what the one-stream form leaves (same launches, same order per buffer)
    pipelined_ok = None
    if tail_stream is not None and not args.no_collector:
        mean, std = gae_adv_mean_std(eng, 0.99, 0.95)
        pipelined_ok = bool(torch.equal(eng.adv_normalize(mean, std), adv))
    # ... and the library's one-call tail what the torch route leaves
    lib_route_ok = None
    if lib_exchange is not None and not args.no_collector and tail_stream is None:
        mom, _, _ = eng.gae_moments(0.99, 0.95)
        dist.all_gather_into_tensor(gather_buf.view(-1), mom.view(-1))
        lib_route_ok = bool(torch.equal(eng.adv_merge_normalize(gather_buf)[0], adv))
    env_steps = world * E * T * steps
    value = env_steps / elapsed
    launches_per_rollout = T if graph is not None else 1
    roll_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    launch_s = roll_ms * 1e-3 / launches_per_rollout
    bytes_per_launch = algorithmic_bytes_per_env_step(N) * E * (T // launches_per_rollout)
    achieved = bytes_per_launch / launch_s / 1e9

    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    # same command (tools/profile_gpu.sh -> tools/summarize_prof.py; FETCH_SIZE x2 + WRITE_SIZE,
    # MI355X_MICROARCH.md); only attached when the run uses the profiled configuration.
    traffic, traffic_src = None, None
    kernel_name = eng.step_variant(T // launches_per_rollout)   # which step kernel these launches ran
    closed = None
    if not args.no_closed_loop:
        closed = closed_loop(fa, args, rank, local_rank, world, dev, barrier)
    # a default 3v3 run on one GPU also carries the 5v5 shapes one GPU can run (BASELINE config 5's per-GPU shape)
    extra = {}
    if world == 1 and (G, A) == (3, 3) and not args.no_5v5:
        extra["single_env_facade"] = facade_record(fa, G, A, dev)
        extra["fused_5v5"] = fused_record(fa, 5, 5, E, T, dev)
        if not args.no_closed_loop:
            extra["closed_loop_5v5"] = closed_loop(fa, args, rank, local_rank, world, dev, barrier, G=5, A=5, rollouts=10, updates=2)
            extra["closed_loop_5v5_ens5"] = closed_loop(fa, args, rank, local_rank, world, dev, barrier, G=5, A=5, ensemble=5,
                                                        rollouts=10, updates=2)
    esweep_rec = esweep(fa, G, A, dev) if (rank == 0 and world == 1 and not args.no_esweep and (G, A) == (3, 3)) else None

    # the PMC passes run BEHIND every timed section (they start two more processes on this GPU)
    if rank == 0 and world == 1 and graph is None and not args.no_live_traffic:
        torch.cuda.synchronize()
        traffic, traffic_src = live_traffic(kernel_name.split("/")[0], args, device=local_rank)
        live_note = traffic_src
    else:
        live_note = "not attempted (several ranks, per-step launches, or --no-live-traffic)"
    if traffic is None and (E, G, A, T) == (4096, 3, 3, 128):
        for rnd in ("r05", "r04", "r03", "r02", "r01"):
            prof = os.path.join(ROOT, "profiles", "%s_%s_summary.json" % (rnd, "fused" if graph is None else "perstep"))
            if not os.path.isfile(prof):
                continue
            try:
                ks = json.load(open(prof))["kernels"]
                k = [v for n, v in ks.items() if n.split("<")[0].endswith(kernel_name.split("/")[0]) and "<3, 3," in n
                     and "hbm_bytes_per_launch" in v]
                if k:
                    traffic = k[0]["hbm_bytes_per_launch"]
                    traffic_src = "%s (committed record, git blob %s; NOT measured in this run: %s)" % (
                        os.path.relpath(prof, ROOT), git_blob_hash(prof), live_note)
                    break
            except Exception:
                pass

    if rank == 0:
        res = {
            "metric": "env-steps/sec FortAttack %dv%d, %d parallel envs per GPU" % (G, A, E),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": steps, "steps_requested": steps_requested,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / steps, "timed_seconds": elapsed,
            "host_enqueue_ms_per_step": host_enqueue * 1e3 / host_steps,   # < ms_per_step: the GPU, not the Python loop, sets the pace
            "host_enqueue_window_steps": host_steps,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "FortAttack %dv%d, %d envs/GPU, %d-step rollout, open-loop uniform-random actions "
                            "(BASELINE config 2), %s; %s" % (
                                G, A, E, T,
                                "fa_step fused over the rollout in one launch" if graph is None else
                                "one fa_step launch per env-step replayed from a hipGraph",
                                "step kernel only" if args.no_collector else
                                "+ fused RolloutStorage write, GAE scan with the one-pass fp64 advantage moments, fold "
                                "(all-gather of N x 3 f64 + merge when n_gpus > 1) and normalisation" + (
                                    "" if tail_stream is None else "; --two-stream-tail: the moments / exchange / normalisation of a rollout run "
                                    "on a second stream under the next rollout (the next GAE pass waits for them)")),
                "envs_per_gpu": E, "rollout_steps": T, "num_guards": G, "num_attackers": A,
                "max_time_steps": 100, "rng": "mt19937 (reference-parity reset stream)",
                "agent_counters": "off" if args.no_counters else "on (numHit / numWasHit / evaluation counters)",
                "launch": args.launch, "parallelism": "env shards, %d rank(s)" % world},
            "roofline": {
                "bound": "hbm", "kernel": "%s<%d,%d>" % (kernel_name, G if (G, A) in ((3, 3), (5, 5)) else 0,
                                                       A if (G, A) in ((3, 3), (5, 5)) else 0),
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_src,
                # what actually crosses HBM (PMC bytes) over the same launch time: the fused launch keeps the
                # world state in registers, so `frac` (algorithmic bytes, the agreed accounting) is NOT HBM
                # utilisation -- this is
                "traffic_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                "limiter": "instruction issue of the state's wave + two barrier round trips per step, not bandwidth, at this batch size: "
                           "see `secondary` and DESIGN.md 3.1 / 7",
                "measured_stream_GBps": measured_stream(dev),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(N),
                "env_steps_per_launch": E * (T // launches_per_rollout),
                "avg_launch_us": launch_s * 1e6, "timed_by": "hipEvents on the launch stream, %d launches%s" % (
                    len(ev) * launches_per_rollout, "" if hot_graph is None else
                    " (eager launches right behind the timed region: the timed region itself replays one hipGraph per step)"),
                "secondary": (issue_model(G, A, E, T, launch_s, getattr(torch.cuda.get_device_properties(local_rank), "clock_rate", 2400000),
                                          torch.cuda.get_device_properties(local_rank).multi_processor_count, variant=kernel_name)
                              if graph is None and kernel_name.startswith("fa_step_pipe_kernel") else None)},
            "env_rollout_ms": roll_ms,
            "steady_state": steady,
            "rank_ms_per_step": {"per_rank": rank_ms, "min": min(rank_ms), "max": max(rank_ms),
                                 "spread_pct": 100.0 * (max(rank_ms) - min(rank_ms)) / max(min(rank_ms), 1e-12)},
        }
        # which exchange carried the advantage statistics (and the closed loop's gradients), seen by how many ranks
        res["collective"] = {"ranks": dist.get_world_size() if exchanging else 1,
                             "backend": dist.get_backend() if exchanging else None,
                             "rccl_ranks": (dist.get_world_size() if (exchanging and dist.get_backend() == "nccl") else 0),
                             "rank_binding": binding,
                             "per_rollout": "one all-gather of N x 3 f64 (advantage moments) + exact merge, between the "
                                            "moments sweep and the normalisation",
                             "exchange_route": exchange_route,
                             "library_rccl_ranks": lib_exchange.ranks() if lib_exchange is not None else 0,
                             "hot_path_in_one_graph": hot_graph is not None,
                             "library_exchange_equals_torch_route": lib_route_ok,
                             "forced_on_one_rank": bool(args.force_collective),
                             "second_stream_exchange_equals_one_stream": pipelined_ok,
                             "per_optimizer_step": "one all_reduce of the flat f32 gradient buffer (149 908 floats) per team"}
        if world > 1 and args.backend == "nccl" and res["collective"]["rccl_ranks"] != world:
            raise SystemExit("bench.py --gpus %d: RCCL saw %d ranks" % (world, res["collective"]["rccl_ranks"]))
        if args.smoke:
            res["smoke"] = True
        if closed is not None:
            res["closed_loop"] = closed
        if extra:
            res.update(extra)
        if esweep_rec is not None:
            res["esweep"] = esweep_rec
        if world == 1 and not args.no_cpu_baseline:
            now = os.sched_getaffinity(0)
            os.sched_setaffinity(0, orig_affinity)     # the CPU baseline gets every core the process started with, not the rank's slice
            try:
                res["cpu_baseline"] = cpu_baseline(E, G, A, T)
            finally:
                os.sched_setaffinity(0, now)
        sys.stdout.flush()
        try:                                   # what C libraries printf'ed while fd 1 pointed at stderr (RCCL's version
            import ctypes                      # banner sits in the C stdio buffer until exit) must not follow the line
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        sys.stdout.flush()
        os.dup2(2, 1)                          # anything printed after the line (process-group teardown) goes to stderr again
    if lib_exchange is not None:
        lib_exchange.close()
    if exchanging:
        dist.destroy_process_group()


def esweep(fa, G, A, dev):
    """BASELINE.md section 5 / SURVEY 8(d): the same fused launch on more envs per GPU (the bandwidth regime).
    T = 128 up to 262 144 envs; 32 steps at 1 048 576 envs (12 GB of rollout buffers instead of 48)."""
    import torch
    N, out = G + A, []
    for E, T in ((32768, 128), (262144, 128), (1048576, 32)):
        eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, device=dev.index, track_counters=True)
        st = fa.JointRolloutStorage(T, E, N, device=dev)
        eng.bind_storage(st)
        gen = torch.Generator(device=dev).manual_seed(E)
        st.actions.copy_(torch.randint(0, 8, st.actions.shape, device=dev, generator=gen))
        eng.collect_reset()
        for _ in range(2):
            eng.collect_rollout(0, T)
        torch.cuda.synchronize()
        iters = 5
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            eng.collect_rollout(0, T)
        b.record()
        torch.cuda.synchronize()
        sec = a.elapsed_time(b) * 1e-3 / iters
        gbps = algorithmic_bytes_per_env_step(N) * E * T / sec / 1e9
        out.append({"envs": E, "rollout_steps": T, "kernel": eng.step_variant(T), "launch_us": sec * 1e6,
                    "env_steps_per_s": E * T / sec, "algorithmic_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS})
        del eng, st
        torch.cuda.empty_cache()
    return out


def attacker_pool(fa, G, A, K):
    """K frozen attacker strategies for the ensemble records (BASELINE config 5): the reference's published 5v5 policies
    (marlsave/tmp_1/ep{220,650,1240,1600,2520}.pt, exported as data by oracle/gen_golden.py into
    tests/golden/attackers_tmp1.npz) when the shape is theirs, else K randomly initialised ones."""
    import numpy as np
    import torch
    path = os.path.join(ROOT, "tests", "golden", "attackers_tmp1.npz")
    if (G, A, K) == (5, 5, 5) and os.path.isfile(path):
        z = np.load(path)
        eps = [int(e) for e in z["episodes"]]
        pool = [{k[len("ep%d." % e):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ep%d." % e) and ".out." not in k} for e in eps]
        return pool, "the reference's published attackers marlsave/tmp_1/ep{%s}.pt (tests/golden/attackers_tmp1.npz)" % ",".join(map(str, eps))
    return ([fa.MPNN(num_agents=A, num_opp_agents=G, num_actions=8).state_dict() for _ in range(K)],
            "%d randomly initialised attacker strategies" % K)


def closed_loop(fa, args, rank, local_rank, world, dev, barrier, G=None, A=None, ensemble=0, rollouts=None, updates=None):
    """BASELINE config 3 (config 4 when world > 1; config 5's per-GPU shape with G = A = 5 and `ensemble` = 5): the MPNN
    actor-critic (h = 128) in the loop.  A rollout = T x (two-team forward + sampling + fa_collect_step), V(obs[T]), GAE
    and the advantage moments (all-gathered over ranks); then one JointPPO update (4 epochs x 32 minibatches per
    trained team; flat gradient all-reduce per optimizer step when world > 1).  With an attacker ensemble every env
    plays one of the frozen strategies, re-drawn at each reset, and only the guards are trained (learner.py:119-140,177)."""
    import torch
    import torch.distributed as dist
    E, T = args.envs, args.rollout
    G = args.guards if G is None else G
    A = args.attackers if A is None else A
    torch.manual_seed(0)                                      # identical initial policies on every rank
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, env_offset=rank * E, device=local_rank,
                               track_counters=not args.no_counters)
    L = fa.BatchedLearner(eng, num_steps=T, use_graph=True)
    pool_note = None
    if ensemble:
        pool, pool_note = attacker_pool(fa, G, A, ensemble)
        L.load_attacker_ensemble(pool)
    guards_only = bool(ensemble)
    torch.manual_seed(1 + rank)                               # different action sampling per rank
    L.reset()
    L.collect()
    L.update(train_guards_only=guards_only)                   # untimed: captures the update's hipGraphs
    L.after_update()
    torch.cuda.synchronize()
    barrier()
    R = max(1, args.closed_loop_rollouts if rollouts is None else rollouts)
    t0 = time.perf_counter()
    for _ in range(R):
        L.collect()
        L.after_update()
    torch.cuda.synchronize()
    barrier()
    t_roll = time.perf_counter() - t0
    L.collect()
    torch.cuda.synchronize()
    barrier()
    U = max(1, args.closed_loop_updates if updates is None else updates)
    t0 = time.perf_counter()
    for _ in range(U):                                        # (each a full JointPPO update from this rollout)
        L.update(train_guards_only=guards_only)
    torch.cuda.synchronize()
    barrier()
    t_upd = (time.perf_counter() - t0) / U
    roof = closed_loop_rooflines(fa, L, E, G, A, T) if rank == 0 else None
    identical = None
    if world > 1:
        tt = torch.tensor([t_roll, t_upd], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_roll, t_upd = float(tt[0]), float(tt[1])
        # data-parallel invariant: after the updates every rank holds the same parameters, bit for bit
        import hashlib
        h = hashlib.sha256()
        for pol in L.policies:
            for p in pol.parameters():
                h.update(p.detach().cpu().numpy().tobytes())
        box = [None] * world
        dist.all_gather_object(box, h.hexdigest())
        identical = len(set(box)) == 1
    per_rollout = t_roll / R
    teams = 1 if guards_only else 2
    rec = {
        "workload": "FortAttack %dv%d, %d envs/GPU, %d-step rollout, MPNN h=128 actor-critic in the loop "
                    "(BASELINE config %s): two-team forward + sampling + fa_collect_step per env-step replayed from "
                    "a hipGraph, V(obs[T]), GAE, advantage moments%s%s" % (
                        G, A, E, T, ("5's per-GPU shape" if ensemble else ("3" if world == 1 else "4")),
                        "" if world == 1 else " all-gathered over the ranks",
                        "; attackers = %s, one per env, re-drawn at every reset; guards only trained" % pool_note if ensemble else ""),
        "rollout_env_steps_per_s": world * E * T / per_rollout, "rollout_ms": per_rollout * 1e3,
        "ms_per_env_step_launch": per_rollout * 1e3 / T, "rollouts_timed": R, "updates_timed": U,
        "update_s": t_upd, "train_env_steps_per_s": world * E * T / (per_rollout + t_upd),
        "roofline": roof["policy"] if roof else None, "update_roofline": roof["train"] if roof else None,
        "update": "JointPPO: 4 epochs x 32 minibatches x %d team%s, Adam, grad-clip; every optimizer step (fused forward + "
                  "losses + backward kernels, fold / unfold, clip + Adam on flat buffers) replayed from a hipGraph, %s" % (
            teams, "" if teams == 1 else "s",
            "one chain" if teams == 1 else
            "the two teams as concurrent chains on two streams" if world == 1 else
            "one flat gradient all-reduce per optimizer step between the two graphs of a step; the two teams as concurrent "
            "chains on two streams whose all-reduces are event-ordered (one order on every rank)"),
        "dtype": "f32 policy / f64 env", "unit": "env-steps/s"}
    if identical is not None:
        rec["ranks_hold_identical_parameters"] = identical
    L.close()
    del L, eng
    torch.cuda.empty_cache()
    return rec


def facade_record(fa, G, A, dev, steps=400):
    """The single-env façade (make_fortattack_env: the reference's surface over the same GPU engine with E = 1, one launch and
    one host synchronisation per env.step) -- what a reference script gets when it runs unmodified: env-steps/s."""
    import numpy as np
    env = fa.make_fortattack_env(100, num_guards=G, num_attackers=A, seed=0, device=dev.index)
    env.reset()
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 8, size=(steps + 20, G + A))
    for k in range(20):
        if env.step(acts[k])[2]:
            env.reset()
    t0 = time.perf_counter()
    for k in range(20, steps + 20):
        if env.step(acts[k])[2]:
            env.reset()
    sec = time.perf_counter() - t0
    env.close()
    return {"workload": "make_fortattack_env(...).step on one env (%dv%d), E = 1 on the GPU, numpy in / numpy out, %d steps incl. the "
                        "resets they trigger" % (G, A, steps), "env_steps_per_s": steps / sec, "us_per_step": sec / steps * 1e6,
            "reference_python_env_steps_per_s": 2580, "note": "latency-bound by design (a launch + a sync + four small copies per "
            "step); the batched engine is the product path"}


def fused_record(fa, G, A, E, T, dev, iters=200):
    """The headline workload (fused rollout launch + GAE with the one-pass moments + fold and normalisation) at another team
    size, one GPU."""
    import torch
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, device=dev.index, track_counters=True)
    st = fa.JointRolloutStorage(T, E, N, device=dev)
    eng.bind_storage(st)
    gen = torch.Generator(device=dev).manual_seed(99)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device=dev, generator=gen))
    st.value_preds.copy_(torch.randn(st.value_preds.shape, device=dev, generator=gen))
    adv = torch.empty((T, E, N, 1), device=dev)
    eng.collect_reset()

    def hot():
        eng.collect_rollout(0, T)
        eng.gae_normalize(0.99, 0.95, out=adv)

    for _ in range(5):
        hot()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    t0 = time.perf_counter()
    for k in range(iters):
        ev[k][0].record()
        eng.collect_rollout(0, T)
        ev[k][1].record()
        eng.gae_normalize(0.99, 0.95, out=adv)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / iters
    launch_s = sum(a.elapsed_time(b) for a, b in ev) / iters * 1e-3
    gbps = algorithmic_bytes_per_env_step(N) * E * T / launch_s / 1e9
    rec = {"workload": "FortAttack %dv%d, %d envs, %d-step rollout, open-loop uniform-random actions: fused launch + GAE with the "
                       "one-pass moments + fold and normalisation" % (G, A, E, T),
           "value": E * T / sec, "unit": "env-steps/s", "ms_per_step": sec * 1e3, "steps": iters,
           "roofline": {"bound": "hbm", "kernel": eng.step_variant(T) + "<%d,%d>" % (G, A), "achieved": gbps, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "avg_launch_us": launch_s * 1e6,
                        "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(N),
                        "timed_by": "hipEvents on the launch stream, %d launches" % iters}}
    del eng, st, adv
    torch.cuda.empty_cache()
    return rec


def closed_loop_rooflines(fa, L, E, G, A, T):
    """The two MFMA kernels of the closed loop against the fp32 MFMA peak, timed live with HIP events on the launch
    stream: fa_policy_kernel (one launch = both teams' forward + sampling for every env: the dominant kernel of the
    rollout) and fa_ppo_grad (mask sums + fa_train_kernel + slab reduction: the dominant launches of the update), each
    as eager back-to-back launches of the SAME shape the hipGraphs replay."""
    import torch
    from emergent_multiagent_strategies_amd import mpnn_pack
    from emergent_multiagent_strategies_amd.env import ppo_grad
    N = G + A

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / iters

    out = {}
    if L.policy_backend == "hip":
        sec = timed(lambda: L._hip_act(0), 200)
        flops = E * (G * policy_flops_per_row(G, A) + A * policy_flops_per_row(A, G))
        out["policy"] = {"bound": "mfma", "kernel": policy_kernel_label(E, G, A), "achieved": flops / sec / 1e12,
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS,
                         # the dense layers run on the bf16 matrix cores with every fp32 operand split exactly into three bf16
                         # terms (six of the nine cross products, fp32 accumulate: fp32-class results, csrc/fa_mfma.h gemm_cb3).
                         # `peak` stays the fp32-MFMA peak -- what an fp32 kernel is priced against; the split form's own
                         # ceiling is the bf16 dense peak / 6 (six MFMAs per fp32-equivalent block)
                         "dtype": "f32 (bf16x3 split MFMA, fp32 accumulate)",
                         "peak_split_form": MFMA_BF16_PEAK_TFLOPS / 6.0,
                         "frac_of_split_form_peak": flops / sec / 1e12 / (MFMA_BF16_PEAK_TFLOPS / 6.0),
                         "traffic": None, "flops_per_launch": flops, "avg_launch_us": sec * 1e6,
                         "timed_by": "hipEvents on the launch stream, 200 eager launches of fa_collect_act",
                         "share_of_env_step": "one launch per env-step next to one fa_step_kernel launch (~6 us)"}
    else:
        out["policy"] = None
    st = L.storage
    flat = lambda t: t.view(T * E, *t.shape[2:])
    rows = (flat(st.obs[:-1]), flat(st.actions), flat(st.value_preds[:-1]), flat(st.returns[:-1]),
            flat(st.action_log_probs), flat(L.adv))
    mb = (T * E) // L.num_mini_batch
    if L._flat and mb > 0:
        fp = L._flat[0]
        w, wt = fp.fold_pack()
        idx = torch.randperm(T * E, device=rows[0].device)[:mb]
        outbuf = torch.zeros(mpnn_pack.SLAB_FLOATS, device=rows[0].device)
        scratch = [None]

        def grad():
            # (as the update's graphs call it: advantages normalised inside the kernel from the rollout's mean / std)
            _, scratch[0] = ppo_grad(*rows[:5], None, w, wt, None, 0, G, A, L.clip_param, L.value_loss_coef, L.entropy_coef,
                                     L.clipped_value_loss, scratch=scratch[0], out=outbuf, idx=idx, normalize=True,
                                     adv_stats=(L._adv_mean, L._adv_std))

        sec = timed(grad, 20)
        flops = mb * G * train_flops_per_row(G, A)
        out["train"] = {"bound": "mfma", "kernel": "fa_ppo_grad = fa_mask_part_kernel + fa_train_kernel<true, %d> (32-row tiles, two workgroups per CU) + "
                                                   "fa_train_dw3_kernel (weight gradients: split-K GEMM over all rows on the bf16 matrix cores, operands "
                                                   "split exactly into three bf16 terms, fp32 accumulate) + fa_train_mred_kernel + "
                                                   "fa_train_reduce_kernel" % (4 if max(G, A) <= 4 else (6 if max(G, A) <= 6 else 8)),
                        "achieved": flops / sec / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "dtype": "f32 (tile kernel: fp32 MFMA; weight-gradient GEMM: bf16x3 split MFMA, fp32 accumulate)",
                        "frac": flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None, "flops_per_launch": flops,
                        "avg_launch_us": sec * 1e6, "minibatch_rows": mb * G,
                        "timed_by": "hipEvents on the launch stream, 20 eager fa_ppo_grad calls (five launches each; the "
                                    "kernels one by one are in profiles/ rocprofv3 stats)"}
    else:
        out["train"] = None
    return out


if __name__ == "__main__":
    main()
