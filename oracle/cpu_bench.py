"""Time the CPU oracle (oracle/fa_oracle.c) on the host cores: the cpu_baseline leg of
bench.py.  TEST INFRASTRUCTURE (a reported baseline, never the product path).

Two forms: `--mode rollout` (default) = fao_rollout, T steps of every env inside one OpenMP region,
each thread stepping its own slice of envs (the CPU counterpart of the fused launch);
`--mode step` = one fao_step call (one parallel region) per env-step from a Python loop, which is
how a trainer with a policy between steps would have to call it.

Run as a subprocess so that OpenMP is configured before libgomp starts:
    OMP_NUM_THREADS=K OMP_PROC_BIND=close OMP_PLACES=cores python oracle/cpu_bench.py --envs 4096 --seconds 3
Prints one JSON line: {"threads": K, "env_steps_per_s": V, "envs": E, "steps": n, "seconds": s, "mode": m}
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fa_oracle import OracleEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--guards", type=int, default=3)
    ap.add_argument("--attackers", type=int, default=3)
    ap.add_argument("--rollout", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--mode", choices=["rollout", "step"], default="rollout")
    a = ap.parse_args()
    N, T = a.guards + a.attackers, a.rollout
    rng = np.random.RandomState(0)
    acts = np.ascontiguousarray(rng.randint(0, 8, size=(T, a.envs, N)).astype(np.int64))
    env = OracleEnv(a.envs, a.guards, a.attackers, 100, base_seed=0)
    env.reset()
    if a.mode == "rollout":
        env.rollout_noout(acts[:8])
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < a.seconds:
            env.rollout_noout(acts)
            n += T
    else:
        for k in range(4):
            env.step_noout(acts[k])
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < a.seconds:
            for k in range(16):
                env.step_noout(acts[k])
            n += 16
    dt = time.perf_counter() - t0
    print(json.dumps({"threads": int(os.environ.get("OMP_NUM_THREADS", "0")), "env_steps_per_s": a.envs * n / dt,
                      "envs": a.envs, "steps": n, "seconds": dt, "mode": a.mode}))


if __name__ == "__main__":
    main()
