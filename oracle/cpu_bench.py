"""Time the CPU oracle (oracle/fa_oracle.c) on the host cores: the cpu_baseline leg of
bench.py.  TEST INFRASTRUCTURE (a reported baseline, never the product path).

Run as a subprocess so that OpenMP is configured before libgomp starts:
    OMP_NUM_THREADS=K OMP_WAIT_POLICY=passive python oracle/cpu_bench.py --envs 4096 --seconds 3
Prints one JSON line: {"threads": K, "env_steps_per_s": V, "envs": E, "steps": n, "seconds": s}
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
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    N = a.guards + a.attackers
    rng = np.random.RandomState(0)
    acts = [np.ascontiguousarray(rng.randint(0, 8, size=(a.envs, N)).astype(np.int64)) for _ in range(16)]
    env = OracleEnv(a.envs, a.guards, a.attackers, 100, base_seed=0)
    env.reset()
    for k in range(4):
        env.step_noout(acts[k])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        for k in range(16):
            env.step_noout(acts[k])
        n += 16
    dt = time.perf_counter() - t0
    print(json.dumps({"threads": int(os.environ.get("OMP_NUM_THREADS", "0")), "env_steps_per_s": a.envs * n / dt,
                      "envs": a.envs, "steps": n, "seconds": dt}))


if __name__ == "__main__":
    main()
