#!/usr/bin/env python
"""BASELINE config 1 counterpart: one env, one CPU thread, 128-step rollout --
the CPU oracle env (oracle/fa_oracle.c, test infrastructure) + this repo's MPNN on the CPU +
the numpy collector oracle, i.e. the reference's train_fortattack.py:51-110 loop shape without
the reference.  A reported CPU baseline (cf. the reference's own 278 env-steps/s rollout and
2 580 env-steps/s env-only on one Xeon thread); never the product path.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import torch

torch.set_num_threads(1)
from fa_oracle import OracleEnv
import collector_oracle as co
from emergent_multiagent_strategies_amd.mpnn import MPNN

G = A = 3
N, T, iters = G + A, 128, 5
torch.manual_seed(0)
pols = [MPNN(num_agents=G, num_opp_agents=A, num_actions=8), MPNN(num_agents=A, num_opp_agents=G, num_actions=8)]
env = OracleEnv(1, G, A, 100, base_seed=0)
st = [co.StorageOracle(T, 1) for _ in range(N)]
obs = env.reset()[0]
t_env = t_act = 0.0
t0 = time.perf_counter()
for _ in range(iters):
    for s in range(T):
        ta = time.perf_counter()
        o = torch.from_numpy(obs.astype(np.float32))[None]
        with torch.no_grad():
            vg, ag, lg = pols[0].act(o[:, :G], o[:, G:])
            va, aa, la = pols[1].act(o[:, G:], o[:, :G])
        act = torch.cat([ag, aa], 1)[0, :, 0].numpy()
        tb = time.perf_counter()
        out = env.step(act[None], auto_reset=True, want_flags=False)
        tc = time.perf_counter()
        val = torch.cat([vg, va], 1)[0].numpy()
        lp = torch.cat([lg, la], 1)[0].numpy()
        masks = obs[:, 0].astype(np.float32)
        obs = out["obs"][0]
        for i in range(N):
            st[i].insert(obs[i].astype(np.float32), 0.0, act[i], lp[i], val[i], np.float32(out["reward"][0, i]), masks[i])
        t_act += tb - ta
        t_env += tc - tb
    for i in range(N):
        st[i].compute_returns(np.float32(0), 0.99, 0.95, 0, T)
        st[i].after_update()
dt = time.perf_counter() - t0
print(json.dumps({"config": "3v3, 1 env, 1 CPU thread, 128-step rollout: C oracle env + MPNN(h=128) on CPU + numpy collector",
                  "rollout_env_steps_per_s": iters * T / dt, "env_only_env_steps_per_s": iters * T / t_env,
                  "policy_ms_per_step": 1e3 * t_act / (iters * T),
                  "reference_python": {"rollout_env_steps_per_s": 278, "env_only_env_steps_per_s": 2580,
                                       "source": "BASELINE.md section 3 (survey container, 1 Xeon 2.1 GHz thread)"}}))
