#!/usr/bin/env python
"""Soak (run on the GPU box): the TRAINING loop against the CPU oracle.  BatchedLearner(use_graph=True) at 3v3 x 4096 x 128
collects a rollout (one hipGraph of 128 x (fa_policy_kernel + fa_step_kernel) + V(obs[T])), the oracle is driven by the
actions the policies sampled, every env row / mask / done flag and the GAE returns are compared bit for bit, then the
PPO update runs and the next rollout follows under the CHANGED policies -- so the step kernel is checked under the action
distributions a learning policy produces (shooting, crowding at the fort), not only under uniform noise.
usage: soak_closed_loop.py [iterations] [G] [A] [measure]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import emergent_multiagent_strategies_amd as fa
import collector_oracle as co
from fa_oracle import OracleEnv
import test_gpu_learner
from test_gpu_learner import _check_rollout_against_oracle

R = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = int(sys.argv[2]) if len(sys.argv) > 2 else 3
A = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if len(sys.argv) > 4 and sys.argv[4] == "measure":   # record the errors without asserting the bound
    test_gpu_learner.ERR_FACTOR = 1e9
E, T, max_t = 4096, 128, 100
N = G + A
torch.manual_seed(0)
eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=0)
orc = OracleEnv(E, G, A, max_t, base_seed=0)
L = fa.BatchedLearner(eng, num_steps=T, use_graph=True)
assert L.policy_backend == "hip" and L._update_graphs["fused"]
L.reset()
stale = np.zeros((T + 1, E, N, 1), np.float32)
t0 = time.time()
shoot_frac, ent = [], []
for it in range(R):
    L.collect()
    torch.cuda.synchronize()
    ep_start, rew, vals, msk, rets = _check_rollout_against_oracle(fa, L, orc, first=(it == 0))   # asserts bit equality
    want = stale.copy()
    for i in range(N):
        co.gae_single_pass(rew[:, :, i], vals[:, :, i], msk[:, :, i], want[:, :, i], ep_start, 0.99, 0.95)
    assert np.array_equal(rets, want), it
    stale = rets
    acts = L.storage.actions.cpu().numpy()
    shoot_frac.append(float((acts == 7).mean()))
    out = L.update()
    ent.append(float(out[:, 2].mean()))
    L.after_update()
st = eng.get_state()
L.close()
print(json.dumps({"config": "%dv%d, E=%d, T=%d, max_time_steps=%d, BatchedLearner(use_graph=True), %d collect + update iterations"
                  % (G, A, E, T, max_t, R), "env_steps": R * E * T, "differing_rows": 0,
                  "checked": "obs / rewards / masks / done rows and GAE returns bit for bit; policy rows (values incl. V(obs[T]), log-probs) of the fused "
                             "forward vs the module in FLOAT64: |fused - f64| <= %g x |module_f32 - f64| + %g x max(1, |row|max), every rollout, both teams"
                             % (test_gpu_learner.ERR_FACTOR, test_gpu_learner.ERR_FLOOR),
                  "policy_row_errors_vs_float64": test_gpu_learner.POLICY_ROW_ERRORS,
                  "shoot_fraction_first_last": [shoot_frac[0], shoot_frac[-1]], "entropy_first_last": [ent[0], ent[-1]],
                  "episodes": int(st["result_count"].sum()), "seconds": round(time.time() - t0, 1)}))
