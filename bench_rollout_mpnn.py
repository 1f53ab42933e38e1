#!/usr/bin/env python
"""BASELINE config 3: FortAttack 3v3, 4096 envs, MPNN actor-critic (PyTorch-ROCm) in the
loop: per env-step two MPNN forwards + sampling + fa_collect_step, then V(obs[T]), GAE and
the advantage statistics.  Reports env-steps/s for the rollout alone and for rollout + PPO
update.  (bench.py measures config 2, the step kernel with open-loop actions.)

    python bench_rollout_mpnn.py [--envs 4096] [--rollout 128] [--iters 3] [--graph 0|1] [--update 0|1]
"""
import argparse
import json
import time

import torch

import emergent_multiagent_strategies_amd as fa


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--rollout", type=int, default=128)
    ap.add_argument("--guards", type=int, default=3)
    ap.add_argument("--attackers", type=int, default=3)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--update", type=int, default=1)
    ap.add_argument("--backend", default="auto", choices=["auto", "hip", "torch"], help="who runs the rollout forwards")
    ap.add_argument("--update-backend", default="auto", choices=["auto", "fused", "torch"])
    ap.add_argument("--ensemble", type=int, default=0, help="K frozen attacker strategies, one per env, re-drawn at "
                                                              "every reset (BASELINE config 5); guards only are trained")
    ap.add_argument("--epochs", type=int, default=4, help="ppo_epoch (4 = the reference's default)")
    ap.add_argument("--sequential-teams", action="store_true", help="update the teams one after the other (profiling: "
                                                                    "every launch then has the GPU to itself)")
    a = ap.parse_args()
    torch.manual_seed(0)
    eng = fa.BatchedFortAttack(a.envs, a.guards, a.attackers, 100, base_seed=0, track_counters=False)
    L = fa.BatchedLearner(eng, num_steps=a.rollout, use_graph=bool(a.graph), policy_backend=a.backend, update_backend=a.update_backend,
                          ppo_epoch=a.epochs)
    if a.sequential_teams and L._update_graphs is not None:
        L._update_graphs["teams_together"] = False
    if a.ensemble:
        L.load_attacker_ensemble([fa.MPNN(num_agents=a.attackers, num_opp_agents=a.guards, num_actions=8).state_dict()
                                  for _ in range(a.ensemble)])
    L.reset()
    L.collect()
    if a.update:                       # untimed: captures the update's hipGraphs
        L.update(train_guards_only=bool(a.ensemble))
    L.after_update()
    torch.cuda.synchronize()
    t_roll = t_upd = 0.0
    for _ in range(a.iters):
        t0 = time.perf_counter()
        L.collect()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if a.update:
            L.update(train_guards_only=bool(a.ensemble))
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        L.after_update()
        t_roll += t1 - t0
        t_upd += t2 - t1
    steps = a.envs * a.rollout * a.iters
    backend = L.policy_backend
    L.close()
    print(json.dumps({
        "config": "FortAttack %dv%d, %d envs, %d-step rollout, MPNN h=128 policy in the loop (%s)" % (
            a.guards, a.attackers, a.envs, a.rollout, ("one hipGraph per rollout" if a.graph else "eager") + ", forwards: " + backend +
            (", ensemble of %d attacker strategies" % a.ensemble if a.ensemble else "")),
        "rollout_env_steps_per_s": steps / t_roll, "rollout_ms_per_env_step_launch": t_roll / (a.iters * a.rollout) * 1e3,
        "train_env_steps_per_s": steps / (t_roll + t_upd) if a.update else None,
        "update_s": t_upd / a.iters if a.update else None,
        "reference_python_rollout_env_steps_per_s": 278, "note": "reference figure: BASELINE.md section 3, 1 CPU thread"}))


if __name__ == "__main__":
    main()
