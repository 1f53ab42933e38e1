"""Experiment (GPU box): seconds per PPO update at config 3 (3v3 x 4096 x 128) for the update's scheduling variants:
two concurrent chains with / without the register-capped tile kernel, and the teams one after the other."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, opts in (("together, capped kernel", {}), ("together, uncapped kernel", {"share_cu": False}), ("teams in sequence", {"teams_together": False})):
    torch.manual_seed(0)
    eng = fa.BatchedFortAttack(4096, G, G, 100, track_counters=False)
    L = fa.BatchedLearner(eng, num_steps=128, use_graph=True)
    L._update_graphs.update(opts)
    L.reset(); L.collect(); L.update()
    ts = []
    for _ in range(3):
        L.collect(); torch.cuda.synchronize()
        t0 = time.perf_counter(); L.update(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"variant": name, "teams": "%dv%d" % (G, G), "update_s": round(min(ts), 4)}), flush=True)
    L.close(); del L, eng
