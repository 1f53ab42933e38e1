"""Experiment (GPU box): where a single-step launch's time goes -- graph replay of 128 single-step launches
per kernel build, with and without the agent counters, next to an empty-kernel floor (torch no-op launches)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa

E, G, A, T = 4096, 3, 3, 128
def run(kernel, counters, E=E):
    eng = fa.BatchedFortAttack(E, G, A, 100, track_counters=counters, step_kernel=kernel)
    st = fa.JointRolloutStorage(T, E, G + A, device="cuda")
    eng.bind_storage(st)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
    eng.collect_reset()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for s in range(T):
            eng.collect_step(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for s in range(T):
            eng.collect_step(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 / T * 1e3

for kernel in ("waves3", "waves2", "waves1"):
    for counters in (True, False):
        print(json.dumps({"kernel": kernel, "counters": counters, "us_per_launch": round(run(kernel, counters), 3)}), flush=True)
print(json.dumps({"kernel": "waves3", "E": 256, "us_per_launch": round(run("waves3", True, 256), 3)}))
# floor: 128 trivial kernels in a graph
x = torch.zeros(64, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    x.add_(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(T):
        x.add_(1)
g.replay(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    g.replay()
b.record(); torch.cuda.synchronize()
print(json.dumps({"kernel": "trivial torch add_ (64 floats)", "us_per_launch": round(a.elapsed_time(b) / 20 / T * 1e3, 3)}))
