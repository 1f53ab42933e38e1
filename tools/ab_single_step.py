import json, os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else ".")
import torch
import emergent_multiagent_strategies_amd as fa
lib = sys.argv[1]
if lib != "product":
    fa._lib._build.LIB = os.path.join("tools", "_build", "lib_%s.so" % lib)
out = {"lib": lib}
for G, A, E, T in ((3, 3, 4096, 128), (5, 5, 4096, 128), (3, 3, 262144, 128)):
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)))
    eng.collect_reset()
    if E <= 8192:
        fn = lambda: [eng.collect_step(s) for s in range(T)]    # T single-step launches (the closed loop's step kernel)
        div = T
    else:
        fn = lambda: eng.collect_rollout(0, T)                  # the one-wave kernel's regime
        div = 1
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph() if E <= 8192 else None
    if g is not None:
        with torch.cuda.graph(g):
            fn()
        run = g.replay
    else:
        run = fn
    best = 1e9
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 5 / div * 1e3)
    out["%dv%d_%d_%s_us" % (G, A, E, eng.step_variant(1 if E <= 8192 else T).replace(" ", ""))] = round(best, 3)
    del eng, st
print(json.dumps(out))
