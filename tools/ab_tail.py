"""Experiment (run ON THE GPU BOX): the collector tail of the headline workload (3v3 x 4096 x 128) by form, microseconds
per call from hipEvents over back-to-back calls, alone and behind the fused rollout launch:
  old      fa_gae + fa_adv_moments_onepass + fa_adv_normalize   (five launches; FA_GAE_MOMENTS_SEPARATE=1)
  moments  fa_gae_moments                                       (scan with moment partials + fold)
  fused    fa_gae_normalize                                     (scan with moment partials, fold + normalisation)
usage: ab_tail.py [E] [iters] [team size: 3 = 3v3, 5 = 5v5]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(E, iters, G=3, A=3):
    sys.path.insert(0, ROOT)
    import torch
    import emergent_multiagent_strategies_amd as fa
    if os.environ.get("FA_AB_LIB"):   # a tools/build_variant.py library instead of the product
        fa._lib._build.LIB = os.path.join(ROOT, "tools", "_build", "lib_%s.so" % os.environ["FA_AB_LIB"])
    T = 128
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    gen = torch.Generator(device="cuda").manual_seed(1)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda", generator=gen))
    st.value_preds.copy_(torch.randn(st.value_preds.shape, device="cuda", generator=gen))
    adv = torch.empty((T, E, N, 1), device="cuda")
    eng.collect_reset()
    eng.collect_rollout(0, T)

    def old():
        eng.gae(0.99, 0.95)
        _, mean, std = eng.adv_moments_onepass()
        eng.adv_normalize(mean, std, out=adv)

    forms = {"gae_only": lambda: eng.gae(0.99, 0.95), "old": old, "moments": lambda: eng.gae_moments(0.99, 0.95),
             "fused": lambda: eng.gae_normalize(0.99, 0.95, out=adv)}
    out = {"lib": os.environ.get("FA_AB_LIB", "product"), "separate": os.environ.get("FA_GAE_MOMENTS_SEPARATE", "0"), "E": E, "team": G}
    for name, fn in forms.items():
        for with_rollout in (False, True):
            def call():
                if with_rollout:
                    eng.collect_rollout(0, T)
                fn()
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                call()
            b.record()
            torch.cuda.synchronize()
            out[name + ("+rollout" if with_rollout else "")] = round(a.elapsed_time(b) * 1e3 / iters, 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("FA_AB_TAIL_CHILD"):
        one(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[3]))
    else:
        E = sys.argv[1] if len(sys.argv) > 1 else "4096"
        iters = sys.argv[2] if len(sys.argv) > 2 else "300"
        team = sys.argv[3] if len(sys.argv) > 3 else "3"
        for sep in ("0", "1"):
            env = dict(os.environ, FA_AB_TAIL_CHILD="1", FA_GAE_MOMENTS_SEPARATE=sep)
            subprocess.run([sys.executable, os.path.abspath(__file__), E, iters, team], env=env, check=False)
