"""GPU: the split-bf16 GEMMs that ship against the fp32-MFMA forms they replaced, on the same inputs.

  * fa_policy_kernel: the product library (gemm_cb3: bf16 matrix cores, three-way exact operand split, fp32 accumulate) vs a
    variant library built with -DFA_POLICY_X3=0 (the v_mfma_f32_32x32x2_f32 chain of rounds 2-5), each in its own process:
        python tools/build_variant.py policy_f32 fa_policy.hip -DFA_POLICY_X3=0       (or FA_BUILD_EXPERIMENTS=1 build())
    values / log-probs of a 3v3 x 4096 and a 5v5 x 1000 batch under fresh AND the published (large-logit) policies, deterministic
    actions: the two forms differ by float32 rounding only (<= 5e-6 relative to the rows' scale, observed <= 2.3e-6; identical argmax wherever the
    top-2 logits differ by more than 1e-3).  Skipped when the variant library was not built (it is not part of the product).
  * fa_train_dw3_kernel vs fa_train_dw_kernel (FA_DW_GEMM=f32: both live in the product library): tests/test_gpu_policy.py
    test_dw_gemm_split_bf16_is_fp32_class_against_an_fp64_gemm.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
VARIANT = os.path.join(ROOT, "tools", "_build", "lib_policy_f32.so")

_CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
import emergent_multiagent_strategies_amd as fa
from test_gpu_policy import _policies, _obs
from test_mpnn_cpu import attacker_pool_from_golden
out = {}
for tag, G, A, E in (("3v3", 3, 3, 4096), ("5v5", 5, 5, 1000)):
    pols, packed = _policies(fa, G, A, 7)
    eng = fa.BatchedFortAttack(E, G, A, 20)
    obs = _obs(E, G + A, 11)
    v, act, lp = eng.policy_act(obs, packed[0], packed[1], deterministic=True)
    torch.cuda.synchronize()
    out[tag] = [v.cpu().numpy(), act.cpu().numpy(), lp.cpu().numpy()]
z, pool, G, A = attacker_pool_from_golden(fa.MPNN, os.path.join(%(root)r, "tests", "golden"))
from emergent_multiagent_strategies_amd import mpnn_pack as mp_
obs = torch.from_numpy(z["obs"]).cuda().contiguous()
eng = fa.BatchedFortAttack(obs.shape[0], G, A, 20)
pg = mp_.pack_policy(fa.MPNN(num_agents=G, num_opp_agents=A, num_actions=8).cuda())
for k, p in enumerate(pool):
    v, act, lp = eng.policy_act(obs, pg, mp_.pack_policy(p.cuda()), deterministic=True)
    torch.cuda.synchronize()
    out["published%%d" %% k] = [v.cpu().numpy(), act.cpu().numpy(), lp.cpu().numpy()]
np.savez(sys.argv[1], **{"%%s.%%d" %% (k, i): a for k, v in out.items() for i, a in enumerate(v)})
"""


def test_split_policy_kernel_vs_the_fp32_mfma_form(tmp_path):
    if not os.path.isfile(VARIANT):
        pytest.skip("tools/_build/lib_policy_f32.so not built (python tools/build_variant.py policy_f32 fa_policy.hip -DFA_POLICY_X3=0)")
    import emergent_multiagent_strategies_amd as fa
    if os.path.getmtime(VARIANT) < os.path.getmtime(fa._lib.lib_path()):
        pytest.skip("tools/_build/lib_policy_f32.so predates the product library: rebuild it")
    res = {}
    for name, lib in (("split", None), ("f32", VARIANT)):
        env = dict(os.environ)
        env.pop("FA_LIBRARY", None)
        if lib:
            env["FA_LIBRARY"] = lib
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = np.load(path)
    worst = {}
    for key in sorted({k.rsplit(".", 1)[0] for k in res["split"].files}):
        v_s, a_s, lp_s = [res["split"]["%s.%d" % (key, i)] for i in range(3)]
        v_f, a_f, lp_f = [res["f32"]["%s.%d" % (key, i)] for i in range(3)]
        dv = float(np.abs(v_s - v_f).max() / max(1.0, np.abs(v_f).max()))
        same = a_s == a_f
        dlp = float(np.abs(lp_s - lp_f)[same].max() / max(1.0, np.abs(lp_f).max()))
        worst[key] = (dv, dlp, float(1.0 - same.mean()))
        assert dv <= 5e-6 and dlp <= 5e-6, (key, worst[key])
        assert same.mean() >= 0.999, (key, worst[key])          # (an argmax may flip only between near-tied logits)
    print(json.dumps({k: ["%.1e" % x for x in v] for k, v in worst.items()}))
