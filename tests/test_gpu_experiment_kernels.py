"""GPU: the two round-4 experiment step kernels (lane = (agent, partner); one workgroup barrier per step) live in
csrc/experiments/ and are NOT in the product library -- fa_create refuses their step_kernel values there.  Their parity
tests (tests/experiment_kernels_cases.py: every-build, coincident agents, the one-barrier kernel at 3v3 x 4096 x 128) run
against the variant library tools/_build/lib_experiments.so:
    python tools/build_variant.py experiments --add experiments/fa_step_experiments.hip
(built in the build container; tools/_build/ travels to the GPU box).  Skipped when that library was not built."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VARIANT = os.path.join(os.path.dirname(HERE), "tools", "_build", "lib_experiments.so")


def test_product_library_refuses_the_experiment_kernels():
    import emergent_multiagent_strategies_amd as fa
    assert torch.cuda.is_available()
    assert "lib_experiments" not in fa._lib.lib_path()
    for k in ("pairs", "chain"):
        with pytest.raises(fa.FaError, match="experiment step kernels"):
            fa.BatchedFortAttack(16, 3, 3, 10, step_kernel=k)


def test_experiment_kernels_from_the_variant_library_vs_oracle():
    if not os.path.isfile(VARIANT):
        pytest.skip("tools/_build/lib_experiments.so not built (tools/build_variant.py experiments --add experiments/fa_step_experiments.hip)")
    import ctypes
    import emergent_multiagent_strategies_amd as fa
    lib = ctypes.CDLL(VARIANT)
    missing = [n for n in fa._lib.EXPORTS if not hasattr(lib, n)]
    if missing or os.path.getmtime(VARIANT) < os.path.getmtime(fa._lib.lib_path()):
        pytest.skip("tools/_build/lib_experiments.so is older than the product library (lacks %s): rebuild it with "
                    "__graft_entry__.build() or tools/build_variant.py" % (missing[:3] or "nothing, but predates it"))
    env = dict(os.environ, FA_LIBRARY=VARIANT)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(HERE, "experiment_kernels_cases.py")],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=os.path.dirname(HERE))
    print(r.stdout[-2000:])
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-4000:]
