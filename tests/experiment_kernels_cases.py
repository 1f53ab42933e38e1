"""Cases of tests/test_gpu_experiment_kernels.py (run by it in a subprocess with FA_LIBRARY = the variant library that
carries csrc/experiments/fa_step_experiments.hip; the file name keeps pytest from collecting it with the product suite):
the product suite's step-kernel parity tests, pinned to the two round-4 experiment kernels."""
import pytest
import torch

import test_gpu_shipped_kernels as shipped

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    assert torch.cuda.is_available()
    lib = m._lib.load()
    assert "lib_experiments" in m._lib.lib_path()
    return m


@pytest.mark.parametrize("G,A,kernel,name", [(3, 3, "pairs", "fa_step_pair_kernel"), (3, 3, "chain", "fa_step_chain_kernel"),
                                             (5, 5, "chain", "fa_step_chain_kernel")])
@pytest.mark.parametrize("collect", [False, True])
def test_experiment_build_vs_oracle(fa, G, A, kernel, name, collect):
    shipped.test_every_step_kernel_build_vs_oracle(fa, G, A, kernel, name, collect)


@pytest.mark.parametrize("kernel", ["pairs", "chain"])
@pytest.mark.parametrize("partner_shot", [True, False])
def test_experiment_coincident_agents(fa, kernel, partner_shot):
    shipped.test_exactly_coincident_agents(fa, kernel, partner_shot)


def test_chain_kernel_full_size_vs_oracle(fa):
    """the one-barrier kernel at config 2's size: its tagged LDS hand-offs under two workgroups per CU"""
    shipped.test_collect_rollout_full_size_vs_oracle(fa, 3, 3, 4096, 128, "fa_step_chain_kernel", "chain")


def test_product_kernels_of_the_variant_library_are_the_products(fa):
    eng = fa.BatchedFortAttack(4096, 3, 3, 100)
    assert eng.step_variant(128) == "fa_step_pipe_kernel"
