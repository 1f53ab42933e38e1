"""CPU: the C-ABI library loads, exports every symbol include/fortattack.h declares, its
structs have the layout the Python binding assumes, and the product package has no CPU
fallback and never touches oracle/."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fortattack.h")
PKG = os.path.join(ROOT, "emergent-multiagent-strategies_amd")


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    from emergent_multiagent_strategies_amd import build
    build.build()  # hipcc cross-compiles for gfx950 without a GPU
    return m


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(fa):
    lib = C.CDLL(fa._lib.lib_path())
    syms = _declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), "libfortattack_hip.so lacks %s" % s
    assert sorted(fa._lib.EXPORTS) == syms  # the binding covers exactly the header


def test_struct_layout_matches_header(fa, tmp_path):
    """sizeof / offsetof from a C translation unit including the header == ctypes."""
    fields = {
        "fa_world_consts": ["agent_size", "shoot_win"],
        "fa_config": ["num_envs", "base_seed", "env_offset", "rng_skip_doubles", "track_counters", "world"],
        "fa_step_io": ["actions", "obs_f32", "done", "was_hit", "auto_reset", "num_steps", "act_stride_step"],
        "fa_storage": ["num_steps", "obs", "actions", "done"],
        "fa_state_host": ["pos_x", "alive", "result_count"],
    }
    prog = ["#include <stdio.h>", "#include <stddef.h>", '#include "fortattack.h"', "int main(void){"]
    for st, fs in fields.items():
        prog.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            prog.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    prog.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    py = {"fa_world_consts": fa._lib.WorldConsts, "fa_config": fa._lib.Config, "fa_step_io": fa._lib.StepIO,
          "fa_storage": fa._lib.Storage, "fa_state_host": fa._lib.StateHost}
    for st, fs in fields.items():
        assert int(got[st]) == C.sizeof(py[st]), st
        for f in fs:
            assert int(got["%s.%s" % (st, f)]) == getattr(py[st], f).offset, (st, f)


def test_default_config_is_the_reference_literals(fa):
    cfg = fa._lib.default_config()
    w = cfg.world
    assert (w.agent_size, w.accel, w.max_speed, w.max_rot) == (0.05, 3.0, 3.0, 0.17)  # core.py:32, env_v1:33-35
    assert (w.fort_dim, w.door_x, w.door_y) == (0.15, 0.0, 0.8)                          # env_v1:16-17
    assert (w.dt, w.damping, w.contact_force, w.contact_margin) == (0.1, 0.25, 100.0, 1e-10)  # core.py:121-126
    assert (w.wall_xmin, w.wall_xmax, w.wall_ymin, w.wall_ymax) == (-1.0, 1.0, -0.8, 0.8)     # core.py:128
    assert (w.shoot_rad, w.shoot_win) == (0.8, 3.141592653589793 / 4)                          # core.py:100-101
    assert cfg.rng_mode == fa._lib.FA_RNG_MT19937 and cfg.rng_skip_doubles == -1


def test_no_gpu_means_a_loud_error_not_a_fallback(fa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fa.FaError):
        fa.BatchedFortAttack(4, 3, 3, 10)
    with pytest.raises(fa.FaError):
        fa.make_fortattack_env(100)
    cfg = fa._lib.default_config()
    h = C.c_void_p()
    rc = fa._lib.load().fa_create(C.byref(cfg), C.byref(h))
    assert rc < 0 and fa._lib.load().fa_last_error()


def test_invalid_arguments_are_reported(fa):
    lib = fa._lib.load()
    assert lib.fa_config_default(None) < 0 and b"null" in lib.fa_last_error()
    assert lib.fa_create(None, None) < 0
    assert lib.fa_step(None, None, None) < 0
    assert lib.fa_num_agents(None) < 0


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: no import, path or dlopen of it in the product tree."""
    bad = []
    for dp, _, fns in os.walk(PKG):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"fa_oracle|collector_oracle|ref_harness|libfa_oracle|/oracle", txt):
                    bad.append(fn)
    assert not bad, bad
    # and importing the package does not pull it in
    code = "import sys; import emergent_multiagent_strategies_amd as m; " \
           "print(any('oracle' in k for k in sys.modules))"
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, text=True)
    assert out.strip() == "False"


def test_step_kernel_names_match_the_header_enum(fa):
    """fa_config.step_kernel: the binding's names are the header's FA_KERNEL_* values (a build pinned by name in the tests and
    the A/B tools is the one the library dispatches on)."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    enum = dict((k.lower(), int(v)) for k, v in re.findall(r"FA_KERNEL_([A-Z0-9]+)\s*=\s*(\d+)", text))
    assert enum and enum == fa._lib.STEP_KERNELS
    assert sorted(enum.values()) == list(range(len(enum)))            # dense: fa_create range-checks AUTO .. the last one
    # the experiment kernels are not part of the public enum: their values live beside their source
    exp = open(os.path.join(PKG, "csrc", "experiments", "fa_step_experiments.h")).read()
    exp = dict((k.lower(), int(v)) for k, v in re.findall(r"FA_KERNEL_EXP_(PAIRS|CHAIN)\s*=\s*(\d+)", exp))
    assert exp == fa._lib.EXPERIMENT_STEP_KERNELS and min(exp.values()) > max(enum.values())


def test_product_library_has_no_experiment_kernels(fa):
    """Only what AUTO (or a pinned product build) can launch ships: the round-4 experiment kernels' code objects are absent from
    the product library (they are built into tools/_build/lib_experiments.so by tools/build_variant.py)."""
    blob = open(fa._lib.lib_path(), "rb").read()
    assert b"fa_step_pipe_kernel" in blob and b"fa_step_kernel" in blob
    assert b"_Z19fa_step_pair_kernel" not in blob and b"_Z20fa_step_chain_kernel" not in blob   # (the mangled kernel symbols)
    assert b"_Z19fa_step_pipe_kernel" in blob
    src = open(os.path.join(PKG, "build.py")).read()
    assert "experiments/fa_step_experiments.hip" not in src


def test_buffer_sizes_of_the_binding_match_the_library(fa):
    """The Python side's buffer sizes and pack offsets are the library's (csrc/fa_policy.h, fa_train.h): the packed weights
    (float32 sections + the bf16x3 half), the plain / gradient layout, the transposed pack, the gradient slab."""
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    lib = fa._lib.load()
    assert lib.fa_policy_weight_floats() == mp_.WEIGHT_FLOATS and lib.fa_policy_plain_floats() == mp_.PLAIN_FLOATS
    assert lib.fa_policy_weight_t_floats() == mp_.TRANS_FLOATS and lib.fa_ppo_grad_floats() == mp_.SLAB_FLOATS
    hdr = open(os.path.join(PKG, "csrc", "fa_policy.h")).read()
    off = dict((k, int(v)) for k, v in re.findall(r"#define FA_POFF_([A-Z0-9]+)\s+(\d+)", hdr))
    off3 = dict((k, int(v)) for k, v in re.findall(r"#define FA_POFF3_([A-Z0-9]+)\s+(\d+)", hdr))
    assert off == mp_.POFF and off3 == mp_.POFF3
    sizes = dict(AO=64 * 64, BO=64 * 64, AM=128 * 128, W7=256 * 128, W8=128 * 256, W9=256 * 32)
    at = mp_.PLAIN_FLOATS
    for k in ("AO", "BO", "AM", "W7", "W8", "W9"):          # back to back, 1.5 floats per weight, 16-byte aligned
        assert off3[k] == at and at % 4 == 0, k
        at += sizes[k] * 3 // 2
    assert at == mp_.WEIGHT_FLOATS


def test_one_call_tail_refuses_null_arguments(fa):
    """fa_gae_allreduce_normalize (the several-rank collector tail as one call): a null handle / communicator / buffer is an error
    code with a message, not a crash -- no GPU needed to say so."""
    lib = fa._lib.load()
    rc = lib.fa_gae_allreduce_normalize(None, 0.99, 0.95, None, None, None, None, None, None, None)
    assert rc < 0 and b"fa_gae_allreduce_normalize" in lib.fa_last_error()
