"""isa_lint (build.py runs it on every translation unit's gfx950 assembly): flags vector instructions placed ahead
of the exec restore of a join block that waves / lanes reach by skipping a divergent region, and only those."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lint():
    spec = importlib.util.spec_from_file_location("_isa_lint", os.path.join(ROOT, "emergent-multiagent-strategies_amd", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


BAD = """
_Z6kernelv:
	v_mov_b32_e32 v1, v0
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	global_store_dword v[6:7], v0, off
.LBB0_2:
	v_accvgpr_write_b32 a132, v186
	s_mov_b32 s86, s77
	s_or_b64 exec, exec, s[2:3]
	v_readlane_b32 s0, v255, 33
	s_endpgm
"""

GOOD = """
_Z6kernelv:
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_f32_e32 v0, v0, v1
	global_store_dword v[6:7], v0, off
.LBB0_2:
	v_writelane_b32 v255, s12, 35
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_write_b32 a132, v186
.LBB0_3:
	v_mul_f32_e32 v166, v166, v158
	s_or_b64 exec, exec, s[4:5]
	s_endpgm
"""


def test_misplaced_split_copy_is_flagged(tmp_path):
    lint = _lint()
    p = tmp_path / "bad.s"
    p.write_text(BAD)
    bad = lint.lint(str(p))
    assert len(bad) == 1
    kernel, block, (ln, line) = bad[0]
    assert kernel == "_Z6kernelv" and block == ".LBB0_2" and line.startswith("v_accvgpr_write_b32 a132")
    assert lint.lint(str(p), only="other_kernel") == []


def test_clean_patterns_pass(tmp_path):
    """Lane ops that ignore exec ahead of the restore, vector code after it, and an if-body that falls through
    into its own exec restore (a block no execz branch targets) are all fine."""
    lint = _lint()
    p = tmp_path / "good.s"
    p.write_text(GOOD)
    assert lint.lint(str(p)) == []


def test_build_runs_the_lint_on_every_source():
    src = open(os.path.join(ROOT, "emergent-multiagent-strategies_amd", "build.py")).read()
    assert "isa_lint.lint(" in src and "--cuda-device-only" in src and "raise RuntimeError" in src


def _model():
    spec = importlib.util.spec_from_file_location("_isa_model", os.path.join(ROOT, "emergent-multiagent-strategies_amd", "isa_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_issue_model_of_the_shipped_step_kernel_is_generated_from_this_build_and_has_no_regression():
    """csrc/fa_isa_model.json (what bench.py quotes as roofline.secondary) is what isa_model.py reads off the assembly of the
    current build, every wave role's step loop is found, and the regression gate of build() is clean: no FLAT memory
    instruction, no spill reload in wave 0's / the walls wave's loop, wave 0's loop within its recorded instruction count."""
    import json
    from emergent_multiagent_strategies_amd import build
    build.build()
    im = _model()
    asm = os.path.join(ROOT, "emergent-multiagent-strategies_amd", "csrc", "_obj", "fa_step_pipe.s")
    if not os.path.isfile(asm):
        build.build(force=True)
    m = im.model(asm)
    assert json.load(open(im.OUT)) == json.loads(json.dumps(m))
    assert im.check(m) == []
    for tag in ("3v3", "5v5_3percu"):
        k = m[tag]
        assert set(k["loops"]) >= {"wave0", "pairs", "walls"}, k["loops"].keys()
        w0 = k["loops"]["wave0"]
        assert w0["barriers"] == 2 and w0.get("spill_reloads", 0) == 0 and w0["model_cycles_per_step"] == round(4 * w0["instructions_per_step"], 1)
        assert k["kernel_totals"].get("flat", 0) == 0 and k["vgprs"] <= 168 and k["scratch_bytes"] == 0


def test_issue_model_gate_flags_regressions():
    im = _model()
    ok = {"3v3": {"kernel_totals": {"flat": 0}, "vgpr_spill_count": 0,
                  "loops": {"wave0": {"instructions": 491, "spill_reloads": 0}, "walls": {"instructions": 387}}}}
    assert im.check(ok) == []
    bad = {"3v3": {"kernel_totals": {"flat": 2}, "vgpr_spill_count": 3,
                   "loops": {"wave0": {"instructions": 600, "spill_reloads": 4}, "walls": {"instructions": 400, "spill_reloads": 16}}}}
    assert len(im.check(bad)) == 5
