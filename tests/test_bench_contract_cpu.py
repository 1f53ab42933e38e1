"""CPU: the committed record of the driver's bench command (profiles/r06_bench_final.json, written on the GPU box by
`python bench.py --gpus 1 --steps 20 --warmup 5`) carries every field the bench contract names, with consistent arithmetic --
so that an edit of bench.py that drops or renames one shows up here, not at the end of a round."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def E_T(d):
    return d["config"]["envs_per_gpu"] * d["config"]["rollout_steps"]


def _rec():
    return json.load(open(os.path.join(ROOT, "profiles", "r06_bench_final.json")))


def test_bench_line_has_the_contract_fields():
    d = _rec()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and d["steps"] == d["steps_requested"] == 20 and d["warmup"] == 5
    # value = env-steps of all ranks / wall time of exactly `steps` rollouts
    E, T = d["config"]["envs_per_gpu"], d["config"]["rollout_steps"]
    assert abs(d["value"] - E * T / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_roofline_and_cpu_baseline_objects():
    d = _rec()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / the launch's average duration (hipEvents over the timed region)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == r["algorithmic_bytes_per_env_step"] * r["env_steps_per_launch"]
    assert r["traffic"] is None or 0 < r["traffic"] < r["algorithmic_bytes_per_launch"]   # the state stays in registers
    assert r["traffic_source"].startswith("live:")                                         # measured in the run (two PMC passes)
    # the issue model of the step kernel, generated from the build's assembly, beside the measured cycles per step
    s2 = r["secondary"]
    for k in ("bound", "instrs_per_step", "fp64_instrs_per_step", "model_cycles_per_step", "measured_cycles_per_step", "frac",
              "cus_with_two_workgroups", "generated_from"):
        assert k in s2, k
    assert abs(s2["model_cycles_per_step"] - 4 * s2["instrs_per_step"]) < 1e-6 and 0.5 < s2["frac"] < 1.0
    assert abs(s2["frac"] - s2["model_cycles_per_step"] / s2["measured_cycles_per_step"]) < 1e-9
    assert s2["cus_with_two_workgroups"] == s2["workgroups"] - s2["cus"] == 154
    assert s2["spill_reloads_in_step_loops"]["wave0"] == 0 and s2["spill_reloads_in_step_loops"]["walls"] == 0
    st = d["steady_state"]
    assert st["timed_seconds"] >= 0.5 and abs(st["value"] - E_T(d) * st["steps"] / st["timed_seconds"]) <= 1e-6 * st["value"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_closed_loop_records_carry_their_rooflines():
    d = _rec()
    for k in ("closed_loop", "closed_loop_5v5", "closed_loop_5v5_ens5"):
        c = d[k]
        assert c["roofline"]["bound"] == "mfma" and c["roofline"]["peak"] == 157.3
        assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / 157.3) < 1e-9
        # the policy kernel's dense layers run on the bf16 matrix cores with the three-way operand split: the record says so and
        # prices the launch against the split form's own ceiling as well (bf16 dense peak / 6)
        assert c["roofline"]["dtype"].startswith("f32 (bf16x3 split MFMA")
        assert abs(c["roofline"]["frac_of_split_form_peak"] - c["roofline"]["achieved"] / c["roofline"]["peak_split_form"]) < 1e-9
        assert "bf16x3" in c["update_roofline"]["dtype"]
        assert c["update_s"] > 0 and c["train_env_steps_per_s"] > 0
