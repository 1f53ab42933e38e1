"""GPU: the reference-flag training entry point runs end to end (two updates, checkpoint in the
reference's wire format, JSON log lines)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_script_two_updates(tmp_path):
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "train_fortattack_amd.py"), "--num-guards", "3", "--num-attackers", "3",
         "--num-processes", "256", "--num-steps", "16", "--num-env-steps", "12", "--num-frames", str(2 * 256 * 16),
         "--num-mini-batch", "2", "--ppo-epoch", "1", "--save-dir", str(tmp_path), "--save-interval", "1"],
        cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [l["update"] for l in lines] == [0, 1] and lines[1]["num_timesteps"] == 2 * 256 * 16
    assert all(abs(l["dist_entropy"]) < 10 and l["fps"] > 0 for l in lines)
    ck = torch.load(os.path.join(str(tmp_path), "ep1.pt"), weights_only=False)
    assert len(ck["models"]) == 6 and ck["ob_rms"] == (None, None)
    assert "encoder.0.weight" in ck["models"][0] and "dist.linear.bias" in ck["models"][-1]
