"""Convergence-level evidence for the arithmetic bench.py times (tools/trajectory.py): 300 steps of the FULL-SIZE model at the LJSpeech shape,
dropout on with identical masks, the bf16 engine and the fp32 engine in lockstep from the same initial parameters over the same batches, plus
the bf16 engine with an fp32 residual gradient (B2S_DX_BF16=0).  Gate: the mean loss over the last 50 steps of every bf16 arm within 2 % of the
fp32 engine's, the loss has come down, and the bf16 parameters have stayed within the fp32 run's own movement (the per-100-step divergence of
record is profiles/r06_bf16_trajectory.json).  Reference: the training loop of train.py:165-191."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_training_tracks_fp32_over_300_full_size_steps(tmp_path):
    out = str(tmp_path / "traj.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trajectory.py"), "--steps", "300", "--out", out], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stderr or r.stdout)[-2000:]
    d = json.load(open(out))
    arms = d["arms"]
    fp32 = arms["fp32"]
    assert len(fp32["losses"]) == 300 and all(x == x and abs(x) < 1e6 for a in arms.values() for x in a["losses"])
    assert fp32["mean_last_50"] < 0.8 * fp32["losses"][0], "the fp32 run itself must have trained"
    print({k: v for k, v in d.items() if k != "arms"})
    assert d["bf16_vs_fp32_loss_rel_last_50"] < 0.02, d["bf16_vs_fp32_loss_rel_last_50"]
    assert d["bf16_dx_fp32_vs_fp32_loss_rel_last_50"] < 0.02, d["bf16_dx_fp32_vs_fp32_loss_rel_last_50"]
    # the bf16 run stays close to the fp32 run in parameter space: its distance never exceeds the fp32 run's own movement from the initial point
    for step, v in d["bf16_vs_fp32_parameters"].items():
        assert v["all"]["dist_over_movement"] < 1.0, (step, v["all"])
