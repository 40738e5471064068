"""The two example programs run end to end on the device (child processes, small sizes) and report what they claim to report.
examples/evaluate_baseline.py = the reference's experiments/evaluate_baseline.py driven in batch; examples/device_rollout.py = the data path
of SB3's collect_rollouts (gl_gym/RL/experiment_manager.py:95-147) with tensors resident in HBM."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def run(*argv):
    r = subprocess.run([sys.executable, *argv], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_device_rollout_example_collects_a_rollout():
    out = run("examples/device_rollout.py", "--n-envs", "4096", "--n-steps", "8")
    m = re.search(r"4096 envs x 8 steps collected in ([\d.]+) ms: ([\d.e+]+) env-steps/s", out)
    assert m, out
    assert float(m.group(2)) > 1e6                                   # a 4096-env rollout is launch-bound; the point is that it is not host-bound
    assert re.search(r"ODE failures 0\b", out), out
    assert "nan" not in out.lower()


def test_evaluate_baseline_example_runs_a_season():
    out = run("examples/evaluate_baseline.py", "--n-envs", "64", "--season", "2")
    assert re.search(r"64 envs x 193 steps in", out), out
    m = re.search(r"batch-mean cumulative reward ([-\d.]+), EPI ([-\d.]+) EUR/m2, ODE failures (\d+)", out)
    assert m, out
    assert int(m.group(3)) == 0
    assert "nan" not in out.lower()
