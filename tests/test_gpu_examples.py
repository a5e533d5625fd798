"""The example scripts (the reference's examples on this backend) run end to end on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["neighbor_sampling.py", "transforms_and_loader.py", "random_walk.py"])
def test_example_runs(script):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], cwd=os.path.join(ROOT, "examples"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(r.stdout.strip().splitlines()) >= 3
