"""The backward instantiations of the forward kernel against round 5's kernels on random shapes (tools/bwd_stress.py: each
side in its own process, MGP_BACKWARD_DLT switches the new path off): every cotangent, the three noise models, one /
several responses, batches of 1 .. 1000 neighbourhoods, feature counts on both sides of the 33 .. 40 window in which the
fp32 kernel runs without the LDS behind tile and image."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_agree_with_the_round_5_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bwd_stress.py"), "--cases", "10"], capture_output=True, text=True,
                       cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
