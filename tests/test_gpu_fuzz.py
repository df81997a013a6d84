"""-m gpu: random small scenes through every scheduling of the integrator (tools/fuzz_kernels.py): item loop,
while-while BVH, traversal restart, wavefront, with overlapping launches and other work-item cuts.  Matte scenes agree
bit for bit, general ones to a last bit, the item-loop family with the BVH family and the oracle statistically (T1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_scenes_agree_across_schedulings():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_kernels.py"), "12"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "all scenes agree" in p.stdout
