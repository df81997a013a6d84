"""-m gpu: random small scenes through every scheduling of the integrator (tools/fuzz_kernels.py): item loop,
while-while BVH, traversal restart, wavefront, with the (ignored) overlap flag and other work-item cuts.  Matte scenes agree
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


def test_random_volpath_scenes_agree_between_restart_and_while_while_kernels():
    """tools/fuzz_volpath.py: media behind None-material boxes and spheres, distant light and / or emitter, trees deep enough for
    the traversal-restart kernel: bit-identical with the while-while kernel across launch splits and work-item cuts; T1 against
    the oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_volpath.py"), "8"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "all scenes agree" in p.stdout
