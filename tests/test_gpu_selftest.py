"""-m gpu: the hardware property behind the work-item hand-off of the render_kernel family, checked on the device
the tests run on (rene_amd/csrc/selftest/record_tear.hip): an aligned 16-byte sc0 sc1 store of one lane is seen by an
aligned 16-byte sc0 sc1 load of a lane on another CU / XCD entirely or not at all."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rene_amd", "csrc", "selftest", "record_tear")


def test_sixteen_byte_records_are_never_torn():
    assert os.path.exists(BIN), "build it: make -C rene_amd/csrc"
    p = subprocess.run([BIN, "20000"], capture_output=True, text=True, timeout=120)
    reads, torn = (int(x) for x in p.stdout.split())
    assert p.returncode == 0 and torn == 0, (p.stdout, p.stderr)
    assert reads > 10_000_000  # readers and writers did overlap on the same records: millions of loads raced stores
