"""-m gpu: the multi-GPU exchange step inside the C ABI (rene_comm_* / rene_reduce / rene_gather_tiles, RCCL).  A 1-GPU
box admits one rank per communicator (RCCL refuses two ranks on one device), which still runs every call on the real
path: communicator set-up from a unique id, ncclReduce on the context's stream behind both launch streams, the packing
kernels of the tile gather, the reset the exchange demands.  World sizes 2 and 3 run over gloo on the CPU
(tests/test_dist_gloo.py); the 8-GPU run is the driver's."""
import numpy as np
import pytest

from rene_amd import abi, api, scenes

pytestmark = pytest.mark.gpu


def test_reduce_and_gather_on_a_communicator_of_one():
    s = scenes.cornell_box(96, 80)  # ragged against the 32x32 tiles
    with api.Renderer(s, flags=abi.FLAG_OVERLAP) as r:
        r.render(0, 3)
        r.render(3, 3)
        want = [r.download(l) for l in range(3)]
        r.reset()
        r.comm_init(1, 0, api.comm_unique_id())
        r.render(0, 3)
        r.render(3, 3)
        r.reduce(0)
        got = [r.download(l) for l in range(3)]
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        with pytest.raises(api.ReneError) as e:  # the exchange has touched the records' version words
            r.render(6, 1)
        assert e.value.code == -1 and "rene_reset" in str(e.value)
        # the tile gather: a communicator of one packs its owned tiles, clears the image and places them again -- the
        # two kernels of the exchange on a ragged image, all three layers
        r.reset()
        r.render(0, 3)
        r.render(3, 3)
        r.gather_tiles(0)
        for layer in range(3):
            assert np.array_equal(r.download(layer), want[layer])


def test_comm_errors():
    s = scenes.cornell_box(64, 64)
    with api.Renderer(s) as r:
        for fn in (r.reduce, r.gather_tiles):
            with pytest.raises(api.ReneError) as e:
                fn(0)
            assert e.value.code == -1 and "rene_comm_init" in str(e.value)
        uid = api.comm_unique_id()
        with pytest.raises(api.ReneError):
            r.comm_init(2, 2, uid)
        r.comm_init(1, 0, uid)
        with pytest.raises(api.ReneError):
            r.comm_init(1, 0, uid)  # already in a communicator
        with pytest.raises(api.ReneError):
            r.reduce(1)  # root out of range
