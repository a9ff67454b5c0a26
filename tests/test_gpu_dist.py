"""RCCL path on the 1-GPU box: a world of one rank still goes through
ncclCommInitRank / ncclBroadcast / ncclAllReduce in librccl (dlopen'ed by the
library), the file rendezvous, and Communicator.load_weights."""
import numpy as np
import pytest

from planer_amd.irgen import customnet
from tests.conftest import RTOL, load_golden
from tests.test_gpu_nets import check_packed

pytestmark = pytest.mark.gpu


def test_rccl_world_of_one(tmp_path):
    import planer_amd
    from planer_amd import dist
    ctx = planer_amd.hip.context()
    comm = dist.RcclCommunicator(ctx, 0, 1, rdzv_path=str(tmp_path / "rdzv"))
    try:
        assert comm.max_over_ranks(3.5) == 3.5
        comm.barrier()
        g, b = customnet.build()
        net = planer_amd.Net(ctx)
        net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
        comm.load_weights(net, b)                       # upload + ncclBroadcast of the device blob
        y = net(customnet.make_input(1))
        check_packed([y], load_golden("customnet_b1.npz"), RTOL)
        rows = planer_amd.asarray(np.arange(12, dtype=np.float32).reshape(3, 4))
        np.testing.assert_array_equal(comm.allgather_rows(rows).get(), rows.get())
    finally:
        comm.close()
