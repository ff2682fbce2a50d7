"""Launched under torch.distributed.run by test_host_cpu.py (world_size 2, gloo, NO GPU): the status word of the library's checked all-gather
(vpbs_comm_allgather_checked: what the IVC driver wraps its sharded constants / sigmas commitment in) -- a rank that reports a failure still
takes part, every rank returns an error at once, and the communicator stays usable."""
import ctypes as C
import datetime
import os
import sys
import time

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from vpbs_amd import api, sharding  # noqa: E402


def main():
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=30))
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = sharding.make_comm()
    L = api.lib()
    local = np.arange(4, dtype=np.uint64) + 10 * rank
    full = np.zeros(4 * world, np.uint64)
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    # everybody fine: the plain all-gather
    assert L.vpbs_comm_allgather_checked(C.byref(comm), ptr(local), 4, ptr(full), 0) == 0
    assert (full == np.concatenate([np.arange(4, dtype=np.uint64) + 10 * r for r in range(world)])).all()
    # rank 1 reports a failure (-2): it gets its own status back, every other rank VPBS_ERR_PEER (-5) -- within the call, nobody waits
    t = time.perf_counter()
    rc = L.vpbs_comm_allgather_checked(C.byref(comm), ptr(local), 4, ptr(full), -2 if rank == 1 else 0)
    assert rc == (-2 if rank == 1 else -5), (rank, rc)
    assert time.perf_counter() - t < 10
    # ... and the communicator is in step afterwards
    assert L.vpbs_comm_allgather_checked(C.byref(comm), ptr(local), 4, ptr(full), 0) == 0
    dist.barrier()
    if rank == 0:
        print("COMM_FAILURE_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
