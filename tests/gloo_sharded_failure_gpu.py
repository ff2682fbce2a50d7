"""Launched under torch.distributed.run by test_gpu_parity.py (world_size 2 / 4, gloo with a timeout, all ranks on the one GPU of the test
box): failure semantics of the sharded step proof (include/vpbs_prover.h).  A rank that fails between two collectives (VPBS_FAULT_INJECT, or a
rank that cannot start the step at all: vpbs_prove_step_sharded_fail) returns its own error, every other rank returns VPBS_ERR_PEER, all
within seconds and without a timeout firing; the communicator is in step afterwards -- the next sharded proof is the single-GPU proof again."""
import ctypes as C
import datetime
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import vpbs_amd  # noqa: E402
from vpbs_amd import api, sharding, synth  # noqa: E402


def main():
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    rank, world = dist.get_rank(), dist.get_world_size()
    log_n = 10
    torch.cuda.set_device(0)
    ctx = vpbs_amd.Context(0, log_n_max=16)
    inputs = synth.step_inputs(log_n)
    digest = np.array([5, 6, 7, 8], np.uint64)
    pis = synth.field_elements(4242, 33)
    n_constants, n_routed = 5, 80
    sig = np.ascontiguousarray(inputs["constants_sigmas"][n_constants:n_constants + n_routed])
    cs_full = ctx.commit_values(inputs["constants_sigmas"])
    want = ctx.prove_step(ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs_full, digest, pis, sigmas=sig, n_routed=n_routed,
                                               n_constants=n_constants))
    dev_cs = torch.from_numpy(inputs["constants_sigmas"].view(np.int64)).cuda()
    torch.cuda.synchronize()
    cs_shard, _ = sharding.sharded_commit(ctx, dev_cs.data_ptr(), 85, log_n)
    comm = sharding.make_comm(stage_words=(2 << (log_n + 3)) // world)
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs_shard, digest, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)

    def good():
        got = ctx.prove_step(si, comm)
        for key in ("caps", "openings", "fri"):
            assert (got[key] == want[key]).all(), (rank, key)

    good()
    bad_rank = world - 1
    for stage in (1, 2, 3):   # the failing rank throws before its wires / Z / quotient commitment: 0, 1 and 3 collectives into the step
        os.environ["VPBS_FAULT_INJECT"] = "%d:%d" % (bad_rank, stage)
        dist.barrier()
        t = time.perf_counter()
        try:
            ctx.prove_step(si, comm)
            raise AssertionError("rank %d: a proof came back from a step in which rank %d failed" % (rank, bad_rank))
        except api.VpbsError as e:
            took = time.perf_counter() - t
            msg = str(e)
        assert took < 10, (rank, stage, took)
        if rank == bad_rank:
            assert "status -2" in msg and "injected failure before commitment %d" % stage in msg, msg
        else:
            assert "status -5" in msg and "another rank failed" in msg, msg
        del os.environ["VPBS_FAULT_INJECT"]
        good()                # every rank left the failed step through all of its collectives: the communicator is in step
    # a rank that cannot even start the step (its witness generation failed): it walks the step's collectives on the failing side
    dist.barrier()
    t = time.perf_counter()
    if rank == bad_rank:
        rc = api.lib().vpbs_prove_step_sharded_fail(ctx.h, C.byref(si), C.byref(comm), -1)
        assert rc == 0, rc
    else:
        try:
            ctx.prove_step(si, comm)
            raise AssertionError("rank %d: a proof came back although rank %d never proved" % (rank, bad_rank))
        except api.VpbsError as e:
            assert "status -5" in str(e), str(e)
    assert time.perf_counter() - t < 10
    good()
    dist.barrier()
    ctx.close()
    if rank == 0:
        print("SHARDED_FAILURE_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
