"""Launched under torch.distributed.run by test_gpu_parity.py::test_native_communicator_world_gt1_on_one_gpu (worlds 2 / 4 / 8, all ranks on
the one GPU of the test box): the library's NATIVE communicator -- vpbs_comm_rccl_create, csrc/comm_rccl.hip: staging, stream polling, the
pinned device-to-host path, timeout + ncclCommAbort -- with more than one rank.  The collective library it binds is the test-only stand-in
tests/fake_rccl.c (VPBS_RCCL_LIB; the ranks exchange through shared memory, stream-ordered like the real calls); torch.distributed over gloo
only carries the 128-byte id.  Scenario (VPBS_TEST_SCENARIO):

  parity   a sharded step proof on every rank = the single-GPU proof bit for bit (words; bytes for the frozen regression circuit), the
           device all-gather of the quotient values included; then a rank that fails between two collectives (VPBS_FAULT_INJECT): its own
           error there, VPBS_ERR_PEER on the others, and the communicator in step afterwards
  absent   the last rank never enters the step: the survivors' first collective gets no answer, times out after VPBS_COMM_TIMEOUT_S,
           the communicator is aborted (ncclCommAbort), they return an error and every later call on it fails at once
"""
import ctypes as C
import datetime
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import vpbs_amd  # noqa: E402
from vpbs_amd import api, sharding, synth  # noqa: E402


def main():
    scenario = os.environ.get("VPBS_TEST_SCENARIO", "parity")
    assert os.environ.get("VPBS_RCCL_LIB", "").endswith("libfake_rccl.so"), "this script is for the stand-in library only"
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = vpbs_amd.Context(0, log_n_max=16)
    log_n = 10
    inputs = synth.step_inputs(log_n)
    digest = np.array([5, 6, 7, 8], np.uint64)
    pis = synth.field_elements(4242, 33)
    n_constants, n_routed = 5, 80
    sig = np.ascontiguousarray(inputs["constants_sigmas"][n_constants:n_constants + n_routed])
    cs_full = ctx.commit_values(inputs["constants_sigmas"])
    # the quotient on the device (zs / quotient = None): the sharded step then uses all three collectives, allgather_dev included
    want = ctx.prove_step(ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs_full, digest, pis, sigmas=sig, n_routed=n_routed,
                                               n_constants=n_constants))
    dev_cs = torch.from_numpy(inputs["constants_sigmas"].view(np.int64)).cuda()
    torch.cuda.synchronize()
    comm = sharding.make_comm_rccl(ctx, stage_words=(2 << (log_n + 3)) // world)
    assert comm.rank == rank and comm.world == world
    fake = C.CDLL(os.environ["VPBS_RCCL_LIB"])           # the copy the library bound (same path: same handle)
    assert fake.ncclAllGather is not None
    cs_shard, local_cap = ctx.commit_sharded_dev(dev_cs.data_ptr(), 85, log_n, rank, world)
    local = np.ascontiguousarray(local_cap).reshape(-1)
    full = np.zeros(local.size * world, np.uint64)
    assert comm.allgather(comm.user, api._ptr(local), local.size, api._ptr(full)) == 0
    assert (full.reshape(-1, 4) == cs_full.cap()).all(), "the all-gathered cap of the sharded commitment is not the single-GPU cap"
    # more than one staging block through the all-reduce: every rank contributes its own vector
    rec = synth.field_elements(100 + rank, 40000)
    total = np.zeros(40000, np.uint64)
    for r in range(world):
        total += synth.field_elements(100 + r, 40000)      # wrapping u64 sum, as ncclSum on ncclUint64
    assert comm.allreduce_sum(comm.user, api._ptr(rec), rec.size) == 0 and (rec == total).all()
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs_shard, digest, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)

    def good():
        got = ctx.prove_step(si, comm)
        for key in ("caps", "openings", "fri"):
            assert (got[key] == want[key]).all(), (rank, key)
        return got

    if scenario == "parity":
        got = good()
        assert ctx.step_proof_to_bytes(si, n_constants, got) == ctx.step_proof_to_bytes(si, n_constants, want)
        bad_rank = world - 1
        for stage in (1, 3):
            os.environ["VPBS_FAULT_INJECT"] = "%d:%d" % (bad_rank, stage)
            dist.barrier()
            t = time.perf_counter()
            try:
                ctx.prove_step(si, comm)
                raise AssertionError("rank %d: a proof came back from a step in which rank %d failed" % (rank, bad_rank))
            except api.VpbsError as e:
                msg = str(e)
            assert time.perf_counter() - t < 10, (rank, stage)
            if rank == bad_rank:
                assert "status -2" in msg and "injected failure" in msg, msg
            else:
                assert "status -5" in msg and "another rank failed" in msg, msg
            del os.environ["VPBS_FAULT_INJECT"]
            good()
        # the frozen regression circuit (whole device prover: gate constraints, quotient), words and bytes
        import regression_cases as rc
        for case in rc.cases(full_size=False):
            if case["kind"] == "synthetic":
                continue
            b = rc.build(case)
            ln, nconst = b["log_n"], b["n_constants"]
            cs_values = np.ascontiguousarray(b["inputs"]["constants_sigmas"])
            d_cs = torch.from_numpy(cs_values.view(np.int64)).cuda()
            torch.cuda.synchronize()
            comm2 = sharding.make_comm_rccl(ctx, stage_words=(2 << (ln + 3)) // world)    # a second communicator beside the first
            shard, _ = ctx.commit_sharded_dev(d_cs.data_ptr(), cs_values.shape[0], ln, rank, world)
            gates = api.GateSet(b["gates"])
            s2 = ctx.make_step_inputs(ln, b["inputs"]["wires"], None, None, shard, np.array(b.get("digest", rc.DIGEST), np.uint64), b["pis"],
                                      sigmas=b["sigma"], n_routed=80, n_constants=nconst, gates=gates)
            g2 = ctx.prove_step(s2, comm2)
            rc.check(case, g2)
            rc.check_bytes(case, ctx.step_proof_to_bytes(s2, nconst, g2))
            shard.free()
            sharding.free_comm_rccl(comm2)
        dist.barrier()
        sharding.free_comm_rccl(comm)
        cs_shard.free()
        ctx.close()
        if rank == 0:
            print("FAKE_RCCL_WORLD_OK world=%d" % world)
    elif scenario == "absent":
        limit = float(os.environ["VPBS_COMM_TIMEOUT_S"])
        good()
        dist.barrier()
        if rank == world - 1:
            # never enters the step; stays alive until the others have reported (a dead PROCESS looks the same to them: no answer)
            dist.barrier()
        else:
            t = time.perf_counter()
            try:
                ctx.prove_step(si, comm)
                raise AssertionError("rank %d: a proof came back although rank %d never joined" % (rank, world - 1))
            except api.VpbsError as e:
                took = time.perf_counter() - t
                msg = str(e)
            assert limit <= took < limit + 8, (rank, took)
            assert "no answer from the other ranks" in msg or "a collective itself failed" in msg, msg
            # the communicator is dead: the next call fails at once, without another timeout
            t = time.perf_counter()
            one = np.zeros(4, np.uint64)
            out = np.zeros(4 * world, np.uint64)
            assert comm.allgather(comm.user, api._ptr(one), 4, api._ptr(out)) != 0 and time.perf_counter() - t < 0.5
            dist.barrier()
        sharding.free_comm_rccl(comm)      # survivors: the dead path -- bounded wait for the stream, then the buffers go back
        if rank == 0:
            print("FAKE_RCCL_ABSENT_OK world=%d" % world)
        # a rank whose communicator timed out must exit without touching the context again (include/vpbs_prover.h); the absent rank is healthy
        sys.stdout.flush()
        dist.destroy_process_group()
        os._exit(0)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
