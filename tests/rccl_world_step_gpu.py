"""Launched under torch.distributed.run by test_gpu_parity.py::test_native_rccl_world_gt1 when the box shows at least two devices: one rank
per GPU, the library's NATIVE communicator on every rank (vpbs_comm_rccl_create: ncclAllGather / ncclAllReduce over xGMI on the prover's
stream; torch.distributed over gloo only carries the ncclUniqueId), the constants / sigmas commitment coset-sharded over the ranks, then ONE
sharded step proof of a frozen regression case (tests/golden/regression_step_proofs.json): every rank must end with the frozen words and
bytes -- the single-GPU proof, bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import regression_cases as rc  # noqa: E402
import vpbs_amd  # noqa: E402
from vpbs_amd import api, sharding  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(device)
    ctx = vpbs_amd.Context(device, log_n_max=16)
    want_full = os.environ.get("VPBS_TEST_FULL_SIZE") == "1"
    for case in rc.cases(full_size=want_full):
        if case["kind"] == "synthetic":
            continue                                   # the circuit cases run the whole device prover (gate constraints, quotient)
        b = rc.build(case)
        log_n, nconst = b["log_n"], b["n_constants"]
        cs_values = np.ascontiguousarray(b["inputs"]["constants_sigmas"])
        dev_cs = torch.from_numpy(cs_values.view(np.int64)).cuda()
        torch.cuda.synchronize()
        comm = sharding.make_comm_rccl(ctx, stage_words=(2 << (log_n + 3)) // world)
        assert comm.rank == rank and comm.world == world
        # the cap of the sharded commitment: this rank's entries, all-gathered through the native communicator
        cs_shard, local_cap = ctx.commit_sharded_dev(dev_cs.data_ptr(), cs_values.shape[0], log_n, rank, world)
        local = np.ascontiguousarray(local_cap).reshape(-1)
        full = np.zeros(local.size * world, np.uint64)
        assert comm.allgather(comm.user, api._ptr(local), local.size, api._ptr(full)) == 0
        gates = api.GateSet(b["gates"])
        digest = np.array(b.get("digest", rc.DIGEST), np.uint64)
        si = ctx.make_step_inputs(log_n, b["inputs"]["wires"], None, None, cs_shard, digest, b["pis"], sigmas=b["sigma"], n_routed=80,
                                  n_constants=nconst, gates=gates)
        got = ctx.prove_step(si, comm)
        rc.check(case, got)                            # caps, transcript challenges, openings, FRI words = the frozen single-GPU proof
        rc.check_bytes(case, ctx.step_proof_to_bytes(si, nconst, got))
        assert api.verify_step(got, full.reshape(-1, 4), [cs_values.shape[0], 135, 20, 16], digest, b["pis"], log_n, n_constants=nconst,
                               n_routed=80, gates=gates)
        cs_shard.free()
        sharding.free_comm_rccl(comm)
    dist.barrier()
    ctx.close()
    if rank == 0:
        print("RCCL_WORLD_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
