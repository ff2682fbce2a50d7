"""Launched under torch.distributed.run by test_gpu_parity.py (world_size 2, gloo, both ranks on the one GPU of the test
box): a step proof with every commitment coset-sharded over the ranks is bit-identical, on every rank, to the
single-GPU step proof, and verifies under the oracle's verifier."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import step_oracle  # noqa: E402
import vpbs_amd  # noqa: E402
from vpbs_amd import sharding, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    log_n = int(os.environ.get("VPBS_TEST_LOG_N", "10"))
    torch.cuda.set_device(0)
    ctx = vpbs_amd.Context(0, log_n_max=16)
    inputs = synth.step_inputs(log_n)
    digest = np.array([5, 6, 7, 8], np.uint64)
    pis = synth.field_elements(4242, 33)
    n_constants, n_routed = 5, 80
    sig = np.ascontiguousarray(inputs["constants_sigmas"][n_constants:n_constants + n_routed])
    # reference: the unsharded proof on this rank
    cs_full = ctx.commit_values(inputs["constants_sigmas"])
    si_full = ctx.make_step_inputs(log_n, inputs["wires"], None, inputs["quotient"], cs_full, digest, pis, sigmas=sig, n_routed=n_routed)
    want = ctx.prove_step(si_full)
    # sharded: constants_sigmas committed per rank, then the sharded step
    dev_cs = torch.from_numpy(inputs["constants_sigmas"].view(np.int64)).cuda()
    torch.cuda.synchronize()
    cs_shard, cs_cap = sharding.sharded_commit(ctx, dev_cs.data_ptr(), 85, log_n)
    assert (cs_cap == cs_full.cap()).all()
    comm = sharding.make_comm(stage_words=(2 << (log_n + 3)) // world)
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, inputs["quotient"], cs_shard, digest, pis, sigmas=sig, n_routed=n_routed)
    got = ctx.prove_step(si, comm)
    for key in ("caps", "challenges", "openings", "fri"):
        assert (got[key] == want[key]).all(), (rank, key)
    assert got["challenger"].state_words() == want["challenger"].state_words()
    assert step_oracle.verify_step(got, cs_cap, [85, 135, 20, 16], digest, pis, log_n)
    # a replicated (unsharded) constants_sigmas batch is accepted too: rank 0 answers its queries
    si2 = ctx.make_step_inputs(log_n, inputs["wires"], None, inputs["quotient"], cs_full, digest, pis, sigmas=sig, n_routed=n_routed)
    got2 = ctx.prove_step(si2, comm)
    assert (got2["fri"] == want["fri"]).all()
    # quotient evaluated on the device: sharded (values all-gathered between the ranks) == single GPU
    si3 = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs_full, digest, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
    want3 = ctx.prove_step(si3)
    si4 = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs_shard, digest, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
    got4 = ctx.prove_step(si4, comm)
    for key in ("caps", "openings", "fri"):
        assert (got4[key] == want3[key]).all(), (rank, "device quotient", key)
    # ... and with the gate constraints of all 14 gate types in the quotient (each rank evaluates them on its own cosets)
    gates = vpbs_amd.api.GateSet(["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext",
                                  "mul_ext", "reducing", "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"])
    assert gates.num_selectors + gates.num_constants <= n_constants + 1
    nc6 = gates.num_selectors + gates.num_constants
    cs6_vals = np.ascontiguousarray(synth.trace(0xBEEF, nc6 + n_routed, log_n))
    sig6 = np.ascontiguousarray(cs6_vals[nc6:])
    cs6_full = ctx.commit_values(cs6_vals)
    dev_cs6 = torch.from_numpy(cs6_vals.view(np.int64)).cuda()
    torch.cuda.synchronize()
    cs6_shard, cs6_cap = sharding.sharded_commit(ctx, dev_cs6.data_ptr(), nc6 + n_routed, log_n)
    si5 = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs6_full, digest, pis, sigmas=sig6, n_routed=n_routed, n_constants=nc6, gates=gates)
    want5 = ctx.prove_step(si5)
    si6 = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs6_shard, digest, pis, sigmas=sig6, n_routed=n_routed, n_constants=nc6, gates=gates)
    got6 = ctx.prove_step(si6, comm)
    for key in ("caps", "openings", "fri"):
        assert (got6[key] == want5[key]).all(), (rank, "device quotient with gates", key)
    assert not (want5["caps"][2] == want3["caps"][2]).all()   # the gate terms really are in the quotient
    dist.barrier()
    ctx.close()
    if rank == 0:
        print("SHARDED_STEP_OK world=%d log_n=%d" % (world, log_n))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
