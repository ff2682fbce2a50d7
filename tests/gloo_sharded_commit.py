"""Launched by test_host_cpu.py under torch.distributed.run (gloo, world_size 2).

Exercises verifiable-fhe-paper_amd/sharding.py end to end on CPU: coset ownership, per-rank subtree hashing (the CPU
oracle stands in for the device kernels here -- test infrastructure only), all_gather of the cap hashes."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen")]
import oracle as orc  # noqa: E402
import vpbs_amd  # noqa: E402
from vpbs_amd import sharding, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    log_n, ncols, rate_bits, cap_height = 6, 7, 3, 4
    n = 1 << log_n
    values = synth.trace(0xC05E7, ncols, log_n)
    full = orc.Batch(values, rate_bits, cap_height, from_values=True)
    coeffs = full.coeffs()
    # rank-local work: the LDE rows of my cosets = my contiguous leaf blocks
    my_cosets = sharding.coset_assignment(rate_bits, rank, world)
    w_big = orc.lib().orc_gl_root_of_unity(log_n + rate_bits)
    blocks = []
    for r in my_cosets:
        shift = orc.lib().orc_gl_mul(7, orc.lib().orc_gl_exp(w_big, r))
        cols = np.stack([orc.coset_lde(coeffs[c], 0, shift) for c in range(ncols)])      # natural order within the coset
        brev = [sharding.brev(q, log_n) for q in range(n)]
        blocks.append(cols[:, brev].T.copy())                                             # leaf order, row-major leaves
    leaves = np.concatenate(blocks)
    lo, hi = sharding.cap_slice(rate_bits, cap_height, rank, world)
    local_tree = orc.Merkle(leaves, cap_height - (world.bit_length() - 1))                # my share of the cap
    local_cap = local_tree.cap()
    assert local_cap.shape[0] == hi - lo
    gathered = sharding.all_gather_cap(torch.from_numpy(local_cap.view(np.int64)))
    cap = gathered.numpy().view(np.uint64)
    assert (cap == full.cap()).all(), "sharded cap differs from the single-process cap"
    # my leaves are exactly the full tree's leaves in my block range
    per = len(my_cosets) * n
    assert (leaves == full.leaves()[rank * per:(rank + 1) * per]).all()
    dist.barrier()
    if rank == 0:
        print("SHARDED_COMMIT_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
