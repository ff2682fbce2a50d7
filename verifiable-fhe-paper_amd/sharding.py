"""Per-rank work plan for multi-GPU runs.

Two modes:
  * replicas  -- independent step proofs / vPBS chains per GPU (BASELINE config 3 across GPUs): no collective at all.
  * coset     -- one commitment sharded by LDE coset: with rate 8 the LDE domain 7<w_18> is the union of 8 cosets of
                 size n; coset r lands in the contiguous leaf range [brev3(r)*n, (brev3(r)+1)*n) = two of the 16 cap
                 subtrees, so each rank hashes its own cosets and the only exchange is an all-gather of cap hashes
                 (2^cap_height * 32 B per tree in total).
This module is pure index arithmetic + one torch.distributed all_gather; the compute backend is injected.
"""
import numpy as np


def group_timeout():
    """The timeout a host gives its torch.distributed process group (and the library's RCCL communicator takes from the same variable):
    VPBS_COMM_TIMEOUT_S seconds, default 60.  The status words of the sharded step make every rank leave a failed step by itself; the timeout is
    the backstop for a peer PROCESS that died -- the survivors then fail instead of waiting for ever, and exit non-zero (a process that has
    touched the GPU is never restarted in place)."""
    import datetime
    import os
    try:
        s = float(os.environ.get("VPBS_COMM_TIMEOUT_S", "60"))
    except ValueError:
        s = 60.0
    return datetime.timedelta(seconds=s if s > 0 else 60.0)


def brev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def replica_assignment(n_items, rank, world_size):
    """Items (independent proofs) owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def coset_assignment(rate_bits, rank, world_size):
    """Cosets r (natural index residue mod 2^rate_bits) owned by `rank`, chosen so that each rank owns a contiguous
    range of leaf blocks (hence whole cap subtrees)."""
    n_cosets = 1 << rate_bits
    if n_cosets % world_size:
        raise ValueError("world_size must divide the number of cosets (%d)" % n_cosets)
    per = n_cosets // world_size
    blocks = range(rank * per, (rank + 1) * per)            # leaf-block indices
    return [brev(b, rate_bits) for b in blocks]             # coset whose rows fill that block


def cap_slice(rate_bits, cap_height, rank, world_size):
    """Half-open range of cap entries a rank produces."""
    if cap_height < rate_bits:
        raise ValueError("cap_height < rate_bits: a cap entry would span several cosets")
    per_block = 1 << (cap_height - rate_bits)
    per = ((1 << rate_bits) // world_size) * per_block
    return rank * per, (rank + 1) * per


def all_gather_cap(local_cap, group=None):
    """RCCL/gloo all-gather of the cap entries each rank owns -> the full cap on every rank.

    local_cap: torch.int64/uint64-compatible tensor [entries_per_rank, 4] on the rank's device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = [torch.empty_like(local_cap) for _ in range(world)]
    dist.all_gather(out, local_cap, group=group)
    return torch.cat(out, dim=0)


def sharded_commit(ctx, d_values_ptr, ncols, log_n, is_values=True, group=None, device=None):
    """Coset-sharded PolynomialBatch commit across the ranks of `group` (one GPU per rank).

    Every rank holds the full input matrix (device pointer); it runs the (cheap) iNTT for all columns, the LDE, leaf
    hashing and Merkle subtrees of ITS cosets only (vpbs_commit_sharded_dev), then one all-gather of the
    2^cap_height / world cap hashes per rank assembles the cap on every rank.  Returns (local batch, full cap ndarray).
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    batch, local_cap = ctx.commit_sharded_dev(d_values_ptr, ncols, log_n, rank, world, is_values)
    t = torch.from_numpy(local_cap.view(np.int64))
    if device is not None:
        t = t.to(device)
    cap = all_gather_cap(t, group).cpu().numpy().view(np.uint64)
    return batch, cap


def make_comm(group=None, device=None, stage_words=0, stage_device=None):
    """vpbs_comm backed by torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).

    The host collectives of a sharded step proof move a few KB: an all-gather of cap hashes per commitment and one
    sum-all-reduce of the query records.  `device`: torch device for their staging tensors (required for nccl).
    stage_words > 0 additionally provides the device-resident all-gather used by the on-device quotient (per rank
    stage_words u64 on `stage_device`, e.g. 2 * 2^18 / world for the N = 1024 step): with nccl it is one
    all_gather_into_tensor between device buffers; with gloo the payload takes a detour through host memory."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import api
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def _allgather(user, local, local_words, full):
        try:
            src = np.ctypeslib.as_array(local, shape=(local_words,)).view(np.int64)
            t = torch.from_numpy(src.copy())
            if device is not None:
                t = t.to(device)
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t, group=group)
            dst = np.ctypeslib.as_array(full, shape=(world * local_words,)).view(np.int64)
            dst[:] = torch.cat(out).cpu().numpy()
            return 0
        except Exception:  # must not unwind through the C frame
            import traceback
            traceback.print_exc()
            return -1

    def _allreduce(user, inout, words):
        try:
            buf = np.ctypeslib.as_array(inout, shape=(words,)).view(np.int64)
            t = torch.from_numpy(buf.copy())
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)   # int64 wrap-around sum == u64 sum
            buf[:] = t.cpu().numpy()
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -1

    comm = api.CommC()
    comm.rank, comm.world = rank, world
    comm.allgather = api.ALLGATHER_FN(_allgather)
    comm.allreduce_sum = api.ALLREDUCE_FN(_allreduce)
    comm.user = None
    keep = [_allgather, _allreduce]
    if stage_words:
        sdev = stage_device if stage_device is not None else torch.device("cuda", torch.cuda.current_device())
        local_t = torch.zeros(stage_words, dtype=torch.int64, device=sdev)
        full_t = torch.zeros(stage_words * world, dtype=torch.int64, device=sdev)
        on_gpu_collective = dist.get_backend(group) == "nccl"

        def _allgather_dev(user, local_words):
            try:
                src, dst = local_t[:local_words], full_t[:local_words * world]
                if on_gpu_collective:
                    dist.all_gather_into_tensor(dst, src, group=group)
                    torch.cuda.synchronize(sdev)
                else:
                    parts = [torch.empty(local_words, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(parts, src.cpu(), group=group)
                    dst.copy_(torch.cat(parts))
                    torch.cuda.synchronize(sdev)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return -1

        cb = api.ALLGATHER_DEV_FN(_allgather_dev)
        comm.allgather_dev = cb
        comm.d_stage_local = local_t.data_ptr()
        comm.d_stage_full = full_t.data_ptr()
        comm.stage_capacity_words = stage_words
        keep += [cb, _allgather_dev, local_t, full_t]
    comm._keep = tuple(keep)
    return comm


def make_comm_rccl(ctx, group=None, stage_words=0):
    """vpbs_comm with the library's NATIVE collectives (vpbs_comm_rccl_create: RCCL bound with dlopen, ncclAllGather / ncclAllReduce between
    device buffers on the context's stream).  torch.distributed is used once, to hand rank 0's ncclUniqueId to the other ranks; with a
    single rank (group is None and torch.distributed not initialised) nothing is exchanged.  Call free_comm_rccl(comm) before ctx.close()."""
    import ctypes as C
    import torch
    from . import api
    rank, world = 0, 1
    dist = None
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    lib = api.lib()
    if not lib.vpbs_rccl_available():
        raise api.VpbsError("librccl.so is not loadable")
    uid = (C.c_uint8 * 128)()
    if rank == 0 and lib.vpbs_rccl_unique_id(uid):
        raise api.VpbsError("ncclGetUniqueId failed")
    if world > 1:
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
    comm = api.CommC()
    ctx._check(lib.vpbs_comm_rccl_create(ctx.h, uid, rank, world, stage_words, C.byref(comm)))
    return comm


def free_comm_rccl(comm):
    import ctypes as C
    from . import api
    api.lib().vpbs_comm_rccl_destroy(C.byref(comm))
