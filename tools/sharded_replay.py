#!/usr/bin/env python3
"""Per-rank compute time of the coset-sharded step proof (BASELINE config 4), measured on ONE GPU.

The test pool has one-GPU boxes, so the sharded step's scaling over 2 / 4 / 8 GPUs cannot be timed directly.  What can be measured is the
critical path of ONE rank: every rank of a world runs the same transcript, and between the collectives a rank's work does not depend on
where the other ranks' results come from.  So
  1. record  -- the `world` ranks of one sharded step run side by side in this process (one thread + one context per rank, an in-process
                communicator with the vpbs_comm contract: all-gather of cap hashes, device all-gather of quotient values, all-reduce of query
                records); every rank's proof is checked word for word against the single-GPU proof, and every collective's RESULT is logged
                per rank;
  2. replay  -- each rank then runs ALONE on the device, its collectives answered from the log (no waiting, no communication): K timed
                steps with the per-kernel-group HIP-event breakdown.
T_rank(w) is therefore compute + transcript round trips of one rank with the device to itself -- what one GPU of a node would spend per
step before any communication cost; it is NOT a scaling curve (the collectives' latency and the slowest rank's jitter come on top:
three 512-byte all-gathers, one 8 MiB device all-gather, one ~150 kB all-reduce per step over xGMI).

usage: tools/sharded_replay.py [log_n=16] [steps=10] [worlds=2,4,8]  ->  one JSON line  (bench.py --mode sharded-replay prints the same)"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vpbs_amd  # noqa: E402
from vpbs_amd import api, synth  # noqa: E402

GATES = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "poseidon_mds", "arithmetic_ext", "mul_ext", "reducing",
         "reducing_ext", ("random_access", 4), "exponentiation", "coset_interpolation"]
N_CONSTANTS, N_ROUTED, N_PUBLIC_INPUTS = 6, 80, 4173
COLS = dict(synth.STEP_COLS, constants_sigmas=N_CONSTANTS + N_ROUTED)
# kernel groups whose work is split over the cosets a rank owns (the rest is computed by every rank)
SHARDED_GROUPS = ("coset_lde", "leaf_hash", "gate_constraints", "quotient_permutation")


class ThreadWorld:
    """the vpbs_comm contract between `world` threads of this process (the ranks of the recording run)"""

    def __init__(self, world, stage_words, device):
        self.world, self.stage_words = world, stage_words
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = [torch.zeros(stage_words, dtype=torch.int64, device=device) for _ in range(world)]
        self.full = [torch.zeros(stage_words * world, dtype=torch.int64, device=device) for _ in range(world)]
        self.logs = [[] for _ in range(world)]
        self.keep = []

    def comm(self, rank):
        world, log = self.world, self.logs[rank]

        def guard(fn):
            def wrapped(*a):
                try:
                    return fn(*a)
                except BaseException:   # noqa: BLE001 -- must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    self.barrier.abort()
                    return -1
            return wrapped

        @guard
        def allgather(user, local, n, full):
            self.slots[rank] = np.ctypeslib.as_array(local, shape=(n,)).copy()
            self.barrier.wait()
            out = np.concatenate(self.slots)
            np.ctypeslib.as_array(full, shape=(world * n,))[:] = out
            self.barrier.wait()
            log.append(("allgather", out))
            return 0

        @guard
        def allreduce(user, inout, n):
            buf = np.ctypeslib.as_array(inout, shape=(n,))
            self.slots[rank] = buf.copy()
            self.barrier.wait()
            out = self.slots[0].copy()
            for q in range(1, world):
                out += self.slots[q]          # u64 wrap-around sum
            buf[:] = out
            self.barrier.wait()
            log.append(("allreduce", out))
            return 0

        @guard
        def allgather_dev(user, n):
            self.barrier.wait()               # every rank's words are in its local staging buffer (the library synchronised its stream)
            torch.cat([self.local[q][:n] for q in range(world)], out=self.full[rank][:n * world])
            torch.cuda.synchronize()
            self.barrier.wait()
            log.append(("allgather_dev", self.full[rank][:n * world].clone()))
            return 0

        c = api.CommC()
        c.rank, c.world = rank, world
        c.allgather, c.allreduce_sum, c.allgather_dev = api.ALLGATHER_FN(allgather), api.ALLREDUCE_FN(allreduce), api.ALLGATHER_DEV_FN(allgather_dev)
        c.d_stage_local, c.d_stage_full, c.stage_capacity_words = self.local[rank].data_ptr(), self.full[rank].data_ptr(), self.stage_words
        self.keep.append((c, allgather, allreduce, allgather_dev))
        return c


def replay_comm(rank, world, log, stage_words, device):
    """a vpbs_comm for ONE rank running alone: every collective returns what the recording run's collective returned, in call order"""
    local = torch.zeros(stage_words, dtype=torch.int64, device=device)
    full = torch.zeros(stage_words * world, dtype=torch.int64, device=device)
    pos = [0]

    def take(kind):
        k, data = log[pos[0] % len(log)]
        pos[0] += 1
        assert k == kind, "collective order differs from the recording: %s vs %s" % (kind, k)
        return data

    def allgather(user, local_p, n, full_p):
        try:
            np.ctypeslib.as_array(full_p, shape=(world * n,))[:] = take("allgather")
            return 0
        except BaseException:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return -1

    def allreduce(user, inout, n):
        try:
            np.ctypeslib.as_array(inout, shape=(n,))[:] = take("allreduce")
            return 0
        except BaseException:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return -1

    def allgather_dev(user, n):
        try:
            full[:n * world].copy_(take("allgather_dev"))
            torch.cuda.synchronize()
            return 0
        except BaseException:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return -1

    c = api.CommC()
    c.rank, c.world = rank, world
    c.allgather, c.allreduce_sum, c.allgather_dev = api.ALLGATHER_FN(allgather), api.ALLREDUCE_FN(allreduce), api.ALLGATHER_DEV_FN(allgather_dev)
    c.d_stage_local, c.d_stage_full, c.stage_capacity_words = local.data_ptr(), full.data_ptr(), stage_words
    c._keep = (allgather, allreduce, allgather_dev, local, full)
    return c


def timed_steps(ctx, si, comm, steps, warmup=2):
    for _ in range(warmup):
        ctx.prove_step(si, comm)
    ctx.synchronize()
    ctx.timing_enable(1)
    ctx.timing_report()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.prove_step(si, comm)
    ctx.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / steps
    groups = {k: v["ms"] / steps for k, v in ctx.timing_report().items()}
    ctx.timing_enable(0)
    return wall, groups


def run(log_n=16, steps=10, worlds=(2, 4, 8), device=0):
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    n = 1 << log_n
    gates = api.GateSet(GATES)
    digest = np.array([11, 22, 33, 44], np.uint64)
    inputs = synth.step_inputs(log_n, cols=COLS)
    pis = synth.field_elements(0xABCD, N_PUBLIC_INPUTS)
    d_wires = torch.from_numpy(inputs["wires"].view(np.int64)).to(dev)
    d_cs = torch.from_numpy(inputs["constants_sigmas"].view(np.int64)).to(dev)
    sig_ptr = d_cs.data_ptr() + 8 * N_CONSTANTS * n
    torch.cuda.synchronize()

    def step_inputs(ctx, cs):
        return ctx.make_step_inputs(log_n, d_wires.data_ptr(), None, None, cs, digest, pis, on_device=True,
                                    shapes=(COLS["wires"], COLS["zs_partial_products"], COLS["quotient"]), sigmas=sig_ptr, n_routed=N_ROUTED,
                                    n_constants=N_CONSTANTS, gates=gates)

    # the single-GPU proof: the reference every rank must reproduce, and the w = 1 row of the table
    ctx0 = vpbs_amd.Context(device, log_n_max=max(16, log_n))
    cs0 = ctx0.commit_values(inputs["constants_sigmas"])
    si0 = step_inputs(ctx0, cs0)
    want = ctx0.prove_step(si0)
    wall1, groups1 = timed_steps(ctx0, si0, None, steps)
    out = {"what": "per-rank compute time of the coset-sharded step proof, one rank at a time ALONE on one MI355X with its collectives answered "
                   "from a recording (tools/sharded_replay.py): per-kernel-group HIP-event ms per step and wall ms per step of each rank; "
                   "communication time and the wait for the slowest rank are NOT included -- this is not a scaling curve",
           "workload": "synthetic N=1024 step (degree 2^%d, 135/20/16 columns, 14 gate types, %d public inputs), quotient on the device" % (log_n, N_PUBLIC_INPUTS),
           "steps": steps, "single_gpu": {"wall_ms": wall1, "groups_ms": groups1}, "worlds": {}}
    cs0.free()
    ctx0.close()
    stage_words_full = 2 << (log_n + 3)
    for world in worlds:
        tw = ThreadWorld(world, stage_words_full // world, dev)
        ctxs = [vpbs_amd.Context(device, log_n_max=max(16, log_n)) for _ in range(world)]
        errs, proofs, css = [], [None] * world, [None] * world

        def rank_thread(r):
            try:
                torch.cuda.set_device(device)
                ctx, comm = ctxs[r], tw.comm(r)
                cs, local_cap = ctx.commit_sharded_dev(d_cs.data_ptr(), COLS["constants_sigmas"], log_n, r, world)
                css[r] = cs
                tw.logs[r].clear()
                proofs[r] = ctx.prove_step(step_inputs(ctx, cs), comm)
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
                tw.barrier.abort()
        ts = [threading.Thread(target=rank_thread, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise errs[0]
        for r in range(world):
            for key in ("caps", "challenges", "openings", "fri"):
                assert (proofs[r][key] == want[key]).all(), "rank %d of %d: sharded proof differs from the single-GPU proof in %s" % (r, world, key)
        ranks = {}
        for r in range(world):
            comm = replay_comm(r, world, tw.logs[r], stage_words_full // world, dev)
            si = step_inputs(ctxs[r], css[r])
            got = ctxs[r].prove_step(si, comm)
            assert all((got[k] == want[k]).all() for k in ("caps", "openings", "fri")), "replayed rank %d of %d differs" % (r, world)
            wall, groups = timed_steps(ctxs[r], si, comm, steps)
            ranks[str(r)] = {"wall_ms": wall, "groups_ms": groups}
        for r in range(world):
            css[r].free()
            ctxs[r].close()
        slowest = max(ranks, key=lambda k: ranks[k]["wall_ms"])
        g = ranks[slowest]["groups_ms"]
        sharded = sum(v for k, v in g.items() if k in SHARDED_GROUPS)
        out["worlds"][str(world)] = {
            "ranks": ranks, "slowest_rank": int(slowest), "T_rank_ms": ranks[slowest]["wall_ms"],
            "kernel_ms_sharded_groups": sharded, "kernel_ms_replicated_groups": sum(g.values()) - sharded,
            "compute_speedup_vs_single_gpu": wall1 / ranks[slowest]["wall_ms"],
            "collectives_per_step": [k for k, _ in tw.logs[0]],
            "bytes_per_step": {k: int(sum((d.numel() if hasattr(d, "numel") else d.size) * 8 for kk, d in tw.logs[0] if kk == k))
                               for k in ("allgather", "allgather_dev", "allreduce")}}
        del tw
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    res = run(int(a[0]) if len(a) > 0 else 16, int(a[1]) if len(a) > 1 else 10,
              tuple(int(x) for x in a[2].split(",")) if len(a) > 2 else (2, 4, 8), int(os.environ.get("VPBS_PBS_DEVICE", "0")))
    print(json.dumps(res))
