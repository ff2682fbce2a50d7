"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on identical inputs.
Bit-exact everywhere (64-bit modular integer work: no tolerance)."""
import ctypes
import glob
import json
import os

import numpy as np
import pytest

import oracle as orc
import step_oracle
import vpbs_amd
from vpbs_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
P = api.P
rng = np.random.default_rng(20240807)


def rand_field(*shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = vpbs_amd.Context(0, log_n_max=16)
    yield c
    c.close()


# ---------- Poseidon / sponge / Merkle ----------
def test_poseidon_kats_and_random(ctx):
    kat = json.load(open(os.path.join(GOLD, "poseidon_kat.json")))
    states = np.array([k["input"] for k in kat["kats"]], np.uint64)
    out = ctx.poseidon_batch(states)
    for i, k in enumerate(kat["kats"]):
        assert [int(v) for v in out[i]] == k["output"], k["name"]
    edge = np.array([[P - 1] * 12, [0] * 11 + [P - 1], [0xFFFFFFFF] * 12, [0xFFFFFFFF00000000] * 12, [1 << 63] * 12], np.uint64)
    states = np.concatenate([edge, rand_field(3000, 12)])
    want = states.copy()
    orc.lib().orc_poseidon_batch(orc.ptr(want), want.shape[0])
    assert (ctx.poseidon_batch(states) == want).all()


@pytest.mark.parametrize("length", [5, 8, 9, 16, 20, 32, 85, 135])
def test_hash_rows(ctx, length):
    rows = rand_field(300, length)
    # edge rows: the sponge carries u64 RESIDUES from one absorb to the next (only the digest is made canonical), so the values that sit at the
    # wrap-around boundaries of the field go through it as inputs too
    edge = [P - 1, 0, 1, (1 << 32) - 1, 1 << 32, P - (1 << 32), (1 << 63), P - 2]
    rows[2] = P - 1
    rows[3] = 0
    rows[4] = [edge[k % len(edge)] for k in range(length)]
    rows[5] = [edge[(3 * k + 1) % len(edge)] for k in range(length)]
    got = ctx.hash_rows(rows)
    for i in (0, 1, 2, 3, 4, 5, 150, 299):
        assert list(got[i]) == list(orc.hash_no_pad(rows[i]))


@pytest.mark.parametrize("leaf_len,n_leaves,cap_h", [(3, 64, 2), (4, 32, 4), (7, 256, 4), (20, 128, 0), (135, 512, 4), (32, 16, 4)])
def test_merkle_cap(ctx, leaf_len, n_leaves, cap_h):
    leaves = rand_field(n_leaves, leaf_len)
    assert (ctx.merkle_cap(leaves, cap_h) == orc.Merkle(leaves, cap_h).cap()).all()


# ---------- NTT ----------
@pytest.mark.parametrize("log_n", [1, 2, 3, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])   # 19 = the quotient's transform at degree 2^16; 22 = the launchers' limit
def test_intt(ctx, log_n):
    vals = rand_field(3, 1 << log_n)
    got = ctx.intt(vals)
    for c in range(3):
        assert (got[c] == orc.fft(vals[c], inverse=True)).all()


@pytest.mark.parametrize("log_n,rate_bits,shift", [(1, 3, 7), (3, 3, 7), (7, 3, pow(7, 256, P)), (11, 3, pow(7, 16, P)), (12, 3, 7),
                                                   (13, 0, 7), (15, 3, 7), (10, 1, 49), (14, 2, 7), (16, 3, 7), (17, 1, 7), (18, 0, 49),
                                                   (19, 1, pow(7, 16, P)), (20, 0, 7), (21, 0, 7), (22, 0, 7)])   # 21 / 22: the strided NR = 3 shapes
def test_coset_lde_leaf_order(ctx, log_n, rate_bits, shift):
    coeffs = rand_field(2, 1 << log_n)
    got = ctx.coset_lde(coeffs, rate_bits, shift)
    log_big = log_n + rate_bits
    idx = np.array([int(format(j, "0%db" % log_big)[::-1], 2) if log_big else 0 for j in range(1 << log_big)])
    for c in range(2):
        nat = orc.coset_lde(coeffs[c], rate_bits, shift)
        assert (got[c] == nat[idx]).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ntt_params_*.json"))))
def test_negacyclic_golden(ctx, path):
    """reference KAT (crypto/poly.rs:195-208) on the device kernel, batched with random rows around it"""
    g = json.load(open(path))
    n = g["N"]
    batch = np.concatenate([rand_field(2, n), np.array([g["TESTG"]], np.uint64), rand_field(1, n)])
    fw = ctx.negacyclic_ntt(batch)
    assert [int(v) for v in fw[2]] == g["TESTGHAT"]
    bw = ctx.negacyclic_ntt(fw, inverse=True)
    assert (bw == batch).all()
    roots, inv, ninv = orc.negacyclic_params(g["LOGN"])
    assert (fw[0] == orc.negacyclic_forward(batch[0], roots)).all()


# ---------- PolynomialBatch ----------
@pytest.mark.parametrize("log_n,ncols,from_values", [(5, 3, True), (6, 4, True), (10, 9, True), (12, 5, False), (13, 135, True),
                                                     (16, 3, True), (16, 2, False)])
def test_commit_matches_oracle(ctx, log_n, ncols, from_values):
    data = rand_field(ncols, 1 << log_n)
    data[0] = P - 1   # a column of the largest field element (as coefficients: every term of the openings' lazy sums at its maximum)
    want = orc.Batch(data, 3, 4, from_values=from_values)
    got = (ctx.commit_values if from_values else ctx.commit_coeffs)(data)
    assert (got.cap_at_commit == want.cap()).all() and (got.cap() == want.cap()).all()
    assert (got.coeffs() == want.coeffs()).all()
    L = 1 << (log_n + 3)
    for idx in (0, 1, L // 2 + 3, L - 1):
        leaf, sib = got.open(idx)
        wleaf, wsib = want.open(idx)
        assert (leaf == wleaf).all() and (sib == wsib).all()
    rows = got.lde_rows(3, 5, step=2)
    for k in range(5):
        assert (rows[k] == want.lde_row(3 + k, 2)).all()
    zeta = rand_field(2)
    assert (got.eval_ext(zeta) == want.eval_ext(zeta)).all()
    got.free()


@pytest.mark.parametrize("n_shards", [2, 4, 8])
def test_coset_sharded_commit_matches_unsharded(ctx, n_shards):
    """each rank commits only its cosets; caps concatenate to the full cap; local leaves open to it.
    (All shards run one after the other on the single GPU of the test box; the collective is covered by the gloo test.)"""
    import torch
    log_n, ncols = 10, 11
    vals = rand_field(ncols, 1 << log_n)
    full = ctx.commit_values(vals)
    want_cap = full.cap()
    assert (want_cap == orc.Batch(vals, 3, 4, True).cap()).all()
    dev = torch.from_numpy(vals.view(np.int64)).cuda()
    torch.cuda.synchronize()
    L = 1 << (log_n + 3)
    caps = []
    for shard in range(n_shards):
        b, local_cap = ctx.commit_sharded_dev(dev.data_ptr(), ncols, log_n, shard, n_shards)
        caps.append(local_cap)
        lo = shard * L // n_shards
        for idx in (lo, lo + 17, lo + L // n_shards - 1):
            leaf, sib = b.open(idx)
            wleaf, wsib = full.open(idx)
            assert (leaf == wleaf).all() and (sib == wsib).all()
            assert orc.merkle_verify(leaf, idx, want_cap, 4, sib)
        with pytest.raises(api.VpbsError):
            b.open((lo + L // n_shards) % L)   # a leaf owned by another rank
        b.free()
    assert (np.concatenate(caps) == want_cap).all()
    full.free()


@pytest.mark.parametrize("log_n,ncols", [(1, 3), (2, 4), (3, 1), (4, 2), (6, 5), (2, 9)])
def test_commit_edge_shapes(ctx, log_n, ncols):
    """tiny degrees (tree barely larger than its cap) and leaves of <= 4 elements (hash_or_noop does not hash them)"""
    data = rand_field(ncols, 1 << log_n)
    want = orc.Batch(data, 3, 4, from_values=True)
    got = ctx.commit_values(data)
    assert (got.cap() == want.cap()).all()
    L = 1 << (log_n + 3)
    for idx in (0, L - 1):
        leaf, sib = got.open(idx)
        wleaf, wsib = want.open(idx)
        assert (leaf == wleaf).all() and sib.shape == wsib.shape and (sib == wsib).all()
    zeta = rand_field(2)   # the openings' power table at tiny degrees: fewer than 16 entries per thread of its walk, or per table
    assert (got.eval_ext(zeta) == want.eval_ext(zeta)).all()
    got.free()


def test_invalid_arguments_are_errors(ctx):
    with pytest.raises(api.VpbsError):
        ctx.commit_values(rand_field(2, 1 << 17))           # log_n > log_n_max of the context
    b = ctx.commit_values(rand_field(2, 16))
    with pytest.raises(api.VpbsError):
        b.open(1 << 20)                                      # leaf index out of range
    with pytest.raises(api.VpbsError):
        b.lde_rows(0, 4, step=1 << 10)                       # row * step beyond the LDE
    b.free()


@pytest.mark.parametrize("world,log_n", [(2, 10), (4, 12), (8, 9), (2, 16), (8, 16)])   # 16: the degree of the N = 1024 step circuit
def test_sharded_step_proof_multi_rank(world, log_n):
    """SURVEY.md 8e / BASELINE config 4: one step proof sharded over `world` ranks (gloo collectives, all ranks on this
    box's single GPU) == the single-GPU proof, bit for bit, on every rank."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tests", "gloo_sharded_step_gpu.py")
    env = dict(os.environ, VPBS_TEST_LOG_N=str(log_n))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), script],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARDED_STEP_OK world=%d" % world in r.stdout


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_failure_semantics(world):
    """a rank of a sharded step that fails outside a collective (before its 1st / 2nd / 3rd commitment, or before the step
    starts) does not leave the others hanging: it returns its own error, every other rank VPBS_ERR_PEER, all within 10 s, no rank returns a
    proof, and the next step on the same communicator is the single-GPU proof again (gloo, all ranks on this box's GPU)."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tests", "gloo_sharded_failure_gpu.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29570 + world), script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARDED_FAILURE_OK world=%d" % world in r.stdout


def test_a_failed_launch_is_not_swallowed_by_the_next_wait():
    """vpbs::stream_sync (every wait of the library: d2h_sync, vpbs_ctx_synchronize, the destructors) used to call hipGetLastError
    after queuing its completion marker and so CLEARED a launch failure left pending on the thread -- the stage-end checks then saw success
    and a proof could be built on buffers no kernel had written.  A launch that fails ahead of a wait (here: a null kernel through the same
    HIP runtime, on this thread) must come back from the wait as an error -- from that ONE call (reporting it consumes it)."""
    import torch  # noqa: F401  (its bundled HIP runtime is the one the library is bound to)
    c = vpbs_amd.Context(0, log_n_max=10)
    c.synchronize()

    class Dim3(ctypes.Structure):
        _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, Dim3, Dim3, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipLaunchKernel.restype = ctypes.c_int
    hip.hipPeekAtLastError.restype = ctypes.c_int
    hip.hipGetLastError.restype = ctypes.c_int
    assert hip.hipPeekAtLastError() == 0
    rc = hip.hipLaunchKernel(None, Dim3(1, 1, 1), Dim3(1, 1, 1), None, 0, None)
    assert rc != 0 and hip.hipPeekAtLastError() == rc, rc            # a launch-class error is pending on this thread
    with pytest.raises(api.VpbsError):
        c.synchronize()
    # the call that reported the failure consumed it -- exactly one call fails, and a host that knows nothing of hipGetLastError
    # (C++, Rust) goes on with the same context
    assert hip.hipPeekAtLastError() == 0
    c.synchronize()
    data = rand_field(3, 1 << 8)
    assert (c.commit_values(data).cap() == orc.Batch(data, 3, 4, True).cap()).all()
    c.close()


def test_hand_scheduled_arithmetic_on_edge_values():
    """the non-canonical-residue paths of the inline-asm products (gl::mul_nc, dot2_nc, mad_nc, add_nn, fold96), of the gate
    kernels' lazy algebra products (times7, mul_lazy, fma2, select_lerp) and the permutation built from them, over every pair of
    {0, 1, p-1, p, 2^64-1, 2^32-1, 2^32, 2^64-2^32, 2^63} plus 65 536 random operands, against big-integer arithmetic: tools/test_asm (built by
    __graft_entry__.build()).  Random field elements reach a residue >= p with probability 2^-32: without this the wrap branches never run."""
    import subprocess
    import __graft_entry__ as entry
    exe = entry.build_asm_edge_tool()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASM_EDGE_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    for name in ("mul_nc", "mul2_nc", "dot2_nc", "mad_nc", "fold96", "add_nn", "add_a", "sub_a", "sub (asm)", "reduce96 / reduce128", "times7", "mul_lazy / fma2", "select_lerp",
                 "permute"):
        assert name + ": 0 mismatch" in r.stdout, (name, r.stdout)


# ---------- FRI ----------
def _fri_case(ctx, log_n, cols, **over):
    datas = [rand_field(nc, 1 << log_n) for nc in cols]
    o_batches = [orc.Batch(d, 3, 4, from_values=(i != 3)) for i, d in enumerate(datas)]
    g_batches = [(ctx.commit_values if i != 3 else ctx.commit_coeffs)(d) for i, d in enumerate(datas)]
    ch = orc.ChallengerState()
    for o in o_batches:
        ch.observe(o.cap())
    zeta = ch.get_ext()
    batches, zeta_next = step_oracle.step_batches(list(cols), 2, zeta, log_n)
    openings = np.concatenate([o.eval_ext(zeta) for o in o_batches] + [o_batches[2].eval_ext(zeta_next)[:2]])
    ch.observe(openings)
    gch = api.ChallengerState()
    for o in g_batches:
        gch.observe(o.cap())
    assert list(gch.get_ext()) == list(zeta)
    gch.observe(openings)
    return o_batches, g_batches, ch, gch, batches, openings, orc.fri_params(log_n, **over), api.fri_params(log_n, **over)


@pytest.mark.parametrize("log_n,over", [(5, {}), (6, {}), (9, {}), (12, {}), (8, {"mul_final_by_x": 1}), (7, {"pow_bits": 5, "num_query_rounds": 3}),
                                        (10, {"pow_bits": 0}), (14, {}), (16, {})])
def test_fri_prove_bit_exact(ctx, log_n, over):
    """log_n = 5: ConstantArityBits(4,5) gives zero folding rounds (the final polynomial is the whole polynomial)."""
    ob, gb, ch, gch, batches, openings, op, gp = _fri_case(ctx, log_n, (4, 6, 3, 2), **over)
    ch_v = ch.clone()
    want = orc.prove_openings(ob, batches, ch, op, log_n)
    got = ctx.fri_prove(gb, batches, gch, gp)
    assert got.shape == want.shape
    assert (got == want).all()
    assert gch.state_words() == ch.state_words()
    total = sum(o.ncols for o in ob)
    assert orc.verify_fri([o.cap() for o in ob], [o.ncols for o in ob], batches, [openings[:total], openings[total:]], ch_v, op, log_n, got)


@pytest.mark.parametrize("log_n,over", [(9, {}), (7, {"pow_bits": 5, "num_query_rounds": 3}), (12, {"pow_bits": 17})])
def test_fri_prove_bit_exact_with_the_settings_of_a_shared_gpu(ctx, log_n, over):
    """VPBS_OPT_WIDE_THRESHOLD below its default: the one-lane Poseidon form on small tree levels and the proof-of-work search in rounds of 2^15
    candidates with early exits -- the smallest valid nonce, the same proof (pow_bits 17: the first range of 2^18 takes up to eight rounds)"""
    before = ctx.get_option("wide_threshold")
    ctx.set_option("wide_threshold", 2048)
    try:
        test_fri_prove_bit_exact(ctx, log_n, over)
    finally:
        ctx.set_option("wide_threshold", before)


def test_fri_forced_pow_and_invalid(ctx):
    ob, gb, ch, gch, batches, openings, op, gp = _fri_case(ctx, 6, (4, 6, 3, 2), pow_bits=6)
    g2 = gch.clone()
    got = ctx.fri_prove(gb, batches, gch, gp)
    w = int(got[-1])
    again = ctx.fri_prove(gb, batches, g2.clone(), gp, forced_pow=w)
    assert (again == got).all()
    bad = next(x for x in range(1, 1000) if x != w and not _pow_valid(g2, gb, batches, gp, ctx, x))
    with pytest.raises(api.VpbsError):
        ctx.fri_prove(gb, batches, g2.clone(), gp, forced_pow=bad)


def _pow_valid(ch, gb, batches, gp, ctx, w):
    try:
        ctx.fri_prove(gb, batches, ch.clone(), gp, forced_pow=w)
        return True
    except api.VpbsError:
        return False


# ---------- permutation argument (a12) ----------
@pytest.mark.parametrize("log_n,n_routed,deg,nc", [(3, 10, 4, 2), (8, 80, 8, 2), (11, 80, 8, 2), (9, 17, 8, 1), (10, 8, 8, 3), (16, 80, 8, 2),
                                                   (7, 80, 8, 1), (6, 80, 8, 3), (5, 72, 8, 2), (5, 80, 4, 2)])   # (80, 8): one kernel, one inversion per row; else three kernels
def test_partial_products_match_oracle(ctx, log_n, n_routed, deg, nc):
    wires, sig = rand_field(n_routed + 3, 1 << log_n), rand_field(n_routed, 1 << log_n)
    betas, gammas = [int(x) for x in rand_field(nc)], [int(x) for x in rand_field(nc)]
    # rows of boundary values: the one-kernel path multiplies and inverts on u64 residues
    wires[:, 1], sig[:, 1] = P - 1, P - 1
    wires[:, 2], sig[:, 2] = 0, (1 << 32) - 1
    wires[:, 3], sig[:, 3] = P - (1 << 32), 1 << 32
    got = ctx.partial_products(wires[:n_routed], sig, betas, gammas, deg)
    want = orc.partial_products(wires[:n_routed], sig, betas, gammas, deg)
    assert got.shape == want.shape and (got == want).all()
    assert (got[:nc, 0] == 1).all()  # Z(1) = 1


@pytest.mark.parametrize("n_routed,col", [(8, 3), (80, 3), (80, 79)])   # three-kernel path; the one-kernel path (shared inversion), first / last chunk
def test_partial_products_zero_denominator_is_an_error(ctx, n_routed, col):
    wires, sig = rand_field(n_routed, 16), rand_field(n_routed, 16)
    beta, gamma = 5, 9
    wires[col][7] = (-(beta * int(sig[col][7]) + gamma)) % P   # den_col(row 7) = 0
    with pytest.raises(api.VpbsError):
        ctx.partial_products(wires, sig, [beta], [gamma])
    ctx.partial_products(rand_field(n_routed, 16), rand_field(n_routed, 16), [beta], [gamma])   # the flag does not stick to the context


# ---------- quotient, permutation part (a13) ----------
def _leaf_order(nat, log_big):
    idx = np.array([int(format(j, "0%db" % log_big)[::-1], 2) for j in range(1 << log_big)])
    return np.ascontiguousarray(nat[:, idx])


@pytest.mark.parametrize("log_n,n_routed,n_constants,with_gates", [(4, 8, 0, False), (6, 20, 3, True), (9, 80, 5, False), (8, 80, 5, True),
                                                                  (16, 80, 4, False)])
def test_quotient_permutation_matches_oracle(ctx, log_n, n_routed, n_constants, with_gates, nc=2):
    import torch
    n = 1 << log_n
    wires_v, sig_v, const_v = rand_field(n_routed + 4, n), rand_field(n_routed, n), rand_field(n_constants, n)
    betas, gammas, alphas = ([int(x) for x in rand_field(nc)] for _ in range(3))
    wires_v[:, 1], sig_v[:, 1] = P - 1, P - 1          # boundary values: the kernel's inner loop runs on u64 residues
    wires_v[:, 2], sig_v[:, 2] = 0, (1 << 32) - 1
    zs_v = orc.partial_products(wires_v[:n_routed], sig_v, betas, gammas)
    cs = ctx.commit_values(np.concatenate([const_v, sig_v]) if n_constants else sig_v)
    wb, zb = ctx.commit_values(wires_v), ctx.commit_values(zs_v)
    gate_nat, gate_dev = None, None
    if with_gates:
        gate_nat = rand_field(nc, 8 * n)
        gate_dev = torch.from_numpy(_leaf_order(gate_nat, log_n + 3).view(np.int64)).cuda()
        torch.cuda.synchronize()
    got = ctx.quotient_permutation(cs, n_constants, wb, zb, n_routed, betas, gammas, alphas,
                                   gate_terms_dev=gate_dev.data_ptr() if with_gates else None)
    want = orc.quotient_permutation(wb.coeffs()[:n_routed], cs.coeffs()[n_constants:], zb.coeffs(), betas, gammas, alphas,
                                    gate_terms=gate_nat)
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("nc,with_gates", [(1, True), (3, False), (4, True)])
def test_quotient_permutation_other_challenge_counts(ctx, nc, with_gates):
    """plonky2's standard two challenges run a kernel specialised for them; any other count the run-time form"""
    test_quotient_permutation_matches_oracle(ctx, 7, 80, 3, with_gates, nc=nc)


def _copy_constraint_instance(log_n, n_routed):
    n = 1 << log_n
    w = orc.lib().orc_gl_root_of_unity(log_n)
    perm = rng.permutation(n_routed * n)
    vals = np.zeros(n_routed * n, dtype=np.uint64)
    seen = np.zeros(n_routed * n, bool)
    for s0 in range(n_routed * n):
        if not seen[s0]:
            v = rand_field(1)[0]
            t = s0
            while not seen[t]:
                seen[t] = True; vals[t] = v; t = perm[t]
    sig = np.zeros((n_routed, n), np.uint64)
    wp = [pow(w, r, P) for r in range(n)]
    kp = [pow(7, c, P) for c in range(n_routed)]
    for pos in range(n_routed * n):
        tc, tr = divmod(int(perm[pos]), n)
        sig[pos // n][pos % n] = kp[tc] * wp[tr] % P
    return vals.reshape(n_routed, n), sig


@pytest.mark.parametrize("log_n", [6, 10])
def test_copy_constraint_proof_end_to_end(ctx, log_n):
    """A complete proof of a copy-constraint-only circuit, every prover stage on the GPU (wires commit -> Z / partial
    products -> commit -> quotient -> commit -> openings -> FRI), checked by the restated plonky2 verifier: the FRI
    proof verifies AND vanishing(zeta) == Z_H(zeta) * t(zeta).  A witness that violates one copy constraint still yields
    consistent commitments but fails the vanishing identity."""
    n_routed, n_constants, n_wires = 80, 5, 135
    n = 1 << log_n
    routed, sig = _copy_constraint_instance(log_n, n_routed)
    consts = rand_field(n_constants, n)
    pis = synth.field_elements(77, 12)

    def prove_and_check(routed_vals):
        wires = np.concatenate([routed_vals, rand_field(n_wires - n_routed, n)])   # advice wires are unconstrained
        cs = ctx.commit_values(np.concatenate([consts, sig]))
        si = ctx.make_step_inputs(log_n, wires, None, None, cs, DIGEST, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
        proof = ctx.prove_step(si)
        ncols = [n_constants + n_routed, n_wires, 20, 16]
        assert step_oracle.verify_step(proof, cs.cap(), ncols, DIGEST, pis, log_n)
        op = proof["openings"]
        cs_z, w_z = op[:ncols[0]], op[ncols[0]:ncols[0] + n_wires]
        zs_all = op[ncols[0] + n_wires:ncols[0] + n_wires + 20]
        q_z = op[ncols[0] + n_wires + 20:ncols[0] + n_wires + 36]
        zs_next = op[ncols[0] + n_wires + 36:]
        ch = proof["challenges"]
        betas, gammas, alphas, zeta = ch[0:2], ch[2:4], ch[4:6], ch[6:8]
        ok = orc.check_vanishing_at_zeta(w_z[:n_routed], cs_z[n_constants:], zs_all[:2], zs_next, zs_all[2:], q_z, log_n,
                                         [int(b) for b in betas], [int(g) for g in gammas], [int(a) for a in alphas], zeta)
        # the product's own host verifier must reach the same verdict (FRI alone is fine either way)
        assert api.verify_step_fri_only(proof, cs.cap(), ncols, DIGEST, pis, log_n)
        assert api.verify_step(proof, cs.cap(), ncols, DIGEST, pis, log_n, check_permutation=True, n_constants=n_constants,
                               n_routed=n_routed) == ok
        cs.free()
        return ok

    assert prove_and_check(routed)
    bad = routed.copy(); bad[7][5] = (int(bad[7][5]) + 1) % P
    assert not prove_and_check(bad)


@pytest.mark.parametrize("log_n,cols", [(8, None), (16, {"constants_sigmas": 12, "wires": 12, "zs_partial_products": 4, "quotient": 16})])
def test_step_proof_with_device_quotient_bit_exact(ctx, log_n, cols):
    """log_n = 16: the degree of the N = 1024 step circuit, with fewer columns so that the oracle prover finishes in seconds --
    the 2^19-point inverse transform of the quotient stage included"""
    n_constants, n_routed = (5, 80) if cols is None else (2, 10)
    inputs = synth.step_inputs(log_n, cols=cols)
    inputs["quotient"] = None
    pis = synth.field_elements(0xD00D, 20)
    sig = np.ascontiguousarray(inputs["constants_sigmas"][n_constants:n_constants + n_routed])
    cs = ctx.commit_values(inputs["constants_sigmas"])
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs, DIGEST, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
    got = ctx.prove_step(si)
    want = step_oracle.prove_step(inputs, DIGEST, pis, log_n, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
    for key in ("caps", "challenges", "openings", "fri"):
        assert (got[key] == want[key]).all(), key


# ---------- step proof ----------
DIGEST = np.array([11, 22, 33, 44], np.uint64)


def _step(ctx, log_n, cols=None, instance=0):
    inputs = synth.step_inputs(log_n, instance, cols)
    pis = synth.field_elements(0xABCD + instance, 77)
    cs = ctx.commit_values(inputs["constants_sigmas"])
    si = ctx.make_step_inputs(log_n, inputs["wires"], inputs["zs_partial_products"], inputs["quotient"], cs, DIGEST, pis)
    return inputs, pis, cs, si, ctx.prove_step(si)


@pytest.mark.parametrize("log_n,cols", [(6, {"constants_sigmas": 5, "wires": 9, "zs_partial_products": 4, "quotient": 3}),
                                        (9, None), (12, None),
                                        (16, {"constants_sigmas": 5, "wires": 9, "zs_partial_products": 4, "quotient": 3})])
def test_step_proof_bit_exact(ctx, log_n, cols):
    """BASELINE config 1 sizes at log_n = 12 (N = 8 ring: degree 2^12, 135/20/16/85 columns)."""
    inputs, pis, cs, si, got = _step(ctx, log_n, cols)
    want = step_oracle.prove_step(inputs, DIGEST, pis, log_n)
    assert (cs.cap() == want["cs_cap"]).all()
    for key in ("caps", "challenges", "openings", "fri"):
        assert (got[key] == want[key]).all(), key
    assert got["challenger"].state_words() == want["challenger"].state_words()
    assert step_oracle.verify_step(got, want["cs_cap"], want["ncols"], DIGEST, pis, log_n)
    n_constants = min(5, want["ncols"][0])
    blob = ctx.step_proof_to_bytes(si, n_constants, got)
    assert blob == step_oracle.to_bytes(want, want["ncols"], n_constants, pis, log_n)
    if cols is None:   # standard column counts: the bytes parse back into the same proof, which the host verifier accepts
        back, back_pis = api.step_proof_from_bytes(blob, want["ncols"], log_n, n_constants)
        assert all((back[k].reshape(-1) == got[k].reshape(-1)).all() for k in ("caps", "openings", "fri")) and (back_pis == pis).all()
        assert api.verify_step_fri_only(back, want["cs_cap"], want["ncols"], DIGEST, back_pis, log_n)


@pytest.mark.parametrize("pos", [dict(fri_mul_final_by_x=m, bytes_pi_len_prefix=b, digest_domain_separator=d) for m in (0, 1) for b in (0, 1)
                                 for d in (0, 1)], ids=lambda p: "x%d_pi%d_ds%d" % tuple(p.values()))
def test_every_compat_position_bit_exact(ctx, pos):
    """The switch table of include/vpbs_prover.h (vpbs_compat): under every position the device prover, the serialiser, the oracle and the
    product's host verifier agree word for word -- one capture of a real plonky2 proof then picks the position, nothing else has to move.
    The whole prover on the device (partial products, gate constraints, quotient) on a satisfiable circuit, so the full verifier runs."""
    import random
    import gates_oracle as go
    gate_spec = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon"]
    gs, ps = go.GateSet(gate_spec), api.GateSet(gate_spec)
    rnd = random.Random(11)
    log_c = 6
    cpis = [rnd.randrange(api.P) for _ in range(4)]
    constants, wires, sigma, _ = go.demo_circuit(rnd, gs, log_c, cpis)
    cs_values = np.concatenate([constants, sigma])
    nconst = constants.shape[0]
    ko = orc.compat(**pos)
    kp = ctx.set_compat(**pos)
    try:
        assert api.compat_dict(ctx.get_compat()) == api.compat_dict(kp)
        cs = ctx.commit_values(cs_values)
        digest = api.circuit_digest(cs.cap(), log_c, kp)
        assert digest.tolist() == orc.circuit_digest(cs.cap(), log_c, ko).tolist()
        si = ctx.make_step_inputs(log_c, wires, None, None, cs, digest, cpis, sigmas=sigma, n_routed=80, n_constants=nconst, gates=ps)
        got = ctx.prove_step(si)
        want = step_oracle.prove_step({"constants_sigmas": cs_values, "wires": wires, "quotient": None}, digest, cpis, log_c, sigmas=sigma,
                                      n_routed=80, n_constants=nconst, gates=gs, compat=ko)
        for key in ("caps", "challenges", "openings", "fri"):
            assert (got[key] == want[key]).all(), key
        blob = ctx.step_proof_to_bytes(si, nconst, got)
        assert blob == step_oracle.to_bytes(want, want["ncols"], nconst, cpis, log_c, compat=ko)
        back, back_pis = api.step_proof_from_bytes(blob, want["ncols"], log_c, nconst, compat=kp)
        assert back_pis.tolist() == cpis
        full = dict(check_permutation=True, n_constants=nconst, n_routed=80, gates=ps)
        assert api.verify_step(back, cs.cap(), want["ncols"], digest, back_pis, log_c, compat=kp, **full)
        assert step_oracle.verify_step(got, want["cs_cap"], want["ncols"], digest, cpis, log_c, compat=ko)
        assert not api.verify_step(back, cs.cap(), want["ncols"], digest, back_pis, log_c,
                                   compat=api.compat(**{**pos, "fri_mul_final_by_x": 1 - pos["fri_mul_final_by_x"]}), **full)
    finally:
        ctx.set_compat()
    with pytest.raises(api.VpbsError):     # "first found" nonces are not a position of this build
        ctx.set_compat(pow_smallest_nonce=0)


@pytest.mark.parametrize("log_n", [7, 12])
def test_step_proof_with_device_partial_products(ctx, log_n):
    """The step with a12 inside: Z / partial products computed on the GPU from the wires, the sigma columns of the
    constants_sigmas matrix and the transcript's betas/gammas -- bit-exact against the oracle doing the same."""
    inputs = synth.step_inputs(log_n)
    pis = synth.field_elements(0xBEEF, 40)
    n_constants, n_routed = 5, 80
    sig = np.ascontiguousarray(inputs["constants_sigmas"][n_constants:n_constants + n_routed])
    cs = ctx.commit_values(inputs["constants_sigmas"])
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, inputs["quotient"], cs, DIGEST, pis, sigmas=sig, n_routed=n_routed)
    got = ctx.prove_step(si)
    want = step_oracle.prove_step(inputs, DIGEST, pis, log_n, sigmas=sig, n_routed=n_routed)
    for key in ("caps", "challenges", "openings", "fri"):
        assert (got[key] == want[key]).all(), key
    assert step_oracle.verify_step(got, want["cs_cap"], want["ncols"], DIGEST, pis, log_n)


def test_step_proof_device_inputs_match_host_inputs(ctx):
    import torch
    log_n = 8
    inputs, pis, cs, si, want = _step(ctx, log_n)
    dev = {k: torch.from_numpy(inputs[k].view(np.int64)).cuda() for k in ("wires", "zs_partial_products", "quotient")}
    torch.cuda.synchronize()
    si2 = ctx.make_step_inputs(log_n, dev["wires"].data_ptr(), dev["zs_partial_products"].data_ptr(), dev["quotient"].data_ptr(),
                               cs, DIGEST, pis, on_device=True, shapes=(135, 20, 16))
    got = ctx.prove_step(si2)
    for key in ("caps", "openings", "fri"):
        assert (got[key] == want[key]).all()


def test_step_properties_2pow16(ctx):
    """BASELINE config 5 shape (N = 2048 ring: degree 2^16 assumed, LDE 2^19, same column counts): the proof verifies
    under the restated plonky2 verifier; FRI schedule [4,4,4] with a 16-coefficient final polynomial."""
    log_n = 16
    inputs, pis, cs, si, got = _step(ctx, log_n)
    assert step_oracle.verify_step(got, cs.cap(), [85, 135, 20, 16], DIGEST, pis, log_n)
    assert got["fri"].size == api.lib().vpbs_fri_proof_words(ctypes.byref(api.fri_params(log_n)), log_n, (ctypes.c_size_t * 4)(85, 135, 20, 16), 4)
    cs.free()


def test_step_properties_2pow17():
    """one size above the N = 1024 step circuit (the N = 2048 ring would land here if its step circuit is padded like the N = 1024 one:
    degree 2^17, LDE 2^20): the proof verifies under the restated plonky2 verifier and under the product's own."""
    log_n = 17
    c = vpbs_amd.Context(0, log_n_max=17)
    inputs, pis, cs, si, got = _step(c, log_n)
    assert step_oracle.verify_step(got, cs.cap(), [85, 135, 20, 16], DIGEST, pis, log_n)
    assert api.verify_step_fri_only(got, cs.cap(), [85, 135, 20, 16], DIGEST, pis, log_n)
    cs.free()
    c.close()


def test_full_size_step_properties(ctx):
    """BASELINE config 2 (N = 1024: degree 2^15, LDE 2^18, 135/20/16/85 columns).  The oracle prover would take
    minutes here, so parity is carried by size-independent properties: the proof verifies under the restated
    plonky2 verifier; Merkle paths open to the caps; openings are consistent with the committed LDE."""
    log_n = 15
    inputs, pis, cs, si, got = _step(ctx, log_n)
    ncols = [85, 135, 20, 16]
    assert step_oracle.verify_step(got, cs.cap(), ncols, DIGEST, pis, log_n)
    assert api.verify_step_fri_only(got, cs.cap(), ncols, DIGEST, pis, log_n)     # the product's own verifier agrees
    # determinism
    again = ctx.prove_step(si)
    assert (again["fri"] == got["fri"]).all() and (again["caps"] == got["caps"]).all()
    # tampered opening is rejected
    bad = dict(got); bad["openings"] = got["openings"].copy(); bad["openings"][100][0] ^= np.uint64(1)
    assert not step_oracle.verify_step(bad, cs.cap(), ncols, DIGEST, pis, log_n)
    # wires commitment: leaves open to the cap, and the LDE row equals the polynomial evaluated at that point
    wires = ctx.commit_values(inputs["wires"])
    assert (wires.cap() == got["caps"][0]).all()
    L = 1 << 18
    w18 = orc.lib().orc_gl_root_of_unity(18)
    for idx in (0, 12345, L - 1):
        leaf, sib = wires.open(idx)
        assert orc.merkle_verify(leaf, idx, got["caps"][0], 4, sib)
        nat = int(format(idx, "018b")[::-1], 2)
        x = orc.lib().orc_gl_mul(7, orc.lib().orc_gl_exp(w18, nat))
        ev = wires.eval_ext(np.array([x, 0], np.uint64))
        assert (ev[:, 0] == leaf).all() and (ev[:, 1] == 0).all()
    # iNTT really inverts: coefficients evaluated back on H reproduce the trace column (spot check through the oracle FFT)
    coeffs = wires.coeffs()
    for c in (0, 134):
        assert (orc.fft(coeffs[c]) == inputs["wires"][c]).all()
    wires.free()


def test_cxx_host_example_matches_python_path(ctx):
    """examples/prove_step.cpp drives the C ABI from plain C++ (no Python, no torch in that process): same seeded step,
    same proof bytes."""
    import subprocess
    import __graft_entry__ as entry
    exe = entry.build_example()
    log_n = 9
    r = subprocess.run([exe, str(log_n)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split())
    inputs = synth.step_inputs(log_n)
    pis = synth.field_elements(0xABCD, 77)
    sig = np.ascontiguousarray(inputs["constants_sigmas"][5:85])
    cs = ctx.commit_values(inputs["constants_sigmas"])
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, inputs["quotient"], cs, DIGEST, pis, sigmas=sig, n_routed=80)
    got = ctx.prove_step(si)
    blob = ctx.step_proof_to_bytes(si, 5, got)
    h = 0xcbf29ce484222325
    for b in blob:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    assert int(fields["proof_bytes"]) == len(blob)
    assert int(fields["pow"]) == int(got["fri"][-1])
    assert fields["fnv1a"] == "%016x" % h


# ---------- native TFHE step (witness-generation core) ----------
@pytest.mark.parametrize("log_N,K,ELL,LOGB,batch", [(3, 2, 8, 8, 3), (6, 2, 4, 5, 2), (10, 2, 4, 5, 2), (5, 3, 3, 7, 2), (11, 2, 4, 5, 1)])
def test_blind_rotate_step_matches_oracle(ctx, log_N, K, ELL, LOGB, batch):
    """paper parameters (N=1024, K=2, ELL=4, LOGB=5: reference src/main.rs:23-30) and the reference's test parameters
    (N=8, LOGB=8, ELL=8: mod.rs:224-227); first / middle / last step semantics of ivc_based_vpbs.rs:99-125"""
    import tfhe_oracle as T
    N = 1 << log_N
    ring = T.Ring(log_N)
    acc = rand_field(batch, K, N)
    masks = rand_field(batch)
    ggsw = rand_field(batch, K * ELL * K * N)
    hat = lambda b: [[[list(map(int, ggsw[b][((p * ELL + l) * K + r) * N:((p * ELL + l) * K + r + 1) * N])) for r in range(K)]
                      for l in range(ELL)] for p in range(K)]
    for first, last in ((False, False), (True, False), (False, True)):
        got = ctx.blind_rotate_step(acc, masks, ggsw, K, ELL, LOGB, first_step=first, last_step=last)
        for b in range(batch):
            want = T.step(ring, [list(map(int, acc[b][p])) for p in range(K)], int(masks[b]), hat(b), K, ELL, LOGB, first, last)
            assert [[int(v) for v in got[b][p]] for p in range(K)] == want, (first, last, b)
    # one GGSW shared by every instance
    got = ctx.blind_rotate_step(acc, masks, ggsw[0], K, ELL, LOGB)
    b = batch - 1
    want = T.step(ring, [list(map(int, acc[b][p])) for p in range(K)], int(masks[b]), hat(0), K, ELL, LOGB)
    assert [[int(v) for v in got[b][p]] for p in range(K)] == want


@pytest.mark.parametrize("log_N,ELL,LOGB", [(3, 8, 8), (6, 8, 8)])
def test_blind_rotate_step_decrypts_like_the_reference_test(ctx, log_N, ELL, LOGB):
    """the reference's own check (test_blind_rot_step, mod.rs:223-279): noise-free GLWE / GGSW encryptions, one CMUX
    step on the device, decrypt: bit = 0 leaves the message, bit = 1 rotates it by the mod-switched mask"""
    import tfhe_oracle as T
    K, N = 2, 1 << log_N
    ring = T.Ring(log_N)
    for bit in (0, 1):
        s = [[int(v) for v in rng.integers(0, 2, size=N)] for _ in range(K - 1)]
        m = list(range(N))
        ct = T.glwe_encrypt(ring, rng, s, m, K)
        gg = T.ggsw_encrypt_hat(ring, rng, s, [bit] + [0] * (N - 1), K, ELL, LOGB)
        ai = int(rand_field(1)[0])
        out = ctx.blind_rotate_step(np.array([ct], dtype=np.uint64), [ai], T.flatten_ggsw(gg), K, ELL, LOGB)
        m_out = T.glwe_decrypt(ring, s, [[int(v) for v in out[0][p]] for p in range(K)], K)
        assert m_out == (m if bit == 0 else T.rotate(m, T.mod_switch(ai, log_N)))


def test_pbs_accumulator_chain_end_to_end(ctx):
    """the native side of src/main.rs:40-65 with noise-free keys: LWE encryption of a bit, the n + 2 accumulators on the
    device (= the accumulator public inputs of the n + 2 step proofs), bit-exact against the oracle chain; decrypting the
    key-switched output under the partial key gives the bit back."""
    import tfhe_oracle as T
    log_N, K, ELL, LOGB, n, p = 6, 2, 8, 8, 5, 2
    ring = T.Ring(log_N)
    s_to, s_lwe, s_glwe, bsk, ksk = T.pbs_setup(ring, rng, n, K, ELL, LOGB, p)
    delta = T.get_delta(2 * p)
    testv = T.get_testv(ring, p, delta)
    acc_init = [[0] * ring.n for _ in range(K - 1)] + [testv]
    for m in (0, 1):
        ct = T.lwe_encrypt(rng, s_lwe, delta * m % P)
        got = ctx.pbs_accumulator_chain(np.array(acc_init, np.uint64), ct, np.stack([T.flatten_ggsw(g) for g in bsk]), T.flatten_ggsw(ksk), K, ELL, LOGB)
        want = T.pbs_chain(ring, acc_init, ct, bsk, ksk, K, ELL, LOGB)
        assert got.shape == (n + 2, K, ring.n)
        for step_i in range(n + 2):
            assert [[int(v) for v in got[step_i][q]] for q in range(K)] == want[step_i], step_i
        m_bar = T.glwe_decrypt(ring, s_to, [[int(v) for v in got[-1][q]] for q in range(K)], K)[0]
        assert round(m_bar / delta) % (2 * p) == m          # src/main.rs:59-65


@pytest.mark.parametrize("kind", ["zeros", "pminus1", "same_column", "edge_mix"])
def test_step_proof_degenerate_inputs(ctx, kind):
    """all-zero / all-(p-1) / identical / edge-valued trace columns through the whole device path (partial products, quotient,
    commitments, openings, FRI) against the oracle: canonical-form and carry corner cases of the field arithmetic."""
    log_n, n_constants, n_routed = 6, 5, 80
    n = 1 << log_n
    inputs = synth.step_inputs(log_n)
    if kind == "zeros":
        wires = np.zeros((135, n), np.uint64)
    elif kind == "pminus1":
        wires = np.full((135, n), P - 1, np.uint64)
    elif kind == "same_column":
        wires = np.tile(rand_field(1, n), (135, 1))
    else:
        edge = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63], np.uint64)
        wires = edge[rng.integers(0, edge.size, size=(135, n))]
    inputs["wires"] = np.ascontiguousarray(wires)
    inputs["quotient"] = None
    pis = synth.field_elements(0xED6E, 5)
    sig = np.ascontiguousarray(inputs["constants_sigmas"][n_constants:n_constants + n_routed])
    cs = ctx.commit_values(inputs["constants_sigmas"])
    si = ctx.make_step_inputs(log_n, inputs["wires"], None, None, cs, DIGEST, pis, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
    got = ctx.prove_step(si)
    want = step_oracle.prove_step(inputs, DIGEST, pis, log_n, sigmas=sig, n_routed=n_routed, n_constants=n_constants)
    for key in ("caps", "challenges", "openings", "fri"):
        assert (got[key] == want[key]).all(), key
    cs.free()


def test_batch_of_128_independent_proofs():
    """BASELINE config 3: 128 independent step proofs of one circuit at the N = 1024 shape (degree 2^16, 135/20/16/86 columns, all 14 gate
    types) on one GPU through a bounded pool of prover contexts.  Every proof verifies (transcript, PoW, Merkle paths, FRI), all 128 are
    distinct, a proof does not depend on which context of the pool produced it, and two sampled instances are bit-identical to the CPU
    oracle's proofs of the same wires."""
    import queue
    import threading
    import torch
    import gates_oracle as go
    import regression_cases as rc
    B = rc.BENCH
    log_n, nc, nr, n = B["log_n"], B["n_constants"], B["n_routed"], 1 << B["log_n"]
    proofs, pool = 128, 4
    gates = api.GateSet(rc.GATES)
    cs_values = synth.step_inputs(log_n, cols=B["cols"])["constants_sigmas"]
    d_cs = torch.from_numpy(cs_values.view(np.int64)).cuda()
    sig_ptr = d_cs.data_ptr() + 8 * nc * n
    gen = torch.Generator(device="cuda")
    wires, pis = [], []
    for i in range(proofs):
        gen.manual_seed(0x5EED0000 + 16 * i)
        wires.append(torch.randint(0, P >> 1, (135, n), dtype=torch.int64, device="cuda", generator=gen))
        pis.append(synth.field_elements(0xABCD + i, 64))
    torch.cuda.synchronize()
    ctxs = [vpbs_amd.Context(0, log_n_max=16) for _ in range(pool)]
    css = [c.commit_values(cs_values) for c in ctxs]
    todo, out, errs = queue.Queue(), [None] * proofs, []
    for i in range(proofs):
        todo.put(i)

    def prove(k, i):
        si = ctxs[k].make_step_inputs(log_n, wires[i].data_ptr(), None, None, css[k], B["digest"], pis[i], on_device=True, shapes=(135, 20, 16),
                                      sigmas=sig_ptr, n_routed=nr, n_constants=nc, gates=gates)
        return ctxs[k].prove_step(si)

    def work(k):
        try:
            while True:
                try:
                    i = todo.get_nowait()
                except queue.Empty:
                    return
                out[i] = prove(k, i)
        except Exception as e:
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(pool)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    cap = css[0].cap()
    assert len({p["caps"].tobytes() for p in out}) == proofs
    for i in range(0, proofs, 9):
        assert api.verify_step_fri_only(out[i], cap, [nc + nr, 135, 20, 16], B["digest"], pis[i], log_n), i
    # another context of the pool, alone on the device, reproduces the proof of a concurrently proven instance
    for i, k in ((5, 3), (77, 0)):
        again = prove(k, i)
        assert all((again[key] == out[i][key]).all() for key in ("caps", "openings", "fri")), i
    # sampled subset against the oracle
    sig = np.ascontiguousarray(cs_values[nc:nc + nr])
    for i in (17, 101):
        w = wires[i].cpu().numpy().view(np.uint64)
        want = step_oracle.prove_step({"constants_sigmas": cs_values, "wires": w, "quotient": None}, B["digest"], pis[i], log_n, sigmas=sig,
                                      n_routed=nr, n_constants=nc, gates=go.GateSet(rc.GATES))
        for key in ("caps", "challenges", "openings", "fri"):
            assert (np.asarray(out[i][key]).reshape(-1) == np.asarray(want[key]).reshape(-1)).all(), (i, key)
    for cs, c in zip(css, ctxs):
        cs.free()
        c.close()


# ---------- seeded key generation ----------
@pytest.mark.parametrize("log_ring,n_lwe,K,ELL,LOGB", [(3, 6, 2, 4, 5), (6, 40, 3, 3, 7), (10, 728, 2, 4, 5)])
def test_keygen_matches_oracle(ctx, log_ring, n_lwe, K, ELL, LOGB):
    """vpbs_keygen (GGSW encryptions generated on the device, directly in the NTT domain) against the oracle's literal restatement of
    Glwe::encrypt -> Glev / Ggsw::encrypt -> ntt_forward (crypto/glwe.rs:49-57, glev.rs:26-38, ggsw.rs:26-48, mod.rs:29-45) on the same
    seeded streams, with the paper's noise levels: bit-exact.  At N = 1024 / n = 728 a sample of the 728 GGSWs is compared."""
    import tfhe_oracle as T
    N, seed = 1 << log_ring, 0xC0FFEE + log_ring
    sg, sl = 4.99027217501041e-8, 1.17021618159313e-5       # main.rs:29-30
    keys = ctx.keygen(N, K, ELL, LOGB, n_lwe, seed, sg, sl)
    ring = T.Ring(log_ring)
    sample = list(range(n_lwe)) if n_lwe <= 40 else [0, 1, 357, 727]
    s_to, s_lwe, s_glwe, bsk, ksk = T.seeded_pbs_keys(ring, seed, n_lwe, K, ELL, LOGB, sg, sl, bsk_indices=sample)
    assert (keys["s_to"] == np.array(s_to, np.uint64)).all() and (keys["s_lwe"] == np.array(s_lwe, np.uint64)).all()
    assert (keys["s_glwe"] == np.array(s_glwe, np.uint64)).all()
    assert (keys["ksk"] == ksk).all()
    for i in sample:
        assert (keys["bsk"][i] == bsk[i]).all(), i
    # the noise is there and small: decrypting GLWE (p = K-1, l = ELL-1) of a GGSW gives s_i * B^(first + ELL - 1) up to ~6 sigma q
    i = sample[-1]
    g = keys["bsk"][i].reshape(K, ELL, K, N)
    ct = np.stack([np.array(ring.bw([int(v) for v in g[K - 1, ELL - 1, r]]), np.uint64) for r in range(K)])
    m = ctx.glwe_decrypt(keys["s_glwe"], ct)
    want = int(keys["s_lwe"][i]) * pow(2, LOGB * (T.num_limbs(LOGB) - 1), P) % P
    err = [min((int(v) - (want if j == 0 else 0)) % P, (-(int(v) - (want if j == 0 else 0))) % P) for j, v in enumerate(m)]
    assert 0 < max(err) < 7 * T.sigma_to_int(sg)


def test_seeded_pbs_decrypts_to_the_message(ctx):
    """main.rs:40-65 end to end at the paper's parameters (N = 1024, K = 2, ELL = 4, LOGB = 5, n = 728, p = 2, the paper's noise): seeded keys
    from vpbs_keygen, an LWE encryption of delta * m, the accumulator chain of the 730 steps on the device, Glwe::decrypt under the partial
    key: the bootstrapped message is m, for both messages."""
    N, K, ELL, LOGB, n, p = 1024, 2, 4, 5, 728, 2
    keys = ctx.keygen(N, K, ELL, LOGB, n, 0x5EED, 4.99027217501041e-8, 1.17021618159313e-5)
    testv, delta = api.testv(N, p)
    acc_init = np.concatenate([np.zeros((K - 1, N), np.uint64), testv.reshape(1, N)])
    for m in (0, 1):
        ct = api.lwe_encrypt(keys["params"], keys["s_lwe"], delta * m % P, nonce=m)
        accs = ctx.pbs_accumulator_chain(acc_init, ct, keys["bsk"], keys["ksk"], K, ELL, LOGB)
        m_bar = ctx.glwe_decrypt(keys["s_to"], accs[-1])
        assert round(int(m_bar[0]) / delta) % (2 * p) == m, (m, int(m_bar[0]) / delta)


def test_native_rccl_collectives_single_rank(ctx):
    """vpbs_comm_rccl_create (RCCL bound with dlopen): with one rank -- all a one-GPU box allows -- the three collectives of the sharded
    step run through ncclAllGather / ncclAllReduce on the context's stream and return their input; a step proof handed that communicator
    equals the plain one.  (The multi-rank behaviour of the same vpbs_comm contract is covered with the callback communicator over gloo.)"""
    import torch
    from vpbs_amd import sharding
    assert api.lib().vpbs_rccl_available() == 1
    stage = 4096
    comm = sharding.make_comm_rccl(ctx, stage_words=stage)
    assert comm.rank == 0 and comm.world == 1 and comm.stage_capacity_words == stage
    local = synth.field_elements(7, 64)
    full = np.zeros(64, np.uint64)
    assert comm.allgather(comm.user, api._ptr(local), 64, api._ptr(full)) == 0 and (full == local).all()
    rec = synth.field_elements(8, 40000)          # more than one staging block
    want = rec.copy()
    assert comm.allreduce_sum(comm.user, api._ptr(rec), rec.size) == 0 and (rec == want).all()
    src = torch.from_numpy(synth.field_elements(9, stage).view(np.int64))
    ctypes.cdll.LoadLibrary("libamdhip64.so")
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(comm.d_stage_local), ctypes.c_void_p(src.data_ptr()), stage * 8, 1) == 0
    assert comm.allgather_dev(comm.user, stage) == 0
    back = torch.zeros(stage, dtype=torch.int64)
    assert hip.hipMemcpy(ctypes.c_void_p(back.data_ptr()), ctypes.c_void_p(comm.d_stage_full), stage * 8, 2) == 0
    assert (back == src).all()
    inputs, pis, cs, si, want_proof = _step(ctx, 8)
    got = ctx.prove_step(si, comm)
    assert all((got[k] == want_proof[k]).all() for k in ("caps", "openings", "fri"))
    cs.free()
    sharding.free_comm_rccl(comm)


def test_context_options_choose_between_bit_identical_arrangements(ctx):
    """vpbs_ctx_set_option: the launch heuristics are per-context settings (the environment variables only set their defaults) and never
    change a result -- one launch per gate type vs the one-launch kernel, one or three gate streams, per-level Merkle launches vs the fused
    climb, the 16-lane Poseidon form everywhere / nowhere: the same proof, word for word"""
    import random
    import gates_oracle as go
    defaults = {name: ctx.get_option(name) for name in ctx.OPTIONS}
    assert defaults == {"gate_lanes": 1, "gates_fused": 1, "gate_items": 5, "wide_threshold": 1 << 14, "merkle_climb": 1, "gates_tile": 1}
    gate_spec = ["noop", "constant", "public_input", "arithmetic", "base_sum", "poseidon", "reducing", ("random_access", 4), "coset_interpolation"]
    gs, ps = go.GateSet(gate_spec), api.GateSet(gate_spec)
    rnd = random.Random(5)
    log_c = 7
    cpis = [rnd.randrange(api.P) for _ in range(4)]
    constants, wires, sigma, _ = go.demo_circuit(rnd, gs, log_c, cpis)
    nconst = constants.shape[0]
    cs = ctx.commit_values(np.concatenate([constants, sigma]))
    si = ctx.make_step_inputs(log_c, wires, None, None, cs, DIGEST, cpis, sigmas=sigma, n_routed=80, n_constants=nconst, gates=ps)
    want = ctx.prove_step(si)
    try:
        for over in ({"gates_tile": 0}, {"gates_fused": 0}, {"gate_lanes": 3}, {"gates_fused": 0, "gate_lanes": 3}, {"gates_tile": 0, "gate_lanes": 3}, {"gates_tile": 0, "gate_items": 2}, {"gates_tile": 0, "gate_items": 8}, {"merkle_climb": 0},
                     {"wide_threshold": 0}, {"wide_threshold": 1 << 30}):
            for name, v in over.items():
                ctx.set_option(name, v)
                assert ctx.get_option(name) == v
            got = ctx.prove_step(si)
            for key in ("caps", "challenges", "openings", "fri"):
                assert (got[key] == want[key]).all(), (over, key)
            for name in over:
                ctx.set_option(name, defaults[name])
        for name, bad in (("gate_lanes", 2), ("gate_items", 0), ("gate_items", 9)):
            with pytest.raises(api.VpbsError):
                ctx.set_option(name, bad)
    finally:
        for name, v in defaults.items():
            ctx.set_option(name, v)
    cs.free()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_communicator_world_gt1_on_one_gpu(world):
    """csrc/comm_rccl.hip with MORE than one rank before hardware day.  The library binds the collective library named by
    VPBS_RCCL_LIB -- here tests/fake_rccl.c, a test-only stand-in whose ranks are processes sharing this box's GPU and exchanging through
    shared memory, stream-ordered like the real calls -- and vpbs_prove_step_sharded runs through vpbs_comm_rccl_create at world 2 / 4 / 8:
    proofs bit-identical to the single-GPU proof, a failing rank => its own error there and VPBS_ERR_PEER everywhere else."""
    import subprocess
    import sys
    import __graft_entry__ as entry
    fake = entry.build_fake_rccl()
    script = os.path.join(ROOT, "tests", "fake_rccl_world_gpu.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                        "--master-port", str(29580 + world), script], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VPBS_RCCL_LIB=fake, VPBS_TEST_SCENARIO="parity", VPBS_COMM_TIMEOUT_S="60"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "FAKE_RCCL_WORLD_OK world=%d" % world in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_native_communicator_times_out_on_an_absent_peer(world):
    """A peer that never arrives: the survivors' collective polls its stream, gives up after VPBS_COMM_TIMEOUT_S = 3 s, aborts the
    communicator (ncclCommAbort) and the sharded step returns an error on every surviving rank; later calls on the dead communicator fail at
    once; vpbs_comm_rccl_destroy's bounded wait lets the stream drain before the staging buffers go back to the pool."""
    import subprocess
    import sys
    import __graft_entry__ as entry
    fake = entry.build_fake_rccl()
    script = os.path.join(ROOT, "tests", "fake_rccl_world_gpu.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                        "--master-port", str(29590 + world), script], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, VPBS_RCCL_LIB=fake, VPBS_TEST_SCENARIO="absent", VPBS_COMM_TIMEOUT_S="3"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "FAKE_RCCL_ABSENT_OK world=%d" % world in r.stdout


@pytest.mark.gpu
def test_native_rccl_world_gt1():
    """The library's RCCL communicator with MORE than one rank -- one process per GPU, ncclAllGather / ncclAllReduce over xGMI: a sharded step
    proof of the frozen regression circuits ends, on every rank, with the frozen single-GPU words and bytes.  Runs by itself wherever at least
    two devices are visible (an 8-GPU node: 2, 4 and 8 ranks); the one-GPU boxes of the test pool skip it -- there the same vpbs_comm contract
    is covered with one native rank (test_native_rccl_collectives_single_rank) and with 2 / 4 / 8 callback ranks over gloo."""
    import subprocess
    import sys
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs at least two GPUs (this box shows %d)" % n_dev)
    script = os.path.join(ROOT, "tests", "rccl_world_step_gpu.py")
    for world in [w for w in (2, 4, 8) if w <= n_dev]:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                            "--master-port", str(29560 + world), script], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert "RCCL_WORLD_OK world=%d" % world in r.stdout


@pytest.mark.gpu
def test_host_to_device_helpers():
    """vpbs_device_upload_bg (the context's upload stream, callable beside a running proof) and vpbs_device_upload_rows (a row range of every
    column of a column-major matrix): the device ends up with exactly the bytes asked for, nothing else touched"""
    import threading
    import torch
    c = vpbs_amd.Context(0, log_n_max=12)
    cols, n = 7, 1 << 10
    rng = np.random.default_rng(3)
    host = torch.from_numpy(rng.integers(0, 1 << 62, size=(cols, n), dtype=np.int64)).pin_memory()
    dev = torch.full((cols, n), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    c.upload_rows(dev.data_ptr(), host.data_ptr(), cols, n, 100, 357)
    got = dev.cpu()
    assert (got[:, 100:357] == host[:, 100:357]).all() and (got[:, :100] == -1).all() and (got[:, 357:] == -1).all()
    c.upload_rows(dev.data_ptr(), host.data_ptr(), cols, n, 5, 5)                      # empty range: nothing moves
    assert (dev.cpu() == got).all()
    # vpbs_device_scatter: packed values (the late witness phase's) put in place through uint32 positions resident on the device
    for count in (1, 2, 4097):
        pos = rng.choice(cols * n, count, replace=False).astype(np.uint32)
        vals = torch.from_numpy(rng.integers(0, 1 << 62, size=count, dtype=np.int64)).pin_memory()
        packed = np.zeros((count + 1) // 2, np.uint64)
        packed.view(np.uint32)[:count] = pos
        d_pos = torch.from_numpy(packed.view(np.int64)).cuda()
        d_stage = torch.empty(count, dtype=torch.int64, device="cuda")
        before = dev.cpu().reshape(-1).clone()
        torch.cuda.synchronize()
        c.scatter(dev.data_ptr(), d_pos.data_ptr(), vals.data_ptr(), count, d_stage.data_ptr())
        after = dev.cpu().reshape(-1)
        before[torch.from_numpy(pos.astype(np.int64))] = vals
        assert (after == before).all()
    with pytest.raises(api.VpbsError):
        c.upload_rows(dev.data_ptr(), host.data_ptr(), cols, n, 10, n + 1)
    # the background upload from a second host thread while the context proves
    inputs = synth.step_inputs(10)
    cs = c.commit_values(inputs["constants_sigmas"])
    si = c.make_step_inputs(10, inputs["wires"], inputs["zs_partial_products"], inputs["quotient"], cs, np.array([1, 2, 3, 4], np.uint64),
                            synth.field_elements(9, 8))
    want = c.prove_step(si)
    errors = []

    def uploader():
        try:
            for _ in range(20):
                c.upload_bg(dev.data_ptr(), host.data_ptr(), cols * n)
        except Exception as e:                                                        # noqa: BLE001
            errors.append(e)
    t = threading.Thread(target=uploader)
    t.start()
    for _ in range(5):
        p = c.prove_step(si)
        assert all((p[k] == want[k]).all() for k in ("caps", "openings", "fri"))
    t.join()
    assert not errors and (dev.cpu() == host).all()
    cs.free()
    c.close()
