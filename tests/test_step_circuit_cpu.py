"""CPU tests (-m "not gpu"): the reference's step circuit without the recursive verifier (build_step_circuit,
/root/reference/src/vtfhe/ivc_based_vpbs.rs:80-155) described by the test-side builder (circuitgen/step_circuit.py), witness generated and
checked by the PRODUCT's host code (vpbs_generate_witness / vpbs_check_witness), public inputs compared with the native restatement
of the same step (tests/tfhe_oracle.py) and the native hash chain.  No device compute."""
import numpy as np
import pytest

import oracle as orc
import step_circuit as sc
import tfhe_oracle as tf
from vpbs_amd import api

P = orc.P


def _ring(log_n):
    roots, inv, ninv = orc.negacyclic_params(log_n)
    return roots, inv, ninv


@pytest.fixture(scope="module")
def small():
    N, K, ELL, LOGB, n_lwe = 8, 2, 4, 5, 6
    return sc.StepCircuit(api, N, K, ELL, LOGB, n_lwe, _ring(3))


def _inputs(seed, N, K, ELL):
    rng = np.random.default_rng(seed)
    f = lambda *shape: rng.integers(0, P, size=shape, dtype=np.uint64)
    return dict(acc_init=f(K, N), acc_in=f(K, N), ggsw=f(K, ELL, K, N), mask=int(f(1)[0]), bsk_hash_in=f(4), lwe_hash_in=f(4))


def _expected(circ, x, counter):
    N, K, ELL, LOGB, n_lwe = circ.shape
    ring = tf.Ring(N.bit_length() - 1)
    ggsw_hat = [[[list(map(int, x["ggsw"][p][l][r])) for r in range(K)] for l in range(ELL)] for p in range(K)]
    acc_out = tf.step(ring, [list(map(int, p)) for p in x["acc_in"]], x["mask"], ggsw_hat, K, ELL, LOGB, first_step=counter == 1,
                      last_step=counter == n_lwe + 2)
    bsk_hash = orc.hash_no_pad(np.concatenate([x["bsk_hash_in"], x["ggsw"].reshape(-1)]))
    lwe_hash = orc.hash_no_pad(np.concatenate([x["lwe_hash_in"], np.array([x["mask"]], np.uint64)]))
    return ([int(v) for v in x["acc_init"].reshape(-1)] + [counter] + [int(v) % P for p in acc_out for v in p] + [int(v) for v in bsk_hash] +
            [int(v) for v in lwe_hash])


@pytest.mark.parametrize("which", ["first", "middle", "last"])
def test_step_circuit_public_inputs_match_the_native_step(small, which):
    N, K, ELL, LOGB, n_lwe = small.shape
    counter = {"first": 1, "middle": 3, "last": n_lwe + 2}[which]
    x = _inputs(100 + counter, N, K, ELL)
    wires = small.witness(x["acc_init"], x["ggsw"].reshape(-1), x["acc_in"], counter, x["mask"], x["bsk_hash_in"], x["lwe_hash_in"])
    pis = small.public_inputs(wires)
    assert pis == _expected(small, x, counter)
    ok, msg = small.built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
    assert ok, msg
    # the public-input hash row feeds the PublicInputGate
    assert small.built.row_kinds[-1] == "public_input"


def test_step_circuit_rejects_a_wrong_witness(small):
    N, K, ELL, LOGB, n_lwe = small.shape
    x = _inputs(7, N, K, ELL)
    wires = small.witness(x["acc_init"], x["ggsw"].reshape(-1), x["acc_in"], 2, x["mask"], x["bsk_hash_in"], x["lwe_hash_in"])
    pis = small.public_inputs(wires)
    h = api.hash_no_pad(np.array(pis, np.uint64))
    col, row = small.built.pos(small.acc_out[0][0])
    bad = wires.copy()
    bad[col, row] ^= np.uint64(1)
    ok, msg = small.built.circuit.check_witness(bad, h)
    assert not ok and msg
    # a preset that contradicts the circuit: acc_out is determined by the inputs
    a = small.built.presets({small.acc_out[0][0]: (pis[K * N + 1] + 1) % P})
    presets = small.built.presets(dict(zip([t for p in small.acc_in for t in p], x["acc_in"].reshape(-1))))
    with pytest.raises(api.VpbsError):
        small.built.circuit.generate_witness({**presets, **a})


def test_step_circuit_other_parameters():
    """K = 3 polynomials per GLWE (two GLEVs summed before the subtraction, ggsw_ct.rs:106-111), base 2^8, two levels"""
    N, K, ELL, LOGB, n_lwe = 8, 3, 2, 8, 4
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n_lwe, _ring(3))
    for counter in (1, 2, n_lwe + 2):
        x = _inputs(50 + counter, N, K, ELL)
        wires = circ.witness(x["acc_init"], x["ggsw"].reshape(-1), x["acc_in"], counter, x["mask"], x["bsk_hash_in"], x["lwe_hash_in"])
        pis = circ.public_inputs(wires)
        assert pis == _expected(circ, x, counter)
        ok, msg = circ.built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
        assert ok, msg


def _assignment(circ, x, counter):
    a = {}
    for targets, values in ((circ.acc_init, x["acc_init"]), (circ.acc_in, x["acc_in"])):
        for tp, vp in zip(targets, values):
            a.update(zip(tp, vp))
    a.update(zip(circ.ggsw_flat, x["ggsw"].reshape(-1)))
    a[circ.counter], a[circ.mask] = counter, x["mask"]
    a.update(zip(circ.bsk_hash_in, x["bsk_hash_in"]))
    a.update(zip(circ.lwe_hash_in, x["lwe_hash_in"]))
    return circ.built.presets(a)


def test_witness_plan_replays_the_generators(small):
    """vpbs_witness_plan: created once for the step circuit's PartialWitness targets, run for the first / CMUX / last step -- the
    same wires as vpbs_generate_witness, for any thread count, also when several host threads share the plan"""
    import threading
    N, K, ELL, LOGB, n_lwe = small.shape
    cases = [(c, _inputs(900 + c, N, K, ELL)) for c in (1, 2, n_lwe + 2, 4)]
    presets = [_assignment(small, x, c) for c, x in cases]
    positions = list(presets[0])
    assert all(list(p) == positions for p in presets)
    plan = small.built.circuit.witness_plan(positions)
    st = plan.stats()
    assert st["generators"] > 1000 and 0 < st["levels"] < st["generators"] and st["slots"] < st["positions"]
    expected = [small.built.circuit.generate_witness(p) for p in presets]
    for p, want, (c, x) in zip(presets, expected, cases):
        for threads in (0, 1, 3):
            got = plan.run(list(p.values()), threads=threads)
            assert (got == want).all()
        assert small.public_inputs(got) == _expected(small, x, c)
    results = [None] * len(presets)

    def work(i):
        for _ in range(5):
            results[i] = plan.run(list(presets[i].values()), threads=2)

    pool = [threading.Thread(target=work, args=(i,)) for i in range(len(presets))]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    assert all((r == w).all() for r, w in zip(results, expected))
    plan.free()


def test_witness_plan_errors(small):
    N, K, ELL, LOGB, n_lwe = small.shape
    x = _inputs(31, N, K, ELL)
    presets = _assignment(small, x, 2)
    # a target the generators need is not part of the PartialWitness: found when the plan is made
    missing = dict(presets)
    del missing[small.built.pos(small.mask)]
    with pytest.raises(api.VpbsError, match="weren't run"):
        small.built.circuit.witness_plan(list(missing))
    # a value that contradicts the circuit: found by the run
    wires = small.built.circuit.generate_witness(presets)
    out_pos = small.built.pos(small.acc_out[1][3])
    plan = small.built.circuit.witness_plan(list(presets) + [out_pos])
    good = list(presets.values()) + [int(wires[out_pos])]
    assert (plan.run(good) == wires).all()
    with pytest.raises(api.VpbsError, match="set twice"):
        plan.run(good[:-1] + [(good[-1] + 1) % P])
    with pytest.raises(api.VpbsError):
        small.built.circuit.witness_plan([(0, small.built.n)])          # position out of range


def test_pbs_chain_of_step_witnesses():
    """verified_pbs (ivc_based_vpbs.rs:276-371) at witness level, N = 8, n = 3, noise-free keys: the n + 2 step witnesses, each fed
    with the previous step's public inputs (accumulator, counter + 1, hash chains); accumulators equal the native chain, hashes equal
    verify_hash_output's chain, the key-switched output decrypts to the message.  (The proofs of the same chain: test_gpu_step_circuit.)"""
    rng = np.random.default_rng(77)
    log_N, K, ELL, LOGB, n, p = 3, 2, 8, 8, 3, 2
    ring = tf.Ring(log_N)
    N = ring.n
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n, _ring(log_N))
    s_to, s_lwe, s_glwe, bsk, ksk = tf.pbs_setup(ring, rng, n, K, ELL, LOGB, p)
    delta = tf.get_delta(2 * p)
    acc_init = np.array([[0] * N for _ in range(K - 1)] + [tf.get_testv(ring, p, delta)], np.uint64)
    bsk_flat, ksk_flat = np.stack([tf.flatten_ggsw(g) for g in bsk]), tf.flatten_ggsw(ksk)
    for m in (0, 1):
        ct = tf.lwe_encrypt(rng, s_lwe, delta * m % P)
        accs = tf.pbs_chain(ring, [list(map(int, q)) for q in acc_init], ct, bsk, ksk, K, ELL, LOGB)
        ggsws = [np.zeros(K * ELL * K * N, np.uint64)] + list(bsk_flat) + [ksk_flat]    # Ggsw::dummy_ct() in step 0
        masks = [int(ct[n])] + [int(v) for v in ct[:n]] + [0]
        acc_in, bsk_hash, lwe_hash = acc_init, np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        for step in range(n + 2):
            wires = circ.witness(acc_init, ggsws[step], acc_in, step + 1, masks[step], bsk_hash, lwe_hash)
            pis = circ.public_inputs(wires)
            acc_out = np.array(pis[K * N + 1:2 * K * N + 1], np.uint64).reshape(K, N)
            assert [[int(v) for v in q] for q in acc_out] == accs[step], step
            bsk_hash, lwe_hash = np.array(pis[-8:-4], np.uint64), np.array(pis[-4:], np.uint64)
            assert (bsk_hash == api.hash_chain(np.stack(ggsws[:step + 1]))[0]).all()
            assert (lwe_hash == api.hash_chain(np.array(masks[:step + 1], np.uint64).reshape(-1, 1))[0]).all()
            ok, msg = circ.built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
            assert ok, msg
            acc_in = acc_out
        m_bar = tf.glwe_decrypt(ring, s_to, [[int(v) for v in acc_in[q]] for q in range(K)], K)[0]
        assert round(m_bar / delta) % (2 * p) == m


def test_cxx_host_generates_the_exported_witness(tmp_path):
    """examples/prove_step_circuit.cpp up to the device boundary, on a box without a GPU: the C++ host loads the exported circuit,
    makes the plan, generates the witness, reproduces the exported public inputs and checks every constraint -- then fails loudly
    (exit code 2) because there is no device to prove on."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by test_gpu_step_circuit on a GPU box")
    import __graft_entry__ as entry
    sys.path.insert(0, entry.ROOT + "/tools")
    import export_step_circuit as ex
    path = str(tmp_path / "step.bin")
    ex.export(path, N=8)
    exe = entry.build_example("prove_step_circuit")
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "context creation failed" in r.stderr, r.stdout + r.stderr
    with open(path, "r+b") as f:                      # a sample value changed: the witness no longer reproduces the exported public inputs
        f.seek(-8 * (2 * 41 + 3), 2)
        f.write((12345).to_bytes(8, "little"))
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "public inputs differ" in r.stderr, r.stdout + r.stderr


def test_split_plan_levels_threads_and_concurrent_runs():
    """vpbs_witness_plan_split on the step circuit with the GGSW (the key material) arriving late: the early phase alone, on 2 and on 5
    threads, gives the matrix of the one-shot plan once the late phase has run; the late phase only writes the rows
    vpbs_witness_plan_late_rows reports; a recycled matrix gives the same result; four host threads running the same plan at once (one gets the plan's thread pool, the others go
    alone) each get their own correct matrix."""
    import threading
    N, K, ELL, LOGB, n_lwe = 64, 2, 4, 5, 6
    circ = sc.StepCircuit(api, N, K, ELL, LOGB, n_lwe, _ring(6))
    b = circ.built

    def presets(seed, counter):
        x = _inputs(seed, N, K, ELL)
        a = {}
        for targets, values in ((circ.acc_init, x["acc_init"]), (circ.acc_in, x["acc_in"])):
            for tp, vp in zip(targets, values):
                a.update(zip(tp, vp))
        a.update(zip(circ.ggsw_flat, x["ggsw"].reshape(-1)))
        a[circ.counter], a[circ.mask] = counter, x["mask"]
        a.update(zip(circ.bsk_hash_in, x["bsk_hash_in"]))
        a.update(zip(circ.lwe_hash_in, x["lwe_hash_in"]))
        return b.presets(a)

    first = presets(1, 3)
    positions = list(first)
    late_pos = {b.pos(t) for t in circ.ggsw_flat}
    late = np.array([1 if q in late_pos else 0 for q in positions], np.uint8)
    plan, whole = b.circuit.witness_plan(positions), b.circuit.witness_plan(positions)
    with pytest.raises(api.VpbsError):
        plan.late_rows()                                   # not split yet
    plan.split(late)
    lo, hi = plan.late_rows()
    assert 0 <= lo < hi <= b.circuit.n
    vals = lambda pre: np.array([pre[q] for q in positions], np.uint64)
    for threads in (1, 2, 5):
        v = vals(presets(10 + threads, 2 + threads % 3))
        want = whole.run(v)
        out = np.full_like(want, 0xABCD)
        early_only = v.copy()
        early_only[late == 1] = 0xDEAD                      # the late entries are not read by the early phase
        st = plan.run_early(early_only, out, threads=threads)
        before = out.copy()
        plan.run_late(st, v, out)
        assert (out == want).all()
        changed = np.nonzero((before != out).any(axis=0))[0]
        assert changed.size and lo <= changed.min() and changed.max() < hi
    # the late phase without the matrix (vpbs_witness_plan_run_late_packed): the same values, packed in the order of late_positions()
    pos = plan.late_positions()
    assert pos.size and np.unique(pos).size == pos.size and lo <= (pos % b.circuit.n).min() and (pos % b.circuit.n).max() < hi
    v = vals(presets(31, 2))
    want = whole.run(v)
    out = np.full_like(want, 0xABCD)
    st = plan.run_early(v, out)
    early_matrix = out.copy()
    packed = plan.run_late_packed(st, v)
    assert (out == early_matrix).all()                                      # the host matrix is left alone
    assert (packed == want.reshape(-1)[pos]).all()
    out.reshape(-1)[pos] = packed
    assert (out == want).all()                                              # early matrix + scattered late values = the witness
    # the late phase seeded from outside (an early phase that ran elsewhere, e.g. on the device): the early-known values it touches, read
    # from the early matrix at late_input_positions(), give the same late values
    lin = plan.late_input_positions()
    assert lin.size and (lin != 0xFFFFFFFF).all() and np.unique(lin).size == lin.size
    st2 = plan.state_from_late_inputs(early_matrix.reshape(-1)[lin])
    assert (plan.run_late_packed(st2, v) == packed).all()
    bad_in = early_matrix.reshape(-1)[lin].copy()
    bad_in[0] = np.uint64(P)                                                # not canonical
    with pytest.raises(api.VpbsError):
        plan.state_from_late_inputs(bad_in)
    # a recycled matrix (it holds the previous run's result): only the positions that carry values are rewritten
    v = vals(presets(77, 4))
    plan.run_late(plan.run_early(v, out, recycled=True), v, out)
    assert (out == whole.run(v)).all()
    results, errors = {}, []

    def worker(i):
        try:
            v = vals(presets(50 + i, 1 + i % 4))
            out = np.empty((b.circuit.n_wires, b.circuit.n), np.uint64)
            for _ in range(3):
                plan.run_late(plan.run_early(v, out), v, out)
            results[i] = (out, v)
        except Exception as e:                              # noqa: BLE001
            errors.append(e)

    pool = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    assert not errors, errors
    for out, v in results.values():
        assert (out == whole.run(v)).all()
    # a late word changed after the early phase: the late phase follows the values it is given
    v = vals(first)
    out = np.empty((b.circuit.n_wires, b.circuit.n), np.uint64)
    st = plan.run_early(v, out)
    v2 = v.copy()
    v2[np.nonzero(late)[0][5]] ^= np.uint64(1)
    plan.run_late(st, v2, out)
    assert (out == whole.run(v2)).all()
    plan.free()
    whole.free()


def test_split_le_asserts_its_unused_limbs_zero():
    """gadgets/split_join.rs split_le: the limbs of the last BaseSumGate beyond num_bits are asserted zero -- split_le(x, 20) IS the range
    check x < 2^20 (the in-circuit proof-of-work check and the decomposition's digits rest on it), not x < 2^63"""
    cb = sc.Builder()
    x, y = cb.virtual(), cb.virtual()
    bits = cb.split_le(x, 20)
    wide = cb.split_le(y, 70)                                # two rows: 63 limbs + 7, the other 56 limbs of the second row are zero
    assert len(bits) == 20 and len(wide) == 70
    cb.register_public_inputs(bits[:4] + wide[60:])
    built = cb.build(api)

    def run(xv, yv):
        w = built.circuit.generate_witness(built.presets({x: xv, y: yv}))
        ok, msg = built.circuit.check_witness(w, api.hash_no_pad(np.array(built.values(w, built.public_inputs), np.uint64)))
        assert ok, msg
        return built.values(w, bits), built.values(w, wide)
    b, wv = run((1 << 20) - 1, P - 1)
    assert b == [1] * 20 and sum(v << i for i, v in enumerate(wv)) == P - 1
    assert run(0b1011, 5)[0][:4] == [1, 1, 0, 1]
    for too_big in (1 << 20, (1 << 62) + 3, 1 << 47):
        with pytest.raises(api.VpbsError, match="set twice"):
            run(too_big, 0)


def test_a_late_check_against_an_early_value_keeps_the_early_phase_early():
    """vpbs_witness_plan_split: a late generator that writes into a copy class the early phase already knows (the limbs a late range check
    connects to the constant zero, a late value connected to an early one) is a comparer -- it must not drag every reader of that class
    into the late phase.  The late rows are the range check's own; a late value that violates the check is still caught."""
    cb = sc.Builder()
    early = cb.virtuals(40)
    late_v = cb.virtual()
    zero = cb.zero()
    acc = zero
    for t in early:                                           # early logic that reads the constant zero all over
        acc = cb.mul_add(acc, t, cb.add(t, zero))
    cb.split_le(late_v, 10)                                   # late: its 53 unused limbs are connected to zero
    cb.connect(cb.mul(late_v, late_v), cb.mul(early[0], early[0]))   # ... and its square is compared with an early value
    cb.register_public_inputs([acc])
    built = cb.build(api)
    positions = [built.pos(t) for t in early + [late_v]]
    plan, whole = built.circuit.witness_plan(positions), built.circuit.witness_plan(positions)
    late = np.zeros(len(positions), np.uint8)
    late[-1] = 1
    plan.split(late)
    lo, hi = plan.late_rows()
    vals = np.array([7 + 3 * i for i in range(40)] + [7], np.uint64)
    want = whole.run(vals)
    out = np.full_like(want, 0xABCD)
    early_only = vals.copy()
    early_only[-1] = 0xDEAD
    st = plan.run_early(early_only, out)
    acc_pos = built.pos(acc)
    assert out[acc_pos[0], acc_pos[1]] == want[acc_pos[0], acc_pos[1]]      # the early chain is complete before the late value exists
    before = out.copy()
    plan.run_late(st, vals, out)
    assert (out == want).all()
    changed = np.nonzero((before != out).any(axis=0))[0]
    assert 0 < changed.size <= 4 and lo <= changed.min() and changed.max() < hi  # the range-check row, the preset, the square: nothing else
    for bad in (8, (1 << 10) + 7):                            # another square; beyond the range
        v = vals.copy()
        v[-1] = bad
        with pytest.raises(api.VpbsError):
            plan.run_late(plan.run_early(early_only, out), v, out)
    plan.free()
    whole.free()


def test_cxx_ivc_host_builds_and_fails_loudly_without_a_device(tmp_path):
    """examples/prove_ivc.cpp (the IVC chain from a plain C++ host) compiles against include/vpbs_prover.h with g++ alone, rejects a file
    that is not a circuit, and without a GPU stops at context creation instead of computing anything on the CPU."""
    import subprocess
    import torch
    import __graft_entry__ as entry
    exe = entry.build_example("prove_ivc")
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"\0" * 256)
    r = subprocess.run([exe, str(junk), str(junk)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "not a circuit file" in r.stderr
    if torch.cuda.is_available():
        return
    head = tmp_path / "head.bin"
    head.write_bytes(b"".join(int(v).to_bytes(8, "little") for v in [0x5354455043495243, 13] + [0] * 14))
    r = subprocess.run([exe, str(head), str(head)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "context creation failed" in r.stderr, r.stdout + r.stderr


@pytest.mark.parametrize("seed", range(6))
def test_random_builder_programs(seed):
    """Random straight-line programs over the builder's gadgets (arithmetic in every form, select, is_equal, split_le / le_sum,
    in-circuit hashing) evaluated side by side in Python integers: the product's witness (one-shot and compiled plan) must give every
    target the value the program computes, satisfy every gate and copy constraint, and agree between the two paths."""
    import random
    r = random.Random(1000 + seed)
    cb = sc.Builder()
    inputs = cb.virtuals(6)
    val = {t: r.randrange(P) for t in inputs}
    val[inputs[4]] = val[inputs[3]]                       # an equal pair for is_equal
    val[inputs[5]] = r.randrange(1 << 40)                 # a small value for narrow splits
    pool = list(inputs)
    const = lambda c: (cb.constant(c), c % P)

    def pick():
        t = r.choice(pool)
        return t, val[t]

    for _ in range(r.randrange(40, 120)):
        kind = r.choice(["add", "sub", "mul", "neg", "mul_add", "mul_sub", "mul_const_add", "select", "is_equal", "split", "hash", "const"])
        (x, vx), (y, vy), (z, vz) = pick(), pick(), pick()
        if kind == "add":
            t, v = cb.add(x, y), (vx + vy) % P
        elif kind == "sub":
            t, v = cb.sub(x, y), (vx - vy) % P
        elif kind == "mul":
            t, v = cb.mul(x, y), vx * vy % P
        elif kind == "neg":
            t, v = cb.neg(x), (-vx) % P
        elif kind == "mul_add":
            t, v = cb.mul_add(x, y, z), (vx * vy + vz) % P
        elif kind == "mul_sub":
            t, v = cb.mul_sub(x, y, z), (vx * vy - vz) % P
        elif kind == "mul_const_add":
            c = r.randrange(P)
            t, v = cb.mul_const_add(c, x, y), (c * vx + vy) % P
        elif kind == "const":
            t, v = const(r.choice([0, 1, 2, P - 1, r.randrange(P)]))
        elif kind == "is_equal":
            t = cb.is_equal(x, y)
            v = 1 if vx == vy else 0
        elif kind == "select":
            other, vo = r.choice([(x, vx), (y, vy)])
            b, vb = cb.is_equal(x, other), 1 if vx == vo else 0
            val[b] = vb
            t, v = cb.select(b, y, z), (vy if vb else vz)
        elif kind == "split":
            nbits = r.choice([40, 64, 65, 70])
            src, vs = (inputs[5], val[inputs[5]]) if nbits == 40 else (x, vx)
            bits = cb.split_le(src, nbits)
            for i, bt in enumerate(bits):
                val[bt] = (vs >> i) & 1
            lo = r.randrange(0, 30)
            t, v = cb.le_sum(bits[lo:lo + 8]), (vs >> lo) & 0xFF
        else:
            items = [pick() for _ in range(r.randrange(1, 20))]
            out = cb.hash_no_pad([a for a, _ in items])
            hv = orc.hash_no_pad(np.array([b for _, b in items], np.uint64))
            for o, h in zip(out, hv):
                val[o] = int(h)
            t, v = out[0], int(hv[0])
        val[t] = v
        pool.append(t)
    cb.register_public_inputs(r.sample(pool, 5))
    built = cb.build(api)
    presets = built.presets({t: val[t] for t in inputs})
    wires = built.circuit.generate_witness(presets)
    pis = built.values(wires, built.public_inputs)
    ok, msg = built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
    assert ok, msg
    checked = 0
    for t in pool:
        try:
            col, row = built.pos(t)
        except KeyError:
            continue                                       # a target no wire carries (an input the program never used)
        assert int(wires[col, row]) == val[t], t
        checked += 1
    assert checked > len(pool) // 2
    plan = built.circuit.witness_plan(list(presets))
    assert (plan.run(list(presets.values())) == wires).all()


@pytest.mark.parametrize("N", [8, 64, 1024])
def test_in_circuit_ntt_reproduces_the_references_vectors(N):
    """The reference's own tests of its NTT gadget (ntt/mod.rs test_ntt_forward / test_ntt_backward: TESTG <-> TESTGHAT of
    params_{N}.rs, held in tests/golden/ntt_params_{N}.json) on the builder's restatement of that gadget, with the witness generated
    by the product: forward(TESTG) = TESTGHAT and backward(TESTGHAT) = TESTG as public inputs, every constraint satisfied."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ntt_params_%d.json" % N)))
    if len(gold["TESTG"]) != N:
        pytest.skip("the fixture holds hashes only for this N")
    if "ROOTS" in gold:
        ring = (np.array(gold["ROOTS"], np.uint64), np.array(gold["INVROOTS"], np.uint64), int(gold["NINV"]))
    else:                                           # large N: the fixture pins the oracle's tables by their SHA-256 (test_oracle_cpu)
        ring = _ring(N.bit_length() - 1)
        assert ring[2] == int(gold["NINV"])
    cb = sc.Builder()
    x, y = cb.virtuals(N), cb.virtuals(N)
    cb.register_public_inputs(sc.ntt_forward(cb, x, ring[0]))
    cb.register_public_inputs(sc.ntt_backward(cb, y, ring[1], ring[2]))
    built = cb.build(api)
    a = dict(zip(x, gold["TESTG"]))
    a.update(zip(y, gold["TESTGHAT"]))
    wires = built.circuit.generate_witness(built.presets(a))
    pis = built.values(wires, built.public_inputs)
    assert pis[:N] == [int(v) for v in gold["TESTGHAT"]] and pis[N:] == [int(v) for v in gold["TESTG"]]
    ok, msg = built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
    assert ok, msg


def test_in_circuit_decompose_like_the_references_test():
    """glwe_poly.rs test_decompose / test_vec_decompose: the LOGB = 8 digits of x recombine to x (sum of out[i] * B^i); inputs as in the
    reference's test (2^63 + a random u32) plus edge values"""
    import random
    r = random.Random(5)
    logb, limbs = 8, 8
    xs = [(1 << 63) + r.randrange(1 << 32) for _ in range(3)] + [0, 1, P - 1, (1 << 63) - 1, 1 << 63, r.randrange(P)]
    cb = sc.Builder()
    targets = cb.virtuals(len(xs))
    digits = [sc.decompose(cb, t, limbs, logb) for t in targets]
    cb.register_public_inputs([d for ds in digits for d in ds])
    built = cb.build(api)
    wires = built.circuit.generate_witness(built.presets(dict(zip(targets, xs))))
    pis = built.values(wires, built.public_inputs)
    ok, msg = built.circuit.check_witness(wires, api.hash_no_pad(np.array(pis, np.uint64)))
    assert ok, msg
    for k, x in enumerate(xs):
        out = pis[limbs * k:limbs * (k + 1)]
        assert sum(o * pow(1 << logb, i, P) for i, o in enumerate(out)) % P == x % P
        assert out == tf.decompose(x, logb)
        assert all(min(o, P - o) <= (1 << (logb - 1)) for o in out)          # centred digits


def test_in_circuit_rotation_like_the_references_test():
    """mod.rs test_poly_rotate + check_rotation: rotate_poly by a mask element = multiplication by X^shift with shift the rounded top
    log2(2N) bits of the mask, negacyclic (out[i + shift] = in[i], wrapped coefficients negated, shift > N: the negated polynomial)"""
    import random
    r = random.Random(6)
    N = 16
    for mask in [r.randrange(P) for _ in range(4)] + [0, P - 1, 1 << 58, (1 << 59) - 1]:
        poly = [r.randrange(P) for _ in range(N)]
        cb = sc.Builder()
        pt, mt = cb.virtuals(N), cb.virtual()
        cb.register_public_inputs(sc.rotate_poly(cb, pt, mt))
        built = cb.build(api)
        a = dict(zip(pt, poly))
        a[mt] = mask
        wires = built.circuit.generate_witness(built.presets(a))
        out = built.values(wires, built.public_inputs)
        assert out == tf.rotate(poly, tf.mod_switch(mask, 4))
        shift = mask >> (64 - 4 - 2)                                          # check_rotation (mod.rs:154-183)
        carry = shift % 2
        shift = (shift >> 1) + carry
        src = poly
        if shift == 2 * N:
            continue      # X^(2N) = 1: the reference's check_rotation would negate here (its random test hits this with probability 1/64)
        if shift > N:
            shift, src = shift % N, [(P - c) % P for c in poly]
        for i in range(N - shift):
            assert src[i] == out[i + shift]
        for i in range(shift):
            assert src[N - shift + i] == (P - out[i]) % P
