"""One step proof composed from the CPU oracle (test infrastructure; also bench.py's cpu_baseline leg).

Transcript order of plonky2 0.2.0 plonk/prover.rs `prove` with the host-only stages
(partial products, quotient evaluation) replaced by supplied data, exactly like vpbs_prove_step.
"""
import numpy as np

import oracle as orc
import pymodel
from pymodel import P


def step_batches(ncols, num_challenges, zeta, log_n):
    g = pymodel.root_of_unity(log_n)
    zeta_next = np.array([int(zeta[0]) * g % P, int(zeta[1]) * g % P], np.uint64)
    all_polys = [(o, p) for o in range(len(ncols)) for p in range(ncols[o])]
    return [(zeta, all_polys), (zeta_next, [(2, p) for p in range(num_challenges)])], zeta_next


def commit_inputs(inputs, rate_bits=3, cap_height=4):
    return {
        "constants_sigmas": orc.Batch(inputs["constants_sigmas"], rate_bits, cap_height, from_values=True),
        "wires": orc.Batch(inputs["wires"], rate_bits, cap_height, from_values=True),
        "zs_partial_products": orc.Batch(inputs["zs_partial_products"], rate_bits, cap_height, from_values=True),
        "quotient": orc.Batch(inputs["quotient"], rate_bits, cap_height, from_values=False),
    }


def prove_step(inputs, circuit_digest, public_inputs, log_n, num_challenges=2, forced_pow=orc.POW_ANY, cs_batch=None,
               sigmas=None, n_routed=0, n_constants=0, gates=None, compat=None):
    """sigmas given: the Z / partial-product matrix is computed from the wires and the transcript's betas/gammas
    (all_wires_permutation_partial_products) instead of being read from inputs["zs_partial_products"].
    compat: an orc.Compat (the switch table, oracle/vpbs_oracle.h); None = the defaults."""
    rate_bits, cap_height = 3, 4
    cs = cs_batch if cs_batch is not None else orc.Batch(inputs["constants_sigmas"], rate_bits, cap_height, True)
    pi_hash = orc.hash_no_pad(public_inputs)
    wires = orc.Batch(inputs["wires"], rate_bits, cap_height, True)
    ch = orc.ChallengerState()
    ch.observe(circuit_digest)
    ch.observe(pi_hash)
    ch.observe(wires.cap())
    betas = ch.get_n(num_challenges)
    gammas = ch.get_n(num_challenges)
    zs_values = inputs["zs_partial_products"] if sigmas is None else \
        orc.partial_products(inputs["wires"][:n_routed], sigmas, betas, gammas)
    zs = orc.Batch(zs_values, rate_bits, cap_height, True)
    ch.observe(zs.cap())
    alphas = ch.get_n(num_challenges)
    if inputs.get("quotient") is None:   # compute_quotient_polys, permutation-argument constraints only
        sig_c = cs.coeffs()[n_constants:n_constants + n_routed]
        gate_terms = None
        if gates is not None:   # a gates_oracle.GateSet: evaluate_gate_constraints_base_batch folded with the alphas
            gate_terms = gates.terms_coset(cs.coeffs()[:n_constants], wires.coeffs(), pi_hash, alphas)
        q_coeffs = orc.quotient_permutation(wires.coeffs()[:n_routed], sig_c, zs.coeffs(), betas, gammas, alphas, gate_terms=gate_terms)
    else:
        q_coeffs = inputs["quotient"]
    quot = orc.Batch(q_coeffs, rate_bits, cap_height, False)
    ch.observe(quot.cap())
    zeta = ch.get_ext()
    oracles = [cs, wires, zs, quot]
    ncols = [o.ncols for o in oracles]
    batches, zeta_next = step_batches(ncols, num_challenges, zeta, log_n)
    openings = np.concatenate([o.eval_ext(zeta) for o in oracles] + [zs.eval_ext(zeta_next)[:num_challenges]])
    ch.observe(openings)
    params = orc.fri_params(log_n, mul_final_by_x=(compat.fri_mul_final_by_x if compat is not None else 0))
    fri = orc.prove_openings(oracles, batches, ch, params, log_n, forced_pow)
    return {"caps": np.stack([wires.cap(), zs.cap(), quot.cap()]), "openings": openings, "fri": fri,
            "challenger": ch, "challenges": np.array(betas + gammas + alphas + [int(zeta[0]), int(zeta[1])], np.uint64),
            "cs_cap": cs.cap(), "ncols": ncols}


def verify_step(proof, cs_cap, ncols, circuit_digest, public_inputs, log_n, num_challenges=2, compat=None):
    """Verifier side of the same transcript + verify_fri_proof (checks a proof without recomputing any commitment)."""
    ch = orc.ChallengerState()
    ch.observe(circuit_digest)
    ch.observe(orc.hash_no_pad(public_inputs))
    ch.observe(proof["caps"][0])
    ch.get_n(2 * num_challenges)
    ch.observe(proof["caps"][1])
    ch.get_n(num_challenges)
    ch.observe(proof["caps"][2])
    zeta = ch.get_ext()
    batches, _ = step_batches(ncols, num_challenges, zeta, log_n)
    total = sum(ncols)
    openings = [proof["openings"][:total], proof["openings"][total:]]
    ch.observe(proof["openings"])
    caps = [cs_cap, proof["caps"][0], proof["caps"][1], proof["caps"][2]]
    params = orc.fri_params(log_n, mul_final_by_x=(compat.fri_mul_final_by_x if compat is not None else 0))
    return orc.verify_fri(caps, ncols, batches, openings, ch, params, log_n, proof["fri"])


def to_bytes(proof, ncols, n_constants, public_inputs, log_n, num_challenges=2, cap_height=4, rate_bits=3, compat=None):
    """ProofWithPublicInputs::to_bytes restated in Python from the flat proof pieces."""
    import struct
    out = bytearray()
    put = lambda a: out.extend(np.ascontiguousarray(a, dtype="<u8").tobytes())
    put(proof["caps"])
    n_cs, n_w, n_z, n_q = ncols
    op = proof["openings"]
    cs, w, z, q, zn = op[:n_cs], op[n_cs:n_cs + n_w], op[n_cs + n_w:n_cs + n_w + n_z], \
        op[n_cs + n_w + n_z:n_cs + n_w + n_z + n_q], op[n_cs + n_w + n_z + n_q:]
    for part in (cs[:n_constants], cs[n_constants:], w, z[:num_challenges], zn, z[num_challenges:], q):
        put(part)
    params = orc.fri_params(log_n)
    fri = proof["fri"]
    log_lde = log_n + rate_bits
    pos = 0

    def take(k):
        nonlocal pos
        v = fri[pos:pos + k]
        pos += k
        return v
    put(take(params.n_rounds * (4 << cap_height)))
    for _ in range(params.num_query_rounds):
        for nc in ncols:
            put(take(nc))
            nsib = log_lde - cap_height
            out.append(nsib)
            put(take(4 * nsib))
        lg = log_lde
        for r in range(params.n_rounds):
            ab = params.arity_bits[r]
            lg -= ab
            put(take(2 << ab))
            out.append(lg - cap_height)
            put(take(4 * (lg - cap_height)))
    put(take(fri.size - pos))
    if compat is None or compat.bytes_pi_len_prefix:       # write_usize(public_inputs.len())
        out.extend(struct.pack("<Q", len(public_inputs)))
    put(np.asarray(public_inputs, dtype=np.uint64))
    return bytes(out)
