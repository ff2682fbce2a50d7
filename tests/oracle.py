"""ctypes binding of the CPU oracle (oracle/_build/libvpbs_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libvpbs_oracle.so")

P = 0xFFFFFFFF00000001
U64P = C.POINTER(C.c_uint64)
U32P = C.POINTER(C.c_uint32)
POW_ANY = 0xFFFFFFFFFFFFFFFF


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


class Challenger(C.Structure):
    _fields_ = [("sponge", C.c_uint64 * 12), ("input", C.c_uint64 * 8), ("output", C.c_uint64 * 8),
                ("input_len", C.c_uint32), ("output_len", C.c_uint32)]


class FriParams(C.Structure):
    _fields_ = [("rate_bits", C.c_uint), ("cap_height", C.c_uint), ("pow_bits", C.c_uint),
                ("num_query_rounds", C.c_uint), ("n_rounds", C.c_uint), ("arity_bits", C.c_uint * 16),
                ("mul_final_by_x", C.c_int)]


class Compat(C.Structure):
    """orc_compat = the oracle's copy of include/vpbs_prover.h vpbs_compat (the switch table of the unpinned plonky2 choices)"""
    _fields_ = [("fri_mul_final_by_x", C.c_int), ("bytes_pi_len_prefix", C.c_int), ("digest_domain_separator", C.c_int),
                ("pow_smallest_nonce", C.c_int)]


COMPAT_FIELDS = ("fri_mul_final_by_x", "bytes_pi_len_prefix", "digest_domain_separator", "pow_smallest_nonce")


def compat(**over):
    """the default table (plonky2 0.2.0 as restated) with the given switches changed"""
    k = Compat()
    lib().orc_compat_default(C.byref(k))
    for name, v in over.items():
        assert name in COMPAT_FIELDS, name
        setattr(k, name, int(v))
    return k


def compat_dict(k=None):
    k = k if k is not None else compat()
    return {name: int(getattr(k, name)) for name in COMPAT_FIELDS}


class FriBatchInfo(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("n_polys", C.c_size_t), ("oracle_index", U32P), ("poly_index", U32P)]


_lib = None


def effective_cpus():
    """CPUs this process can really use: the scheduler affinity capped by the cgroup CPU quota (containers on many-core hosts)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u64, sz, vp, ui = C.c_uint64, C.c_size_t, C.c_void_p, C.c_uint
        sig = {
            "orc_gl_add": (u64, [u64, u64]), "orc_gl_sub": (u64, [u64, u64]), "orc_gl_mul": (u64, [u64, u64]),
            "orc_gl_inv": (u64, [u64]), "orc_gl_exp": (u64, [u64, u64]), "orc_gl_root_of_unity": (u64, [ui]),
            "orc_ext_mul": (None, [U64P, U64P, U64P]), "orc_ext_inv": (None, [U64P, U64P]),
            "orc_poseidon": (None, [U64P]), "orc_poseidon_batch": (None, [U64P, sz]),
            "orc_hash_no_pad": (None, [U64P, sz, U64P]), "orc_hash_or_noop": (None, [U64P, sz, U64P]),
            "orc_hash_pad": (None, [U64P, sz, U64P]), "orc_compat_default": (None, [C.POINTER(Compat)]),
            "orc_circuit_digest": (None, [C.POINTER(Compat), U64P, sz, ui, U64P]),
            "orc_two_to_one": (None, [U64P, U64P, U64P]), "orc_hash_chain": (None, [U64P, sz, sz, U64P]),
            "orc_fft": (None, [U64P, ui]), "orc_ifft": (None, [U64P, ui]),
            "orc_coset_lde": (None, [U64P, ui, ui, u64, U64P]),
            "orc_merkle_new": (vp, [U64P, sz, sz, ui]), "orc_merkle_free": (None, [vp]),
            "orc_merkle_cap": (None, [vp, U64P]), "orc_merkle_proof_len": (sz, [vp]),
            "orc_merkle_leaf": (None, [vp, sz, U64P]), "orc_merkle_prove": (None, [vp, sz, U64P]),
            "orc_merkle_verify": (C.c_int, [U64P, sz, sz, U64P, ui, U64P, sz]),
            "orc_batch_from_values": (vp, [U64P, sz, ui, ui, ui]), "orc_batch_from_coeffs": (vp, [U64P, sz, ui, ui, ui]),
            "orc_batch_free": (None, [vp]), "orc_batch_cap": (None, [vp, U64P]),
            "orc_batch_coeffs": (U64P, [vp]), "orc_batch_leaves": (U64P, [vp]), "orc_batch_ncols": (sz, [vp]),
            "orc_batch_lde_row": (None, [vp, sz, sz, U64P]), "orc_batch_eval_ext": (None, [vp, U64P, U64P]),
            "orc_batch_open": (None, [vp, sz, U64P, U64P]),
            "orc_challenger_init": (None, [C.POINTER(Challenger)]),
            "orc_challenger_observe": (None, [C.POINTER(Challenger), U64P, sz]),
            "orc_challenger_get": (u64, [C.POINTER(Challenger)]),
            "orc_challenger_get_n": (None, [C.POINTER(Challenger), U64P, sz]),
            "orc_fri_params_standard": (None, [ui, C.POINTER(FriParams)]),
            "orc_fri_proof_words": (sz, [C.POINTER(FriParams), ui, C.POINTER(sz), sz]),
            "orc_prove_openings": (C.c_int, [C.POINTER(vp), sz, C.POINTER(FriBatchInfo), sz, C.POINTER(Challenger),
                                            C.POINTER(FriParams), ui, u64, U64P]),
            "orc_verify_fri": (C.c_int, [C.POINTER(U64P), C.POINTER(sz), sz, C.POINTER(FriBatchInfo), C.POINTER(U64P), sz,
                                        C.POINTER(Challenger), C.POINTER(FriParams), ui, U64P]),
            "orc_partial_products": (C.c_int, [U64P, U64P, sz, ui, U64P, U64P, sz, sz, U64P]),
            "orc_quotient_permutation": (C.c_int, [U64P, U64P, U64P, sz, ui, U64P, U64P, U64P, sz, sz, U64P, U64P]),
            "orc_check_vanishing_at_zeta": (C.c_int, [U64P, U64P, U64P, U64P, U64P, U64P, sz, ui, U64P, U64P, U64P, sz, sz, U64P, U64P]),
            "orc_negacyclic_params": (None, [ui, U64P, U64P, U64P]),
            "orc_negacyclic_forward": (None, [U64P, ui, U64P]),
            "orc_negacyclic_backward": (None, [U64P, ui, U64P, u64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L.orc_set_num_threads.restype, L.orc_set_num_threads.argtypes = None, [C.c_int]
        L.orc_set_num_threads(effective_cpus())
        _lib = L
    return _lib


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


def u64arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


# ---- thin functional wrappers ----
def poseidon(state):
    s = u64arr(state).copy()
    lib().orc_poseidon(ptr(s))
    return s


def hash_no_pad(x):
    x = u64arr(x); out = np.zeros(4, np.uint64)
    lib().orc_hash_no_pad(ptr(x), x.size, ptr(out))
    return out


def hash_pad(x):
    x = u64arr(x).reshape(-1); out = np.zeros(4, np.uint64)
    lib().orc_hash_pad(ptr(x) if x.size else None, x.size, ptr(out))
    return out


def circuit_digest(cs_cap, log_n, k=None):
    """CircuitBuilder::build's circuit_digest (formula: compat.digest_domain_separator)"""
    cap = u64arr(cs_cap).reshape(-1); out = np.zeros(4, np.uint64)
    lib().orc_circuit_digest(C.byref(k) if k is not None else None, ptr(cap), cap.size, log_n, ptr(out))
    return out


def hash_or_noop(x):
    x = u64arr(x); out = np.zeros(4, np.uint64)
    lib().orc_hash_or_noop(ptr(x), x.size, ptr(out))
    return out


def two_to_one(l, r):
    l, r = u64arr(l), u64arr(r); out = np.zeros(4, np.uint64)
    lib().orc_two_to_one(ptr(l), ptr(r), ptr(out))
    return out


def fft(a, inverse=False):
    a = u64arr(a).copy(); log_n = a.size.bit_length() - 1
    (lib().orc_ifft if inverse else lib().orc_fft)(ptr(a), log_n)
    return a


def coset_lde(coeffs, rate_bits, shift=7):
    c = u64arr(coeffs); log_n = c.size.bit_length() - 1
    out = np.zeros(c.size << rate_bits, np.uint64)
    lib().orc_coset_lde(ptr(c), log_n, rate_bits, shift, ptr(out))
    return out


def negacyclic_params(log_n):
    n = 1 << log_n
    roots, inv, ninv = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(1, np.uint64)
    lib().orc_negacyclic_params(log_n, ptr(roots), ptr(inv), ptr(ninv))
    return roots, inv, int(ninv[0])


def negacyclic_forward(a, roots):
    a = u64arr(a).copy()
    lib().orc_negacyclic_forward(ptr(a), a.size.bit_length() - 1, ptr(u64arr(roots)))
    return a


def negacyclic_backward(a, invroots, ninv):
    a = u64arr(a).copy()
    lib().orc_negacyclic_backward(ptr(a), a.size.bit_length() - 1, ptr(u64arr(invroots)), ninv)
    return a


def partial_products(wires, sigmas, betas, gammas, max_degree=8):
    w, sg = u64arr(wires), u64arr(sigmas)
    n_routed, n = sg.shape
    nc = len(betas)
    chunks = (n_routed + max_degree - 1) // max_degree
    out = np.zeros((nc * chunks, n), np.uint64)
    rc = lib().orc_partial_products(ptr(w), ptr(sg), n_routed, n.bit_length() - 1, ptr(u64arr(betas)), ptr(u64arr(gammas)), nc, max_degree, ptr(out))
    assert rc == 0, rc
    return out


def quotient_permutation(wires_coeffs, sigmas_coeffs, zs_pp_coeffs, betas, gammas, alphas, max_degree=8, gate_terms=None):
    """compute_quotient_polys restricted to the permutation argument: -> [nc * 8][n] coefficient chunks."""
    w, sg, zp = u64arr(wires_coeffs), u64arr(sigmas_coeffs), u64arr(zs_pp_coeffs)
    n_routed, n = sg.shape
    nc = len(betas)
    out = np.zeros((nc * 8, n), np.uint64)
    gt = ptr(u64arr(gate_terms)) if gate_terms is not None else None
    rc = lib().orc_quotient_permutation(ptr(w), ptr(sg), ptr(zp), n_routed, n.bit_length() - 1, ptr(u64arr(betas)), ptr(u64arr(gammas)),
                                        ptr(u64arr(alphas)), nc, max_degree, gt, ptr(out))
    assert rc == 0
    return out


def check_vanishing_at_zeta(wires_z, sigmas_z, zs_z, zs_next_z, pps_z, quotient_z, log_n, betas, gammas, alphas, zeta, max_degree=8,
                            gate_terms_zeta=None):
    a = [u64arr(x) for x in (wires_z, sigmas_z, zs_z, zs_next_z, pps_z, quotient_z)]
    n_routed = a[1].shape[0]
    pps_ptr = ptr(a[4]) if a[4].size else ptr(np.zeros(2, np.uint64))
    gt = ptr(u64arr(gate_terms_zeta)) if gate_terms_zeta is not None else None
    return bool(lib().orc_check_vanishing_at_zeta(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), pps_ptr, ptr(a[5]), n_routed, log_n,
                                                  ptr(u64arr(betas)), ptr(u64arr(gammas)), ptr(u64arr(alphas)), len(betas), max_degree,
                                                  ptr(u64arr(zeta)), gt))


def eval_coeffs_ext(coeffs, zeta):
    """[ncols][n] coefficient matrix -> [ncols][2] evaluations at the extension point zeta (Horner, C)."""
    c = u64arr(coeffs)
    out = np.zeros((c.shape[0], 2), np.uint64)
    z = (int(zeta[0]), int(zeta[1]))
    P_ = P
    for k in range(c.shape[0]):
        a0 = a1 = 0
        for v in c[k][::-1]:
            a0, a1 = (a0 * z[0] + 7 * a1 * z[1] + int(v)) % P_, (a0 * z[1] + a1 * z[0]) % P_
        out[k] = (a0, a1)
    return out


class Merkle:
    def __init__(self, leaves, cap_height):
        leaves = u64arr(leaves)
        self.n, self.leaf_len = leaves.shape
        self.cap_height = cap_height
        self.h = lib().orc_merkle_new(ptr(leaves), self.n, self.leaf_len, cap_height)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_merkle_free(self.h); self.h = None

    def cap(self):
        out = np.zeros((1 << self.cap_height, 4), np.uint64)
        lib().orc_merkle_cap(self.h, ptr(out)); return out

    def prove(self, idx):
        out = np.zeros((lib().orc_merkle_proof_len(self.h), 4), np.uint64)
        lib().orc_merkle_prove(self.h, idx, ptr(out)); return out


def merkle_verify(leaf, idx, cap, cap_height, siblings):
    leaf, cap, siblings = u64arr(leaf), u64arr(cap), u64arr(siblings)
    return bool(lib().orc_merkle_verify(ptr(leaf), leaf.size, idx, ptr(cap), cap_height, ptr(siblings), siblings.shape[0]))


class Batch:
    """PolynomialBatch restated (fri/oracle.rs)."""

    def __init__(self, data, rate_bits=3, cap_height=4, from_values=True):
        data = u64arr(data)
        self.ncols, self.n = data.shape
        self.log_n = self.n.bit_length() - 1
        self.rate_bits, self.cap_height = rate_bits, cap_height
        f = lib().orc_batch_from_values if from_values else lib().orc_batch_from_coeffs
        self.h = f(ptr(data), self.ncols, self.log_n, rate_bits, cap_height)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_batch_free(self.h); self.h = None

    def cap(self):
        out = np.zeros((1 << self.cap_height, 4), np.uint64)
        lib().orc_batch_cap(self.h, ptr(out)); return out

    def coeffs(self):
        p = lib().orc_batch_coeffs(self.h)
        return np.ctypeslib.as_array(p, shape=(self.ncols, self.n)).copy()

    def leaves(self):
        p = lib().orc_batch_leaves(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n << self.rate_bits, self.ncols)).copy()

    def lde_row(self, index, step=1):
        out = np.zeros(self.ncols, np.uint64)
        lib().orc_batch_lde_row(self.h, index, step, ptr(out)); return out

    def eval_ext(self, zeta):
        z = u64arr(zeta); out = np.zeros((self.ncols, 2), np.uint64)
        lib().orc_batch_eval_ext(self.h, ptr(z), ptr(out)); return out

    def open(self, idx):
        nsib = self.log_n + self.rate_bits - self.cap_height
        leaf, sib = np.zeros(self.ncols, np.uint64), np.zeros((nsib, 4), np.uint64)
        lib().orc_batch_open(self.h, idx, ptr(leaf), ptr(sib)); return leaf, sib


class ChallengerState:
    def __init__(self):
        self.c = Challenger(); lib().orc_challenger_init(C.byref(self.c))

    def clone(self):
        o = ChallengerState(); C.memmove(C.byref(o.c), C.byref(self.c), C.sizeof(Challenger)); return o

    def observe(self, elems):
        e = u64arr(elems).reshape(-1)
        lib().orc_challenger_observe(C.byref(self.c), ptr(e), e.size)

    def get(self):
        return int(lib().orc_challenger_get(C.byref(self.c)))

    def get_n(self, n):
        return [self.get() for _ in range(n)]

    def get_ext(self):
        return np.array(self.get_n(2), np.uint64)

    def state_words(self):
        """(sponge[12], input[8], input_len, output[8], output_len) as plain python for comparisons."""
        c = self.c
        return (list(c.sponge), list(c.input)[:c.input_len], list(c.output)[:c.output_len])


def fri_params(degree_bits, **over):
    p = FriParams(); lib().orc_fri_params_standard(degree_bits, C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def make_batch_infos(batches):
    """batches: list of (point(2), [(oracle_index, poly_index), ...]) -> (ctypes array, keepalive list)."""
    arr = (FriBatchInfo * len(batches))()
    keep = []
    for i, (point, polys) in enumerate(batches):
        oi = np.array([p[0] for p in polys], np.uint32); pi = np.array([p[1] for p in polys], np.uint32)
        keep += [oi, pi]
        arr[i].point[0], arr[i].point[1] = int(point[0]), int(point[1])
        arr[i].n_polys = len(polys)
        arr[i].oracle_index = oi.ctypes.data_as(U32P); arr[i].poly_index = pi.ctypes.data_as(U32P)
    return arr, keep


def prove_openings(oracles, batches, challenger, params, degree_bits, forced_pow=POW_ANY):
    ncols = (C.c_size_t * len(oracles))(*[o.ncols for o in oracles])
    words = lib().orc_fri_proof_words(C.byref(params), degree_bits, ncols, len(oracles))
    proof = np.zeros(words, np.uint64)
    handles = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    infos, keep = make_batch_infos(batches)
    rc = lib().orc_prove_openings(handles, len(oracles), infos, len(batches), C.byref(challenger.c), C.byref(params),
                                  degree_bits, forced_pow, ptr(proof))
    assert rc == 0, rc
    return proof


def verify_fri(caps, ncols, batches, openings, challenger, params, degree_bits, proof):
    caps = [u64arr(c) for c in caps]; openings = [u64arr(o) for o in openings]
    cap_ptrs = (U64P * len(caps))(*[ptr(c) for c in caps])
    open_ptrs = (U64P * len(openings))(*[ptr(o) for o in openings])
    nc = (C.c_size_t * len(ncols))(*ncols)
    infos, keep = make_batch_infos(batches)
    proof = u64arr(proof)
    return bool(lib().orc_verify_fri(cap_ptrs, nc, len(caps), infos, open_ptrs, len(batches), C.byref(challenger.c),
                                     C.byref(params), degree_bits, ptr(proof)))
