"""Circuits that arrive as data: the flat description a host exports from its circuit builder after build() -- gates, gate per row, constants
columns, copy constraints, gadget generators, the targets the PartialWitness sets and the public-input targets -- in the little-endian
u64 format of examples/prove_step_circuit.cpp (the C++ reader) and tools/export_step_circuit.py (the exporter that stands in for the Rust
side, /root/reference/src/vtfhe/ivc_based_vpbs.rs:80-155 `build_step_circuit` -> CircuitData).  The prover-side tools
(bench.py, tools/prove_pbs.py, tools/prove_ivc.py) load a circuit through this module and the C ABI only; they never import or run a circuit
builder: find_* locate exported files and raise when one is missing (tools/export_circuits.py, run by __graft_entry__.build(), makes them).

File: header {magic, log_n, n_wires, n_routed, n_gates, n_constants_cols, n_copies, n_generators, generator_words, n_preset, n_public_inputs};
gates [n_gates][kind, p0, p1, p2]; row_gate [n]; constants [cols][n]; copies [n_copies][2]; generators {kind, p0, n_in, n_out, in.., out..}*;
preset positions; public-input positions; sample preset values; expected public inputs; optional trailer {N, K, ELL, LOGB, n_lwe, used_rows
[, kind, proof_words]} (kind 1: the cyclic step circuit, 2: its dummy circuit -- no sample witness in those files).
Positions are column * n + row."""
import os

import numpy as np

from . import api

MAGIC = 0x5354455043495243  # "STEPCIRC"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "circuits")


class CircuitDescription:
    """What load() returns: `circuit` (api.Circuit: sigma_values, witness_plan, ...), `gates` (api.GateSet), the constants columns, the
    PartialWitness positions in the exporter's order, the public-input positions, one sample PartialWitness with its public inputs."""

    def __init__(self, words):
        if words.size < 11 or int(words[0]) != MAGIC:
            raise ValueError("not a step-circuit file")
        (self.log_n, self.n_wires, self.n_routed, n_gates, n_cols, n_copies, n_generators, gen_words, n_preset, n_pi) = (int(x) for x in words[1:11])
        self.n = n = 1 << self.log_n
        pos = 11

        def take(k):
            nonlocal pos
            v = words[pos:pos + k]
            if v.size != k:
                raise ValueError("truncated circuit file")
            pos += k
            return v
        g = take(4 * n_gates).reshape(n_gates, 4)
        self.gates = api.GateSet([(api.GATE_KINDS[int(k)], int(p0), int(p1), int(p2)) for k, p0, p1, p2 in g])
        row_gate = take(n).astype(np.uint32)
        self.constants = np.ascontiguousarray(take(n_cols * n).reshape(n_cols, n))
        copies = take(2 * n_copies).astype(np.uint32).reshape(-1, 2)
        gw = take(gen_words)
        gens, q = [], 0
        unpos = lambda v: [(int(x) // n, int(x) % n) for x in v]
        for _ in range(n_generators):
            kind, p0, n_in, n_out = (int(x) for x in gw[q:q + 4])
            gens.append((api.GENERATOR_KINDS[kind], p0, unpos(gw[q + 4:q + 4 + n_in]), unpos(gw[q + 4 + n_in:q + 4 + n_in + n_out])))
            q += 4 + n_in + n_out
        self.preset_flat = take(n_preset).astype(np.int64)
        self.pi_flat = take(n_pi).astype(np.int64)
        self.sample_values = take(n_preset).copy()
        self.sample_public_inputs = take(n_pi).copy()
        self.meta = {}
        if words.size - pos >= 6:
            self.meta = dict(zip(("N", "K", "ELL", "LOGB", "n_lwe", "used_rows"), (int(x) for x in take(6))))
        if words.size - pos >= 2:       # cyclic / dummy circuits: kind (1 / 2) and the number of proof words among the presets
            self.meta.update(zip(("kind", "proof_words"), (int(x) for x in take(2))))
        self.preset_pos = [(int(x) // n, int(x) % n) for x in self.preset_flat]
        self.pi_pos = [(int(x) // n, int(x) % n) for x in self.pi_flat]
        self.circuit = api.Circuit(self.gates, self.log_n, row_gate, self.constants, copies, self.n_wires, self.n_routed, gens)
        self.n_constants = self.constants.shape[0]
        self.used_rows = self.meta.get("used_rows", int((row_gate != self.gates.by_kind("noop").index).sum()) if any(
            gg.kind == api.GATE_KINDS.index("noop") for gg in self.gates) else n)


def load(path):
    return CircuitDescription(np.fromfile(path, dtype="<u8"))


def step_circuit_path(N, K, ELL, LOGB, n_lwe):
    return os.path.join(DIR, "step_N%d_K%d_ELL%d_LOGB%d_n%d_v2.bin" % (N, K, ELL, LOGB, n_lwe))   # v2: split_le asserts its unused limbs zero


def _missing(what, paths, args):
    return FileNotFoundError(
        "%s: %s not found.  Circuit files are produced OUTSIDE the product package by the circuit builder's exporter -- here "
        "`python tools/export_circuits.py %s` (what __graft_entry__.build() runs for the standard parameter sets); with the reference, "
        "tools/plonky2_capture exports the Rust builder's own circuit in the same format" % (what, ", ".join(paths), args))


def find_step_circuit(N=1024, K=2, ELL=4, LOGB=5, n_lwe=728):
    """Path of the exported step circuit at these parameters; FileNotFoundError when it has not been exported (this module only LOCATES
    circuit files: it neither runs nor imports a circuit builder)."""
    path = step_circuit_path(N, K, ELL, LOGB, n_lwe)
    if not os.path.exists(path):
        raise _missing("step circuit", [path], "--step %d %d %d %d %d" % (N, K, ELL, LOGB, n_lwe))
    return path


def circuit_digest(cs_cap, log_n, compat=None):
    """verifier_only.circuit_digest as CircuitBuilder::build derives it: hash_no_pad(constants/sigmas cap || hash_pad(domain separator = [])
    || degree bits) (vpbs_circuit_digest; compat: an api.CompatC for the older formula without the separator)."""
    return api.circuit_digest(cs_cap, log_n, compat)


def verifier_data_words(cs_cap, log_n, compat=None):
    """the 68 public-input words of add_verifier_data_public_inputs: circuit digest [4], then the cap [16][4]"""
    return np.concatenate([circuit_digest(cs_cap, log_n, compat), np.asarray(cs_cap, np.uint64).reshape(-1)])


def cyclic_circuit_paths(N, K, ELL, LOGB, n_lwe, log_n):
    stem = "N%d_K%d_ELL%d_LOGB%d_n%d_deg%d_slots2.bin" % (N, K, ELL, LOGB, n_lwe, log_n)   # slots2: the two-proof-slot PartialWitness layout
    return os.path.join(DIR, "cyclic_" + stem), os.path.join(DIR, "dummy_" + stem)


def find_cyclic_circuit(N=1024, K=2, ELL=4, LOGB=5, n_lwe=728, log_n=16):
    """(cyclic circuit file, dummy circuit file) of the IVC chain at these parameters; FileNotFoundError when they have not been exported"""
    path, dummy = cyclic_circuit_paths(N, K, ELL, LOGB, n_lwe, log_n)
    if not (os.path.exists(path) and os.path.exists(dummy)):
        raise _missing("cyclic step circuit", [path, dummy], "--cyclic %d %d %d %d %d %d" % (N, K, ELL, LOGB, n_lwe, log_n))
    return path, dummy
