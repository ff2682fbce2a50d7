"""verifiable-fhe-paper_amd: MI355X-native prover hot path for the vPBS step circuit.

Python here is binding + plumbing only (ctypes over the C ABI of include/vpbs_prover.h, synthetic inputs, the
per-rank sharding plan).  The compute path is the HIP library `libvpbs_hip.so`; there is no CPU fallback: every
entry point raises if the library is missing or no device is present.

The directory name is not a valid Python identifier; import it through the root-level `vpbs_amd` loader.
"""
from . import api, synth, sharding, circuit_file  # noqa: F401
from .api import Context, Batch, ChallengerState, FriParams, VpbsError, lib, build_library  # noqa: F401
