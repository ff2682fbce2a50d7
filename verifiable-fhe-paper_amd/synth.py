"""Seeded synthetic inputs for the vPBS step-proof workload.

The reference draws every input from unseeded RNGs (/root/reference/src/vtfhe/crypto/poly.rs:72-88,
/root/reference/src/main.rs:50), so reproducible inputs have to be injected: splitmix64 stream, values >= p rejected.
"""
import numpy as np

P = 0xFFFFFFFF00000001
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, count):
    """First `count` outputs of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def field_elements(seed, count):
    """`count` canonical Goldilocks elements: the splitmix64(seed) stream with values >= p dropped."""
    out = np.empty(0, dtype=np.uint64)
    drawn = 0
    while out.size < count:
        need = count - out.size + 16
        z = splitmix64(np.uint64((int(seed) + drawn * int(_GOLDEN)) & 0xFFFFFFFFFFFFFFFF), need)
        drawn += need
        out = np.concatenate([out, z[z < np.uint64(P)]])
    return np.ascontiguousarray(out[:count])


def trace(seed, ncols, log_n):
    """Column-major [ncols][2^log_n] matrix of field elements."""
    return field_elements(seed, ncols << log_n).reshape(ncols, 1 << log_n)


# BASELINE.md 2 / SURVEY.md 8d config 2: the four polynomial batches of one step proof
STEP_COLS = {"constants_sigmas": 85, "wires": 135, "zs_partial_products": 20, "quotient": 16}
STEP_SEED = 0x5EED0000


def step_inputs(log_n, instance=0, cols=None):
    cols = dict(STEP_COLS if cols is None else cols)
    s = STEP_SEED + 16 * instance
    return {
        "wires": trace(s, cols["wires"], log_n),                              # values  -> from_values
        "zs_partial_products": trace(s + 1, cols["zs_partial_products"], log_n),  # values  -> from_values
        "quotient": trace(s + 2, cols["quotient"], log_n),                    # coeffs  -> from_coeffs
        "constants_sigmas": trace(s + 3, cols["constants_sigmas"], log_n),    # values, committed once (untimed)
    }
