#!/usr/bin/env python3
"""Dynamic VALU-instruction count of the product Poseidon permutation on gfx950 (the number quoted in DESIGN.md / bench.py).

Compiles a one-permutation-per-lane kernel to ISA, counts VALU instructions per loop body and multiplies by the trip
counts (4 + 4 full rounds, 7 fused partial groups, 1 plain partial round).  Needs hipcc only (no GPU)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "verifiable-fhe-paper_amd", "csrc")
SRC = r'''
#include "poseidon.h"
using gl::u64;
__global__ void __launch_bounds__(256) permute_batch_kernel(u64* states, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = states[12 * i + k];
    poseidon::permute(s);
#pragma unroll
    for (int k = 0; k < 12; ++k) states[12 * i + k] = s[k];
}
'''


def main():
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "t.hip"), os.path.join(d, "t.s")
        open(src, "w").write(SRC)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-S", "--cuda-device-only", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        # -S does not ASSEMBLE inline asm: an operand the assembler rejects (gfx950: VGPR pairs must be 64-bit aligned) only shows with -c
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-c", "--cuda-device-only", src, "-o",
                               os.path.join(d, "t.o")], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*permute_batch_kernel.*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the practically-never-taken second corrections of the asm arithmetic (gl.h: .Lgl_* labels): skipped by a forward wave-level branch --
    # not part of the dynamic count
    kept, skip_to, rare = [], None, 0
    for l in body:
        if skip_to is not None:
            if l.strip().startswith(skip_to + ":"):
                skip_to = None
            elif re.match(r"^\s+v_", l):
                rare += 1
            continue
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.Lgl_\w+)", l)
        if m:
            skip_to = m.group(1)
            continue
        kept.append(l)
    print("VALU instructions behind never-taken forward branches (excluded):", rare)
    body = kept
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            loops.append(sum(1 for x in seg if re.match(r"^\s+v_", x)))
    total_static = sum(1 for x in body if re.match(r"^\s+v_", x))
    print("static VALU instructions:", total_static, " loop bodies:", loops)
    if len(loops) == 3:
        full_a, group, full_b = loops
        rest = total_static - sum(loops)   # first constant layer, the plain partial round, canonicalisation, load/store
        dyn = 4 * full_a + 7 * group + 4 * full_b + rest
        print("dynamic VALU instructions per permutation ~ %d  (4 x %d + 7 x %d + 4 x %d + %d)" % (dyn, full_a, group, full_b, rest))


if __name__ == "__main__":
    main()
