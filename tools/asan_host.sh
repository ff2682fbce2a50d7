#!/bin/bash
# AddressSanitizer run of the library's HOST code (witness generation, gate evaluation at zeta, verifier, challenger, layout) through
# the CPU test-suite.  Device code is compiled without sanitizer (GPU ASAN needs xnack+, not available on the pool).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/vpbs_asan; mkdir -p $OUT
cd "$ROOT/verifiable-fhe-paper_amd/csrc"
for f in ntt hash fri permutation quotient gates witness witness_device tfhe keygen comm_rccl api prover verifier ivc; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -Wno-unused-function -Wno-pass-failed -c $f.hip -o $OUT/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -o $OUT/libvpbs_hip.so $OUT/*.o
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
cp "$ROOT/verifiable-fhe-paper_amd/libvpbs_hip.so" $OUT/real.so
cp $OUT/libvpbs_hip.so "$ROOT/verifiable-fhe-paper_amd/libvpbs_hip.so"
trap 'cp $OUT/real.so "$ROOT/verifiable-fhe-paper_amd/libvpbs_hip.so"' EXIT
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$RT python -m pytest tests/test_gates_cpu.py tests/test_host_cpu.py tests/test_golden_plonky2.py tests/test_step_circuit_cpu.py tests/test_cyclic_cpu.py -k "not cxx" -x -q -m "not gpu" -p no:cacheprovider
