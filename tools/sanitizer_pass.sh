#!/bin/bash
# CPU box: the kernel sources of the library under AddressSanitizer + UndefinedBehaviorSanitizer — the host-emulation build
# (tests/emu, make SAN=1: every kernel, the C ABI layer, the arena, the radix layer) through the CPU tier's kernel tests, and
# the C++ host binaries of the reference's GPU tests linked against it.  GPU AddressSanitizer is not available on this pool;
# this is the CPU tier's counterpart of the reference's compute-sanitizer runs (scripts/check_memory_errors.sh:1-60).
#   bash tools/sanitizer_pass.sh [tag] ["<pytest -k expression>"]     -> profiles/<tag>_sanitizer_cpu.txt
#   (CPP_FILTER=<substring>: only the C++ tests whose name holds it; SANOUT: where the instrumented build goes, default /tmp)
tag=${1:-r06}
sel=${2:-"emu"}
cd "$(dirname "$0")/.."
out=profiles/${tag}_sanitizer_cpu.txt
sandir=${SANOUT:-/tmp/tfhe_hip_san}; mkdir -p $sandir
make -C tests/emu -j${SANJOBS:-8} SAN=1 SANOUT=$sandir > /dev/null || exit 1
san=$sandir/libtfhe_hip_backend_emu_san.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=4
{
  echo "== $(date -u +%FT%TZ) g++ $(g++ -dumpversion) -O1 -fsanitize=address,undefined, TFHE_EMU_LIB=$san"
  echo "== pytest tests/test_backend_parity.py tests/test_radix_integer.py tests/test_radix_more_ops.py tests/test_malloc_async_arena.py tests/test_streams_and_graphs.py tests/test_split_engine_worst_case.py tests/test_fourier_entry_points.py tests/test_multi_bit_noise_entry_points.py -m 'not gpu' -k '$sel'"
  LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) TFHE_EMU_LIB=$san TFHE_HIP_ARENA_REDZONE=${REDZONE:-0} \
    python -m pytest tests/test_backend_parity.py tests/test_radix_integer.py tests/test_radix_more_ops.py tests/test_malloc_async_arena.py tests/test_streams_and_graphs.py \
      tests/test_split_engine_worst_case.py tests/test_fourier_entry_points.py tests/test_multi_bit_noise_entry_points.py \
      -m "not gpu" -k "$sel" -q -x -p no:cacheprovider 2>&1 | grep -vE "^\s*$" | tail -15
  echo "== the reference's GPU tests restated in C++ (tests/cpp), compiled with the same flags and linked against the instrumented library"
  for src in reference_gpu_tests reference_integer_gpu_tests; do
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/${src}_san tests/cpp/$src.cpp $san oracle/libtfhe_oracle.so \
      -Wl,-rpath,$sandir -Wl,-rpath,$PWD/oracle || exit 1
    TFHE_FFT_GOLDEN=$PWD/tests/golden/fft16x4x16_golden_v1.json /tmp/${src}_san toy ${CPP_FILTER:-} 2>&1 | grep -E "test result|ERROR|runtime error|FAILED" | tail -5
  done
} > $out 2>&1
cat $out
# the instrumented objects live outside the tree ($sandir): gpurun refuses a snapshot that holds AddressSanitizer code
