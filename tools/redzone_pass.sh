#!/bin/bash
# GPU box: the whole GPU tier (the C++ host binaries of the reference's tests included) with every device block of the
# library between canaries (TFHE_HIP_ARENA_REDZONE=1, csrc/arena.hip) -> gpurun_out/<tag>_redzone.txt
tag=${1:-r06}
out=gpurun_out/${tag}_redzone.txt; mkdir -p gpurun_out
{
  echo "== TFHE_HIP_ARENA_REDZONE=1 python -m pytest tests -m gpu -q -s   ($(date -u +%FT%TZ))"
  TFHE_HIP_ARENA_REDZONE=1 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "arena red zone|passed|failed|FAILED|ERROR|panic" | tail -40
  echo "== the reference's keyswitch determinism test (failed on the runtime's pool: profiles/r04h_ks_cpp_diag2_*), red zones on, three more runs"
  for i in 1 2 3; do TFHE_HIP_ARENA_REDZONE=1 python -m pytest tests/test_reference_gpu_tests_cpp.py -m gpu -q -s -k "reference_parameter_sets" 2>&1 | grep -E "arena red zone|passed|failed|panic"; done
} > $out 2>&1
cat $out
