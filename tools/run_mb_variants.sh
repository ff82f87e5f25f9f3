# usage on the GPU box: bash tools/run_mb_variants.sh name1 name2 ...  (multi-bit PBS g=3 and g=4, batch 4096,
# libraries from tools/build_variants.py; "default" = the in-tree library)
for v in "$@" default; do
  lib=variants/lib_$v.so; [ "$v" = default ] && lib=tfhe_rs_amd/lib/libtfhe_hip_backend.so
  echo "== $v"; TFHE_HIP_BACKEND_LIB=$lib python tools/measure_all.py mb mb4 2>&1 | grep '"batch": 4096' | cut -c40-220
done
