# usage on the GPU box: bash tools/run_split_variants.sh name1 name2 ...  (exact engine, split-key form, batch 4096)
for v in "$@" default; do
  lib=variants/lib_$v.so; [ "$v" = default ] && lib=tfhe_rs_amd/lib/libtfhe_hip_backend.so
  echo "== $v"; TFHE_HIP_BACKEND_LIB=$lib python tools/measure_all.py ntt_split 2>&1 | grep '"batch": 4096' | cut -c60-220
done
