# usage on the GPU box: bash tools/run_wave_variants.sh name1 name2 ...  (classic PBS 2_2, batch 4096; "default" = in-tree)
for v in "$@" default; do
  lib=variants/lib_$v.so; [ "$v" = default ] && lib=tfhe_rs_amd/lib/libtfhe_hip_backend.so
  echo "== $v"; TFHE_HIP_BACKEND_LIB=$lib python tools/measure_all.py wave 2>&1 | grep '"batch": 4096' | sed -e 's/.*"ms": \([0-9.]*\).*pbs_per_s": \([0-9.]*\).*/ms \1 pbs \2/'
done
