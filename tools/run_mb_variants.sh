# usage on the GPU box: bash tools/run_mb_variants.sh name1 name2 ...  (multi-bit PBS, batch 4096, libraries from tools/build_variants.py)
for v in "$@"; do
  echo -n "$v "; TFHE_HIP_BACKEND_LIB=variants/lib_$v.so python tools/measure_all.py mb 2>&1 | tail -1 | cut -c100-200
done
echo -n "default "; python tools/measure_all.py mb 2>&1 | tail -1 | cut -c100-200
