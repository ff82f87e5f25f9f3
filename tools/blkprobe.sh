for v in default blknokey blknobar blkboth; do
  if [ $v = default ]; then L=""; else L="TFHE_HIP_BACKEND_LIB=variants/lib_$v.so"; fi
  echo -n "$v "; env $L python tools/measure_all.py mblat 2>&1 | grep '"requested": 5, "batch": 1,' | grep GROUP_4 | cut -c150-200
done
