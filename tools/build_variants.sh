#!/bin/bash
# Builds timing-ablation variants of the wave kernel into variants/lib_<name>.so (results of the
# variants are WRONG by construction; they exist only to attribute kernel time to phases).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/tfhe-rs_amd/csrc
OUT=$ROOT/variants
mkdir -p $OUT
build() { # name, sed script
  name=$1; shift
  W=/tmp/variant_$name; rm -rf $W; mkdir -p $W; cp $SRC/*.h $SRC/*.hip $W/; mkdir -p $W/../../include 2>/dev/null || true
  sed -i "s|#include \"../../include/tfhe_hip_backend.h\"|#include \"$ROOT/include/tfhe_hip_backend.h\"|" $W/abi.hip
  for s in "$@"; do sed -i "$s" $W/pbs_fft_wave.hip; done
  objs=""
  for f in abi tables pbs_generic pbs_fft_wave keyswitch ciphertext multibit testhooks; do
    if [ $f = pbs_fft_wave ] || [ ! -f /tmp/variant_objs/$f.o ]; then
      mkdir -p /tmp/variant_objs
      o=$W/$f.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $W/$f.hip -o $o
      [ $f != pbs_fft_wave ] && cp $o /tmp/variant_objs/$f.o
    else o=/tmp/variant_objs/$f.o; fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $objs
  echo built $name
}
[ -f $OUT/lib_base.so ] || build base
build nokey 's|b0\[r \* 64\]|cplx{1.0, 0.5}|g' 's|b1\[r \* 64\]|cplx{0.25, 2.0}|g'
build noflags 's|^    flag_wait(f_ready_ot, epoch);||' 's|^    flag_wait(r_done_ot, epoch);.*$||'
build nobfly 's|if (!(r \& (1 << BIT))) bfly(d\[r\], d\[r \| (1 << BIT)\], tw(r));|if (!(r \& (1 << BIT))) { d[r].re += tw(r).re; }|'
build nodecomp 's|decomp_digit(x0, base_log, level, idx)|(int64_t)(x0 >> 41)|' 's|decomp_digit(x1, base_log, level, idx)|(int64_t)(x1 >> 41)|'
build notorus 's|acc_re\[r\] += from_torus(tr);|acc_re[r] += (uint64_t)(int64_t)(int32_t)tr;|' 's|acc_im\[r\] += from_torus(ti);|acc_im[r] += (uint64_t)(int64_t)(int32_t)ti;|'
build nosync 's|HX_WAVE_SYNC();||g'
build notranspose 's|^    for (int r = 0; r < 16; ++r) p1\[68 \* r\] = d\[r\];||' 's|^    for (int r = 0; r < 16; ++r) d\[r\] = p2\[4 \* r\];||' 's|^    for (int r = 0; r < 16; ++r) p2\[4 \* r + (r >> 2)\] = d\[r\];||' 's|^    for (int r = 0; r < 16; ++r) d\[r\] = p3\[r\];||' 's|^    for (int r = 0; r < 16; ++r) p3\[r\] = o\[r\];||' 's|^    for (int r = 0; r < 16; ++r) o\[r\] = p2\[4 \* r + (r >> 2)\];||' 's|^    for (int r = 0; r < 16; ++r) p2\[4 \* r\] = o\[r\];||' 's|^    for (int r = 0; r < 16; ++r) o\[r\] = p1\[68 \* r\];||'
build nostage 's|uint64_t s = buf64\[(t0 + r \* 64) \& (N - 1)\];|uint64_t s = acc_re[(r + 1) \& 15];|' 's|s = buf64\[(t0 + 1024 + r \* 64) \& (N - 1)\];|s = acc_im[(r + 1) \& 15];|' 's|^        p\[r \* 64\] = acc_re\[r\];||' 's|^        p\[1024 + r \* 64\] = acc_im\[r\];||'
build notw 's|= T\[T_\([A-Z0-9]*\) + [^]]*\]|= cplx{0.7, 0.7}|g' 's|{T\[T_[^}]*}|{cplx{0.7,0.7}, cplx{0.6,0.8}, cplx{0.8,0.6}, cplx{0.5,0.5}}|g'
build nopartner 's|const cplx x = q3\[r\];          // partner.s point at the same position|const cplx x = d[15 - r];|' 's|^      for (int r = 0; r < 16; ++r) p3\[r\] = d\[r\];||'
