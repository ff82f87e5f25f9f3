#!/bin/bash
# Host-emulation build of a tuning variant of one kernel file, for the CPU test tier:
#   tools/build_emu_variant.sh name file.hip -DX=1 -DY=2   ->  variants/emu_<name>.so
# run the tests on it with  TFHE_EMU_LIB=variants/emu_<name>.so python -m pytest tests -m "not gpu" -k emu
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
make -C tests/emu -j8 >/dev/null
mkdir -p variants
obj=/tmp/emu_variant_$name.o
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -mfma -fopenmp -DTFHE_HIPEMU -Itests/emu -Itfhe_rs_amd/csrc -Wno-unknown-pragmas \
    "$@" -x c++ -c tfhe_rs_amd/csrc/$src -o $obj
objs=$(ls tests/emu/build/*.o | grep -v "/${src%.hip}.o")
g++ -shared -fopenmp -o variants/emu_$name.so $obj $objs
echo built variants/emu_$name.so
