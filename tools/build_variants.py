#!/usr/bin/env python3
"""Builds tuning variants of one kernel file: variants/lib_<name>.so = the in-tree objects with that file
(default pbs_fft_wave.hip; prefix the flags with "file.hip:" for another) recompiled under extra -D flags.
Usage: build_variants.py name=-DX=1,-DY=2 other=pbs_fft_wave3.hip:-DZ=0 ...
Run a variant with TFHE_HIP_BACKEND_LIB=variants/lib_<name>.so python bench.py ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tfhe_rs_amd", "csrc")
OUT = os.path.join(ROOT, "variants")
os.makedirs(OUT, exist_ok=True)
subprocess.check_call(["make", "-C", SRC], stdout=subprocess.DEVNULL)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    src = "pbs_fft_wave.hip"
    if ".hip:" in flags:
        src, _, flags = flags.partition(":")
    objs = [os.path.join(SRC, "build", f) for f in os.listdir(os.path.join(SRC, "build")) if f.endswith(".o")
            and f != src.replace(".hip", ".o")]
    o = f"/tmp/variant_{name}.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-ffp-contract=off", "-c", os.path.join(SRC, src), "-o", o] +
                          [f for f in flags.split(",") if f])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(OUT, f"lib_{name}.so"), o] + objs)
    print("built", name)
