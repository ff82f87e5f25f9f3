#!/usr/bin/env python3
"""Identity of the kernel build: SHA-256 over the sources libtfhe_hip_backend.so is compiled from
(tfhe_rs_amd/csrc/*.{hip,h}, its Makefile, include/tfhe_hip_backend.h), in name order.  tools/pmc.sh
stamps the counter record it writes (profiles/pmc_latest.json) with it and bench.py refuses a
record whose stamp differs from the tree it runs from — measured HBM traffic is then reported only
for the build that was actually profiled."""
import glob
import hashlib
import os

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))


def source_build_id(root=ROOT):
    files = sorted(glob.glob(os.path.join(root, "tfhe_rs_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "tfhe_rs_amd", "csrc", "*.h")) +
                   [os.path.join(root, "tfhe_rs_amd", "csrc", "Makefile"),
                    os.path.join(root, "include", "tfhe_hip_backend.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_build_id())
