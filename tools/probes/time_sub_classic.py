import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from tests.test_radix_integer import setup, encrypt_radix, decrypt_blocks, recompose
from tfhe_rs_amd import ffi
p, keys, st, sks, igpu = setup("hip")
lib = ffi.default_library()
L = 32
a, b = 0x123456789ABCDEF0, 0x0FEDCBA987654321
for op in ("add", "sub", "gt", "eq"):
    for rep in range(3):
        ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 1), st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [b], L, 2), st)
        st.synchronize(); lib.cuda_synchronize_device(0)
        t0 = time.perf_counter()
        if op == "add": sks.add_assign(ca, cb, st)
        elif op == "sub": sks.sub_assign(ca, cb, st)
        else: out = sks.compare(ca, cb, op, st)
        st.synchronize()
        t1 = time.perf_counter()
        lib.cuda_synchronize_device(0)
        t2 = time.perf_counter()
    print(op, "stream-sync ms", (t1 - t0) * 1e3, "device-sync extra ms", (t2 - t1) * 1e3, "kernel", lib.hip_backend_last_pbs_kernel())
    if op in ("add", "sub"):
        print("  result ok:", recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [(a + b) % 2**64 if op == "add" else (a - b) % 2**64])
