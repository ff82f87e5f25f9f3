#!/usr/bin/env python3
"""Probe: repeated keyswitches of B LWEs with shuffled input / output indexes at the reference's test sizes
(2048 -> 742, base 2^3, 5 levels), every repetition compared with the oracle; per kernel choice and K-split.
   python tools/probes/ks_determinism.py <emu|hip> [B ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, os.pardir)))
from tests import oracle as orc  # noqa: E402
from tests.harness import use_backend  # noqa: E402
from tfhe_rs_amd import core_crypto_gpu as gpu  # noqa: E402

kind = sys.argv[1]
lib = use_backend(kind)
st = gpu.CudaStreams.new_single_gpu(0)
n_in, n_out, bl, lv = 2048, 742, 3, 5
rng = np.random.default_rng(5)
ksk = rng.integers(0, 1 << 64, size=n_in * lv * (n_out + 1), dtype=np.uint64)
d_ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(ksk, n_in, n_out, bl, lv, st)
REPS = 6
for B in [int(a) for a in sys.argv[2:]] or [148, 193, 244, 400]:
    cts = rng.integers(0, 1 << 64, size=(B, n_in + 1), dtype=np.uint64)
    ii = rng.permutation(B).astype(np.uint64)
    oi = rng.permutation(B).astype(np.uint64)
    want = np.zeros((B, n_out + 1), dtype=np.uint64)
    want[oi.astype(np.int64)] = orc.keyswitch_batch(cts[ii.astype(np.int64)], ksk, n_in, n_out, bl, lv)
    s_of_row = np.argsort(oi)          # output row -> sample position s
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, st)
    d_ii = gpu.CudaVec.from_cpu_async(ii, st)
    d_oi = gpu.CudaVec.from_cpu_async(oi, st)
    for label, kernel, kparts in (("default", 0, 8), ("kparts1", 0, 1), ("scalar", 1, 8), ("digits+gemm", 3, 8)):
        lib.hip_backend_set_keyswitch_kernel(kernel)
        lib.hip_backend_set_keyswitch_kparts(kparts)
        t = time.time()
        res = []
        for rep in range(REPS):
            d_out = gpu.CudaLweCiphertextList.new(n_out, B, st)
            gpu.cuda_keyswitch_lwe_ciphertext(d_ksk, d_in, d_out, d_ii, d_oi, False, st, use_gemm_ks=bool(rep & 1))
            o = d_out.to_lwe_ciphertext_list(st)
            bad_rows = np.flatnonzero(np.any(o != want, axis=1))
            tiles = sorted(set((s_of_row[bad_rows] // 32).tolist()))
            cols = np.flatnonzero(np.any(o != want, axis=0))
            res.append((len(bad_rows), tiles, (int(cols.min()), int(cols.max())) if len(cols) else None))
        print(f"B={B} {label} path={lib.hip_backend_last_keyswitch_path()} {time.time() - t:.2f}s:", res, flush=True)
    lib.hip_backend_set_keyswitch_kernel(0)
    lib.hip_backend_set_keyswitch_kparts(8)
