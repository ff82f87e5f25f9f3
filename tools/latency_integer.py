#!/usr/bin/env python3
"""Latency of ONE FheUint64 add / mul on one GPU, phase by phase (scratch, operation, cleanup), through the
integer FFI of the backend.  Random key material (timing is data independent)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tfhe_rs_amd import core_crypto_gpu as gpu, ffi, integer_gpu as igpu  # noqa: E402
from tests.common import C1, C4G4  # noqa: E402

lib = ffi.default_library()
rng = np.random.default_rng(3)
r64 = lambda n: rng.integers(0, 1 << 64, size=n, dtype=np.uint64)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "classic"
    p = {"classic": C1, "multibit_g4": C4G4}[which]
    st = gpu.CudaStreams([0])
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(r64(p.big_n * p.ks_level * (p.n + 1)), p.big_n, p.n,
                                                         p.ks_base_log, p.ks_level, st)
    k1 = p.k + 1
    if p.grouping:
        bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
            r64((p.n // p.grouping) * (1 << p.grouping) * p.pbs_level * k1 * k1 * p.N), p.n, p.k, p.N, p.pbs_base_log,
            p.pbs_level, p.grouping, st)
    else:
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(r64(p.n * p.pbs_level * k1 * k1 * p.N), p.n, p.k, p.N,
                                                             p.pbs_base_log, p.pbs_level, st, ms_noise_reduction=True)
    sks = igpu.CudaServerKey(ksk, bsk, 4, 4)
    L = 32
    blocks = r64(L * (p.big_n + 1)).reshape(1, L, -1)
    s, keep = sks._streams(st)
    ksks, bsks = sks._key_ptrs(st)
    for op in ("add", "mul"):
        for rep in range(3):
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks, st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks, st)
            cin, cout = sks._carry_blocks(ca, None, st), sks._carry_blocks(ca, None, st)
            st.synchronize()
            mem = C.c_void_p()
            t0 = time.perf_counter()
            lib.hip_integer_scratch_batch(1)
            if op == "add":
                lib.scratch_cuda_add_and_propagate_single_carry_64_inplace_async(
                    s, C.byref(mem), sks._bsk_params(), sks._ksk_params(), L, 4, 4, 0, True, sks._noise_reduction())
            else:
                lib.scratch_cuda_integer_mult_inplace_64_async(
                    s, C.byref(mem), False, False, 4, 4, sks._bsk_params(), sks._ksk_params(), L, True,
                    sks._noise_reduction())
            st.synchronize()
            t1 = time.perf_counter()
            if op == "add":
                lib.cuda_add_and_propagate_single_carry_64_inplace_async(
                    s, C.byref(ca._ffi()), C.byref(cb._ffi()), C.byref(cout._ffi()), C.byref(cin._ffi()), mem, bsks,
                    ksks, 0, 0)
            else:
                lib.cuda_integer_mult_inplace_64_async(s, C.byref(ca._ffi()), False, C.byref(cb._ffi()), False, bsks, ksks,
                                                       mem, p.N, L)
            st.synchronize()
            t2 = time.perf_counter()
            if op == "add":
                lib.cleanup_cuda_add_and_propagate_single_carry_64_inplace(s, C.byref(mem))
            else:
                lib.cleanup_cuda_integer_mult_inplace_64(s, C.byref(mem))
            t3 = time.perf_counter()
        print(json.dumps({"op": f"FheUint64 {op}", "params": p.name, "scratch_ms": (t1 - t0) * 1e3,
                          "operation_ms": (t2 - t1) * 1e3, "cleanup_ms": (t3 - t2) * 1e3,
                          "note": "third repetition; one ciphertext pair, one stream"}))
    # the second tranche of the radix layer (round 6), each through its host wrapper: scratch + operation + cleanup
    cond = igpu.CudaUnsignedRadixCiphertext.from_blocks(r64(p.big_n + 1).reshape(1, 1, -1), st)
    blocks2 = r64(L * (p.big_n + 1)).reshape(1, L, -1)
    calls = {"sub": lambda a, b: sks.sub_assign(a, b, st), "bitand": lambda a, b: sks.bitop_assign(a, b, "and", st),
             "eq": lambda a, b: sks.compare(a, b, "eq", st), "gt": lambda a, b: sks.compare(a, b, "gt", st),
             "max": lambda a, b: sks.compare(a, b, "max", st), "if_then_else": lambda a, b: sks.if_then_else(cond, a, b, st)}
    for op, fn in calls.items():
        for rep in range(3):
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks, st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks2, st)  # not the same words: a - a has zero masks, which the
            st.synchronize()                                                # classic loop skips (a_hat == 0): 1.6 ms instead of 23
            t0 = time.perf_counter()
            fn(ca, cb)
            st.synchronize()
            t1 = time.perf_counter()
        print(json.dumps({"op": f"FheUint64 {op}", "params": p.name, "total_ms": (t1 - t0) * 1e3,
                          "note": "third repetition; scratch + operation + cleanup; one ciphertext pair, one stream"}))
    if "--throughput" in sys.argv:
        # the same operations over a batch of independent integers (timing only: the key material is random; the
        # decrypt-checked form of this measurement is tools/bench_integer.py)
        for op, B in (("add", 1024), ("mul", 128), ("sub", 1024)):
            big = r64(B * L * (p.big_n + 1)).reshape(B, L, -1)
            big2 = r64(B * L * (p.big_n + 1)).reshape(B, L, -1) if op == "sub" else big  # a - a has zero masks (see above)
            times = []
            for rep in range(2):  # the first repetition carries the process's one-time costs (code objects, the arena's first growth)
                ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(big, st)
                cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(big2, st)
                st.synchronize()
                t0 = time.perf_counter()
                if op == "add":
                    sks.add_assign(ca, cb, st)
                    pbs = int(lib.hip_integer_propagate_pbs_count(L))
                elif op == "sub":
                    sks.sub_assign(ca, cb, st)
                    pbs = int(lib.hip_integer_propagate_pbs_count(L))
                else:
                    pbs = int(sks.mul_assign(ca, cb, st, return_pbs_count=True))
                st.synchronize()
                times.append(time.perf_counter() - t0)
                del ca, cb
            dt = times[1]
            print(json.dumps({"op": f"FheUint64 {op}", "params": p.name, "batch": B, "seconds": dt, "ops_per_s": B / dt,
                              "pbs_per_op": pbs, "ks_pbs_per_s": B * pbs / dt, "first_repetition_seconds": times[0],
                              "note": "second repetition; scratch, every round and cleanup inside the timed region; random key "
                                      "material"}))


if __name__ == "__main__":
    main()
