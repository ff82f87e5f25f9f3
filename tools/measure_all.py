#!/usr/bin/env python3
"""Secondary measurements on one MI355X (everything except the headline bench.py line):
keyswitch, KS->PBS pipeline, NTT engine, multi-bit PBS, N=1024/k=2 datapoint, batch sweep.
Throughput-only runs use uniform-random key material like the reference's own benches
(tfhe-benchmark/benches/core_crypto/pbs_bench.rs:45-58): timing is data independent.
Prints one JSON object per line."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import tfhe_rs_amd  # noqa: E402,F401
from tfhe_rs_amd import core_crypto_gpu as gpu  # noqa: E402
from tfhe_rs_amd import ffi  # noqa: E402
from tests.common import C1, C1P, C33, C4, C4G2_L1, C4G3, C4G3_L1, C4G4  # noqa: E402

lib = ffi.default_library()
streams = gpu.CudaStreams.new_single_gpu(0)
S, G = streams.ptr[0], 0
rng = np.random.default_rng(7)


def rand_u64(n):
    return rng.integers(0, 1 << 64, size=n, dtype=np.uint64)


def timed(fn, steps=3, warmup=1):
    if any(a in sys.argv for a in ("mb1", "mb4one", "lat1", "ntt1", "n1024x", "wave1", "ntt4096", "nttsplit4096", "n1024x4096", "ks1", "mblat151")):
        warmup = 0   # exactly one launch: the PMC passes of tools/pmc_record.py
    for _ in range(warmup):
        fn()
    lib.cuda_synchronize_device(G)
    e0, e1 = lib.hip_event_create(), lib.hip_event_create()
    lib.hip_event_record(e0, S)
    for _ in range(steps):
        fn()
    lib.hip_event_record(e1, S)
    ms = lib.hip_event_elapsed_ms(e0, e1) / steps
    lib.hip_event_destroy(e0)
    lib.hip_event_destroy(e1)
    return ms


def emit(**kw):
    print(json.dumps(kw), flush=True)


def pbs_case(p, B, engine="fft64", kernel=0, steps=3):
    k1 = p.k + 1
    if p.grouping:
        bsk_h = rand_u64((p.n // p.grouping) * (1 << p.grouping) * p.pbs_level * k1 * k1 * p.N)
        bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
            bsk_h, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping, streams)
    else:
        bsk_h = rand_u64(p.n * p.pbs_level * k1 * k1 * p.N)
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(bsk_h, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                                             streams, ms_noise_reduction=bool(p.ms_type),
                                                             engine=engine)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(rand_u64(B * (p.n + 1)).reshape(B, -1), streams)
    d_out = gpu.CudaLweCiphertextList.new(p.k * p.N, B, streams)
    d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(rand_u64(k1 * p.N), p.k, p.N, streams)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), streams)
    lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), streams)
    buf = C.c_void_p()
    lib.hip_backend_set_fft_kernel(kernel)
    if p.grouping:
        lib.scratch_cuda_multi_bit_programmable_bootstrap_64_async(S, G, C.byref(buf), p.k, p.N, p.pbs_level, B, True)

        def run():
            lib.cuda_multi_bit_programmable_bootstrap_64_async(
                S, G, d_out.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr, d_in.d_vec.ptr, idx.ptr, bsk.d_vec.ptr,
                buf, p.n, p.k, p.N, p.grouping, p.pbs_base_log, p.pbs_level, B, 1, 0)
    else:
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), p.n, p.k, p.N, p.pbs_level, B, True,
                                                         p.ms_type)
        launch = (lib.cuda_programmable_bootstrap_64_async if engine == "fft64"
                  else lib.hip_programmable_bootstrap_ntt64_split_async if bsk.engine_impl == "ntt64_split"
                  else lib.hip_programmable_bootstrap_ntt64_async)

        def run():
            launch(S, G, d_out.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr, d_in.d_vec.ptr, idx.ptr, bsk.d_vec.ptr,
                   buf, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, B, 1, 0)
    ms = timed(run, steps=steps)
    kid = lib.hip_backend_last_pbs_kernel()
    if p.grouping:
        lib.cleanup_cuda_multi_bit_programmable_bootstrap_64(S, G, C.byref(buf))
    else:
        lib.cleanup_cuda_programmable_bootstrap_64(S, G, C.byref(buf))
    lib.hip_backend_set_fft_kernel(0)
    emit(what="pbs", params=p.name, engine=engine if not p.grouping else "multi_bit_fft64", kernel_id=kid, requested=kernel, batch=B,
         ms=ms, pbs_per_s=B / ms * 1e3)
    return ms


def ks_case(p, B, steps=5):
    n_in, n_out = p.k * p.N, p.n
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(rand_u64(n_in * p.ks_level * (n_out + 1)), n_in, n_out,
                                                         p.ks_base_log, p.ks_level, streams)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(rand_u64(B * (n_in + 1)).reshape(B, -1), streams)
    d_out = gpu.CudaLweCiphertextList.new(n_out, B, streams)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), streams)
    ms = timed(lambda: gpu.cuda_keyswitch_lwe_ciphertext(ksk, d_in, d_out, idx, idx, True, streams), steps=steps)
    ksk_bytes = n_in * p.ks_level * (n_out + 1) * 8
    emit(what="keyswitch", params=p.name, batch=B, ms=ms, ks_per_s=B / ms * 1e3,
         algorithmic_GBps=(ksk_bytes + (n_in + 1 + n_out + 1) * 8) * B / ms / 1e6,
         ksk_once_GBps=(ksk_bytes + B * (n_in + n_out + 2) * 8) / ms / 1e6)
    return ms


def ks32_case(p, B, steps=5):
    """64->32 keyswitch (u32 key) on the dimensions of `p`, matrix-core path and scalar kernel."""
    n_in, n_out = p.k * p.N, p.n
    key = rng.integers(0, 1 << 32, size=n_in * p.ks_level * (n_out + 1), dtype=np.uint64).astype(np.uint32)
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(key, n_in, n_out, p.ks_base_log, p.ks_level, streams)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(rand_u64(B * (n_in + 1)).reshape(B, -1), streams)
    d_out = gpu.CudaLweCiphertextList.new(n_out, B, streams, dtype=np.uint32)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), streams)
    for choice, name in ((0, "matrix cores"), (1, "scalar")):
        lib.hip_backend_set_keyswitch_kernel(choice)
        ms = timed(lambda: gpu.cuda_keyswitch_lwe_ciphertext(ksk, d_in, d_out, idx, idx, True, streams), steps=steps)
        emit(what="keyswitch 64->32", path=name, params=p.name, batch=B, ms=ms, ks_per_s=B / ms * 1e3)
    lib.hip_backend_set_keyswitch_kernel(0)


def chain_case(p, B, rounds=6):
    """KS -> PBS rounds in which every round reads what the previous one wrote (hip_keyswitch_programmable_bootstrap_chain):
    plain (digit pass in every keyswitch) against the fused form (the bootstrap's sample extraction emits the next
    keyswitch's operands)."""
    k1 = p.k + 1
    bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(rand_u64(p.n * p.pbs_level * k1 * k1 * p.N), p.n, p.k, p.N,
                                                         p.pbs_base_log, p.pbs_level, streams, ms_noise_reduction=bool(p.ms_type))
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(rand_u64(p.big_n * p.ks_level * (p.n + 1)), p.big_n, p.n,
                                                         p.ks_base_log, p.ks_level, streams)
    d_a = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(rand_u64(B * (p.big_n + 1)).reshape(B, -1), streams)
    d_b = gpu.CudaLweCiphertextList.new(p.big_n, B, streams)
    d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(rand_u64(k1 * p.N), p.k, p.N, streams)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), streams)
    lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), streams)
    buf = C.c_void_p()
    lib.hip_scratch_keyswitch_programmable_bootstrap_64_async(S, G, C.byref(buf), p.n, p.k, p.N, p.pbs_level, B, True, p.ms_type)
    for name, first, later in (("plain", 0, 0), ("sample extraction emits the next keyswitch's digits", 1, 3)):
        def run():
            src, dst = d_a, d_b
            for r in range(rounds):
                lib.hip_keyswitch_programmable_bootstrap_chain_64_async(
                    S, G, dst.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr, src.d_vec.ptr, idx.ptr, ksk.d_vec.ptr,
                    bsk.d_vec.ptr, buf, p.n, p.k, p.N, p.ks_base_log, p.ks_level, p.pbs_base_log, p.pbs_level, B, 1, 0,
                    first if r == 0 else later)
                src, dst = dst, src
        ms = timed(run, steps=2) / rounds
        emit(what="KS -> PBS round in a chain", form=name, params=p.name, batch=B, ms_per_round=ms,
             ks_pbs_per_s=B / ms * 1e3, last_keyswitch_path=int(lib.hip_backend_last_keyswitch_path()))
    lib.cleanup_cuda_programmable_bootstrap_64(S, G, C.byref(buf))


if __name__ == "__main__":
    if os.environ.get("TFHE_KS_CHOICE"):
        lib.hip_backend_set_keyswitch_kernel(int(os.environ["TFHE_KS_CHOICE"]))
    which = sys.argv[1:] or ["ks", "wave", "generic", "ntt", "mb", "n1024", "sweep"]
    if "ks" in which:
        ks_ms = ks_case(C1, 4096)
    if "kscross" in which:  # one-launch kernel with K shared by workgroups (choice 2) against digit pass + GEMM (0)
        for B in (128, 256, 512, 1024, 2048, 4096):
            for choice in (2, 0):
                lib.hip_backend_set_keyswitch_kernel(choice)
                ks_case(C1, B, steps=10)
        lib.hip_backend_set_keyswitch_kernel(0)
    if "kssmall" in which:  # the rounds of one radix operation: 7 .. 32 blocks
        for prm in (C1, C4G4):
            for parts in (1, 8):
                lib.hip_backend_set_keyswitch_kparts(parts)
                for B in (8, 32, 64, 128):
                    ms = ks_case(prm, B, steps=20)
            lib.hip_backend_set_keyswitch_kparts(8)
    if "chain" in which:
        chain_case(C1, 4096)
    if "ks32" in which:
        ks32_case(C1, 4096)
        ks32_case(C1P, 4096)
    if "mb4" in which:  # the reference's GPU default set (grouping factor 4, one level)
        pbs_case(C4G4, 4096, steps=2)
        pbs_case(C4G4, 1, steps=3)
    if "mb3gpu" in which:  # the reference's GPU g = 3 set of the 2_2 precision (n = 879, base_log 14, two levels)
        pbs_case(C4G3, 4096, steps=2)
    if "mbgauss" in which:  # the one-level g = 3 / g = 2 GPU sets of the gaussian families (n = 813 / 820)
        pbs_case(C4G3_L1, 4096, steps=2)
        pbs_case(C4G2_L1, 4096, steps=2)
    if "mbcross" in which:  # where the multi-bit latency path (5) and the throughput kernel (2) cross
        for p in (C4G4, C4):
            for B in (128, 192, 256, 384, 512):
                for kern in (5, 2):
                    pbs_case(p, B, kernel=kern, steps=3)
    if "mbmid" in which:  # mid-size batches (the rounds of ONE multiplication): key loads shared by all waves / by quads (8) / not (7)
        for B in (260, 320, 384, 448, 512, 768, 1024, 2048):
            for kern in (2, 8, 7):
                pbs_case(C4G4, B, kernel=kern, steps=3)
    if "mblat151" in which:  # one launch of the multi-bit latency path (PMC passes over the keybundle kernel)
        pbs_case(C4G4, 151, kernel=5, steps=1)
    if "mblat2" in which:  # the multi-bit latency path at the round sizes of one addition / multiplication
        for B in (16, 20, 32, 64, 82, 151, 256):
            pbs_case(C4G4, B, kernel=5, steps=5)
    if "mblat" in which:  # multi-bit latency path: products on the latency kernel (5) or the generic kernels (6)
        for p in (C4G4, C4):
            for kern in (5, 6):
                for B in (1, 16, 128):
                    pbs_case(p, B, kernel=kern, steps=5)
    if "n8192" in which:  # the 3_3 set: generic kernel with the accumulator in device memory
        ks_case(C33, 1024)
        pbs_case(C33, 1024, steps=2)
    if "ks1024" in which:
        ks_case(C1P, 4096)
    if "wave" in which:
        w_ms = pbs_case(C1, 4096, kernel=2, steps=5)
        if "ks" in which:
            emit(what="ks+pbs pipeline (sum of the two launches)", params=C1.name, batch=4096, ms=ks_ms + w_ms,
                 ks_pbs_per_s=4096 / (ks_ms + w_ms) * 1e3)
    if "generic" in which:
        pbs_case(C1, 4096, kernel=1)
    if "ntt" in which:
        pbs_case(C1, 4096, engine="ntt64", steps=2)       # integer-Goldilocks form
    if "ntt_split" in which:
        pbs_case(C1, 4096, engine="ntt64_split", steps=2)  # split-key f64 form (throughput kernel machinery)
    if "mb" in which:
        pbs_case(C4, 4096, steps=2)
    if "wave1" in which:   # one launch each, no warm-up (PMC passes, tools/pmc_record.py)
        pbs_case(C1, 4096, kernel=2, steps=1)
    if "nttsplit4096" in which:
        pbs_case(C1, 4096, engine="ntt64_split", steps=1)
    if "ntt4096" in which:
        pbs_case(C1, 4096, engine="ntt64", steps=1)
    if "n1024x4096" in which:
        pbs_case(C1P, 4096, steps=1)
    if "ks1" in which:
        ks_case(C1, 4096, steps=1)
    if "lat1" in which:  # one launch of the latency kernel, no warm-up (PMC passes)
        pbs_case(C1, 256, kernel=3, steps=1)
    if "n1024x" in which:  # one launch, no warm-up (PMC passes)
        pbs_case(C1P, 1024, steps=1)
    if "ntt1" in which:  # one launch, no warm-up (PMC passes)
        pbs_case(C1, 1024, engine="ntt64", steps=1)
    if "mb1" in which:   # one launch, no warm-up (PMC passes)
        pbs_case(C4, 4096, steps=1)
    if "mb4one" in which:   # one launch, no warm-up (PMC passes)
        pbs_case(C4G4, 4096, steps=1)
    if "n1024" in which:
        pbs_case(C1P, 4096, steps=3)
    if "latency" in which:
        for kern in (3, 4, 2):
            for B in (1, 128, 256, 512):
                pbs_case(C1, B, kernel=kern, steps=5)
    if "sweep" in which:
        for B in (1, 4, 64, 256, 1024, 2048, 8192):
            pbs_case(C1, B, kernel=2, steps=3)
