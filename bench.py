#!/usr/bin/env python3
"""bench.py — PBS/s on PARAM_MESSAGE_2_CARRY_2 (n=918, k=1, N=2048, l=1, base_log=23, centered
mean modulus switch), batch of 4096 independent LWEs per GPU, f64 FFT external product.

One "step" = one batched PBS launch (modulus switch + 918 CMUXes + sample extract) over the
4096 LWEs resident in HBM on this rank's GPU, through the C ABI
(scratch_/cuda_/cleanup_ programmable_bootstrap_64).  Multi-GPU: one process per GPU, each
rank owns its own 4096-LWE shard and a replica of the key; no data-path collective (weak
scaling); the barrier/max-over-ranks timing uses torch.distributed (RCCL).

Prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.  At N = 1 the line
also carries `extra`: the single-PBS latency, the N=1024/k=2 datapoint, BASELINE.json's configs 3 (NTT engine) and
4 (multi-bit g = 3, plus the reference's GPU default g = 4), each at batch 4096 with its oracle-parity bit, and
config 5 on one GPU (FheUint64 add / mul through the radix layer, results decrypted) — none of them part of `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 4096
# algorithmic bytes per PBS, streaming model with no inter-LWE key reuse:
# BSK n*(k+1)^2*l*N*8 + LWE_in (n+1)*8 + LUT (k+1)*N*8 + LWE_out (kN+1)*8
ALGO_BYTES_PER_PBS = 918 * 4 * 1 * 2048 * 8 + 919 * 8 + 2 * 2048 * 8 + 2049 * 8  # = 60,218,560
# algorithmic f64 flop per PBS = n x 270,336 (2 forward + 2 inverse 1024-point transforms at
# 5 (N/2) log2(N/2), the (k+1)^2 l (N/2) 8-flop MAC and the twists, per external product)
ALGO_FLOP_PER_PBS = 918 * 270336  # = 2.48e8
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md / datasheet: FP64 vector = half of the 157.3 TF FP32 vector rate


def bytes_per_pbs(p):
    """streaming-model bytes of one PBS of parameter set p (multi-bit: the standard-domain key, 2^g GGSW per group)"""
    k1 = p.k + 1
    key = p.n * k1 * k1 * p.pbs_level * p.N * 8
    if p.grouping:
        key = (p.n // p.grouping) * (1 << p.grouping) * k1 * k1 * p.pbs_level * p.N * 8
    return key + (p.n + 1) * 8 + k1 * p.N * 8 + (p.k * p.N + 1) * 8


def pmc_record(target):
    """Counter record of one throughput kernel ("fft", "ntt", "mb_g3"), measured by tools/pmc_record.py on the SAME
    kernel sources this tree holds (profiles/pmc_latest.json carries the build id of what it profiled; a record
    of another build is refused).  Returns (hbm_bytes_per_launch or None, provenance string)."""
    from tools.build_id import source_build_id
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        m = json.load(open(path))
    except Exception as e:
        return None, f"no PMC record ({e.__class__.__name__})"
    mine = source_build_id()
    if m.get("build_id") != mine:
        return None, f"PMC record is of build {m.get('build_id')}, this tree is {mine}: refused as stale"
    k = m.get("kernels", {}).get(target)
    if not k or "hbm_bytes_per_launch" not in k:
        return None, f"PMC record of build {mine} has no '{target}' entry"
    return k["hbm_bytes_per_launch"], (f"profiles/pmc_latest.json, build {mine}, kernel {k.get('kernel')}: "
                                       f"2 x FETCH_SIZE + WRITE_SIZE per launch, L2 hit rate {k.get('l2_hit_rate', 0):.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH, help="LWEs per GPU per step (default: the metric's 4096)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic LDS kernel, 2 throughput kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3/4 and N=1024 datapoints under `extra`")
    ap.add_argument("--no-verify", action="store_true", help="skip the decrypt check (timing-ablation builds only)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="PBS count of the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("TFHE_BENCH_FORCE_DIST"):  # the env knob exercises this path at N = 1
        # import torch first so its HIP runtime is the process's single libamdhip64 instance
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import numpy as np
    import tfhe_rs_amd  # noqa: F401
    from tfhe_rs_amd import core_crypto_gpu as gpu
    from tfhe_rs_amd import ffi
    from tfhe_rs_amd.multi_gpu import shard_range
    from tests import oracle as orc          # checker + cpu_baseline leg only
    from tests.common import C1, make_keys, encrypt_small, decrypt_big

    lib = ffi.default_library()
    assert lib.cuda_is_available() == 1, "no GPU visible: the backend has no CPU fallback"
    lib.hip_backend_set_fft_kernel(args.kernel)
    p = C1
    B = args.batch
    gpu_index = local_rank

    # ---- synthetic inputs: real keys (seeded), fresh encryptions of i mod 16, LUT f(x) = x
    keys = make_keys(p, with_ksk=False)
    streams = gpu.CudaStreams([gpu_index])
    bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                                         streams, ms_noise_reduction=True, engine="fft64")
    lo, hi = shard_range(B * world, rank, world)          # this rank's shard of the global batch
    msgs = [(lo + i) % p.plaintext_modulus for i in range(B)]
    cts = encrypt_small(p, keys, msgs, seed=100 + rank)   # B distinct fresh encryptions
    rng = np.random.default_rng(1234 + rank)
    f = lambda x: x
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, f)

    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, streams)
    d_out = gpu.CudaLweCiphertextList.new(p.k * p.N, B, streams)
    d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, p.k, p.N, streams)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), streams)
    lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), streams)

    s, g = streams.ptr[0], gpu_index
    buf = C.c_void_p()
    lib.scratch_cuda_programmable_bootstrap_64_async(s, g, C.byref(buf), p.n, p.k, p.N, p.pbs_level, B, True, 1)

    def step():
        lib.cuda_programmable_bootstrap_64_async(s, g, d_out.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr,
                                                 d_in.d_vec.ptr, idx.ptr, bsk.d_vec.ptr, buf, p.n, p.k, p.N,
                                                 p.pbs_base_log, p.pbs_level, B, 1, 0)

    def sync_all():
        lib.cuda_synchronize_device(g)
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    ev = [(lib.hip_event_create(), lib.hip_event_create()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:
        lib.hip_event_record(e0, s)
        step()
        lib.hip_event_record(e1, s)
    lib.cuda_synchronize_device(g)
    if dist is not None:
        import torch
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = [lib.hip_event_elapsed_ms(e0, e1) for e0, e1 in ev]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_id = lib.hip_backend_last_pbs_kernel()

    # ---- single-PBS latency (outside the timed region; the reference publishes this figure)
    latency_ms = None
    if rank == 0 and args.kernel == 0:
        d_o1 = gpu.CudaLweCiphertextList.new(p.k * p.N, 1, streams)

        def one():
            lib.cuda_programmable_bootstrap_64_async(s, g, d_o1.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr,
                                                     d_in.d_vec.ptr, idx.ptr, bsk.d_vec.ptr, buf, p.n, p.k, p.N,
                                                     p.pbs_base_log, p.pbs_level, 1, 1, 0)
        one()
        lib.cuda_synchronize_device(g)
        e0, e1 = lib.hip_event_create(), lib.hip_event_create()
        lib.hip_event_record(e0, s)
        for _ in range(5):
            one()
        lib.hip_event_record(e1, s)
        lib.cuda_synchronize_device(g)
        latency_ms = lib.hip_event_elapsed_ms(e0, e1) / 5
        latency_kernel = lib.hip_backend_last_pbs_kernel()
        assert decrypt_big(p, keys, d_o1.to_lwe_ciphertext_list(streams)[0]) == f(msgs[0])

    # ---- validity (outside the timed region): every output of this rank decrypts to f(m)
    out = d_out.to_lwe_ciphertext_list(streams)
    check = rng.choice(B, size=min(B, 256), replace=False)
    bad = [int(i) for i in check if decrypt_big(p, keys, out[i]) != f(msgs[i])]
    assert args.no_verify or not bad, f"PBS outputs failed to decrypt at rows {bad[:8]}"
    lib.cleanup_cuda_programmable_bootstrap_64(s, g, C.byref(buf))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_pbs = B * world * args.steps
    value = total_pbs / elapsed
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
    achieved = ALGO_BYTES_PER_PBS * B / avg_kernel_s / 1e9
    tflops = ALGO_FLOP_PER_PBS * B / avg_kernel_s / 1e12
    traffic, traffic_src = pmc_record("fft")
    # What bounds the kernel: the 60 MB key is shared by all workgroups through L2 / Infinity Cache (measured
    # HBM traffic is ~1 % of the streaming model's bytes), so the binding roof is the FP64 vector pipe.  The
    # streaming-model HBM fraction that SURVEY §8(d) / north_star name is kept under its own name next to it.
    roofline = {"bound": "fp64_valu", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / FP64_PEAK_TFLOPS, "frac_fp64": tflops / FP64_PEAK_TFLOPS,
                "algorithmic_flop_per_launch": ALGO_FLOP_PER_PBS * B,
                "hbm_streaming_model": {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": achieved / HBM_PEAK_GBPS,
                                        "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PBS * B,
                                        "note": "key re-read per LWE (no inter-LWE reuse): a model, not a bound — "
                                                "compare `traffic`"},
                "frac_hbm_streaming_model": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_src,
                "hbm_measured_frac": (traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "note": "achieved = algorithmic f64 flop (SURVEY §8d, 2.48e8 per PBS) / HIP-event launch time; "
                        "peak = FP64 vector 78.6 TFLOP/s; traffic = HBM bytes per launch from PMC passes of this "
                        "exact kernel build (null when the committed record is of another build)"}
    result = {
        "metric": "PBS/sec (shortint PARAM_MESSAGE_2_CARRY_2, classic PBS, f64 FFT external product)",
        "value": value, "unit": "PBS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64 (u64 torus)", "data": "synthetic",
        "config": {"workload": "batch of 4096 independent PBS per GPU, PARAM_MESSAGE_2_CARRY_2 "
                               "(n=918,k=1,N=2048,l=1,base_log=23, centered-mean MS), f64 FFT, inputs resident in HBM",
                   "batch_per_gpu": B, "lwe_dimension": p.n, "glwe_dimension": p.k, "polynomial_size": p.N,
                   "pbs_kernel": {1: "generic_lds", 2: "wave_throughput"}.get(kernel_id, str(kernel_id)),
                   "parallelism": f"batch-sharded x{world}, key replicas, no collective"},
        "roofline": roofline,
    }
    if latency_ms is not None:
        result["extra"] = {"single_pbs_latency_ms": latency_ms,
                           "single_pbs_kernel": {7: "block_latency", 2: "wave_throughput"}.get(latency_kernel,
                                                                                              str(latency_kernel)),
                           "note": "one PBS, batch 1, same key; not part of `value` (the reference publishes "
                                   "4.21 ms on an H100, BASELINE.md)"}
    if world == 1 and args.kernel == 0 and not args.no_extra:
        # the "N=1024" wording of BASELINE.json: the production set with polynomial size 1024 (k = 2, n = 885) on the
        # same GPU, uniform-random key and inputs like the reference's own benches; reported next to `value`, never in it
        from tests.common import C1P
        q = C1P
        r2 = np.random.default_rng(7)
        bsk2 = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(
            r2.integers(0, 1 << 64, size=q.n * (q.k + 1) ** 2 * q.pbs_level * q.N, dtype=np.uint64), q.n, q.k, q.N,
            q.pbs_base_log, q.pbs_level, streams)
        d_in2 = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(
            r2.integers(0, 1 << 64, size=(B, q.n + 1), dtype=np.uint64), streams)
        d_out2 = gpu.CudaLweCiphertextList.new(q.k * q.N, B, streams)
        d_lut2 = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(
            r2.integers(0, 1 << 64, size=(q.k + 1) * q.N, dtype=np.uint64), q.k, q.N, streams)
        buf2 = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(s, g, C.byref(buf2), q.n, q.k, q.N, q.pbs_level, B, True, 0)

        def step2():
            lib.cuda_programmable_bootstrap_64_async(s, g, d_out2.d_vec.ptr, idx.ptr, d_lut2.d_vec.ptr, lidx.ptr,
                                                     d_in2.d_vec.ptr, idx.ptr, bsk2.d_vec.ptr, buf2, q.n, q.k, q.N,
                                                     q.pbs_base_log, q.pbs_level, B, 1, 0)
        step2()
        lib.cuda_synchronize_device(g)
        e0, e1 = lib.hip_event_create(), lib.hip_event_create()
        lib.hip_event_record(e0, s)
        for _ in range(3):
            step2()
        lib.hip_event_record(e1, s)
        lib.cuda_synchronize_device(g)
        ms2 = lib.hip_event_elapsed_ms(e0, e1) / 3
        lib.cleanup_cuda_programmable_bootstrap_64(s, g, C.byref(buf2))
        bytes2 = q.n * (q.k + 1) ** 2 * q.pbs_level * q.N * 8 + (q.n + 1) * 8 + (q.k + 1) * q.N * 8 + (q.k * q.N + 1) * 8
        result.setdefault("extra", {})["n1024_datapoint"] = {
            "params": q.name + " (n=885, k=2, N=1024, l=1)", "batch": B, "ms_per_launch": ms2,
            "pbs_per_s": B / ms2 * 1e3, "frac_hbm_streaming_model": B / ms2 * 1e3 * bytes2 / (HBM_PEAK_GBPS * 1e9),
            "frac_fp64": B / ms2 * 1e3 * 1.8e8 / (FP64_PEAK_TFLOPS * 1e12),
            "pbs_kernel_id": int(lib.hip_backend_last_pbs_kernel())}
    if world == 1 and args.kernel == 0 and not args.no_extra:
        # BASELINE.json configs 3 and 4 next to the headline (never part of `value`): batch 4096 on this GPU, HIP-event
        # time over 3 launches, and the GPU's output words compared with the CPU oracle on the first 64 LWEs.
        from tests.common import C4
        PAR = 64

        def datapoint(q, run_gpu, steps=3):
            run_gpu()
            lib.cuda_synchronize_device(g)
            e0, e1 = lib.hip_event_create(), lib.hip_event_create()
            lib.hip_event_record(e0, s)
            for _ in range(steps):
                run_gpu()
            lib.hip_event_record(e1, s)
            lib.cuda_synchronize_device(g)
            ms = lib.hip_event_elapsed_ms(e0, e1) / steps
            rate = B / ms * 1e3
            return {"params": q.name, "batch": B, "ms_per_launch": ms, "pbs_per_s": rate,
                    "frac_hbm_streaming_model": rate * bytes_per_pbs(q) / (HBM_PEAK_GBPS * 1e9),
                    "pbs_kernel_id": int(lib.hip_backend_last_pbs_kernel())}

        # ---- config 3: 64-bit prime NTT engine (tfhe-ntt semantics), same key, same ciphertexts as the headline
        bsk_n = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                                               streams, ms_noise_reduction=True, engine="ntt64")
        d_out3 = gpu.CudaLweCiphertextList.new(p.k * p.N, B, streams)
        buf3 = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(s, g, C.byref(buf3), p.n, p.k, p.N, p.pbs_level, B, True, 1)
        ntt_launch = (lib.hip_programmable_bootstrap_ntt64_crt_async if bsk_n.engine_impl == "ntt64_crt"
                      else lib.hip_programmable_bootstrap_ntt64_async)
        dp = datapoint(p, lambda: ntt_launch(
            s, g, d_out3.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr, d_in.d_vec.ptr, idx.ptr, bsk_n.d_vec.ptr, buf3,
            p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, B, 1, 0), steps=2)
        out3 = d_out3.to_lwe_ciphertext_list(streams)
        lib.cleanup_cuda_programmable_bootstrap_64(s, g, C.byref(buf3))
        t0 = time.perf_counter()
        ref3 = orc.pbs_batch(orc.ENGINE_NTT, cts[:PAR], lut, orc.convert_bsk_ntt(keys.bsk, p.n, p.k, p.N, p.pbs_level),
                             p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, 1)
        traffic3, src3 = pmc_record("ntt")
        dp.update({"engine": "Goldilocks NTT (p = 2^64 - 2^32 + 1), exact integer arithmetic", "bound": "int64 VALU",
                   "mulmod_per_s": dp["pbs_per_s"] * 5.2e7, "hbm_traffic_bytes_per_launch": traffic3,
                   "traffic_source": src3, "gpu_matches_cpu_bits": bool(np.array_equal(ref3, out3[:PAR])),
                   "parity_sample": f"first {PAR} LWEs, all 2049 words, vs the C oracle's NTT engine "
                                    f"({time.perf_counter() - t0:.1f} s CPU)",
                   "decrypts": all(decrypt_big(p, keys, out3[i]) == f(msgs[i]) for i in range(PAR))})
        result.setdefault("extra", {})["ntt"] = dp
        del bsk_n, d_out3

        # ---- config 4: multi-bit PBS, grouping factor 3; and the reference's GPU default multi-bit set (g = 4).
        # Uniform-random key and inputs like the reference's benches: bit parity with the oracle does not need a
        # valid key, and a real 320 MB key takes minutes to encrypt.
        from tests.common import C4G4
        def multibit_flop(q):
            """f64 flop per multi-bit PBS with the keybundle combined in the Fourier domain: per group l (k+1) forward and
            (k+1) inverse transforms at 5 n log2 n, (2^g - 1) l (k+1)^2 n complex multiply-adds for the combine and
            l (k+1)^2 n for the products at 8 flop each; n / g groups.  (SURVEY §8(d) quotes 3.7e8 for g = 3 with the
            keybundle built in the integer domain and transformed: that formulation is no longer what runs.)"""
            nn, k1 = q.N // 2, q.k + 1
            tr = 5 * nn * (nn.bit_length() - 1)
            per_group = (q.pbs_level * k1 + k1) * tr + ((1 << q.grouping) - 1 + 1) * q.pbs_level * k1 * k1 * nn * 8
            return per_group * (q.n // q.grouping)

        for tag, q in (("multibit_g3", C4), ("multibit_g4", C4G4)):
            flop = multibit_flop(q)
            r4 = np.random.default_rng(11)
            bsk4_h = r4.integers(0, 1 << 64, size=(q.n // q.grouping) * (1 << q.grouping) * q.pbs_level * 4 * q.N,
                                 dtype=np.uint64)
            cts4 = r4.integers(0, 1 << 64, size=(B, q.n + 1), dtype=np.uint64)
            lut4 = r4.integers(0, 1 << 64, size=2 * q.N, dtype=np.uint64)
            bsk4 = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
                bsk4_h, q.n, q.k, q.N, q.pbs_base_log, q.pbs_level, q.grouping, streams)
            d_in4 = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts4, streams)
            d_out4 = gpu.CudaLweCiphertextList.new(q.k * q.N, B, streams)
            d_lut4 = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut4, q.k, q.N, streams)
            buf4 = C.c_void_p()
            lib.scratch_cuda_multi_bit_programmable_bootstrap_64_async(s, g, C.byref(buf4), q.k, q.N, q.pbs_level, B, True)
            dp = datapoint(q, lambda: lib.cuda_multi_bit_programmable_bootstrap_64_async(
                s, g, d_out4.d_vec.ptr, idx.ptr, d_lut4.d_vec.ptr, lidx.ptr, d_in4.d_vec.ptr, idx.ptr, bsk4.d_vec.ptr,
                buf4, q.n, q.k, q.N, q.grouping, q.pbs_base_log, q.pbs_level, B, 1, 0), steps=2)
            out4 = d_out4.to_lwe_ciphertext_list(streams)
            lib.cleanup_cuda_multi_bit_programmable_bootstrap_64(s, g, C.byref(buf4))
            t0 = time.perf_counter()
            ref4 = orc.pbs_multi_bit(orc.ENGINE_FFT, cts4[:PAR], lut4, bsk4_h, q.n, q.k, q.N, q.pbs_base_log,
                                     q.pbs_level, q.grouping)   # key conversion + OpenMP over the LWEs in the C oracle
            traffic4, src4 = pmc_record("mb_g3") if tag == "multibit_g3" else (None, "not profiled")
            dp.update({"engine": f"f64 FFT, multi-bit grouping factor {q.grouping} (Fourier-domain key; keybundle "
                                 "combined in registers per LWE and group)",
                       "frac_fp64": dp["pbs_per_s"] * flop / (FP64_PEAK_TFLOPS * 1e12), "f64_flop_per_pbs": flop,
                       "hbm_traffic_bytes_per_launch": traffic4, "traffic_source": src4,
                       "gpu_matches_cpu_bits": bool(np.array_equal(ref4, out4[:PAR])),
                       "parity_sample": f"first {PAR} LWEs, all 2049 words, vs the C oracle's multi-bit f64 path "
                                        f"({time.perf_counter() - t0:.1f} s CPU); uniform-random key and inputs"})
            result["extra"][tag] = dp
            del bsk4, d_in4, d_out4
        # ---- config 5 on ONE GPU: FheUint64 (32 blocks of the 2_2 set) add and mul through the radix layer of the
        # backend (keyswitch -> PBS rounds, `tfhe_rs_amd/integer_gpu.py` over the reference's integer FFI names);
        # same key as the headline, 32 distinct operand pairs tiled over the batch (timing is data independent),
        # every distinct result decrypted and compared with clear arithmetic outside the timed region.
        from tfhe_rs_amd import integer_gpu as igpu
        ksk_h = orc.gen_ksk(0x74666865 + 2, keys.glwe_sk, keys.lwe_sk, p.ks_base_log, p.ks_level, p.lwe_noise)
        ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(ksk_h, p.k * p.N, p.n, p.ks_base_log, p.ks_level, streams)
        sks = igpu.CudaServerKey(ksk, bsk, 4, 4)
        LB, DISTINCT = 32, 32
        mask64 = (1 << (2 * LB)) - 1
        r5 = np.random.default_rng(5)
        va = [int.from_bytes(r5.bytes(8), "little") for _ in range(DISTINCT)]
        vb = [int.from_bytes(r5.bytes(8), "little") for _ in range(DISTINCT)]
        er = orc.Rng(55)
        enc = lambda vals: np.stack([np.stack([orc.lwe_encrypt(er, keys.glwe_sk, (((v >> (2 * j)) & 3) * p.delta) % (1 << 64),
                                                               p.glwe_noise) for j in range(LB)]) for v in vals])
        ha, hb = enc(va), enc(vb)
        fhe = {}
        for op, nb in (("add", 1024), ("mul", 128)):
            reps = nb // DISTINCT
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(np.tile(ha, (reps, 1, 1)), streams)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(np.tile(hb, (reps, 1, 1)), streams)
            streams.synchronize()
            t0 = time.perf_counter()
            if op == "add":
                sks.add_assign(ca, cb, streams)
                pbs_count = int(lib.hip_integer_propagate_pbs_count(LB))
            else:
                pbs_count = int(sks.mul_assign(ca, cb, streams, return_pbs_count=True))
            streams.synchronize()
            dt = time.perf_counter() - t0
            rows = ca.to_blocks(streams)
            want = [((x + y) if op == "add" else (x * y)) & mask64 for x, y in zip(va, vb)]
            got = [sum(decrypt_big(p, keys, rows[i, j]) << (2 * j) for j in range(LB)) for i in range(DISTINCT)]
            fhe[op] = {"batch": nb, "seconds": dt, "ops_per_s": nb / dt, "pbs_per_op": pbs_count,
                       "ks_pbs_per_s": nb * pbs_count / dt, "results_decrypt_to_clear_arithmetic": got == want}
            del ca, cb
        fhe["note"] = ("one GPU, classic 2_2 set, wall clock including scratch allocation, index uploads and every round; "
                       "the reference publishes 510 add/s and 53.2 mul/s on 8xH100 with multi-bit parameters (BASELINE.md); "
                       "multi-bit sets and several streams: tools/bench_integer.py, profiles/r02_bench_integer_fheuint64.jsonl")
        result["extra"]["fheuint64"] = fhe
        del sks, ksk
    if world == 1 and not args.no_cpu_baseline:
        # CPU leg: the oracle's f64 path on the host cores actually available to this process
        # (affinity mask and cgroup quota, not the machine's nominal thread count), on a sample
        # sized from a short calibration so the leg takes ~15 s.
        cores = min(int(orc.lib().orc_max_threads()), len(os.sched_getaffinity(0)))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except Exception:
            pass
        bsk_f = orc.convert_bsk_fft(keys.bsk, p.n, p.k, p.N, p.pbs_level)
        t0 = time.perf_counter()
        orc.pbs_batch(orc.ENGINE_FFT, cts[:cores], lut, bsk_f, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, 1,
                      threads=cores)
        # second calibration pass: the first one pays thread start-up and cold caches
        t0 = time.perf_counter()
        orc.pbs_batch(orc.ENGINE_FFT, cts[:cores], lut, bsk_f, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, 1,
                      threads=cores)
        calib = time.perf_counter() - t0
        count = args.cpu_sample or int(max(cores, min(B, cores * max(1, round(25.0 / max(calib, 1e-3))))))
        t0 = time.perf_counter()
        ref = orc.pbs_batch(orc.ENGINE_FFT, cts[:count], lut, bsk_f, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, 1,
                            threads=cores)
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": count / dt, "unit": "PBS/s", "cores": cores, "kind": "port",
            "sample": f"{count} PBS of the same batch through the C oracle's f64 FFT path (C restatement with AVX2 butterflies, "
                      f"OpenMP over LWEs, {cores} threads, {dt:.1f} s); the reference's AVX-512 Rust publishes "
                      f"5.64 ms/PBS on one EPYC 9R45 core",
            "gpu_matches_cpu_bits": bool(np.array_equal(ref, out[:count])),
        }
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
