#!/usr/bin/env python3
"""bench.py — PBS/s on PARAM_MESSAGE_2_CARRY_2 (n=918, k=1, N=2048, l=1, base_log=23, centered
mean modulus switch), batch of 4096 independent LWEs per GPU, f64 FFT external product.

One "step" = one batched PBS launch (modulus switch + 918 CMUXes + sample extract) over the
4096 LWEs resident in HBM on this rank's GPU, through the C ABI
(scratch_/cuda_/cleanup_ programmable_bootstrap_64).  Multi-GPU: one process per GPU, each
rank owns its own 4096-LWE shard and a replica of the key; no data-path collective (weak
scaling); the barrier/max-over-ranks timing uses torch.distributed (RCCL).

Prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
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
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def pmc_traffic_bytes(kernel_id):
    """HBM bytes per launch of the dominant kernel, from the PMC passes committed with this kernel build
    (tools/pmc.sh -> profiles/pmc_latest.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    16-byte coalesced reads on gfx950).  None when no measurement of this kernel is on file."""
    try:
        import json
        m = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")))
        if m.get("pbs_kernel_id") != kernel_id:
            return None
        return 2 * m["FETCH_SIZE_KB"] * 1024 + m["WRITE_SIZE_KB"] * 1024
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH, help="LWEs per GPU per step (default: the metric's 4096)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic LDS kernel, 2 throughput kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic_bytes(kernel_id),
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PBS * B,
                     "kernel_ms_avg": avg_kernel_s * 1e3,
                     "note": "achieved = streaming model (key re-read per LWE, SURVEY §8d); traffic = HBM bytes per "
                             "launch from the committed PMC passes of this kernel (profiles/pmc_latest.json: "
                             "2 x FETCH_SIZE + WRITE_SIZE, tools/pmc.sh), not re-measured in this run; "
                             "fp64 ceiling see DESIGN.md"},
    }
    if latency_ms is not None:
        result["extra"] = {"single_pbs_latency_ms": latency_ms,
                           "single_pbs_kernel": {7: "block_latency", 2: "wave_throughput"}.get(latency_kernel,
                                                                                              str(latency_kernel)),
                           "note": "one PBS, batch 1, same key; not part of `value` (the reference publishes "
                                   "4.21 ms on an H100, BASELINE.md)"}
    if world == 1 and args.kernel == 0:
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
            "pbs_per_s": B / ms2 * 1e3, "roofline_frac_streaming_model": B / ms2 * 1e3 * bytes2 / (HBM_PEAK_GBPS * 1e9),
            "pbs_kernel_id": int(lib.hip_backend_last_pbs_kernel())}
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
            "sample": f"{count} PBS of the same batch through the C oracle's f64 FFT path (scalar C restatement, "
                      f"OpenMP over LWEs, {cores} threads, {dt:.1f} s); the reference's AVX-512 Rust publishes "
                      f"5.64 ms/PBS on one EPYC 9R45 core",
            "gpu_matches_cpu_bits": bool(np.array_equal(ref, out[:count])),
        }
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
