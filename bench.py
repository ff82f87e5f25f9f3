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
config 5 on one GPU (FheUint64 add / mul through the radix layer, results decrypted), and the reference's own GPU golden
ciphertexts (its pbs_golden test: distance in phase, tools/golden_datapoint.py) — none of them part of `value`.
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


_PMC_NOW = {}   # counter record measured by THIS run (measure_traffic_now), else the committed one


def measure_traffic_now(targets, budget_s=240):
    """HBM traffic of one launch of each throughput kernel, measured DURING this bench run: tools/pmc_record.py runs
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters only) over one batch-4096 launch
    per kernel in child processes.  Skipped — the committed, build-stamped record is used instead — when rocprofv3 is
    absent or this process is itself being profiled (nested tool libraries)."""
    import shutil
    import subprocess
    if not shutil.which("rocprofv3"):
        return "rocprofv3 not on PATH"
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        return "this process runs under a profiler"
    out = os.path.join(ROOT, "gpurun_out", "pmc_bench_run.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    try:
        os.remove(out)
    except OSError:
        pass
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_record.py"), *targets, "--hbm-only", "--tag",
                        "benchrun", "--out", out, "--timeout", "60"], capture_output=True, text=True, timeout=budget_s)
        _PMC_NOW.update(json.load(open(out)))
        return None
    except Exception as e:  # noqa: BLE001
        return f"in-run PMC passes failed ({e.__class__.__name__})"


def pmc_record(target):
    """HBM bytes per launch of one throughput kernel ("fft", "ntt", "mb_g3", "mb_g4") and where the figure comes from:
    the PMC passes of this very run when they were taken (measure_traffic_now), else the committed record
    profiles/pmc_latest.json — measured by tools/pmc_record.py and stamped with the build id of the kernel sources it
    profiled; a record of another build is refused.  Returns (hbm_bytes_per_launch or None, provenance string)."""
    from tools.build_id import source_build_id
    mine = source_build_id()
    k = _PMC_NOW.get("kernels", {}).get(target)
    if _PMC_NOW.get("build_id") == mine and k and "hbm_bytes_per_launch" in k:
        pv = _PMC_NOW.get("provenance", {})
        return k["hbm_bytes_per_launch"], (f"measured in this bench run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                           f"one batch-4096 launch of {k.get('kernel')}, 2 x FETCH_SIZE + WRITE_SIZE; build "
                                           f"{mine}, {pv.get('gpus')}, ROCm {pv.get('rocm')}, host {pv.get('host')}")
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        m = json.load(open(path))
    except Exception as e:
        return None, f"no PMC record ({e.__class__.__name__})"
    if m.get("build_id") != mine:
        return None, f"PMC record is of build {m.get('build_id')}, this tree is {mine}: refused as stale"
    k = m.get("kernels", {}).get(target)
    if not k or "hbm_bytes_per_launch" not in k:
        return None, f"PMC record of build {mine} has no '{target}' entry"
    pv = m.get("provenance", {})
    return k["hbm_bytes_per_launch"], (f"profiles/pmc_latest.json (committed record, not re-measured in this run"
                                       f"{': ' + _PMC_NOW['skipped'] if 'skipped' in _PMC_NOW else ''}), build {mine}, kernel "
                                       f"{k.get('kernel')}: 2 x FETCH_SIZE + WRITE_SIZE per launch, L2 hit rate "
                                       f"{k.get('l2_hit_rate', 0):.3f}; taken on {pv.get('gpus')}, ROCm {pv.get('rocm')}, "
                                       f"host {pv.get('host')}")


def pmc_counters(target):
    """All counters of `target` from the PMC passes of this run, else from the committed build-stamped record; {} if neither
    is of this build."""
    from tools.build_id import source_build_id
    mine = source_build_id()
    for rec in (_PMC_NOW, None):
        if rec is None:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            except Exception:  # noqa: BLE001
                return {}
        if rec.get("build_id") == mine and target in rec.get("kernels", {}):
            return rec["kernels"][target]
    return {}


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4   # wave-instructions per second: 256 CUs x 4 SIMDs, one 4-cycle VALU instruction at a time


class HeadlineShard:
    """One GPU's part of the headline workload: a replica of the bootstrap key, this GPU's shard of the global batch
    resident in HBM, its scratch, and the launches through the C ABI.  The reference's throughput bench has the same
    shape — one host thread and one stream per GPU, key replicas, contiguous shards
    (tfhe-benchmark/benches/core_crypto/pbs_bench.rs:1050-1160, cuda/src/utils/helper_multi_gpu.cuh:170-294)."""

    def __init__(self, lib, p, keys, device, rank, world, B, kernel=0, f=lambda x: x, global_batch=None):
        import numpy as np
        from tfhe_rs_amd import core_crypto_gpu as gpu
        from tfhe_rs_amd.multi_gpu import shard_range
        from tests import oracle as orc
        from tests.common import encrypt_small
        self.lib, self.p, self.keys, self.g, self.rank, self.B, self.f = lib, p, keys, int(device), rank, B, f
        lib.hip_backend_set_fft_kernel(kernel)
        self.streams = gpu.CudaStreams([self.g])
        self.s = self.streams.ptr[0]
        self.bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                                                  self.streams, ms_noise_reduction=bool(p.ms_type),
                                                                  engine="fft64")
        if global_batch is None:
            lo, hi = shard_range(B * world, rank, world)      # this GPU's shard of the global batch (weak scaling)
            assert hi - lo == B
        else:                                                  # a given global batch, split by the reference's rule
            lo, hi = shard_range(global_batch, rank, world)    # (ragged: the first global_batch % world shards take one more)
            B = self.B = hi - lo
        self.lo = lo
        self.msgs = [(lo + i) % p.plaintext_modulus for i in range(B)]
        self.cts = encrypt_small(p, keys, self.msgs, seed=100 + rank)   # B distinct fresh encryptions
        self.lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, f)
        self.d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(self.cts, self.streams)
        self.d_out = gpu.CudaLweCiphertextList.new(p.k * p.N, B, self.streams)
        self.d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(self.lut, p.k, p.N, self.streams)
        self.idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), self.streams)
        self.lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), self.streams)
        self.buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(self.s, self.g, C.byref(self.buf), p.n, p.k, p.N, p.pbs_level, B,
                                                         True, int(p.ms_type))

    def step(self, out=None, count=None):
        p = self.p
        self.lib.cuda_programmable_bootstrap_64_async(
            self.s, self.g, (out or self.d_out).d_vec.ptr, self.idx.ptr, self.d_lut.d_vec.ptr, self.lidx.ptr,
            self.d_in.d_vec.ptr, self.idx.ptr, self.bsk.d_vec.ptr, self.buf, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
            count or self.B, 1, 0)

    def sync(self):
        self.lib.cuda_synchronize_stream(self.s, self.g)

    def timed(self, steps):
        """EXACTLY `steps` launches, each bracketed by HIP events on the stream it is launched on; returns the
        per-launch milliseconds after draining the stream."""
        lib = self.lib
        ev = [(lib.hip_event_create(), lib.hip_event_create()) for _ in range(steps)]
        for e0, e1 in ev:
            lib.hip_event_record(e0, self.s)
            self.step()
            lib.hip_event_record(e1, self.s)
        self.sync()
        return [lib.hip_event_elapsed_ms(e0, e1) for e0, e1 in ev]

    def outputs(self):
        return self.d_out.to_lwe_ciphertext_list(self.streams)

    def undecryptable_rows(self, sample=256):
        import numpy as np
        from tests.common import decrypt_big
        out = self.outputs()
        rng = np.random.default_rng(1234 + self.rank)
        check = rng.choice(self.B, size=min(self.B, sample), replace=False)
        return [int(i) for i in check if decrypt_big(self.p, self.keys, out[i]) != self.f(self.msgs[i])]

    def close(self):
        if self.buf:
            self.lib.cleanup_cuda_programmable_bootstrap_64(self.s, self.g, C.byref(self.buf))
            self.buf = None


def run_in_process(lib, p, keys, devices, B, steps, warmup, kernel=0, verify=True, global_batch=None):
    """The headline on len(devices) GPUs from ONE process: one host thread, one stream, one key replica and one
    B-LWE shard per GPU (the reference bench's shape, pbs_bench.rs:1050-1160); no exchange between GPUs.  All
    threads meet before the timed region, time their own `steps` launches and drain their stream; the job's time is
    latest finish - earliest start.  `devices` may name one GPU several times (the reference's
    debug-fake-multi-gpu idea: several streams on one device) — the logic is the same, the rate is not.
    Returns (elapsed_s, per_gpu list, shard 0 kept open for the rank-0 datapoints)."""
    import threading
    n = len(devices)
    gate = threading.Barrier(n)
    res, errs = [None] * n, []

    def work(i):
        try:
            sh = HeadlineShard(lib, p, keys, devices[i], i, n, B, kernel, global_batch=global_batch)
            for _ in range(warmup):
                sh.step()
            sh.sync()
            gate.wait()
            t0 = time.perf_counter()
            kernel_ms = sh.timed(steps)
            t1 = time.perf_counter()
            gate.wait()
            bad = sh.undecryptable_rows() if verify else []
            res[i] = {"t0": t0, "t1": t1, "kernel_ms": kernel_ms, "bad": bad, "shard": sh, "device": int(devices[i]),
                      "lwes": sh.B, "lo": sh.lo,
                      "kernel_id": int(lib.hip_backend_last_pbs_kernel())}
            if i:
                sh.close()
        except BaseException as e:  # noqa: BLE001 — a failed GPU must not leave the others parked at the barrier
            errs.append(e)
            gate.abort()

    threads = [threading.Thread(target=work, args=(i,), name=f"gpu{devices[i]}-shard{i}") for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    elapsed = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    per_gpu = [{"shard": i, "device": r["device"], "lwes": r["lwes"], "first_lwe": r["lo"], "seconds": r["t1"] - r["t0"],
                "pbs_per_s": r["lwes"] * steps / (r["t1"] - r["t0"]), "kernel_ms_avg": sum(r["kernel_ms"]) / len(r["kernel_ms"]),
                "verified": not r["bad"]} for i, r in enumerate(res)]
    for r in res:
        assert not verify or not r["bad"], f"PBS outputs failed to decrypt at rows {r['bad'][:8]}"
    return elapsed, per_gpu, res[0]["shard"], res[0]["kernel_ms"], res[0]["kernel_id"]


def fheuint64_datapoint(lib, p, keys, devices, batches, in_library):
    """BASELINE.json config 5: FheUint64 (32 blocks of the 2_2 set) add and mul through the radix layer of the backend
    (keyswitch -> PBS rounds, `tfhe_rs_amd/integer_gpu.py` over the reference's integer FFI names), `batches[op]`
    integers in total over `devices`; 32 distinct operand pairs tiled over the batch (timing is data independent),
    every distinct result of every shard decrypted and compared with clear arithmetic outside the timed region.
    in_library=False: the caller shards, one host thread + stream + key replicas per GPU, batch / N integers each,
    nothing crosses GPUs.  in_library=True: ONE stream set naming all GPUs, the library shards every round."""
    import threading
    import numpy as np
    from tfhe_rs_amd import core_crypto_gpu as gpu
    from tfhe_rs_amd import integer_gpu as igpu
    from tfhe_rs_amd.multi_gpu import get_num_inputs_on_gpu
    from tests import oracle as orc
    from tests.common import decrypt_big
    ksk_h = orc.gen_ksk(0x74666865 + 2, keys.glwe_sk, keys.lwe_sk, p.ks_base_log, p.ks_level, p.lwe_noise)
    LB, DISTINCT = 32, 32
    mask64 = (1 << (2 * LB)) - 1
    r5 = np.random.default_rng(5)
    va = [int.from_bytes(r5.bytes(8), "little") for _ in range(DISTINCT)]
    vb = [int.from_bytes(r5.bytes(8), "little") for _ in range(DISTINCT)]
    er = orc.Rng(55)
    enc = lambda vals: np.stack([np.stack([orc.lwe_encrypt(er, keys.glwe_sk, (((v >> (2 * j)) & 3) * p.delta) % (1 << 64),
                                                           p.glwe_noise) for j in range(LB)]) for v in vals])
    ha, hb = enc(va), enc(vb)
    groups = [list(devices)] if in_library else [[d] for d in devices]
    n = len(groups)
    gate = threading.Barrier(n)
    res, errs = {}, []

    def work(i):
        try:
            streams = gpu.CudaStreams(groups[i])
            bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                                                 streams, ms_noise_reduction=bool(p.ms_type))
            ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(ksk_h, p.k * p.N, p.n, p.ks_base_log, p.ks_level, streams)
            sks = igpu.CudaServerKey(ksk, bsk, 4, 4)
            for op, total in batches.items():
                nb = get_num_inputs_on_gpu(total, i, n)
                tile = lambda h: np.tile(h, ((nb + DISTINCT - 1) // DISTINCT, 1, 1))[:nb]
                ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(tile(ha), streams)
                cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(tile(hb), streams)
                streams.synchronize()
                gate.wait()
                t0 = time.perf_counter()
                if op == "add":
                    sks.add_assign(ca, cb, streams)
                    pbs_count = int(lib.hip_integer_propagate_pbs_count(LB))
                else:
                    pbs_count = int(sks.mul_assign(ca, cb, streams, return_pbs_count=True))
                streams.synchronize()
                t1 = time.perf_counter()
                gate.wait()
                rows = ca.to_blocks(streams)
                m = min(nb, DISTINCT)
                want = [((x + y) if op == "add" else (x * y)) & mask64 for x, y in zip(va[:m], vb[:m])]
                got = [sum(decrypt_big(p, keys, rows[r, j]) << (2 * j) for j in range(LB)) for r in range(m)]
                res[(op, i)] = (t0, t1, nb, pbs_count, got == want)
                del ca, cb
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            gate.abort()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    fhe = {}
    for op, total in batches.items():
        rs = [res[(op, i)] for i in range(n)]
        dt = max(r[1] for r in rs) - min(r[0] for r in rs)
        fhe[op] = {"batch": total, "seconds": dt, "ops_per_s": total / dt, "pbs_per_op": rs[0][3],
                   "ks_pbs_per_s": total * rs[0][3] / dt, "per_shard_integers": [r[2] for r in rs],
                   "results_decrypt_to_clear_arithmetic": all(r[4] for r in rs)}
    fhe["gpus"] = [int(d) for d in devices]
    fhe["sharding"] = ("in the library: one CudaStreamsFFI over all GPUs, every KS -> PBS round split by "
                       "get_num_inputs_on_gpu with peer copies, ciphertexts on the first GPU" if in_library else
                       "by the caller: batch / N integers per GPU, one host thread + stream + key replicas each, nothing "
                       "crosses GPUs")
    fhe["note"] = ("classic 2_2 set, wall clock including scratch allocation, index uploads and every round; the reference "
                   "publishes 510 add/s and 53.2 mul/s on 8xH100 with multi-bit parameters (BASELINE.md); multi-bit sets: "
                   "tools/bench_integer.py")
    return fhe


def pick_devices(lib, n):
    """GPU of each of the n shards.  More shards than GPUs is an error unless TFHE_BENCH_FAKE_MULTI_GPU=1, which maps
    shard i to GPU i mod (GPUs present): a logic check of the multi-GPU path on a smaller box, never a scaling number."""
    have = int(lib.cuda_get_number_of_gpus())
    if n <= have:
        return list(range(n)), False
    if os.environ.get("TFHE_BENCH_FAKE_MULTI_GPU") == "1":
        return [i % have for i in range(n)], True
    raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (TFHE_BENCH_FAKE_MULTI_GPU=1 runs the "
                     f"{n} shards as streams of the GPUs present — logic check only)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH, help="LWEs per GPU per step (default: the metric's 4096)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic LDS kernel, 2 throughput kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3/4/5 and N=1024 datapoints under `extra`")
    ap.add_argument("--no-pmc", action="store_true", help="do not take rocprofv3 counter passes during the run "
                    "(`traffic` then comes from the committed build-stamped record)")
    ap.add_argument("--no-verify", action="store_true", help="skip the decrypt check (timing-ablation builds only)")
    ap.add_argument("--parity-sample", type=int, default=0,
                    help="LWEs of the config-3 / config-4 launches compared word for word with the CPU oracle "
                         "(0 = the whole batch: about 2.5 min of CPU work on 16 threads)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="PBS count of the CPU baseline sample (0 = auto)")
    ap.add_argument("--scale-quick", action="store_true",
                    help="headline only (= --no-extra --no-cpu-baseline --no-pmc): the form for a 1/2/4/8-GPU scaling sweep — no CPU-oracle "
                         "legs (they are 95 %% of the default run's wall time), per-GPU ms_per_step and config 5's active-GPU count in the line")
    ap.add_argument("--fheuint64-worker", default=None, help=argparse.SUPPRESS)  # child process of the N > 1 config-5 datapoints
    args = ap.parse_args()
    if args.scale_quick:
        args.no_extra = args.no_cpu_baseline = args.no_pmc = True
    if args.fheuint64_worker:
        import tfhe_rs_amd  # noqa: F401
        from tfhe_rs_amd import ffi
        from tests.common import C1, make_keys
        w = json.loads(args.fheuint64_worker)
        print(json.dumps(fheuint64_datapoint(ffi.default_library(), C1, make_keys(C1, with_ksk=False), w["devices"],
                                             {"add": 1024, "mul": 128}, in_library=bool(w["in_library"]))))
        return

    # Two ways to N GPUs.  (1) Launched by torch.distributed.run (WORLD_SIZE set): one process per GPU, barrier and
    # max-over-ranks through torch.distributed (RCCL).  (2) Plain `python bench.py --gpus N`: this process drives
    # the N GPUs itself, one host thread + stream + key replica per GPU (run_in_process) — the reference bench's shape.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and world > 1
    dist = None
    if launched or os.environ.get("TFHE_BENCH_FORCE_DIST"):  # the env knob exercises this path at N = 1
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
    from tests import oracle as orc          # checker + cpu_baseline leg only
    from tests.common import C1, make_keys, decrypt_big

    lib = ffi.default_library()
    assert lib.cuda_is_available() == 1, "no GPU visible: the backend has no CPU fallback"
    p = C1
    B = args.batch
    f = lambda x: x

    # ---- synthetic inputs: real keys (seeded), fresh encryptions of i mod 16, LUT f(x) = x
    keys = make_keys(p, with_ksk=False)
    fake = False
    if launched or args.gpus <= 1:
        sh = HeadlineShard(lib, p, keys, local_rank, rank, world, B, args.kernel, f)
        for _ in range(args.warmup):
            sh.step()
        sh.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
        t0 = time.perf_counter()
        kernel_ms = sh.timed(args.steps)
        if dist is not None:
            import torch
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        rank_ms = rank_kernel_ms = None
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.barrier()
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)     # per-rank times for the line (16 bytes per rank; outside the timed region)
            rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
            tk = torch.tensor([sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device="cuda")
            every_k = [torch.zeros_like(tk) for _ in range(world)]
            dist.all_gather(every_k, tk)  # ... and every rank's HIP-event kernel time (its GPU's clock shows in it)
            rank_kernel_ms = [float(x.item()) for x in every_k]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kernel_id = lib.hip_backend_last_pbs_kernel()
        bad = [] if args.no_verify else sh.undecryptable_rows()
        assert not bad, f"PBS outputs failed to decrypt at rows {bad[:8]}"
        per_gpu, n_gpus, devices = None, world, [local_rank]
        mode = f"one process per GPU (torch.distributed, RCCL barrier) x{world}" if launched else "single GPU"
    else:
        devices, fake = pick_devices(lib, args.gpus)
        elapsed, per_gpu, sh, kernel_ms, kernel_id = run_in_process(lib, p, keys, devices, B, args.steps, args.warmup,
                                                                    args.kernel, not args.no_verify)
        n_gpus = args.gpus
        mode = (f"one process, {n_gpus} host threads, one stream + key replica + {B}-LWE shard per GPU"
                + (" — FAKE multi-GPU: shards are streams of the GPUs present, logic check only" if fake else ""))
    streams, s, g = sh.streams, sh.s, sh.g
    bsk, d_in, d_lut, idx, lidx, buf, cts, lut, msgs = (sh.bsk, sh.d_in, sh.d_lut, sh.idx, sh.lidx, sh.buf, sh.cts,
                                                        sh.lut, sh.msgs)
    single = n_gpus == 1

    # ---- single-PBS latency (outside the timed region; the reference publishes this figure)
    latency_ms = None
    if rank == 0 and args.kernel == 0 and not args.no_extra:
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

    out = None if args.no_cpu_baseline or not single else sh.outputs()   # compared with the CPU leg's bits below
    sh.close()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_pbs = B * n_gpus * args.steps
    value = total_pbs / elapsed
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
    achieved = ALGO_BYTES_PER_PBS * B / avg_kernel_s / 1e9
    tflops = ALGO_FLOP_PER_PBS * B / avg_kernel_s / 1e12
    if single and not args.no_pmc and args.kernel == 0:
        # after the timed region: one launch per kernel under `rocprofv3 --pmc`, in child processes
        # (the integer Goldilocks kernel's counters for extra.ntt.roofline come from the committed build-stamped record:
        # three more passes of a 250 ms launch are not worth the wall time of the default run)
        why = measure_traffic_now(["fft"] if args.no_extra else ["fft", "ntt", "mb_g3", "mb_g4"], budget_s=360)
        if why:
            _PMC_NOW["skipped"] = why
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
    ctr = pmc_counters("fft")
    if ctr.get("GRBM_GUI_ACTIVE") and B == BATCH:
        # the launch takes the same number of shader cycles however busy the chip is (profiles/r05_penalty_attribution.txt):
        # cycles of one launch (GRBM_GUI_ACTIVE over the 8 XCCs, a PMC pass of this build) / the timed launches' duration =
        # the shader clock the chip sustained under this kernel (nominal 2.4 GHz; power management lowers it at full load)
        cycles = ctr["GRBM_GUI_ACTIVE"] / 8
        clk = cycles / avg_kernel_s / 1e9
        roofline.update({"shader_cycles_per_launch": cycles, "sustained_clock_ghz": clk,
                         "peak_at_sustained_clock": FP64_PEAK_TFLOPS * clk / 2.4,
                         "frac_at_sustained_clock": tflops / (FP64_PEAK_TFLOPS * clk / 2.4),
                         "valu_busy": (4 * ctr["SQ_ACTIVE_INST_VALU"] / (cycles * 1024)) if ctr.get("SQ_ACTIVE_INST_VALU") else None,
                         "clock_note": "`frac` is against the 2.4 GHz datasheet peak; the chip holds sustained_clock_ghz under this "
                                       "kernel (per-XCC power management: 2.39 GHz with <= 128 CUs busy, 2.0-2.2 GHz with all 256)"})
    result = {
        "metric": "PBS/sec (shortint PARAM_MESSAGE_2_CARRY_2, classic PBS, f64 FFT external product)",
        "value": value, "unit": "PBS/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64 (u64 torus)", "data": "synthetic",
        "config": {"workload": "batch of 4096 independent PBS per GPU, PARAM_MESSAGE_2_CARRY_2 "
                               "(n=918,k=1,N=2048,l=1,base_log=23, centered-mean MS), f64 FFT, inputs resident in HBM",
                   "batch_per_gpu": B, "lwe_dimension": p.n, "glwe_dimension": p.k, "polynomial_size": p.N,
                   "pbs_kernel": {1: "generic_lds", 2: "wave_throughput"}.get(kernel_id, str(kernel_id)),
                   "parallelism": f"batch-sharded x{n_gpus}, key replicas, no collective", "launch": mode},
        "roofline": roofline,
    }
    per_gpu_kernel_ms = None
    if per_gpu is not None:
        result["per_gpu"] = per_gpu
        result["fake_multi_gpu"] = fake
        result["per_gpu_ms_per_step"] = [g_["seconds"] / args.steps * 1e3 for g_ in per_gpu]
        per_gpu_kernel_ms = [g_["kernel_ms_avg"] for g_ in per_gpu]
    elif launched and rank_ms is not None:
        result["per_gpu_ms_per_step"] = rank_ms
        per_gpu_kernel_ms = rank_kernel_ms
    if per_gpu_kernel_ms and roofline.get("shader_cycles_per_launch"):
        # a launch is the same number of shader cycles on every GPU (profiles/r05_penalty_attribution.txt): a GPU's kernel time
        # gives the clock it held — a scaling curve is to be read against these, not against PBS/s alone
        cyc = roofline["shader_cycles_per_launch"]
        result["per_gpu_kernel_ms"] = per_gpu_kernel_ms
        result["per_gpu_sustained_clock_ghz"] = [cyc / (ms * 1e-3) / 1e9 for ms in per_gpu_kernel_ms]
    if args.scale_quick:
        # config 5 on this many GPUs: how many of them a KS -> PBS round of the radix layer would use under the reference's
        # thresholds (helper_multi_gpu.cu:39-101: a GPU is added per `threshold` blocks; classic = compute units + 1,
        # multi-bit = 12) — 1024 FheUint64 of 32 blocks enter the first round of an addition as 32768 blocks
        cus = lib.cuda_get_number_of_sms()
        blocks = 1024 * 32
        act = lambda nb, pbs_type: int(lib.hip_integer_active_gpu_count(nb, n_gpus, pbs_type, 0))  # the library's own rule
        result["config5_active_gpus"] = {"gpus": n_gpus, "blocks_in_first_round": blocks,
                                         "classic": act(blocks, 1), "multi_bit": act(blocks, 0),
                                         "thresholds": {"classic": cus + 1, "multi_bit": 12},
                                         "one_fheuint64_classic": act(32, 1), "one_fheuint64_multi_bit": act(32, 0)}
    if latency_ms is not None:
        result["extra"] = {"single_pbs_latency_ms": latency_ms,
                           "single_pbs_kernel": {7: "block_latency", 2: "wave_throughput"}.get(latency_kernel,
                                                                                              str(latency_kernel)),
                           "note": "one PBS, batch 1, same key; not part of `value` (the reference publishes "
                                   "4.21 ms on an H100, BASELINE.md)"}
    if single and args.kernel == 0 and not args.no_extra:
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
    if single and args.kernel == 0 and not args.no_extra:
        # BASELINE.json configs 3 and 4 next to the headline (never part of `value`): batch 4096 on this GPU, HIP-event
        # time over 3 launches, and the GPU's output words compared with the CPU oracle on the WHOLE batch (BASELINE.json
        # config 3: "batch 4096 ... bit-exact vs CPU NTT") unless --parity-sample asks for a prefix.
        from tests.common import C4
        PAR = B if args.parity_sample <= 0 else min(B, args.parity_sample)
        par_txt = f"all {B} LWEs of the launch" if PAR == B else f"first {PAR} LWEs"

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

        # ---- config 3: 64-bit prime NTT engine (tfhe-ntt semantics), same key, same ciphertexts as the headline.  Two
        # implementations of the same function, both compared word for word with the C oracle's NTT path: the split-key
        # f64 form on the throughput kernel's machinery (the default where it applies) and the integer Goldilocks kernel.
        d_out3 = gpu.CudaLweCiphertextList.new(p.k * p.N, B, streams)
        buf3 = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(s, g, C.byref(buf3), p.n, p.k, p.N, p.pbs_level, B, True, 1)
        t0 = time.perf_counter()
        ref3 = orc.pbs_batch(orc.ENGINE_NTT, cts[:PAR], lut, orc.convert_bsk_ntt(keys.bsk, p.n, p.k, p.N, p.pbs_level),
                             p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, 1)
        t_ref3 = time.perf_counter() - t0
        ntt = {}
        for impl, launch3, label in (
                ("ntt64_split", lib.hip_programmable_bootstrap_ntt64_split_async,
                 "exact products modulo p = 2^64 - 2^32 + 1 on f64 transforms: key in four 16-bit limbs, round-off checked"),
                ("ntt64", lib.hip_programmable_bootstrap_ntt64_async,
                 "Goldilocks NTT (p = 2^64 - 2^32 + 1), integer arithmetic")):
            bsk_n = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
                                                                   streams, ms_noise_reduction=True, engine=impl)
            dp = datapoint(p, lambda: launch3(
                s, g, d_out3.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr, d_in.d_vec.ptr, idx.ptr, bsk_n.d_vec.ptr, buf3,
                p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, B, 1, 0), steps=2)
            out3 = d_out3.to_lwe_ciphertext_list(streams)
            dp.update({"engine": label, "gpu_matches_cpu_bits": bool(np.array_equal(ref3, out3[:PAR])),
                       "parity_sample": PAR,
                       "parity_note": f"{par_txt}, all 2049 words, vs the C oracle's NTT engine ({t_ref3:.1f} s CPU)",
                       "decrypts": all(decrypt_big(p, keys, out3[i]) == f(msgs[i]) for i in range(PAR))})
            ntt[impl] = dp
            del bsk_n
        lib.cleanup_cuda_programmable_bootstrap_64(s, g, C.byref(buf3))
        traffic3, src3 = pmc_record("ntt")
        dp = ntt["ntt64_split"]
        dp.update({"bound": "fp64 VALU + 64-bit integer VALU", "hbm_traffic_bytes_per_launch": traffic3, "traffic_source": src3,
                   "frac_fp64": dp["pbs_per_s"] * 918 * (2 * 25600 + 4 * (2 * 25600 + 4 * 1024 * 8 + 2 * 2048)) / (FP64_PEAK_TFLOPS * 1e12),
                   "f64_flop_model": "per CMUX (k+1) forward transforms + 4 limbs x ((k+1) inverse transforms, (k+1)^2 n "
                                     "multiply-adds at 8 flop, 1 flop per coefficient: the rounding addition whose bit pattern feeds the integer recombination and the round-off check)",
                   "integer_goldilocks_kernel": {k2: ntt["ntt64"][k2] for k2 in ("ms_per_launch", "pbs_per_s",
                                                                               "gpu_matches_cpu_bits", "pbs_kernel_id")}})
        # how far each engine is from its own ceiling (VERDICT r04 #3a): VALU instructions are what both are made of
        roof = {"valu_issue_peak_wave_instructions_per_s": VALU_ISSUE_PEAK,
                "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction; achieved = SQ_INSTS_VALU of one "
                        "launch (PMC pass of this build) / launch time; floor = the least instructions the engine's arithmetic needs"}
        c_split, c_int = pmc_counters("ntt"), pmc_counters("ntt_int")
        if c_split.get("SQ_INSTS_VALU"):
            per_pbs = c_split["SQ_INSTS_VALU"] / B
            flop_split = 918 * (2 * 25600 + 4 * (2 * 25600 + 4 * 1024 * 8 + 2 * 2048))
            roof["split_f64"] = {"valu_wave_instructions_per_pbs": per_pbs, "achieved": per_pbs * dp["pbs_per_s"],
                                 "frac_valu_issue": per_pbs * dp["pbs_per_s"] / VALU_ISSUE_PEAK,
                                 "f64_wave_instructions_floor_per_pbs": flop_split / 2 / 64,
                                 "frac_of_f64_floor": dp["pbs_per_s"] * (flop_split / 2 / 64) / VALU_ISSUE_PEAK,
                                 "floor_note": "f64 flop model / 2 flop per FMA / 64 lanes: (k+1) forward + 4 x (k+1) inverse transforms, "
                                               "4 x (k+1)^2 n complex multiply-adds, one rounding addition per coefficient; the Horner recombination "
                                               "modulo p, digit extraction and lane exchanges come on top"}
        if c_int.get("SQ_INSTS_VALU"):
            per_pbs = c_int["SQ_INSTS_VALU"] / B
            mulmods = 5.2e7   # ~57 k 64-bit modular multiplications per external product x 918
            rate_i = ntt["ntt64"]["pbs_per_s"]
            roof["integer_goldilocks"] = {
                "mulmods_per_pbs": mulmods, "valu_wave_instructions_per_pbs": per_pbs,
                "valu_lane_instructions_per_mulmod_measured": per_pbs * 64 / mulmods,
                "valu_instructions_per_butterfly_floor": 40,
                "floor_note": "one Goldilocks butterfly as this compiler emits it for gfx950: 27 instructions for the modular product "
                              "(5 v_mad_u64_u32 + the reduction by 2^64 = 2^32 - 1) + 13 for the modular add and subtract",
                "achieved": per_pbs * rate_i, "frac_valu_issue": per_pbs * rate_i / VALU_ISSUE_PEAK,
                "ceiling_pbs_per_s_at_floor": VALU_ISSUE_PEAK / (mulmods * 40 / 64),
                "frac_of_ceiling": rate_i / (VALU_ISSUE_PEAK / (mulmods * 40 / 64))}
        dp["roofline"] = roof
        result.setdefault("extra", {})["ntt"] = dp
        del d_out3

        # ---- config 4: multi-bit PBS, grouping factor 3; and the reference's GPU default multi-bit set (g = 4).
        # REAL keys (GGSW encryptions of the products of the key bits, 320 / 241 MB, a few seconds in the C oracle) and
        # fresh encryptions of the messages i mod 16: the outputs are compared with the oracle AND decrypted.
        from tests.common import C4G4, encrypt_small, make_keys
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
            keys4 = make_keys(q, with_ksk=False)
            bsk4_h = keys4.bsk
            msgs4 = [i % q.plaintext_modulus for i in range(B)]
            cts4 = encrypt_small(q, keys4, msgs4, seed=400 + q.grouping)
            f4 = lambda x: (3 * x + 1) % q.plaintext_modulus  # noqa: E731
            lut4 = orc.generate_lut(q.k, q.N, q.plaintext_modulus, q.delta, f4)
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
            traffic4, src4 = pmc_record("mb_g3" if tag == "multibit_g3" else "mb_g4")
            dp.update({"engine": f"f64 FFT, multi-bit grouping factor {q.grouping} (Fourier-domain key; keybundle "
                                 "combined in registers per LWE and group)",
                       "frac_fp64": dp["pbs_per_s"] * flop / (FP64_PEAK_TFLOPS * 1e12), "f64_flop_per_pbs": flop,
                       "hbm_traffic_bytes_per_launch": traffic4, "traffic_source": src4,
                       "gpu_matches_cpu_bits": bool(np.array_equal(ref4, out4[:PAR])),
                       "parity_sample": PAR,
                       "parity_note": f"{par_txt}, all 2049 words, vs the C oracle's multi-bit f64 path "
                                      f"({time.perf_counter() - t0:.1f} s CPU); real key, fresh encryptions",
                       "decrypts": all(decrypt_big(q, keys4, out4[i]) == f4(msgs4[i]) for i in range(B))})
            result["extra"][tag] = dp
            del bsk4, d_in4, d_out4, keys4
        # ---- config 5 on ONE GPU: FheUint64 (32 blocks of the 2_2 set) add and mul through the radix layer
        result["extra"]["fheuint64"] = fheuint64_datapoint(lib, p, keys, [g], {"add": 1024, "mul": 128}, in_library=False)
        # ---- latency of ONE FheUint64 operation on the reference's GPU multi-bit set (what its documentation publishes
        # for 8 x H100: 9.52 / 31.9 ms); uniform-random key material, the third repetition is reported
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "latency_integer.py"), "multibit_g4",
                                "--throughput"], capture_output=True, text=True, timeout=300)
            lat = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
            one = {l["op"].split()[-1]: l for l in lat if "batch" not in l}      # one-operation rows, by operation name
            many = {l["op"].split()[-1]: l for l in lat if "batch" in l}         # --throughput rows
            result["extra"]["fheuint64_single_operation_latency"] = {
                "params": C4G4.name, "add_ms": one["add"]["operation_ms"], "mul_ms": one["mul"]["operation_ms"],
                "note": "one ciphertext pair, one stream, one GPU: six (add) dependent KS -> multi-bit PBS rounds; the "
                        "reference publishes 9.52 / 31.9 ms with the blocks of a round spread over 8 x H100"}
            more = {k + "_ms": v["total_ms"] for k, v in one.items() if "total_ms" in v}
            if more:  # round 6's operations: scratch + operation + cleanup through the host wrapper
                result["extra"]["fheuint64_single_operation_latency"]["with_scratch_and_cleanup"] = more
            if "add" in many and "mul" in many:
                result["extra"]["fheuint64_multibit_g4_throughput"] = {
                    "params": C4G4.name, "add": {k: many["add"][k] for k in ("batch", "seconds", "ops_per_s", "pbs_per_op")},
                    "mul": {k: many["mul"][k] for k in ("batch", "seconds", "ops_per_s", "pbs_per_op")},
                    **({"sub": {k: many["sub"][k] for k in ("batch", "seconds", "ops_per_s", "pbs_per_op")}} if "sub" in many else {}),
                    "note": "ONE GPU, the parameter set the reference's published 510 add/s and 53.2 mul/s (8 x H100) "
                            "use; timing only (uniform-random key material, the timing is data independent), the second "
                            "repetition of each (the first carries the process's one-time costs); "
                            "decrypt-checked: tools/bench_integer.py --params multibit_g4"}
        except Exception as e:  # noqa: BLE001
            result["extra"]["fheuint64_single_operation_latency"] = {"error": f"{e.__class__.__name__}: {e}"[:300]}
        # ---- the reference's own GPU golden ciphertexts (pbs_golden, captured on an H100) at the metric's parameter set and
        # the GPU multi-bit g = 4 set, on keys / inputs regenerated from the test's seed: lanes, oracle bits, distance in phase.
        # In its own process: a surprise there must not cost the line its headline.
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "golden_datapoint.py")], capture_output=True,
                               text=True, timeout=300)
            result["extra"]["reference_gpu_golden"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            result["extra"]["reference_gpu_golden"] = {"error": f"{e.__class__.__name__}: {e}"[:300]}
    if not single and per_gpu is not None and args.kernel == 0 and not args.no_extra:
        # ---- config 5 on the N GPUs: the batch of 1024 FheUint64 sharded (a) by the caller, 1024 / N integers per GPU,
        # every round GPU-local, and (b) inside the library, one CudaStreamsFFI naming the N GPUs: the
        # ciphertexts live on the first GPU and every KS -> PBS round is split over the GPUs with peer copies
        # (helper_multi_gpu.cuh:170-294).
        # Each in its own process: the peer copies of (b) have only ever run between streams of ONE device (the test
        # boxes have one GPU) — a failure there must not cost the line its headline.
        import subprocess
        for key, in_lib in (("fheuint64", False), ("fheuint64_in_library_sharding", True)):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fheuint64-worker",
                                    json.dumps({"devices": devices, "in_library": in_lib})], capture_output=True, text=True,
                                   timeout=900)
                result.setdefault("extra", {})[key] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001
                result.setdefault("extra", {})[key] = {"error": f"{e.__class__.__name__}: {e}"[:400]}
    if single and not args.no_cpu_baseline:
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
        # one PBS on one thread (SURVEY §8d / BASELINE.md §3: the latency next to the throughput), best of three
        single_ms = None
        for _ in range(3):
            t1 = time.perf_counter()
            orc.pbs_batch(orc.ENGINE_FFT, cts[:1], lut, bsk_f, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, 1, threads=1)
            d1 = (time.perf_counter() - t1) * 1e3
            single_ms = d1 if single_ms is None else min(single_ms, d1)
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.lower().startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        result["cpu_baseline"] = {
            "value": count / dt, "unit": "PBS/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model, "host_threads_online": os.cpu_count(),
            "single_thread_ms_per_pbs": single_ms,
            "published_reference": {"ms_per_pbs": 5.64, "where": "tfhe-rs AVX-512 Rust, one EPYC 9R45 core (BASELINE.md)"},
            "sample": f"{count} PBS of the same batch through the C oracle's f64 FFT path (C restatement with AVX2 butterflies, "
                      f"OpenMP over LWEs, {cores} threads, {dt:.1f} s); the reference's AVX-512 Rust publishes "
                      f"5.64 ms/PBS on one EPYC 9R45 core",
            "gpu_matches_cpu_bits": bool(np.array_equal(ref, out[:count])),
        }
    if dist is not None:
        dist.destroy_process_group()
    # the ONE JSON line goes out last: RCCL's version banner sits in the C runtime's stdout buffer (written at communicator
    # set-up, flushed at exit when stdout is a file or a pipe) and would otherwise land behind it
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
