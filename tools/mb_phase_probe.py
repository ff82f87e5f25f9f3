#!/usr/bin/env python3
"""GPU box: where the waves of the multi-bit throughput kernels spend their cycles (VERDICT r05 #1b).
Needs a library built with -DWAVE_MB_PROBE=1 (tools/build_variants.py probe=-DWAVE_MB_PROBE=1 ...):
   TFHE_HIP_BACKEND_LIB=variants/lib_probe.so python tools/mb_phase_probe.py [g3|g4] [batch]
Every wave sums the shader-clock cycles between its phase boundaries (s_memtime) over the whole launch; printed: the mean
per wave and group in cycles, by phase and by wave index."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import tfhe_rs_amd  # noqa: E402,F401
from tfhe_rs_amd import core_crypto_gpu as gpu  # noqa: E402
from tfhe_rs_amd import ffi  # noqa: E402
from tests.common import C4, C4G4  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "g4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
p = C4G4 if which == "g4" else C4
lib = ffi.default_library()
probe = lib.cdll.hip_probe_wave_timestamps
probe.restype = None
probe.argtypes = [C.c_void_p, C.c_uint32]
streams = gpu.CudaStreams.new_single_gpu(0)
S, G = streams.ptr[0], 0
rng = np.random.default_rng(7)
r64 = lambda n: rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
k1 = p.k + 1
bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
    r64((p.n // p.grouping) * (1 << p.grouping) * p.pbs_level * k1 * k1 * p.N), p.n, p.k, p.N, p.pbs_base_log, p.pbs_level,
    p.grouping, streams)
d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(r64(B * (p.n + 1)).reshape(B, -1), streams)
d_out = gpu.CudaLweCiphertextList.new(p.k * p.N, B, streams)
d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(r64(k1 * p.N), p.k, p.N, streams)
idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), streams)
lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), streams)
blocks = (B + 3) // 4
rec = gpu.CudaVec.from_cpu_async(np.zeros(blocks * 64, dtype=np.uint64), streams)
buf = C.c_void_p()
lib.scratch_cuda_multi_bit_programmable_bootstrap_64_async(S, G, C.byref(buf), p.k, p.N, p.pbs_level, B, True)


def run():
    lib.cuda_multi_bit_programmable_bootstrap_64_async(
        S, G, d_out.d_vec.ptr, idx.ptr, d_lut.d_vec.ptr, lidx.ptr, d_in.d_vec.ptr, idx.ptr,
        bsk.d_vec.ptr, buf, p.n, p.k, p.N, p.grouping, p.pbs_base_log, p.pbs_level, B, 1, 0)


run()
lib.cuda_synchronize_device(G)
probe(rec.ptr, 0)
e0, e1 = lib.hip_event_create(), lib.hip_event_create()
lib.hip_event_record(e0, S)
run()
lib.hip_event_record(e1, S)
lib.cuda_synchronize_device(G)
ms = lib.hip_event_elapsed_ms(e0, e1)
out = rec.copy_to_cpu(streams)
r = np.asarray(out, dtype=np.float64).reshape(blocks, 8, 8)
groups = p.n // p.grouping
names = ["pace wait", "digits", "forward", "requests + sync in", "keybundle + products", "sync out", "exchange", "inverse"]
per_phase = r.mean(axis=(0, 1)) / groups
res = {"which": which, "params": p.name, "batch": B, "ms": ms, "kernel_id": lib.hip_backend_last_pbs_kernel(), "groups": groups,
       "cycles_per_group_and_wave": {n: round(float(v), 1) for n, v in zip(names, per_phase)},
       "total_cycles_per_group": round(float(per_phase.sum()), 1),
       "by_wave_total": [round(float(x), 1) for x in r.sum(axis=2).mean(axis=0) / groups],
       "by_wave_sync_in": [round(float(x), 1) for x in r[:, :, 3].mean(axis=0) / groups]}
print(json.dumps(res))
lib.cleanup_cuda_multi_bit_programmable_bootstrap_64(S, G, C.byref(buf))
