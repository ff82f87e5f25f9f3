import sys, os, time, json, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.common import C1, make_keys
from tfhe_rs_amd import core_crypto_gpu as gpu, integer_gpu as igpu
from tfhe_rs_amd.integer_gpu import _lib, OUTPUT_FLAG_NONE
p = C1
keys = make_keys(p)
st = gpu.CudaStreams([0])
ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(keys.ksk, p.big_n, p.n, p.ks_base_log, p.ks_level, st)
bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, st, ms_noise_reduction=True)
sks = igpu.CudaServerKey(ksk, bsk, 4, 4)
B, L = 1024, 32
rng = np.random.default_rng(1)
blocks = rng.integers(0, 1 << 63, size=(B, L, p.big_n + 1), dtype=np.uint64)
for rep in range(3):
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks, st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks, st)
    st.synchronize()
    s, keep = sks._streams(st)
    ksks, bsks = sks._key_ptrs(st)
    mem = C.c_void_p()
    cin, cout = sks._carry_blocks(ca, None, st), sks._carry_blocks(ca, None, st)
    t0 = time.perf_counter()
    _lib().hip_integer_scratch_batch(B)
    _lib().scratch_cuda_add_and_propagate_single_carry_64_inplace_async(s, C.byref(mem), sks._bsk_params(), sks._ksk_params(), L, 4, 4, OUTPUT_FLAG_NONE, True, sks._noise_reduction())
    st.synchronize(); t1 = time.perf_counter()
    _lib().cuda_add_and_propagate_single_carry_64_inplace_async(s, C.byref(ca._ffi()), C.byref(cb._ffi()), C.byref(cout._ffi()), C.byref(cin._ffi()), mem, bsks, ksks, OUTPUT_FLAG_NONE, 0)
    st.synchronize(); t2 = time.perf_counter()
    _lib().cleanup_cuda_add_and_propagate_single_carry_64_inplace(s, C.byref(mem))
    st.synchronize(); t3 = time.perf_counter()
    print(json.dumps({"rep": rep, "scratch_ms": (t1-t0)*1e3, "op_ms": (t2-t1)*1e3, "cleanup_ms": (t3-t2)*1e3}))
