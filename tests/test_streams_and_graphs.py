"""Behaviour of the `_async` entry points as stream work: they only enqueue — so a launch can be
captured into a HIP graph and replayed — and they may be called concurrently from several host threads on
different streams with the keys shared (the reference: backends/tfhe-cuda-backend/cuda/tests_and_benchmarks/
tests/test_concurrent_pbs.cpp).  Every result is compared with the same call made alone."""
import ctypes as C
import threading

import numpy as np
import pytest

from .common import (TOY_2048, TOY_2048_L2, TOY_MB4_2048, TOY_MB_2048, decrypt_big, encrypt_big, encrypt_small,
                     make_keys)
from .harness import use_backend
from . import oracle as orc

pytestmark = pytest.mark.gpu


class Hip:
    """The four graph calls of the HIP runtime the capture test needs (libamdhip64 is already loaded by the
    backend library)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        for f in ("hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch",
                  "hipGraphExecDestroy", "hipGraphDestroy"):
            getattr(self.rt, f).restype = C.c_int

    def check(self, rc, what):
        assert rc == 0, f"{what}: hipError {rc}"

    def capture(self, stream, enqueue):
        self.check(self.rt.hipStreamBeginCapture(C.c_void_p(stream), 0), "hipStreamBeginCapture")  # mode global
        enqueue()
        graph = C.c_void_p()
        self.check(self.rt.hipStreamEndCapture(C.c_void_p(stream), C.byref(graph)), "hipStreamEndCapture")
        exe = C.c_void_p()
        self.check(self.rt.hipGraphInstantiate(C.byref(exe), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        return graph, exe

    def launch(self, exe, stream):
        self.check(self.rt.hipGraphLaunch(exe, C.c_void_p(stream)), "hipGraphLaunch")

    def destroy(self, graph, exe):
        self.rt.hipGraphExecDestroy(exe)
        self.rt.hipGraphDestroy(graph)


class Job:
    """Device buffers + scratch of one PBS launch on its own stream; `enqueue()` is exactly one `_async` call."""

    def __init__(self, lib, gpu, p, keys, bsk, B, seed, ksk=None):
        self.lib, self.gpu, self.p, self.B = lib, gpu, p, B
        self.st = gpu.CudaStreams.new_single_gpu(0)
        s = self.st.ptr[0]
        msgs = [(seed + 3 * m) % p.plaintext_modulus for m in range(B)]
        self.msgs = msgs
        self.f = lambda x: (x * x + seed) % p.plaintext_modulus
        lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, self.f)
        cts = encrypt_big(p, keys, msgs, seed=seed) if ksk is not None else encrypt_small(p, keys, msgs, seed=seed)
        self.d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, self.st)
        self.d_out = gpu.CudaLweCiphertextList.new(p.k * p.N, B, self.st)
        self.d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, p.k, p.N, self.st)
        self.idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), self.st)
        self.lidx = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), self.st)
        self.buf = C.c_void_p()
        self.bsk, self.ksk = bsk, ksk
        if p.grouping:
            lib.scratch_cuda_multi_bit_programmable_bootstrap_64_async(s, 0, C.byref(self.buf), p.k, p.N, p.pbs_level,
                                                                       B, True)
        else:
            scratch = (lib.hip_scratch_keyswitch_programmable_bootstrap_64_async if ksk is not None
                       else lib.scratch_cuda_programmable_bootstrap_64_async)
            scratch(s, 0, C.byref(self.buf), p.n, p.k, p.N, p.pbs_level, B, True, p.ms_type)
        self.st.synchronize()

    def enqueue(self):
        lib, p, s, B = self.lib, self.p, self.st.ptr[0], self.B
        if p.grouping:
            lib.cuda_multi_bit_programmable_bootstrap_64_async(
                s, 0, self.d_out.d_vec.ptr, self.idx.ptr, self.d_lut.d_vec.ptr, self.lidx.ptr, self.d_in.d_vec.ptr,
                self.idx.ptr, self.bsk.d_vec.ptr, self.buf, p.n, p.k, p.N, p.grouping, p.pbs_base_log, p.pbs_level,
                B, 1, 0)
        elif self.ksk is not None:
            lib.hip_keyswitch_programmable_bootstrap_64_async(
                s, 0, self.d_out.d_vec.ptr, self.idx.ptr, self.d_lut.d_vec.ptr, self.lidx.ptr, self.d_in.d_vec.ptr,
                self.idx.ptr, self.ksk.d_vec.ptr, self.bsk.d_vec.ptr, self.buf, p.n, p.k, p.N, p.ks_base_log,
                p.ks_level, p.pbs_base_log, p.pbs_level, B, 1, 0)
        else:
            lib.cuda_programmable_bootstrap_64_async(
                s, 0, self.d_out.d_vec.ptr, self.idx.ptr, self.d_lut.d_vec.ptr, self.lidx.ptr, self.d_in.d_vec.ptr,
                self.idx.ptr, self.bsk.d_vec.ptr, self.buf, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, B, 1, 0)

    def clear_output(self):
        self.lib.cuda_memset_async(self.d_out.d_vec.ptr, 0, self.B * (self.p.k * self.p.N + 1) * 8, self.st.ptr[0], 0)

    def result(self):
        return self.d_out.to_lwe_ciphertext_list(self.st)

    def close(self):
        s = self.st.ptr[0]
        if self.p.grouping:
            self.lib.cleanup_cuda_multi_bit_programmable_bootstrap_64(s, 0, C.byref(self.buf))
        else:
            self.lib.cleanup_cuda_programmable_bootstrap_64(s, 0, C.byref(self.buf))


def upload(gpu, p, keys, with_ksk=False):
    st = gpu.CudaStreams.new_single_gpu(0)
    if p.grouping:
        bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
            keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping, st)
    else:
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, st,
                                                             ms_noise_reduction=bool(p.ms_type))
    ksk = None
    if with_ksk:
        ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(keys.ksk, p.k * p.N, p.n, p.ks_base_log, p.ks_level, st)
    st.synchronize()
    return bsk, ksk


CASES = [("classic_latency_kernel", TOY_2048, 5, False), ("classic_throughput_kernel", TOY_2048, 300, False),
         ("classic_two_levels", TOY_2048_L2, 300, False), ("keyswitch_then_pbs_one_call", TOY_2048, 70, True),
         ("multi_bit_latency_path", TOY_MB4_2048, 7, False), ("multi_bit_throughput_kernel", TOY_MB_2048, 260, False)]


@pytest.mark.parametrize("name,p,B,with_ksk", CASES, ids=[c[0] for c in CASES])
def test_a_launch_is_captured_into_a_hip_graph_and_replayed(name, p, B, with_ksk):
    """One `_async` launch under hipStreamBeginCapture (global mode: any synchronous or allocating runtime call
    inside the entry point would fail the capture), instantiated, replayed twice onto a cleared output: the bytes
    of the direct call.  The keyswitch case is warmed once first: the byte-plane layout of a keyswitch key is
    built by the first launch that sees the key."""
    from tfhe_rs_amd import core_crypto_gpu as gpu
    lib = use_backend("hip")
    keys = make_keys(p, with_ksk=True)
    bsk, ksk = upload(gpu, p, keys, with_ksk)
    job = Job(lib, gpu, p, keys, bsk, B, seed=11, ksk=ksk)
    hip = Hip()
    try:
        job.enqueue()
        direct = job.result()
        assert [decrypt_big(p, keys, o) for o in direct[:8]] == [job.f(m) for m in job.msgs[:8]]
        graph, exe = hip.capture(job.st.ptr[0], job.enqueue)
        for _ in range(2):
            job.clear_output()
            hip.launch(exe, job.st.ptr[0])
            assert np.array_equal(job.result(), direct)
        hip.destroy(graph, exe)
    finally:
        job.close()


def test_a_captured_keyswitch_keeps_its_operand_scratch_when_later_launches_grow_theirs():
    """The large-batch keyswitch (digit pass + GEMM) writes its operands into a scratch of the (device, stream) — the
    reference's keyswitch entry points carry none.  A launch recorded under stream capture bakes that pointer into the
    graph, so it must never be freed or handed to live launches afterwards: the captured launch takes the
    stream's buffer over, a later LARGER live keyswitch on the same stream allocates its own, and replaying the graph
    still reads and writes valid memory and yields the bytes of the direct call."""
    from tfhe_rs_amd import core_crypto_gpu as gpu
    lib = use_backend("hip")
    p = TOY_2048
    keys = make_keys(p, with_ksk=True)
    st = gpu.CudaStreams.new_single_gpu(0)
    s = st.ptr[0]
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(keys.ksk, p.k * p.N, p.n, p.ks_base_log, p.ks_level, st)
    lib.hip_backend_set_keyswitch_kernel(3)   # digit pass + GEMM from 129 LWEs on (the automatic choice starts at 769)
    hip = Hip()
    try:
        def make(B, seed):
            cts = encrypt_big(p, keys, [(seed + m) % p.plaintext_modulus for m in range(B)], seed=seed)
            d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, st)
            d_out = gpu.CudaLweCiphertextList.new(p.n, B, st)
            idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), st)
            run = lambda: lib.cuda_keyswitch_lwe_ciphertext_vector_64_64_async(  # noqa: E731
                s, 0, d_out.d_vec.ptr, idx.ptr, d_in.d_vec.ptr, idx.ptr, ksk.d_vec.ptr, p.k * p.N, p.n, p.ks_base_log,
                p.ks_level, B)
            return cts, d_in, d_out, idx, run
        cts_a, in_a, out_a, idx_a, run_a = make(160, 5)
        run_a()                                   # warms the key layout and creates the stream's live scratch
        st.synchronize()
        assert lib.hip_backend_last_keyswitch_path() == 2
        direct = out_a.to_lwe_ciphertext_list(st)
        assert np.array_equal(direct, orc.keyswitch_batch(cts_a, keys.ksk, p.k * p.N, p.n, p.ks_base_log, p.ks_level))
        graph, exe = hip.capture(s, run_a)        # the captured launch now owns that scratch
        cts_b, in_b, out_b, idx_b, run_b = make(420, 6)
        run_b()                                   # a larger live launch: must not free or reuse the graph's buffer
        st.synchronize()
        assert np.array_equal(out_b.to_lwe_ciphertext_list(st),
                              orc.keyswitch_batch(cts_b, keys.ksk, p.k * p.N, p.n, p.ks_base_log, p.ks_level))
        for _ in range(2):
            lib.cuda_memset_async(out_a.d_vec.ptr, 0, 160 * (p.n + 1) * 8, s, 0)
            run_b()                               # live traffic on the stream between the replays
            hip.launch(exe, s)
            st.synchronize()
            assert np.array_equal(out_a.to_lwe_ciphertext_list(st), direct)
        hip.destroy(graph, exe)
    finally:
        lib.hip_backend_set_keyswitch_kernel(0)


def test_concurrent_host_threads_on_their_own_streams_share_the_keys():
    """Six host threads, one stream and one scratch each, three parameter sets (classic both kernels, multi-bit
    both paths) over shared device keys, all launching at once, ten rounds: every round of every thread gives the
    bytes the same job gives alone.  Exercises the once-per-device state of the library (tables, kernel
    attributes, the keyswitch-key cache) under contention."""
    from tfhe_rs_amd import core_crypto_gpu as gpu
    lib = use_backend("hip")
    plan = [(TOY_2048, 5, False), (TOY_2048, 300, False), (TOY_2048, 70, True), (TOY_MB4_2048, 7, False),
            (TOY_MB_2048, 260, False), (TOY_2048_L2, 33, False)]
    uploaded = {}
    jobs = []
    for i, (p, B, with_ksk) in enumerate(plan):
        keys = make_keys(p, with_ksk=True)
        if p.name not in uploaded:
            uploaded[p.name] = upload(gpu, p, keys, True)
        bsk, ksk = uploaded[p.name]
        jobs.append(Job(lib, gpu, p, keys, bsk, B, seed=20 + i, ksk=ksk if with_ksk else None))
    alone = []
    for j in jobs:
        j.enqueue()
        alone.append(j.result())
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(j, want):
        try:
            start.wait()
            for _ in range(10):
                j.clear_output()
                j.enqueue()
                if not np.array_equal(j.result(), want):
                    errors.append(f"{j.p.name} batch {j.B}: differs from the same job run alone")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(j, w)) for j, w in zip(jobs, alone)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for j in jobs:
        j.close()
    assert not errors, errors


@pytest.mark.parametrize("which", ["classic", "multi_bit_g4"])
def test_a_whole_radix_addition_is_captured_and_replayed(which):
    """`cuda_add_and_propagate_single_carry_64_inplace_async` between its scratch and cleanup calls only enqueues:
    the block additions and the six KS -> PBS rounds of a 16-block addition (index arrays, LUTs and every buffer
    belong to the scratch) are captured into one HIP graph and replayed on fresh operands; each replay decrypts to
    the sum."""
    from tfhe_rs_amd import core_crypto_gpu as gpu
    from tfhe_rs_amd import integer_gpu as igpu
    lib = use_backend("hip")
    p = TOY_2048 if which == "classic" else TOY_MB4_2048
    keys = make_keys(p, with_ksk=True)
    st = gpu.CudaStreams.new_single_gpu(0)
    bsk, ksk = upload(gpu, p, keys, True)
    sks = igpu.CudaServerKey(ksk, bsk, 4, 4)
    L, mask = 16, (1 << 32) - 1
    rng = np.random.default_rng(77)

    def enc(vals, seed):
        flat = encrypt_big(p, keys, [(int(v) >> (2 * j)) & 3 for v in vals for j in range(L)], seed=seed)
        return flat.reshape(len(vals), L, -1)

    def dec(rows):
        return [sum(decrypt_big(p, keys, b) << (2 * j) for j, b in enumerate(row)) for row in rows]

    a0 = [int(x) for x in rng.integers(0, 1 << 32, size=3)]
    b0 = [int(x) for x in rng.integers(0, 1 << 32, size=3)]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(enc(a0, 1), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(enc(b0, 2), st)
    cin, cout = sks._carry_blocks(ca, None, st), sks._carry_blocks(ca, None, st)
    s, keep = sks._streams(st)
    ksks, bsks = sks._key_ptrs(st)
    mem = C.c_void_p()
    lib.hip_integer_scratch_batch(3)
    lib.scratch_cuda_add_and_propagate_single_carry_64_inplace_async(
        s, C.byref(mem), sks._bsk_params(), sks._ksk_params(), L, 4, 4, 0, True, sks._noise_reduction())
    st.synchronize()

    def enqueue():
        lib.cuda_add_and_propagate_single_carry_64_inplace_async(
            s, C.byref(ca._ffi()), C.byref(cb._ffi()), C.byref(cout._ffi()), C.byref(cin._ffi()), mem, bsks, ksks, 0, 0)

    hip = Hip()
    try:
        enqueue()   # warm: the keyswitch key's matrix-core layout is built by the first launch that sees the key
        assert dec(ca.to_blocks(st)) == [(x + y) & mask for x, y in zip(a0, b0)]
        graph, exe = hip.capture(st.ptr[0], enqueue)
        for rep in range(2):
            a1 = [int(x) for x in rng.integers(0, 1 << 32, size=3)]
            ca.d_blocks.copy_from_cpu_async(np.ascontiguousarray(enc(a1, 10 + rep).reshape(-1)), st)
            st.synchronize()
            hip.launch(exe, st.ptr[0])
            assert dec(ca.to_blocks(st)) == [(x + y) & mask for x, y in zip(a1, b0)]
        hip.destroy(graph, exe)
    finally:
        lib.cleanup_cuda_add_and_propagate_single_carry_64_inplace(s, C.byref(mem))
