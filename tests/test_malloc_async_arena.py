"""cuda_malloc_async / cuda_drop as the reference means them (tfhe-cuda-common/cuda/src/device.cu:176-226,457-491; CudaVec::new_async
and Drop, tfhe/src/core_crypto/gpu/vec.rs:94-150,487-495): an allocation is stream work, not a device synchronisation.  Served by
the library's own arena (tfhe_rs_amd/csrc/arena.hip).  [emu] the book-keeping on the host build (no GPU); [hip] stream order across
streams, stream capture and the reference's alloc / drop-per-operation pattern on the MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

from tfhe_rs_amd import core_crypto_gpu as gpu

from .harness import use_backend

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def stats(lib):
    out = (C.c_uint64 * 7)()
    lib.hip_backend_allocator_stats(0, out)
    return dict(zip(("allocations", "reuses", "runtime_allocations", "frees", "cross_stream_waits", "live_bytes", "cached_bytes"), out))


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.skipif(os.environ.get("TFHE_HIP_ARENA_REDZONE", "0") not in ("", "0"),
                    reason="red-zone mode carves blocks out of slabs: nothing goes back to the runtime at a trim")
def test_blocks_are_recycled_by_size_class_and_trimmed(kind):
    lib = use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)
    s = st.ptr[0]
    lib.cuda_synchronize_device(0)
    lib.hip_backend_trim_allocator(0)
    s0 = stats(lib)
    a = lib.cuda_malloc_async(1000, s, 0)          # class 1024
    b = lib.cuda_malloc_async(1024, s, 0)          # class 1024, another block
    c = lib.cuda_malloc_async(3 << 20, s, 0)       # 3 MiB
    d = lib.cuda_malloc_async((3 << 20) + 1, s, 0)  # 4 MiB
    assert len({a, b, c, d}) == 4 and all((a, b, c, d))
    s1 = stats(lib)
    assert s1["live_bytes"] - s0["live_bytes"] == 1024 + 1024 + (3 << 20) + (4 << 20)
    lib.cuda_memset_async(a, 0x0101010101010101, 1000, s, 0)
    lib.cuda_drop(a, 0)
    a2 = lib.cuda_malloc_async(600, s, 0)          # same class, same stream: the block just dropped, no runtime call
    assert a2 == a
    s2 = stats(lib)
    assert s2["reuses"] - s1["reuses"] == 1 and s2["runtime_allocations"] == s1["runtime_allocations"]
    assert s2["cross_stream_waits"] == s1["cross_stream_waits"]
    e = lib.cuda_malloc_async(100, s, 0)            # class 256: nothing cached there
    assert e not in (a, b, c, d)
    for p in (a2, b, c, d, e):
        lib.cuda_drop(p, 0)
    st.synchronize()
    s3 = stats(lib)
    assert s3["live_bytes"] == s0["live_bytes"] and s3["cached_bytes"] - s0["cached_bytes"] == 2048 + (7 << 20) + 256
    released = lib.hip_backend_trim_allocator(0)
    assert released >= 2048 + (7 << 20) + 256
    assert stats(lib)["cached_bytes"] == s3["cached_bytes"] - released
    # cuda_malloc'd memory still goes back to the runtime through the same cuda_drop
    q = lib.cuda_malloc(4096, 0)
    lib.cuda_drop(q, 0)
    assert stats(lib)["frees"] == s3["frees"]


@pytest.mark.parametrize("kind", BACKENDS)
def test_a_destroyed_streams_blocks_become_anybodys(kind):
    lib = use_backend(kind)
    s1 = lib.cuda_create_stream_ffi(0)
    s2 = lib.cuda_create_stream_ffi(0)
    lib.hip_backend_trim_allocator(0)
    p = lib.cuda_malloc_async(5000, s1, 0)
    lib.cuda_drop(p, 0)
    lib.cuda_destroy_stream(s1, 0)   # synchronises: the block is idle now and tied to no stream
    before = stats(lib)
    q = lib.cuda_malloc_async(5000, s2, 0)
    assert q == p
    assert stats(lib)["cross_stream_waits"] == before["cross_stream_waits"]   # nothing to wait for
    lib.cuda_drop(q, 0)
    lib.cuda_destroy_stream(s2, 0)


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("TFHE_HIP_ARENA_REDZONE", "0") not in ("", "0"),
                    reason="red-zone mode synchronises the device at every drop: there is no cross-stream wait left to count")
def test_a_block_changes_streams_in_stream_order():
    """Stream A still works on a block when the host drops it and stream B asks for one of the same class: B's work is queued
    behind A's (an event wait, counted), so what B writes is what B reads back — A's late writes cannot land on top of it."""
    from .common import TOY_2048, encrypt_small, make_keys
    from . import oracle as orc
    lib = use_backend("hip")
    p = TOY_2048
    keys = make_keys(p)
    B = 512
    sa, sb = gpu.CudaStreams.new_single_gpu(0), gpu.CudaStreams.new_single_gpu(0)
    a, b = sa.ptr[0], sb.ptr[0]
    bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, sa, ms_noise_reduction=True)
    cts = encrypt_small(p, keys, [m % 16 for m in range(B)], seed=5)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, sa)
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, lambda x: (x + 1) % 16)
    d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, p.k, p.N, sa)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), sa)
    zero = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), sa)
    buf = C.c_void_p()
    lib.scratch_cuda_programmable_bootstrap_64_async(a, 0, C.byref(buf), p.n, p.k, p.N, p.pbs_level, B, True, p.ms_type)
    out_bytes = B * (p.k * p.N + 1) * 8
    sa.synchronize()
    lib.hip_backend_trim_allocator(0)
    for round_ in range(4):
        out = lib.cuda_malloc_async(out_bytes, a, 0)
        for _ in range(6):   # a few ms of writes into `out` on stream A
            lib.cuda_programmable_bootstrap_64_async(a, 0, out, idx.ptr, d_lut.d_vec.ptr, zero.ptr, d_in.d_vec.ptr, idx.ptr, bsk.d_vec.ptr,
                                                     buf, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, B, 1, 0)
        before = stats(lib)
        lib.cuda_drop(out, 0)                              # the host lets go while A is still writing
        mine = lib.cuda_malloc_async(out_bytes, b, 0)      # the same block, for stream B
        assert mine == out
        assert stats(lib)["cross_stream_waits"] == before["cross_stream_waits"] + 1
        lib.cuda_memset_async(mine, 0, out_bytes, b, 0)
        host = np.empty(out_bytes // 8, dtype=np.uint64)
        lib.cuda_memcpy_async_to_cpu(host.ctypes.data_as(C.c_void_p), mine, out_bytes, b, 0)
        sb.synchronize()
        assert not host.any(), f"round {round_}: stream A's bootstrap wrote into the block after stream B cleared it"
        lib.cuda_drop(mine, 0)
        sa.synchronize()
    lib.cleanup_cuda_programmable_bootstrap_64(a, 0, C.byref(buf))


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("TFHE_HIP_ARENA_REDZONE", "0") not in ("", "0"),
                    reason="red-zone mode synchronises the device at every drop: there is no cross-stream wait left to count")
def test_a_block_that_another_stream_still_uses_is_ordered_behind_that_stream_too():
    """The reference's cuda_drop is a cudaFree: it waits for the whole device.  Here the drop records an event on every OTHER busy
    stream the library made on the device as well (idle ones need none), so a vector allocated for stream A, still being written by
    a bootstrap on stream B when the host drops it, and handed back to stream A at once — its owner: stream order alone would say
    "free" — is cleared by A only after B's writes (a CudaVec shared across the streams of a set)."""
    from .common import TOY_2048, encrypt_small, make_keys
    from . import oracle as orc
    lib = use_backend("hip")
    p = TOY_2048
    keys = make_keys(p)
    B = 512
    sa, sb = gpu.CudaStreams.new_single_gpu(0), gpu.CudaStreams.new_single_gpu(0)
    a, b = sa.ptr[0], sb.ptr[0]
    bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, sb, ms_noise_reduction=True)
    cts = encrypt_small(p, keys, [m % 16 for m in range(B)], seed=5)
    d_in = gpu.CudaLweCiphertextList.from_lwe_ciphertext_list(cts, sb)
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, lambda x: (x + 1) % 16)
    d_lut = gpu.CudaGlweCiphertextList.from_glwe_ciphertext_list(lut, p.k, p.N, sb)
    idx = gpu.CudaVec.from_cpu_async(np.arange(B, dtype=np.uint64), sb)
    zero = gpu.CudaVec.from_cpu_async(np.zeros(B, dtype=np.uint64), sb)
    buf = C.c_void_p()
    lib.scratch_cuda_programmable_bootstrap_64_async(b, 0, C.byref(buf), p.n, p.k, p.N, p.pbs_level, B, True, p.ms_type)
    out_bytes = B * (p.k * p.N + 1) * 8
    sa.synchronize()
    sb.synchronize()
    lib.hip_backend_trim_allocator(0)
    for round_ in range(4):
        out = lib.cuda_malloc_async(out_bytes, a, 0)       # owner: stream A
        sa.synchronize()
        for _ in range(6):   # a few ms of writes into `out` on stream B
            lib.cuda_programmable_bootstrap_64_async(b, 0, out, idx.ptr, d_lut.d_vec.ptr, zero.ptr, d_in.d_vec.ptr, idx.ptr, bsk.d_vec.ptr,
                                                     buf, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, B, 1, 0)
        before = stats(lib)
        lib.cuda_drop(out, 0)                              # the host lets go while B is still writing
        mine = lib.cuda_malloc_async(out_bytes, a, 0)      # the same block, for its owner stream
        assert mine == out
        assert stats(lib)["cross_stream_waits"] == before["cross_stream_waits"] + 1
        lib.cuda_memset_async(mine, 0, out_bytes, a, 0)
        host = np.empty(out_bytes // 8, dtype=np.uint64)
        lib.cuda_memcpy_async_to_cpu(host.ctypes.data_as(C.c_void_p), mine, out_bytes, a, 0)
        sa.synchronize()
        sb.synchronize()
        assert not host.any(), f"round {round_}: stream B's bootstrap wrote into the block after stream A cleared it"
        lib.cuda_drop(mine, 0)
    lib.cleanup_cuda_programmable_bootstrap_64(b, 0, C.byref(buf))


@pytest.mark.gpu
def test_allocations_inside_a_stream_capture():
    """cuda_malloc_async / cuda_drop between hipStreamBeginCapture and hipStreamEndCapture (global mode): no runtime allocator call
    fails the capture, the captured work replays on the captured addresses, and those blocks never go to another stream."""
    from .test_streams_and_graphs import Hip
    lib = use_backend("hip")
    hip = Hip()
    st, other = gpu.CudaStreams.new_single_gpu(0), gpu.CudaStreams.new_single_gpu(0)
    s, o = st.ptr[0], other.ptr[0]
    n = 1 << 16
    src = gpu.CudaVec.from_cpu_async(np.arange(n, dtype=np.uint64), st)
    dst = gpu.CudaVec(n, st)
    warm = lib.cuda_malloc_async(n * 8, s, 0)   # one cached block of the class (dropped on this stream before the capture)
    lib.cuda_drop(warm, 0)
    st.synchronize()
    seen = []

    def enqueue():
        t1 = lib.cuda_malloc_async(n * 8, s, 0)      # the warm block
        t2 = lib.cuda_malloc_async(n * 8, s, 0)      # nothing cached: hipMalloc in relaxed capture mode
        seen.extend([t1, t2])
        lib.cuda_memcpy_async_gpu_to_gpu(t1, src.ptr, n * 8, s, 0)
        lib.cuda_memcpy_async_gpu_to_gpu(t2, t1, n * 8, s, 0)
        lib.cuda_drop(t1, 0)
        t3 = lib.cuda_malloc_async(n * 8, s, 0)      # t1 again: free at this point of the graph's timeline
        seen.append(t3)
        lib.cuda_memcpy_async_gpu_to_gpu(t3, t2, n * 8, s, 0)
        lib.cuda_memcpy_async_gpu_to_gpu(dst.ptr, t3, n * 8, s, 0)
        lib.cuda_drop(t2, 0)
        lib.cuda_drop(t3, 0)

    graph, exe = hip.capture(s, enqueue)
    try:
        assert seen[0] == warm and seen[2] == seen[0] and seen[1] != seen[0]
        # the graph's blocks are not handed to another stream
        x = [lib.cuda_malloc_async(n * 8, o, 0) for _ in range(3)]
        assert not set(x) & set(seen)
        for rep in range(2):
            lib.cuda_memset_async(dst.ptr, 0, n * 8, s, 0)
            hip.launch(exe, s)
            st.synchronize()
            assert np.array_equal(dst.copy_to_cpu(st), np.arange(n, dtype=np.uint64))
        for p in x:
            lib.cuda_drop(p, 0)
    finally:
        hip.destroy(graph, exe)


@pytest.mark.parametrize("kind", BACKENDS)
def test_host_threads_hammer_the_arena_on_their_own_streams(kind):
    """Four host threads, each on its own stream, allocate / fill / read back / drop blocks of a few shared size classes for a
    while: blocks migrate between the threads' streams through the free lists (event waits), and every read-back must be the
    pattern its own thread wrote — a block handed out while its previous owner's work is still queued would show the other
    thread's pattern."""
    import threading
    lib = use_backend(kind)
    errors = []
    rounds = 60 if kind == "hip" else 25
    live_before = stats(lib)["live_bytes"]

    def worker(tid):
        try:
            st = gpu.CudaStreams.new_single_gpu(0)
            s = st.ptr[0]
            rng = np.random.default_rng(tid)
            for it in range(rounds):
                sizes = [int(rng.choice([4096, 70_000, 1 << 20, (2 << 20) + 8])) for _ in range(3)]
                ptrs = [lib.cuda_malloc_async(n, s, 0) for n in sizes]
                pats = [(tid << 56) | (it << 24) | j for j in range(3)]
                for p, n, pat in zip(ptrs, sizes, pats):
                    host = np.full(n // 8, pat, dtype=np.uint64)
                    lib.cuda_memcpy_async_to_gpu(p, host.ctypes.data_as(C.c_void_p), host.nbytes, s, 0)
                    st.synchronize()   # (the source is pageable host memory)
                outs = []
                for p, n in zip(ptrs, sizes):
                    out = np.empty(n // 8, dtype=np.uint64)
                    lib.cuda_memcpy_async_to_cpu(out.ctypes.data_as(C.c_void_p), p, out.nbytes, s, 0)
                    outs.append(out)
                for p in ptrs:
                    lib.cuda_drop(p, 0)   # dropped while the read-backs may still be queued: stream order must protect them
                st.synchronize()
                for out, pat in zip(outs, pats):
                    if not (out == np.uint64(pat)).all():
                        errors.append((tid, it, hex(pat), hex(int(out[np.argmax(out != np.uint64(pat))]))))
                        return
        except BaseException as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t + 1,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert stats(lib)["live_bytes"] == live_before   # every block came back


@pytest.mark.gpu
def test_a_block_of_a_callers_own_stream_that_is_gone_by_the_time_it_is_dropped():
    """The stream argument is the caller's (a hipStream_t made and destroyed outside the library's cuda_create_stream_ffi /
    cuda_destroy_stream): dropping the block afterwards must not trip over the dead handle — the block comes back idle."""
    lib = use_backend("hip")
    rt = C.CDLL("libamdhip64.so")
    own = C.c_void_p()
    assert rt.hipStreamCreate(C.byref(own)) == 0
    st = gpu.CudaStreams.new_single_gpu(0)
    p = lib.cuda_malloc_async(1 << 18, own, 0)
    lib.cuda_memset_async(p, 0x55, 1 << 18, own, 0)
    assert rt.hipStreamSynchronize(own) == 0 and rt.hipStreamDestroy(own) == 0
    before = stats(lib)
    lib.cuda_drop(p, 0)
    q = lib.cuda_malloc_async(1 << 18, st.ptr[0], 0)
    assert q == p and stats(lib)["cross_stream_waits"] == before["cross_stream_waits"]
    lib.cuda_drop(q, 0)


# ---- debug mode TFHE_HIP_ARENA_REDZONE=1 (arena.hip): slabs, canaries around every block, poisoned payloads.  Its counterpart
# in the reference's tree is compute-sanitizer over the GPU tests (scripts/check_memory_errors.sh:1-60, Makefile:955-962).
# Each case runs in its own interpreter: the mode is read once per process and a finding aborts.
_RZ_PRELUDE = """
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import tfhe_rs_amd
from tfhe_rs_amd import core_crypto_gpu as gpu, ffi
lib = ffi.default_library()
st = gpu.CudaStreams.new_single_gpu(0)
S, G = st.ptr[0], 0
"""

_RZ_CASES = {
    "clean": ("""
        a = lib.cuda_malloc_async(1000, S, G); b = lib.cuda_malloc_async(1000, S, G); c = lib.cuda_malloc(5000, G)
        assert b - a == 1024 + 2 * 4096, (a, b)   # neighbours in one slab, a red zone on either side of each
        src = np.arange(1000, dtype=np.uint8)
        for p in (a, b):
            lib.cuda_memcpy_async_to_gpu(p, src.ctypes.data_as(C.c_void_p), 1000, S, G)
        lib.cuda_synchronize_stream(S, G)
        for p in (a, b, c):
            lib.cuda_drop(p, G)
        a2 = lib.cuda_malloc_async(700, S, G)     # the poisoned block, checked and re-armed for 700 bytes
        assert a2 in (a, b)
        lib.cuda_drop(a2, G)
        assert lib.hip_backend_redzone_checks(G) == 4
        print("clean run ok")
        """, None),
    "one byte past the end": ("""
        a = lib.cuda_malloc_async(1000, S, G)
        src = np.zeros(1001, dtype=np.uint8)
        lib.cuda_memcpy_async_to_gpu(a, src.ctypes.data_as(C.c_void_p), 1001, S, G)
        lib.cuda_synchronize_stream(S, G)
        lib.cuda_drop(a, G)
        """, "was written 1 bytes PAST its last byte"),
    "in front of the payload": ("""
        a = lib.cuda_malloc_async(4096, S, G)
        src = np.zeros(8, dtype=np.uint8)
        lib.cuda_memcpy_async_to_gpu(a - 16, src.ctypes.data_as(C.c_void_p), 8, S, G)
        lib.cuda_synchronize_stream(S, G)
        lib.cuda_drop(a, G)
        """, "was written 16 bytes IN FRONT of its payload"),
    "write after the drop": ("""
        a = lib.cuda_malloc_async(2048, S, G)
        lib.cuda_drop(a, G)
        src = np.zeros(4, dtype=np.uint8)
        lib.cuda_memcpy_async_to_gpu(a + 100, src.ctypes.data_as(C.c_void_p), 4, S, G)
        lib.cuda_synchronize_stream(S, G)
        b = lib.cuda_malloc_async(2048, S, G)
        """, "was written at offset 100 AFTER it had been dropped"),
    "a scratch of the library overrun by its neighbour's owner": ("""
        buf = C.c_void_p()
        lib.scratch_cuda_programmable_bootstrap_64_async(S, G, C.byref(buf), 10, 1, 256, 1, 4, True, 0)
        a = lib.cuda_malloc_async(256, S, G)
        src = np.zeros(256 + 4096 + 8, dtype=np.uint8)   # through a's red zone into the next block's
        lib.cuda_memcpy_async_to_gpu(a, src.ctypes.data_as(C.c_void_p), src.size, S, G)
        lib.cuda_synchronize_stream(S, G)
        lib.cleanup_cuda_programmable_bootstrap_64(S, G, C.byref(buf))
        lib.cuda_drop(a, G)
        """, "arena red zone (drop)"),
}


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("case", list(_RZ_CASES))
def test_red_zone_mode_finds_overruns_and_writes_after_a_drop(kind, case):
    import os
    import signal
    import subprocess
    import sys
    import textwrap
    from .harness import EMU_LIB, build_emu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFHE_HIP_ARENA_REDZONE="1")
    if kind == "emu":
        build_emu()
        env["TFHE_HIP_BACKEND_LIB"] = EMU_LIB
    snippet, message = _RZ_CASES[case]
    r = subprocess.run([sys.executable, "-c", _RZ_PRELUDE % root + textwrap.dedent(snippet)], env=env, capture_output=True,
                       text=True, timeout=600)
    if message is None:
        assert r.returncode == 0, r.stderr[-2000:]
        assert "clean run ok" in r.stdout
    else:
        assert r.returncode == -signal.SIGABRT, (r.returncode, r.stderr[-2000:])
        assert message in r.stderr, r.stderr[-2000:]
