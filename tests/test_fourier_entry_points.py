"""The transform of the path as launches of its own: the reference's backend-test entry points
(backends/tfhe-cuda-backend/cuda/include/pbs/programmable_bootstrap.h:8-45, csrc/fourier.hip), with the reference's tests of
them restated:

  cuda_fft_mult                       tests_and_benchmarks/tests/test_fft.cpp:82-118      (sizes 256 .. 16384, schoolbook, 1e-9)
  fft16x4x16 multiplication           tests_and_benchmarks/tests/test_fft16x4x16.cpp      (N = 2048)
  forward_matches_classic_fft         tests_and_benchmarks/tests/test_forward_fft16x4x16.cpp:118-152  (the permutation, 2^-20)
  test_regression_fft16x4x16          tfhe/src/core_crypto/gpu/algorithms/test/fft/mod.rs:268-294 — the committed GOLDEN spectrum
                                      of the deterministic input of :51-71 (tests/golden/fft16x4x16_golden_v1.json, transcribed by
                                      tests/golden/make_fft_golden.py).  The reference asserts its own kernel's bits on an H100;
                                      another operation order cannot reproduce bits, so the gate is the values: within a few ulp
                                      of the spectrum's scale.  The oracle is pinned on the same vector (CPU tier).

[emu] on the host build, [hip] on the MI355X.  Against the oracle the device transforms are BIT-exact (one transform spec)."""
import json
import os
import signal
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from tfhe_rs_amd import core_crypto_gpu as gpu

from . import oracle as orc
from .harness import use_backend

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1


def golden():
    g = json.load(open(os.path.join(HERE, "golden", "fft16x4x16_golden_v1.json")))
    re_ = np.array([int(x, 16) for x in g["expected_re"]], dtype=np.uint64).view(np.float64)
    im_ = np.array([int(x, 16) for x in g["expected_im"]], dtype=np.uint64).view(np.float64)
    return g["polynomial_size"], re_ + 1j * im_


def reference_input(n=2048):
    """fft16x4x16_reference_input (gpu/algorithms/test/fft/mod.rs:51-71): a bijective hash of the index, as i64 / i64::MAX,
    compressed [re, im, re, im, ...] with complex[i] = (poly[i], poly[i + N/2])."""
    def coeff(k):
        bits = (k * 0x517CC1B727220A95) & M64
        bits = ((bits << 17) | (bits >> 47)) & M64
        bits ^= 0xDEADBEEFCAFEBABE
        s = bits - (1 << 64) if bits >= (1 << 63) else bits
        return float(s) / float((1 << 63) - 1)
    half = n // 2
    x = np.zeros(n)
    x[0::2] = [coeff(i) for i in range(half)]
    x[1::2] = [coeff(i + half) for i in range(half)]
    return x


def bitreverse(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def classic_index(f, half=1024):
    """test_forward_fft16x4x16.cpp:24-31: natural frequency f -> index in the classic transform's native order"""
    return bitreverse((half - f) % half, half.bit_length() - 1)


def compress(p):
    n = len(p)
    c = np.zeros(n)
    c[0::2], c[1::2] = p[:n // 2], p[n // 2:]
    return c


def decompress(c):
    return np.concatenate([c[0::2], c[1::2]])


def negacyclic_schoolbook(a, b):
    """test_fft.cpp:52-69"""
    n = len(a)
    full = np.convolve(a, b)
    res = full[:n].copy()
    res[:n - 1] -= full[n:]
    return res


def run(lib, st, name, n, *arrays, total=1):
    """one of the forward / backward entry points on `total` compressed polynomials"""
    d = [gpu.CudaVec.from_cpu_async(np.ascontiguousarray(a, dtype=np.float64).view(np.uint64), st) for a in arrays]
    out = gpu.CudaVec(n * total, st)
    getattr(lib, name)(st.ptr[0], 0, *[v.ptr for v in d], out.ptr, n, total)
    return out.copy_to_cpu(st).view(np.float64)


def test_oracle_forward_transform_matches_the_reference_golden_spectrum():
    """oracle pin: the restated transform at the reference's own golden vector — tree index bitreverse((n - f) mod n) holds
    natural frequency f, value within a few ulp of the spectrum's scale of what the reference's kernel produced"""
    n, G = golden()
    t = orc.fft_forward_f64(reference_input(n))
    T = t[0::2] + 1j * t[1::2]
    nat = np.array([T[classic_index(f)] for f in range(n // 2)])
    scale = np.max(np.abs(G))
    assert np.max(np.abs(nat - G)) < 64 * np.finfo(np.float64).eps * scale, np.max(np.abs(nat - G)) / scale
    # ... and against the definition, in extended precision: F[f] = sum_k (p[k] + i p[k + n]) e^{i pi k / N} e^{-2 pi i k f / n}
    x = reference_input(n)
    z = (x[0::2] + 1j * x[1::2]).astype(np.clongdouble)
    k = np.arange(n // 2, dtype=np.longdouble)
    zz = z * np.exp(1j * np.pi * k / n)
    for f in (0, 1, 2, 511, 512, 1023):
        direct = np.sum(zz * np.exp(-2j * np.pi * k * f / (n // 2)))
        assert abs(complex(direct) - G[f]) < 1e-11 * scale


@pytest.mark.parametrize("kind", BACKENDS)
def test_regression_fft16x4x16_against_the_reference_golden_spectrum(kind):
    lib = use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)
    assert lib.cuda_fft16x4x16_is_supported_async(0) is True
    n, G = golden()
    x = reference_input(n)
    d_input = gpu.CudaVec.from_cpu_async(x.view(np.uint64), st)     # run_fft16x4x16_forward (mod.rs:76-103) on the host mirror
    d_output = gpu.CudaVec(n, st)
    gpu.forward_fft16x4x16_async(st, d_input, d_output, n, 1)
    out = d_output.copy_to_cpu(st).view(np.float64)
    F = out[0::2] + 1j * out[1::2]
    scale = np.max(np.abs(G))
    assert np.max(np.abs(F - G)) < 64 * np.finfo(np.float64).eps * scale, np.max(np.abs(F - G)) / scale
    # the oracle's bits in the natural order
    t = orc.fft_forward_f64(x)
    T = t[0::2] + 1j * t[1::2]
    assert np.array_equal(F, np.array([T[classic_index(f)] for f in range(n // 2)]))


@pytest.mark.parametrize("kind", BACKENDS)
def test_forward_matches_classic_fft_and_backward_inverts(kind):
    """test_forward_fft16x4x16.cpp: random polynomials in [-1, 1), both forward entry points, the permutation between their
    orders (tolerance 2^-20 there; one transform here: equal bits).  backward_fft16x4x16 is the inverse without the 1/1024."""
    lib = use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)
    n, half, samples = 2048, 1024, 100 if kind == "hip" else 3
    rng = np.random.default_rng(11)
    x = np.concatenate([compress(rng.uniform(-1, 1, n)) for _ in range(samples)])
    nat = run(lib, st, "cuda_forward_fft16x4x16_async", n, x, total=samples).reshape(samples, half, 2)
    cla = run(lib, st, "cuda_forward_fft_classic_async", n, x, total=samples).reshape(samples, half, 2)
    perm = np.array([classic_index(f) for f in range(half)])
    assert np.array_equal(nat, cla[:, perm, :])
    assert np.array_equal(cla.reshape(samples, -1), np.stack([orc.fft_forward_f64(x[s * n:(s + 1) * n]) for s in range(samples)]))
    back = run(lib, st, "cuda_backward_fft16x4x16_async", n, nat.reshape(-1), total=samples)
    assert np.max(np.abs(back / half - x)) < 1e-12


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("n,samples", [(256, 100), (512, 100), (1024, 100), (2048, 100), (4096, 100), (8192, 50), (16384, 10)])
def test_cuda_fft_mult(kind, n, samples):
    """test_fft.cpp cuda_fft_mult with its parameter list: the product of two random polynomials in [-1, 1) against the
    schoolbook negacyclic product within 1e-9 (the reference's EXPECT_NEAR), output aliased onto input2 as there; bit for bit
    the oracle's.  input1 comes back holding its spectrum (the reference's documented side effect)."""
    lib = use_backend(kind)
    st = gpu.CudaStreams.new_single_gpu(0)
    if kind == "emu":
        if n > 4096:
            pytest.skip("the host emulation covers the sizes up to 4096; the larger ones run on the GPU")
        samples = 2
    rng = np.random.default_rng(n)
    a = [rng.uniform(-1, 1, n) for _ in range(samples)]
    b = [rng.uniform(-1, 1, n) for _ in range(samples)]
    ca, cb = np.concatenate([compress(p) for p in a]), np.concatenate([compress(p) for p in b])
    d1 = gpu.CudaVec.from_cpu_async(ca.view(np.uint64), st)
    d2 = gpu.CudaVec.from_cpu_async(cb.view(np.uint64), st)
    lib.cuda_fourier_polynomial_mul_async(st.ptr[0], 0, d1.ptr, d2.ptr, d2.ptr, n, samples)
    got = d2.copy_to_cpu(st).view(np.float64).reshape(samples, n)
    spec = d1.copy_to_cpu(st).view(np.float64).reshape(samples, n)
    for s in range(samples if n <= 4096 else 2):
        assert np.max(np.abs(decompress(got[s]) - negacyclic_schoolbook(a[s], b[s]))) < 1e-9
    for s in range(min(samples, 4)):
        assert np.array_equal(got[s], orc.fft_polynomial_mul_f64(compress(a[s]), compress(b[s])))
        assert np.array_equal(spec[s], orc.fft_forward_f64(compress(a[s])))
    if n == 2048:   # test_fft16x4x16.cpp: the same product through the throughput transform's entry point
        d1 = gpu.CudaVec.from_cpu_async(ca.view(np.uint64), st)
        d2 = gpu.CudaVec.from_cpu_async(cb.view(np.uint64), st)
        out = gpu.CudaVec(n * samples, st)
        lib.cuda_fourier_polynomial_mul_fft16x4x16_async(st.ptr[0], 0, d1.ptr, d2.ptr, out.ptr, n, samples)
        assert np.array_equal(out.copy_to_cpu(st).view(np.float64).reshape(samples, n), got)


@pytest.mark.parametrize("name", ["cuda_forward_fft16x4x16_async", "cuda_forward_fft_classic_async", "cuda_backward_fft16x4x16_async"])
def test_the_2048_only_entry_points_abort_on_other_sizes_like_the_reference(name):
    """cuda/src/pbs/bootstrapping_key.cu:408-410, :448-450, :489-491: PANIC("... only supports polynomial_size == 2048")"""
    code = textwrap.dedent(f"""
        import numpy as np
        from tests.harness import use_backend
        from tfhe_rs_amd import core_crypto_gpu as gpu
        lib = use_backend("emu")
        st = gpu.CudaStreams.new_single_gpu(0)
        a, b = gpu.CudaVec(1024, st), gpu.CudaVec(1024, st)
        lib.{name}(st.ptr[0], 0, a.ptr, b.ptr, 1024, 1)
        """)
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=300)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stderr[-300:])
    assert f"{name} only supports polynomial_size == 2048" in r.stderr
