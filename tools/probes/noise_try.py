#!/usr/bin/env python3
"""Probe behind tests/test_pbs_noise.py: the reference's classic noise-distribution test (16 x 1000 bootstraps of
NOISE_TEST_PARAMS_4_BITS_NATIVE_U64_132_BITS_GAUSSIAN) through the oracle's f64 engine, with the variance at 2,000 / 4,000 /
16,000 samples.   python tools/probes/noise_try.py   (about two minutes on 8 cores)"""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, os.pardir)))
import time, math, numpy as np
from tests import oracle as orc
from tests.common import Params, Keys
from tests.harness import oracle_pbs
M64 = (1 << 64) - 1
n, k, N, bl, lv = 841, 1, 2048, 22, 1
lwe_std, glwe_std = 3.1496674685772435e-06, 2.845267479601915e-15
def pbs_variance(n, k, N, B, l, mant, q):
    ln = math.log
    return n * (0.00705 * 2.0 ** (2 * (0.0 if mant - math.log2(math.e) * ln(q) >= 0 else -mant + math.log2(math.e) * ln(q)) + 2.88539008177793 * ln(B) - 2.88539008177793 * ln(q)) * l ** 1.01827 * k ** 1.22003 * N ** 2.22003 * (k + 1) ** 1.01827
                + l * N * (2.0 ** (4.0 - 2.88539008177793 * ln(q)) + 2.0 ** (-0.0497829131652661 * k * N + 5.31469187675068)) * ((1 / 12.0) * B ** 2 + 0.166666666666667) * (k + 1)
                - 1 / 24.0 * q ** -2.0 + 0.5 * k * N * (0.0208333333333333 * q ** -2.0 + 0.0416666666666667 * B ** (-2.0 * l)) + (1 / 24.0) * B ** (-2.0 * l))
def min_var(dim, q):
    return 2.0 ** (4.0 - 2.88539008177793 * math.log(q)) + 2.0 ** (5.31469187675068 - 0.0497829131652661 * dim)
q = 2.0 ** 64
exp_var = pbs_variance(n, k, N, 2.0 ** bl, lv, 53.0, q)
print("expected", exp_var, "minimal", min_var(k * N, q), "glwe var", glwe_std ** 2, "min glwe", min_var(k*N, q))
rng = np.random.default_rng(2024)
small = np.zeros(n, dtype=np.uint64); small[::2] = 1
big = np.zeros(k * N, dtype=np.uint64); big[::2] = 1
def gauss(std, size):
    return np.rint(rng.standard_normal(size) * std * 2.0 ** 64).astype(np.int64).astype(np.uint64)
t = time.time()
rows = []
for i in range(n):
    factor = ((-int(small[i])) << (64 - bl)) & M64
    for row in range(k + 1):
        a = rng.integers(0, 1 << 64, size=k * N, dtype=np.uint64)
        body = gauss(glwe_std, N)
        if row < k:
            body = body + (big[row * N:(row + 1) * N] * np.uint64(factor))
        else:
            body[0] += np.uint64((-factor) & M64)
        for j in range(k):
            orc.negacyclic_mul_add(body, big[j * N:(j + 1) * N].astype(np.int64), a[j * N:(j + 1) * N])
        rows.append(np.concatenate([a, body]))
bsk = np.concatenate(rows)
print("bsk", time.time() - t)
p = Params("noise", n, k, N, bl, lv, 3, 5, 0, 0, 16, ms_type=0)
keys = Keys(p, small, big, bsk, np.zeros(0, dtype=np.uint64))
lut = orc.generate_lut(k, N, 16, 1 << 59, lambda x: x)
delta = 1 << 59
samples = []
NB = 1000
t = time.time()
for msg in range(15, -1, -1):
    a = rng.integers(0, 1 << 64, size=(NB, n), dtype=np.uint64)
    body = gauss(lwe_std, NB) + a[:, small == 1].sum(axis=1, dtype=np.uint64) + np.uint64(msg * delta)
    cts = np.concatenate([a, body[:, None]], axis=1)
    out = oracle_pbs(p, keys, "fft64", cts, lut)
    dec = out[:, -1] - out[:, :-1][:, big == 1].sum(axis=1, dtype=np.uint64)
    decoded = ((dec + np.uint64(delta // 2)) >> np.uint64(59)) % np.uint64(16)
    assert np.all(decoded == msg), (msg, np.count_nonzero(decoded != msg))
    diff = (dec - np.uint64(msg * delta)).astype(np.int64).astype(np.float64) / 2.0 ** 64
    samples.append(diff)
print("pbs", time.time() - t)
s = np.concatenate(samples)
for cnt in (2000, 4000, 16000):
    v = s[:: len(s) // cnt].var(ddof=1)
    print(cnt, "measured", v, "ratio", v / exp_var)
