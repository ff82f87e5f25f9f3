"""The reference's statistical test of the centered-mean modulus switch (tfhe/src/core_crypto/gpu/algorithms/test/
modulus_switch.rs:130-275 `check_centered_modulus_switch_is_centered`; CPU twin algorithms/test/
modulus_switch_noise_reduction.rs): noise-free encryptions of 0 under a key of half Hamming weight (n = 800), switched to
2^12, decrypted to a signed LUT index; with a redundancy of two the index falls left (< -1) or right (>= 1) of the box.  The
plain switch (fft_impl/common.rs:10-23) is NOT centered — its left and right error probabilities differ by more than
max_ratio = 1.05 — the centered switch (algorithms/modulus_switch.rs:57-103) is.  This is the reference's acceptance of the
switch PARAM_MESSAGE_2_CARRY_2 uses, for which no reference bytes exist.

The reference runs 1,000,000 single-ciphertext launches per algorithm; the GPU's switch equals the oracle's word for word
(tests/cpp/reference_gpu_tests.cpp compare_cpu_and_gpu_centered_modulus_switch, tests/test_pins_extra.py), so the statistic
is taken here on a vectorised exact-integer restatement (numpy, checked against the oracle below) over 200,000 ciphertexts per algorithm."""
import numpy as np

from . import oracle as orc

N_LWE, LOG_MOD, SHIFT = 800, 12, 52
MAX_RATIO = 1.05
HALF_REDUNDANCY = 1


def ms(x):
    return (x + np.uint64(1 << (SHIFT - 1))) >> np.uint64(SHIFT)


def switch_batch(a, b, centered):
    """a [B][n] u64 masks, b [B] u64 bodies -> switched masks and bodies (values in [0, 2^12))"""
    a_hat = ms(a)
    if not centered:
        return a_hat, ms(b)
    e = ((a_hat << np.uint64(SHIFT)) - a).astype(np.int64)          # s((ms(a) << 52) - a)
    h = np.where(e < 0, -((-e) // 2), e // 2)                       # Rust's `/ 2`: toward zero
    H = h.astype(np.uint64).sum(axis=1, dtype=np.uint64)            # wrapping u64
    D = (2 * h - e).sum(axis=1, dtype=np.int64)
    half_d = np.where(D < 0, -((-D) // 2), D // 2)
    corr = H - half_d.astype(np.uint64) - np.uint64(1 << (SHIFT - 1))
    return a_hat, ms(b + corr)


def test_vectorised_switch_equals_the_oracle():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 1 << 64, size=(200, N_LWE), dtype=np.uint64)
    b = rng.integers(0, 1 << 64, size=200, dtype=np.uint64)
    for centered in (False, True):
        a_hat, b_hat = switch_batch(a, b, centered)
        for i in range(200):
            want = orc.lwe_modulus_switch(np.concatenate([a[i], b[i:i + 1]]), LOG_MOD, 1 if centered else 0)
            assert np.array_equal(want[:-1], a_hat[i]) and want[-1] == b_hat[i], (centered, i)


def error_probabilities(centered, number_loops, seed):
    rng = np.random.default_rng(seed)
    sk = np.zeros(N_LWE, dtype=bool)
    sk[::2] = True                                                   # sk.iter_mut().step_by(2)
    left = right = 0
    for _ in range(number_loops // 20000):
        a = rng.integers(0, 1 << 64, size=(20000, N_LWE), dtype=np.uint64)
        b = a[:, sk].sum(axis=1, dtype=np.uint64)                    # Plaintext(0), noise 0
        a_hat, b_hat = switch_batch(a, b, centered)
        lut_index = (b_hat - a_hat[:, sk].sum(axis=1, dtype=np.uint64)) % np.uint64(1 << LOG_MOD)
        signed = (lut_index << np.uint64(64 - LOG_MOD)).astype(np.int64) >> np.int64(64 - LOG_MOD)
        left += int(np.count_nonzero(signed < -HALF_REDUNDANCY))
        right += int(np.count_nonzero(signed >= HALF_REDUNDANCY))
    return left / number_loops, right / number_loops


def check_both_ratio_under(a, b, max_ratio):
    return a / b < max_ratio and b / a < max_ratio


def test_check_centered_modulus_switch_is_centered():
    number_loops = 200_000
    p_left, p_right = error_probabilities(False, number_loops, seed=11)
    print(f"regular: p_left_error={p_left}, p_right_error={p_right}")
    assert not check_both_ratio_under(p_left, p_right, MAX_RATIO)   # "does do half case correction so should fail this check"
    p_left, p_right = error_probabilities(True, number_loops, seed=12)
    print(f"centered: p_left_error={p_left}, p_right_error={p_right}")
    assert check_both_ratio_under(p_left, p_right, MAX_RATIO)
