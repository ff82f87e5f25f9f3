"""Radix-integer layer, second tranche (round 6): subtraction, negation, scalar addition, bitwise operations (ciphertext
and scalar forms), bitwise NOT, full propagation — the reference's entry points of cuda/include/integer/integer.h:159-171,
189-198, 312-347, 559-573 on the round driver of the first tranche.

The checker is clear arithmetic on the decrypted blocks, as in the reference's own tests of these operations
(tfhe/src/integer/gpu/server_key/radix/tests_unsigned/{test_sub.rs,test_neg.rs,test_scalar_add.rs,test_bitwise_op.rs,
test_scalar_bitwise_op.rs}: encrypt, operate, decrypt, compare with the clear result).
[emu] runs the kernel sources on the host with a toy key, [hip] on the MI355X with PARAM_MESSAGE_2_CARRY_2."""
import numpy as np
import pytest

from .harness import use_backend
from .test_radix_integer import BACKENDS, MSG, decrypt_blocks, encrypt_radix, recompose, setup


def digits(v, n):
    return [(int(v) >> (2 * j)) & 3 for j in range(n)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_sub_with_and_without_borrow(kind):
    p, keys, st, sks, igpu = setup(kind)
    L = 9 if kind == "emu" else 32
    bits = 2 * L
    mask = (1 << bits) - 1
    rng = np.random.default_rng(91)
    nrand = 1 if kind == "emu" else 4
    a = [int.from_bytes(rng.bytes(8), "little") & mask for _ in range(nrand)] + [0, mask, 1 << (bits - 1), 5]
    b = [int.from_bytes(rng.bytes(8), "little") & mask for _ in range(nrand)] + [1, mask, 1, 5]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 61), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 62), st)
    ca.set_degrees(MSG - 1)
    cb.set_degrees(MSG - 1)
    cout = sks.sub_assign(ca, cb, st, want_carry_out=True)
    rows = decrypt_blocks(p, keys, ca.to_blocks(st))
    assert all(d < MSG for r in rows for d in r)
    assert recompose(rows) == [(x - y) & mask for x, y in zip(a, b)]
    # the carry of lhs + (2^bits - rhs): 1 exactly when no borrow occurred
    assert [r[0] for r in decrypt_blocks(p, keys, cout.to_blocks(st))] == [int(x >= y) for x, y in zip(a, b)]
    assert list(ca.degrees) == [MSG - 1] * ca.total_blocks
    # rhs is untouched
    assert recompose(decrypt_blocks(p, keys, cb.to_blocks(st))) == b


@pytest.mark.parametrize("kind", BACKENDS)
def test_sub_refuses_what_is_not_wired(kind):
    """An input carry and FLAG_OVERFLOW are refused loudly (HX_PANIC aborts the process): checked on the host side of
    the entry point by the degrees it accepts — dirty operands are refused as well; here only the accepting side is run
    (an abort cannot be caught in-process), the refusal text is in the library's source."""
    p, keys, st, sks, igpu = setup(kind)
    a, b = [7], [9]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, 3, 63), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, 3, 64), st)
    assert sks.sub_assign(ca, cb, st) is None
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [(7 - 9) & 63]


@pytest.mark.parametrize("kind", BACKENDS)
def test_negation_with_correcting_term_then_propagation(kind):
    p, keys, st, sks, igpu = setup(kind)
    L = 5 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    for seed, v in ((71, 0x2D3 & mask), (72, 0), (73, mask), (74, 1)):
        ct = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [v], L, seed), st)
        ct.set_degrees(MSG - 1)
        neg = sks.unchecked_neg(ct, st)
        raw = decrypt_blocks(p, keys, neg.to_blocks(st))[0]
        d = digits(v, L)
        # negation.cuh:20-49: block 0 = msg - b0, the others msg - 1 - b_i (the unit borrowed from them)
        assert raw == [MSG - d[0]] + [MSG - 1 - x for x in d[1:]]
        # the reference's degree loop: [msg, msg - 1, msg - 1, ...]
        assert list(neg.degrees) == [MSG] + [MSG - 1] * (L - 1)
        sks.full_propagate_assign(neg, st)
        rows = decrypt_blocks(p, keys, neg.to_blocks(st))
        assert all(x < MSG for x in rows[0])
        assert recompose(rows) == [(-v) & mask]
        # input untouched
        assert recompose(decrypt_blocks(p, keys, ct.to_blocks(st))) == [v]


@pytest.mark.parametrize("kind", BACKENDS)
def test_scalar_addition_and_full_propagation(kind):
    p, keys, st, sks, igpu = setup(kind)
    L = 6 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    v, sc = 0x9A7F3C21E5B4D608 & mask, 0xF3C5A9E7B1D2486F & mask
    ct = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [v], L, 81), st)
    ct.set_degrees(MSG - 1)
    clear = digits(sc, L)
    sks.unchecked_scalar_add_assign(ct, clear, st)
    raw = decrypt_blocks(p, keys, ct.to_blocks(st))[0]
    assert raw == [x + y for x, y in zip(digits(v, L), clear)]
    assert list(ct.degrees) == [MSG - 1 + c for c in clear]
    # fewer scalars than blocks: only the first ones move
    ct2 = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [v], L, 82), st)
    sks.unchecked_scalar_add_assign(ct2, [3, 2], st)
    assert decrypt_blocks(p, keys, ct2.to_blocks(st))[0] == [digits(v, L)[0] + 3, digits(v, L)[1] + 2] + digits(v, L)[2:]
    sks.full_propagate_assign(ct, st)
    rows = decrypt_blocks(p, keys, ct.to_blocks(st))
    assert all(x < MSG for x in rows[0])
    assert recompose(rows) == [(v + sc) & mask]
    assert list(ct.degrees) == [MSG - 1] * L


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("op", ["and", "or", "xor"])
def test_bitwise_operations_on_ciphertexts(kind, op):
    p, keys, st, sks, igpu = setup(kind)
    L = 8 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a, b = 0xC3A5F00F9E3779B9 & mask, 0x5A3CFF00C2B2AE35 & mask
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 101), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [b], L, 102), st)
    ca.set_degrees(MSG - 1)
    cb.set_degrees(MSG - 1)
    sks.bitop_assign(ca, cb, op, st)
    want = {"and": a & b, "or": a | b, "xor": a ^ b}[op]
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [want]
    assert list(ca.degrees) == [MSG - 1] * L  # bitwise_ops.cu:133-185 on (3, 3)
    assert recompose(decrypt_blocks(p, keys, cb.to_blocks(st))) == [b]


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("op", ["and", "or", "xor"])
def test_bitwise_operations_with_a_scalar(kind, op):
    p, keys, st, sks, igpu = setup(kind)
    L = 8 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a, sc = 0x9E3779B97F4A7C15 & mask, 0x6C62272E07BB0142 & mask
    # the scalar decomposes into fewer blocks than the ciphertext: AND clears the rest, OR / XOR leave it
    nclear = L - 3
    sc &= (1 << (2 * nclear)) - 1
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 111), st)
    ca.set_degrees(MSG - 1)
    sks.scalar_bitop_assign(ca, digits(sc, nclear), op, st)
    want = {"and": a & sc, "or": a | sc, "xor": a ^ sc}[op]
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [want]
    if op == "and":
        assert list(ca.degrees[nclear:]) == [0, 0, 0]
        assert list(ca.degrees[:nclear]) == [min(c, MSG - 1) for c in digits(sc, nclear)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_bitnot_is_its_own_inverse(kind):
    p, keys, st, sks, igpu = setup(kind)
    L = 7 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a = 0x0123456789ABCDEF & mask
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 121), st)
    sks.bitnot_assign(ca, st)
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [a ^ mask]
    assert list(ca.degrees) == [MSG - 1] * L
    sks.bitnot_assign(ca, st)
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [a]


@pytest.mark.parametrize("kind", BACKENDS)
def test_size_queries_allocate_nothing(kind):
    """allocate_gpu_memory = false (gpu/ffi.rs "size on gpu" queries) reports the bytes of the new scratches."""
    import ctypes as C
    p, keys, st, sks, igpu = setup(kind)
    lib = igpu._lib()
    s, keep = sks._streams(st)
    for name, args in (("integer_bitop_inplace", (16, MSG, MSG, 1)), ("integer_scalar_bitop_inplace", (16, MSG, MSG, 4)),
                       ("sub_and_propagate_single_carry_64_inplace", (16, MSG, MSG, 0))):
        mem = C.c_void_p()
        fn = getattr(lib, f"scratch_cuda_{name}_64_async" if "sub" not in name else f"scratch_cuda_{name}_async")
        size = fn(s, C.byref(mem), sks._bsk_params(), sks._ksk_params(), *args, False, sks._noise_reduction())
        assert size > (p.k + 1) * p.N * 8  # at least its lookup table(s)
        getattr(lib, f"cleanup_cuda_{name}_64" if "sub" not in name else f"cleanup_cuda_{name}")(s, C.byref(mem))
    for scratch, cleanup, args in (
            ("scratch_cuda_integer_comparison_64_async", "cleanup_cuda_integer_comparison_64", (16, MSG, MSG, 0, False)),
            ("scratch_cuda_integer_comparison_64_async", "cleanup_cuda_integer_comparison_64", (16, MSG, MSG, 2, False)),
            ("scratch_cuda_integer_comparison_64_async", "cleanup_cuda_integer_comparison_64", (16, MSG, MSG, 6, False)),
            ("scratch_cuda_integer_scalar_comparison_64_async", "cleanup_cuda_integer_scalar_comparison_64", (16, MSG, MSG, 3, False)),
            ("scratch_cuda_cmux_64_async", "cleanup_cuda_cmux_64", (16, MSG, MSG)),
            ("scratch_cuda_logical_scalar_shift_64_inplace_async", "cleanup_cuda_logical_scalar_shift_64_inplace", (16, MSG, MSG, 1)),
            ("scratch_cuda_integer_overflowing_sub_64_inplace_async", "cleanup_cuda_integer_overflowing_sub_64_inplace", (16, MSG, MSG, 1))):
        mem = C.c_void_p()
        size = getattr(lib, scratch)(s, C.byref(mem), sks._bsk_params(), sks._ksk_params(), *args, False, sks._noise_reduction())
        assert size > (p.k + 1) * p.N * 8, scratch
        getattr(lib, cleanup)(s, C.byref(mem))
    mem = C.c_void_p()
    size = lib.scratch_cuda_full_propagation_64_inplace_async(s, C.byref(mem), sks._bsk_params(), sks._ksk_params(), MSG,
                                                               MSG, False, sks._noise_reduction())
    assert size > 2 * (p.big_n + 1) * 8
    lib.cleanup_cuda_full_propagation_64_inplace(s, C.byref(mem))


@pytest.mark.parametrize("kind", BACKENDS)
def test_unsigned_comparisons(kind):
    """eq / ne over sums of block results, the orderings over the subtraction's output carry; operands that differ in the
    lowest block only, in the highest only, equal ones, and the extremes."""
    p, keys, st, sks, igpu = setup(kind)
    for L in ((1, 5) if kind == "emu" else (1, 2, 16, 17, 32)):
        mask = (1 << (2 * L)) - 1
        rng = np.random.default_rng(131 + L)
        r1, r2 = (int.from_bytes(rng.bytes(8), "little") & mask for _ in range(2))
        pairs = [(r1, r2), (r1, r1), (0, mask), (mask, 0), (r1, r1 ^ 1), (r1, r1 ^ (1 << (2 * L - 1)))]
        if kind == "emu":
            pairs = pairs[:2] + pairs[4:]
        for n, (a, b) in enumerate(pairs):
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 140 + n), st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [b], L, 150 + n), st)
            ca.set_degrees(MSG - 1)
            cb.set_degrees(MSG - 1)
            ops = ("eq", "ne", "gt", "ge", "lt", "le") if (kind != "emu" or n < 2) else ("eq", "lt")
            for op in ops:
                out = sks.compare(ca, cb, op, st)
                want = {"eq": a == b, "ne": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[op]
                assert decrypt_blocks(p, keys, out.to_blocks(st)) == [[int(want)]], (L, a, b, op)
                assert list(out.degrees) == [1]
            # the operands are untouched
            assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [a]
            assert recompose(decrypt_blocks(p, keys, cb.to_blocks(st))) == [b]


@pytest.mark.parametrize("kind", BACKENDS)
def test_if_then_else_and_max_min(kind):
    p, keys, st, sks, igpu = setup(kind)
    L = 4 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a, b = 0x93C467E37DB0C7A4 & mask, 0xD1B54A32D192ED03 & mask
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 161), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [b], L, 162), st)
    for c in (0, 1):
        cond = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [c], 1, 163 + c), st)
        out = sks.if_then_else(cond, ca, cb, st)
        rows = decrypt_blocks(p, keys, out.to_blocks(st))
        assert all(x < MSG for x in rows[0])
        assert recompose(rows) == [a if c else b]
    for op, want in (("max", max(a, b)), ("min", min(a, b))):
        out = sks.compare(ca, cb, op, st)
        assert recompose(decrypt_blocks(p, keys, out.to_blocks(st))) == [want]
        out = sks.compare(cb, ca, op, st)
        assert recompose(decrypt_blocks(p, keys, out.to_blocks(st))) == [want]
    out = sks.compare(ca, ca, "max", st)
    assert recompose(decrypt_blocks(p, keys, out.to_blocks(st))) == [a]


@pytest.mark.parametrize("kind", BACKENDS)
def test_logical_shifts_by_a_clear_amount(kind):
    """Whole-block moves, bit shifts inside the blocks, both at once, no shift, and the overshift that clears the integer."""
    p, keys, st, sks, igpu = setup(kind)
    L = 6 if kind == "emu" else 32
    bits = 2 * L
    mask = (1 << bits) - 1
    a = 0xB7E151628AED2A6B & mask
    shifts = (0, 1, 2, 5, bits - 1, bits, bits + 3) if kind == "emu" else (0, 1, 2, 3, 8, 17, 31, 32, 33, 62, 63, 64, 65)
    for left in (True, False):
        for sh in shifts:
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 171 + sh), st)
            ca.set_degrees(MSG - 1)
            sks.scalar_shift_assign(ca, sh, st, left=left)
            rows = decrypt_blocks(p, keys, ca.to_blocks(st))
            assert all(x < MSG for x in rows[0])
            want = ((a << sh) & mask) if left else (a >> sh)
            assert recompose(rows) == [want], (left, sh)
            q = min(sh // 2, L)
            zeroed = list(ca.degrees[:q]) if left else list(ca.degrees[L - q:])
            assert zeroed == [0] * q, (left, sh)


@pytest.mark.parametrize("kind", BACKENDS)
def test_comparisons_with_a_clear_scalar(kind):
    """The scalar arrives as its clear blocks up to the last non-zero one (fewer than the ciphertext's, none for zero)."""
    p, keys, st, sks, igpu = setup(kind)
    L = 5 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a = 0x6A09E667F3BCC908 & mask
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, [a], L, 181), st)
    ca.set_degrees(MSG - 1)
    scalars = (a, 0, a + 1, 3) if kind == "emu" else (a, 0, a + 1, a - 1, 3, mask, a ^ (1 << (2 * L - 1)))
    for sc in scalars:
        for op in (("eq", "gt", "le") if kind == "emu" else ("eq", "ne", "gt", "ge", "lt", "le")):
            out = sks.scalar_compare(ca, sc, op, st)
            want = {"eq": a == sc, "ne": a != sc, "gt": a > sc, "ge": a >= sc, "lt": a < sc, "le": a <= sc}[op]
            assert decrypt_blocks(p, keys, out.to_blocks(st)) == [[int(want)]], (sc, op)
    out = sks.scalar_compare(ca, 3, "max", st)
    assert recompose(decrypt_blocks(p, keys, out.to_blocks(st))) == [max(a, 3)]
    out = sks.scalar_compare(ca, 3, "min", st)
    assert recompose(decrypt_blocks(p, keys, out.to_blocks(st))) == [3]


@pytest.mark.parametrize("kind", BACKENDS)
def test_overflowing_sub_returns_the_borrow(kind):
    p, keys, st, sks, igpu = setup(kind)
    L = 4 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a = [5, 0x8000000000000001 & mask, 0, mask]
    b = [9, 0x8000000000000000 & mask, 0, mask]
    if kind == "emu":
        a, b = a[:2], b[:2]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 191), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 192), st)
    borrow = sks.unsigned_overflowing_sub_assign(ca, cb, st)
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [(x - y) & mask for x, y in zip(a, b)]
    assert [r[0] for r in decrypt_blocks(p, keys, borrow.to_blocks(st))] == [int(x < y) for x, y in zip(a, b)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_second_tranche_rounds_sharded_over_the_streams_of_the_set(kind):
    """The rounds of sub / bitxor / gt / eq / if_then_else / shift split over three streams of the CudaStreamsFFI (threshold of 3
    blocks per GPU: ragged shards, peer copies, events — helper_multi_gpu.cuh:170-294) must equal the single-stream run bit for
    bit and decrypt to the clear results."""
    p, keys, st1, sks1, igpu = setup(kind)
    _, _, st3, sks3, _ = setup(kind, gpu_indexes=(0, 0, 0))
    lib = use_backend(kind)
    L = 5 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    a, b = 0xD6E8FEB86659FD93 & mask, 0x3C6EF372FE94F82B & mask
    blocks_a, blocks_b = encrypt_radix(p, keys, [a], L, 201), encrypt_radix(p, keys, [b], L, 202)
    cond_blocks = encrypt_radix(p, keys, [1], 1, 203)
    outs = {}
    for name, st, sks, thr in (("one", st1, sks1, 512), ("three", st3, sks3, 3)):
        lib.hip_integer_set_multi_gpu_threshold(thr)
        try:
            mk = lambda blk: igpu.CudaUnsignedRadixCiphertext.from_blocks(blk, st)
            d, x, sh = mk(blocks_a), mk(blocks_a), mk(blocks_a)
            cb = mk(blocks_b)
            sks.sub_assign(d, cb, st)
            sks.bitop_assign(x, cb, "xor", st)
            gt = sks.compare(mk(blocks_a), cb, "gt", st)
            eq = sks.compare(mk(blocks_a), cb, "eq", st)
            sel = sks.if_then_else(mk(cond_blocks), mk(blocks_a), cb, st)
            sks.scalar_shift_assign(sh, 3, st, left=True)
            outs[name] = [c.to_blocks(st) for c in (d, x, gt, eq, sel, sh)]
        finally:
            lib.hip_integer_set_multi_gpu_threshold(0)
    for one, three in zip(outs["one"], outs["three"]):
        assert np.array_equal(one, three)
    d, x, gt, eq, sel, sh = outs["three"]
    assert recompose(decrypt_blocks(p, keys, d)) == [(a - b) & mask]
    assert recompose(decrypt_blocks(p, keys, x)) == [a ^ b]
    assert decrypt_blocks(p, keys, gt) == [[int(a > b)]] and decrypt_blocks(p, keys, eq) == [[int(a == b)]]
    assert recompose(decrypt_blocks(p, keys, sel)) == [a]
    assert recompose(decrypt_blocks(p, keys, sh)) == [(a << 3) & mask]
