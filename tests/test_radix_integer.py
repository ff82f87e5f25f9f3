"""Radix-integer layer ("next" row N1): batched apply-LUT rounds, carry propagation, add, mul.

The checker is clear arithmetic on the decrypted blocks (decryption by the oracle): the reference's
own tests for these operations do the same (tfhe/src/integer/gpu/server_key/radix/tests_unsigned/
{test_add.rs,test_mul.rs}: encrypt, operate, decrypt, compare with the clear result).
[emu] runs the kernel sources on the host with a toy key, [hip] on the MI355X with
PARAM_MESSAGE_2_CARRY_2 and 64-bit integers."""
import numpy as np
import pytest

from . import oracle as orc
from .common import C1, TOY_2048, TOY_2048_P64, decrypt_big, encrypt_big, make_keys
from .harness import use_backend

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
MSG = 4  # message_modulus = carry_modulus = 4


def setup(kind, p=None, gpu_indexes=(0,), msg=MSG):
    from tfhe_rs_amd import core_crypto_gpu as gpu
    from tfhe_rs_amd import integer_gpu as igpu
    use_backend(kind)
    p = p or (TOY_2048 if kind == "emu" else C1)
    keys = make_keys(p)
    st = gpu.CudaStreams(gpu_indexes)
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(keys.ksk, p.big_n, p.n, p.ks_base_log, p.ks_level, st)
    if p.grouping:
        bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
            keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping, st)
    else:
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, st,
                                                             ms_noise_reduction=bool(p.ms_type))
    return p, keys, st, igpu.CudaServerKey(ksk, bsk, msg, msg), igpu


def encrypt_radix(p, keys, values, num_blocks, seed, msg=MSG):
    """[integer][block] big-key encryptions of the base-msg digits, least significant first."""
    digits = [[(int(v) // msg ** j) % msg for j in range(num_blocks)] for v in values]
    flat = encrypt_big(p, keys, [d for row in digits for d in row], seed=seed)
    return flat.reshape(len(values), num_blocks, -1)


def decrypt_blocks(p, keys, blocks):
    return [[decrypt_big(p, keys, b) for b in row] for row in blocks]


def recompose(rows, msg=MSG):
    return [sum(int(d) * msg ** j for j, d in enumerate(row)) for row in rows]


@pytest.mark.parametrize("kind", BACKENDS)
def test_apply_lookup_table_on_every_block(kind):
    p, keys, st, sks, igpu = setup(kind)
    vals = [0x1B, 0xE4, 0x39]
    ct = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, vals, 4, 11), st)
    lut = orc.generate_lut(p.k, p.N, 16, p.delta, lambda x: (3 * x + 1) % 16)
    out = sks.apply_lookup_table(ct, lut, st, degree=15)
    got = decrypt_blocks(p, keys, out.to_blocks(st))
    want = [[(3 * ((v >> (2 * j)) & 3) + 1) % 16 for j in range(4)] for v in vals]
    assert got == want


@pytest.mark.parametrize("kind", BACKENDS)
def test_add_and_carry_propagation(kind):
    p, keys, st, sks, igpu = setup(kind)
    # 17 blocks: four full groups of the carry look-ahead and a partial one — a two-step scan in which the first
    # step completes one prefix and leaves two open (32 blocks: three steps)
    L = 17 if kind == "emu" else 32
    bits = 2 * L
    rng = np.random.default_rng(5)
    nrand = 1 if kind == "emu" else 3
    a = [int(x) for x in rng.integers(0, 1 << 62, size=nrand)] + [(1 << bits) - 1, 0x5555555555555555,
                                                                  0x0FFF0FFF0FFF0FFF, 0]
    b = [int(x) for x in rng.integers(0, 1 << 62, size=nrand)] + [1, 0xAAAAAAAAAAAAAAAB, 0x0001000100010001, 0]
    if kind != "emu":
        a.append(0x3333333333333333)
        b.append(0xCCCCCCCCCCCCCCCD)
    a = [x & ((1 << bits) - 1) for x in a]
    b = [x & ((1 << bits) - 1) for x in b]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 21), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 22), st)
    # unchecked add leaves carries in the blocks ...
    tmp = ca.duplicate(st)
    sks.unchecked_add_assign(tmp, cb, st)
    raw = decrypt_blocks(p, keys, tmp.to_blocks(st))
    assert raw == [[((x >> (2 * j)) & 3) + ((y >> (2 * j)) & 3) for j in range(L)] for x, y in zip(a, b)]
    # ... which one propagation resolves; the ripple cases (all-ones + 1, 0101.. + 1010..1) cross every block
    sks.propagate_single_carry_assign(tmp, st)
    rows = decrypt_blocks(p, keys, tmp.to_blocks(st))
    assert all(d < MSG for r in rows for d in r)
    assert recompose(rows) == [(x + y) % (1 << bits) for x, y in zip(a, b)]
    # fused entry point, with an input carry per integer and the output carry requested (OutputFlag::Carry):
    # a + b + c_in = result + 2^bits * c_out
    cin_vals = [i & 1 for i in range(len(a))]
    cin = igpu.CudaUnsignedRadixCiphertext.from_blocks(
        encrypt_big(p, keys, cin_vals, seed=23).reshape(len(a), 1, -1), st)
    cout = sks.add_assign(ca, cb, st, carry_in=cin, want_carry_out=True)
    full = [x + y + c for x, y, c in zip(a, b, cin_vals)]
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [f % (1 << bits) for f in full]
    assert [r[0] for r in decrypt_blocks(p, keys, cout.to_blocks(st))] == [f >> bits for f in full]


@pytest.mark.parametrize("kind", BACKENDS)
def test_add_at_the_widths_where_the_carry_tree_changes_shape(kind):
    """The look-ahead is a tree of groups of three under a top level of at most four elements: 1 block (no carry
    at all), 2 and 4 (the top level alone), 10 (four groups under the top), 13 (a group of one block); the 17- and
    32-block cases above have two levels of groups."""
    p, keys, st, sks, igpu = setup(kind)
    rng = np.random.default_rng(77)
    # on the MI355X also widths with three and four levels of groups (37, 64, 100 blocks) and ragged last groups
    widths = (1, 2, 4, 10, 13) if kind == "emu" else (1, 2, 3, 4, 5, 7, 8, 10, 11, 13, 16, 21, 28, 33, 37, 64, 100)
    for L in widths:
        bits = 2 * L
        mask = (1 << bits) - 1
        a = [mask, int.from_bytes(rng.bytes(32), "little") & mask]
        b = [1, int.from_bytes(rng.bytes(32), "little") & mask]
        ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 41 + L), st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 42 + L), st)
        cout = sks.add_assign(ca, cb, st, want_carry_out=True)
        full = [x + y for x, y in zip(a, b)]
        assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [f & mask for f in full], L
        assert [r[0] for r in decrypt_blocks(p, keys, cout.to_blocks(st))] == [f >> bits for f in full], L
        if L in (1, 2, 4, 10, 13):
            assert int(igpu._lib().hip_integer_propagate_pbs_count(L)) == {1: 1, 2: 5, 4: 11, 10: 29, 13: 39}[L]


@pytest.mark.parametrize("kind", BACKENDS)
def test_mul_at_small_and_odd_widths(kind):
    """1 block (one product, no column sum, no carry), 2 and 3 (columns of at most five terms: no reduction step),
    and on the MI355X widths whose last group of three is ragged or whose tree has three levels."""
    p, keys, st, sks, igpu = setup(kind)
    rng = np.random.default_rng(78)
    for L in ((1, 2, 3) if kind == "emu" else (1, 2, 3, 5, 8, 13, 21, 40)):
        mask = (1 << (2 * L)) - 1
        a = [mask, int.from_bytes(rng.bytes(16), "little") & mask]
        b = [mask, int.from_bytes(rng.bytes(16), "little") & mask]
        ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 51 + L), st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 52 + L), st)
        sks.mul_assign(ca, cb, st)
        rows = decrypt_blocks(p, keys, ca.to_blocks(st))
        assert all(d < MSG for r in rows for d in r), L
        assert recompose(rows) == [(x * y) & mask for x, y in zip(a, b)], L


def test_add_and_mul_with_three_message_and_three_carry_bits():
    """message_modulus = carry_modulus = 8 (the MESSAGE_3_CARRY_3 shape) on a toy key: nothing in the carry tree, the
    two-function first bootstrap, the 4 m + z results or the column sums planned on bounds is specific to base 4."""
    m = 8
    p, keys, st, sks, igpu = setup("emu", p=TOY_2048_P64, msg=m)
    L = 5
    top = m ** L
    a = [top - 1, 0o52525, 0o17071]
    b = [1, 0o25253, 0o60707]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 61, m), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 62, m), st)
    cm = ca.duplicate(st)
    cout = sks.add_assign(ca, cb, st, want_carry_out=True)
    full = [x + y for x, y in zip(a, b)]
    rows = decrypt_blocks(p, keys, ca.to_blocks(st))
    assert all(d < m for r in rows for d in r)
    assert recompose(rows, m) == [f % top for f in full]
    assert [r[0] for r in decrypt_blocks(p, keys, cout.to_blocks(st))] == [f // top for f in full]
    sks.mul_assign(cm, cb, st)
    rows = decrypt_blocks(p, keys, cm.to_blocks(st))
    assert all(d < m for r in rows for d in r)
    assert recompose(rows, m) == [(x * y) % top for x, y in zip(a, b)]


@pytest.mark.parametrize("many", [False, True], ids=["few_integers", "many_integers"])
@pytest.mark.parametrize("kind", BACKENDS)
def test_mul(kind, many):
    """few integers: every term of a column is grouped at once (fewest rounds); 8 or more per call: only full
    groups are summed and the rest of a column waits (fewest bootstraps) — both plans, same products."""
    p, keys, st, sks, igpu = setup(kind)
    # 9 blocks: columns of up to 17 terms, several reduction steps
    L = (5 if many else 9) if kind == "emu" else 32
    bits = 2 * L
    mask = (1 << bits) - 1
    rng = np.random.default_rng(9)
    extra = 7 if many else 2
    a = [int(x) & mask for x in rng.integers(0, 1 << 62, size=extra)] + [mask]
    b = [int(x) & mask for x in rng.integers(0, 1 << 62, size=extra)] + [mask]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 31), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 32), st)
    pbs = sks.mul_assign(ca, cb, st, return_pbs_count=True)
    rows = decrypt_blocks(p, keys, ca.to_blocks(st))
    assert all(d < MSG for r in rows for d in r)
    assert recompose(rows) == [(x * y) & mask for x, y in zip(a, b)]
    assert pbs > L * L
    if L == 32:
        assert pbs == (1762 if many else 1804)   # products (1,024) + column sums (641 or 683) + one propagation (97)


@pytest.mark.parametrize("kind", BACKENDS)
def test_radix_ops_on_a_multi_bit_key(kind):
    """pbs_type = MULTI_BIT through the radix FFI (the reference's GPU defaults are multi-bit sets,
    cuda/src/pbs/programmable_bootstrap.cuh:348-535): add with carries and mul on a multi-bit server key."""
    from .common import C4, TOY_MB_2048
    p, keys, st, sks, igpu = setup(kind, TOY_MB_2048 if kind == "emu" else C4)
    L = 6 if kind == "emu" else 16
    bits, mask = 2 * L, (1 << (2 * L)) - 1
    a, b = [0x9E3779B9 & mask, mask, 0x12345678 & mask], [0x7F4A7C15 & mask, 1, 0x0FEDCBA9 & mask]
    ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 41), st)
    cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 42), st)
    cm = ca.duplicate(st)
    sks.add_assign(ca, cb, st)
    assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [(x + y) & mask for x, y in zip(a, b)]
    sks.mul_assign(cm, cb, st)
    assert recompose(decrypt_blocks(p, keys, cm.to_blocks(st))) == [(x * y) & mask for x, y in zip(a, b)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_radix_rounds_sharded_over_the_streams_of_the_set(kind):
    """In-library multi-GPU (helper_multi_gpu.cuh:170-294): every KS -> PBS round of add / mul is split over the
    streams of the CudaStreamsFFI by the reference's get_num_inputs_on_gpu rule, with per-stream key replicas and
    scratch, peer copies and events.  Three streams on GPU 0 (the reference's debug-fake-multi-gpu idea: the box
    has one GPU) and a threshold of 3 blocks per GPU so that even the small rounds are sharded raggedly; results
    must equal the single-stream run bit for bit and decrypt to the clear results."""
    p, keys, st1, sks1, igpu = setup(kind)
    _, _, st3, sks3, _ = setup(kind, gpu_indexes=(0, 0, 0))
    lib = use_backend(kind)
    L = 5 if kind == "emu" else 32
    mask = (1 << (2 * L)) - 1
    rng = np.random.default_rng(77)
    a = [int(x) & mask for x in rng.integers(0, 1 << 62, size=2 if kind == "emu" else 3)] + [mask]
    b = [int(x) & mask for x in rng.integers(0, 1 << 62, size=2 if kind == "emu" else 3)] + [1]
    blocks_a, blocks_b = encrypt_radix(p, keys, a, L, 51), encrypt_radix(p, keys, b, L, 52)
    outs = {}
    for name, st, sks, thr in (("one", st1, sks1, 512), ("three", st3, sks3, 3)):
        lib.hip_integer_set_multi_gpu_threshold(thr)
        try:
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_a, st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_b, st)
            cm = ca.duplicate(st)
            sks.add_assign(ca, cb, st)
            sks.mul_assign(cm, cb, st)
            outs[name] = (ca.to_blocks(st), cm.to_blocks(st))
        finally:
            lib.hip_integer_set_multi_gpu_threshold(0)
    assert np.array_equal(outs["one"][0], outs["three"][0]) and np.array_equal(outs["one"][1], outs["three"][1])
    assert recompose(decrypt_blocks(p, keys, outs["three"][0])) == [(x + y) & mask for x, y in zip(a, b)]
    assert recompose(decrypt_blocks(p, keys, outs["three"][1])) == [(x * y) & mask for x, y in zip(a, b)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_one_operation_spreads_over_the_gpus_by_the_reference_thresholds(kind):
    """No setter: the library's DEFAULT is the reference's rule (12 blocks per GPU with a multi-bit key), so ONE 32-block
    addition on a stream set of three spreads its rounds of 32 / 20 / ... blocks over 3 / 2 / 1 GPUs — what makes the
    reference's published single-operation latencies on 8 GPUs.  Three streams of GPU 0 on the GPU tier, three streams of
    the one pretend device on the CPU tier; same bits as the single-stream run."""
    from .common import TOY_MB4_2048
    p, keys, st1, sks1, igpu = setup(kind, TOY_MB4_2048)
    _, _, st3, sks3, _ = setup(kind, TOY_MB4_2048, gpu_indexes=(0, 0, 0))
    lib = use_backend(kind)
    lib.hip_integer_set_multi_gpu_threshold(0)
    assert [lib.hip_integer_active_gpu_count(b, 3, 0, 0) for b in (32, 20, 7, 3)] == [3, 2, 1, 1]
    L = 32
    mask = (1 << (2 * L)) - 1
    a, b = [0x9E3779B97F4A7C15 & mask, mask], [0x0123456789ABCDEF & mask, 1]
    blocks_a, blocks_b = encrypt_radix(p, keys, a, L, 71), encrypt_radix(p, keys, b, L, 72)
    outs = {}
    for name, st, sks in (("one", st1, sks1), ("three", st3, sks3)):
        ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_a, st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_b, st)
        sks.add_assign(ca, cb, st)
        outs[name] = ca.to_blocks(st)
    assert np.array_equal(outs["one"], outs["three"])
    assert recompose(decrypt_blocks(p, keys, outs["three"])) == [(x + y) & mask for x, y in zip(a, b)]


def test_radix_rounds_on_distinct_devices_record_events_on_their_own_streams(monkeypatch):
    """The same sharded rounds with the stream set naming three DIFFERENT devices.  The CPU tier's runtime stand-in
    models devices (HIPEMU_DEVICES): a stream and an event belong to the device current at their creation and
    hipEventRecord fails — as on real HIP, hipErrorInvalidHandle — for an event recorded on another device's stream;
    HX_CHECK aborts on that.  (A round-2 build created `staged` / `copied` under the shard's GPU and recorded them on
    the first GPU's stream: it only ever ran as several streams of one device.)"""
    monkeypatch.setenv("HIPEMU_DEVICES", "3")
    kind = "emu"
    lib = use_backend(kind)
    assert lib.cuda_get_number_of_gpus() == 3
    p, keys, st1, sks1, igpu = setup(kind)
    _, _, st3, sks3, _ = setup(kind, gpu_indexes=(0, 1, 2))
    L, mask = 5, (1 << 10) - 1
    a, b = [0x2A7, 0x155, mask], [0x1F3, 0x2AB, 1]
    blocks_a, blocks_b = encrypt_radix(p, keys, a, L, 61), encrypt_radix(p, keys, b, L, 62)
    outs = {}
    for name, st, sks, thr in (("one", st1, sks1, 512), ("three", st3, sks3, 3)):
        lib.hip_integer_set_multi_gpu_threshold(thr)
        try:
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_a, st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_b, st)
            cm = ca.duplicate(st)
            sks.add_assign(ca, cb, st)
            sks.mul_assign(cm, cb, st)
            outs[name] = (ca.to_blocks(st), cm.to_blocks(st))
        finally:
            lib.hip_integer_set_multi_gpu_threshold(0)
    assert np.array_equal(outs["one"][0], outs["three"][0]) and np.array_equal(outs["one"][1], outs["three"][1])
    assert recompose(decrypt_blocks(p, keys, outs["three"][0])) == [(x + y) & mask for x, y in zip(a, b)]
    assert recompose(decrypt_blocks(p, keys, outs["three"][1])) == [(x * y) & mask for x, y in zip(a, b)]


def test_radix_rounds_on_devices_without_peer_access_take_the_host_staged_copies(monkeypatch):
    """Three distinct devices that cannot reach each other's memory (HIPEMU_NO_PEER: hipDeviceCanAccessPeer says no,
    hipDeviceEnablePeerAccess fails, a peer copy between them is an error): the library must have asked and must ship
    its shards through the pinned host buffer instead.  With peer access available the same stand-in only lets a peer
    copy through once BOTH directions were enabled (the test above runs that way).  Same bits as the single-device run."""
    monkeypatch.setenv("HIPEMU_DEVICES", "3")
    monkeypatch.setenv("HIPEMU_NO_PEER", "1")
    kind = "emu"
    lib = use_backend(kind)
    p, keys, st1, sks1, igpu = setup(kind)
    _, _, st3, sks3, _ = setup(kind, gpu_indexes=(2, 0, 1))   # the first GPU of the set need not be device 0
    L, mask = 5, (1 << 10) - 1
    a, b = [0x2A7, 0x155, mask], [0x1F3, 0x2AB, 1]
    blocks_a, blocks_b = encrypt_radix(p, keys, a, L, 61), encrypt_radix(p, keys, b, L, 62)
    outs = {}
    for name, st, sks, thr in (("one", st1, sks1, 0), ("three", st3, sks3, 2)):
        lib.hip_integer_set_multi_gpu_threshold(thr)
        try:
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_a, st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(blocks_b, st)
            sks.add_assign(ca, cb, st)
            outs[name] = ca.to_blocks(st)
        finally:
            lib.hip_integer_set_multi_gpu_threshold(0)
    assert np.array_equal(outs["one"], outs["three"])
    assert recompose(decrypt_blocks(p, keys, outs["three"])) == [(x + y) & mask for x, y in zip(a, b)]


def test_active_gpu_count_is_the_reference_rule():
    """get_active_gpu_count (helper_multi_gpu.cu:16-48): a round spreads over min(G, ceil(blocks / threshold)) GPUs,
    threshold 12 for multi-bit keys and (compute units of GPU 0) + 1 for classic ones — so ONE FheUint64 operation on
    the multi-bit set (rounds of 32, 20, 3, 3, 7, 32 blocks) uses 3, 2, 1, 1, 1, 3 GPUs of 8, as on the reference's
    8 x H100 run; the setter overrides the rule, 0 restores it."""
    lib = use_backend("emu")
    MULTI_BIT, CLASSICAL = 0, 1   # pbs/pbs_enums.h:4
    lib.hip_integer_set_multi_gpu_threshold(0)
    assert [lib.hip_integer_active_gpu_count(b, 8, MULTI_BIT, 0) for b in (32, 20, 3, 7, 12, 13, 96, 97, 1024)] == \
        [3, 2, 1, 1, 1, 2, 8, 8, 8]
    cus = lib.cuda_get_number_of_sms()
    assert [lib.hip_integer_active_gpu_count(b, 8, CLASSICAL, 0) for b in (1, cus + 1, cus + 2, 4 * (cus + 1), 32768)] == \
        [1, 1, 2, 4, 8]
    lib.hip_integer_set_multi_gpu_threshold(3)
    try:
        assert lib.hip_integer_active_gpu_count(7, 8, CLASSICAL, 0) == 3
    finally:
        lib.hip_integer_set_multi_gpu_threshold(0)


@pytest.mark.parametrize("kind", BACKENDS)
def test_apply_many_lookup_table_extracts_every_function_from_one_bootstrap(kind):
    """scratch_/cuda_/cleanup_ apply_many_univariate_lut_64 (cuda/include/integer/integer.h:135-160,
    integer.cuh:1002-1110): a shortint ManyLookupTable of three functions over inputs of degree <= 3
    (shortint/engine/mod.rs:169-254), one keyswitch + one PBS per block, three samples extracted at multiples of the
    stride.  Function t of block s must sit in output block t * n + s and decrypt to f_t(m) — the check the reference
    makes in shortint/server_key/tests (apply_many_lookup_table)."""
    from .common import generate_many_lut
    p, keys, st, sks, igpu = setup(kind)
    fns = [lambda x: (x * x) % 4, lambda x: (3 - x) % 4, lambda x: (2 * x + 1) % 8]
    many, max_degree, stride = generate_many_lut(p, fns)
    assert max_degree == 4 and stride == 5 * (p.N // 16)     # 16 / 3 functions -> inputs 0..4
    vals = [0x1B, 0xE4, 0x39, 0xC6, 0x00]                    # every 2-bit digit value in every position
    ct = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, vals, 4, 13), st)
    out = sks.apply_many_lookup_table(ct, many, len(fns), stride, st, degree=7)
    blocks = out.to_blocks(st).reshape(len(fns), len(vals), 4, -1)
    for t, f in enumerate(fns):
        got = decrypt_blocks(p, keys, blocks[t])
        assert got == [[f((v >> (2 * j)) & 3) for j in range(4)] for v in vals], f"function {t}"
    # fewer functions than the scratch holds is allowed, more is refused by the library (abort), like a stride that
    # walks past the polynomial — covered in tests/test_error_behaviour.py style by the size check below
    one = sks.apply_many_lookup_table(ct, many, 1, stride, st, degree=3)
    assert np.array_equal(one.to_blocks(st).reshape(len(vals), 4, -1), blocks[0])


def _signed(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


@pytest.mark.parametrize("kind", BACKENDS)
def test_signed_overflow_flag_of_add_and_propagate(kind):
    """OutputFlag::Overflow of cuda_add_and_propagate_single_carry (integer.h:39,383-407; integer_utilities.h:2311-2357,
    :2383-2412): the flag is 1 exactly when the two's-complement sum of the operands (and the input carry) leaves the
    range of the width — the reference's signed_overflowing_add tests compare with the clear i64 result the same way.
    Widths: a single block (the input carry is the carry into the sign block), a partial group, the carry tree's
    three-level shape.  propagate_single_carry alone refuses the flag like the reference (integer.cuh:2368-2370)."""
    p, keys, st, sks, igpu = setup(kind)
    rng = np.random.default_rng(31)
    for L in ((1, 2, 5) if kind == "emu" else (1, 2, 5, 13, 32)):
        bits = 2 * L
        top = 1 << (bits - 1)
        cases = [(top - 1, 1, 0), (top - 1, 0, 1), (top, top, 0), (top, top - 1, 1), (top - 1, top - 1, 1),
                 ((1 << bits) - 1, 1, 0), (0, 0, 0), (top, (1 << bits) - 1, 0), (top | 1, (1 << bits) - 1, 0)]
        cases += [(int(rng.integers(0, 1 << bits)) if bits < 63 else int(rng.integers(0, 1 << 62)) * 4 + int(rng.integers(0, 4)),
                   int(rng.integers(0, 1 << bits)) if bits < 63 else int(rng.integers(0, 1 << 62)) * 4 + int(rng.integers(0, 4)),
                   int(rng.integers(0, 2))) for _ in range(3 if kind == "emu" else 8)]
        a, b, c = [x for x, _, _ in cases], [y for _, y, _ in cases], [z for _, _, z in cases]
        for use_cin in (True, False):
            ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 41), st)
            cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 42), st)
            ca.set_degrees(MSG - 1)
            cb.set_degrees(MSG - 1)
            cin = igpu.CudaUnsignedRadixCiphertext.from_blocks(
                encrypt_big(p, keys, c, seed=43).reshape(len(a), 1, -1), st) if use_cin else None
            ovf = sks.add_assign(ca, cb, st, carry_in=cin, want_overflow=True)
            cc = c if use_cin else [0] * len(a)
            clear = [_signed(x, bits) + _signed(y, bits) + z for x, y, z in zip(a, b, cc)]
            assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [v % (1 << bits) for v in clear], (L, use_cin)
            want = [int(not (-(1 << (bits - 1)) <= v < (1 << (bits - 1)))) for v in clear]
            assert [r[0] for r in decrypt_blocks(p, keys, ovf.to_blocks(st))] == want, (L, use_cin)
            assert list(ca.degrees) == [MSG - 1] * ca.total_blocks


@pytest.mark.parametrize("kind", BACKENDS)
def test_multiplication_by_an_encrypted_boolean(kind):
    """is_boolean_right / is_boolean_left of cuda_integer_mult_inplace (integer.h:173-187; multiplication.cuh:508-520,
    cmux.cuh:13-46 zero_out_if): every block of the integer is kept where the boolean is 1 and zeroed where it is 0."""
    p, keys, st, sks, igpu = setup(kind)
    L = 4 if kind == "emu" else 32
    bits = 2 * L
    rng = np.random.default_rng(33)
    vals = [int(rng.integers(0, 1 << min(bits, 62))) for _ in range(3 if kind == "emu" else 6)] + [(1 << bits) - 1]
    conds = [i & 1 for i in range(len(vals))]
    for left in (False, True):
        ct = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, vals, L, 51), st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_big(p, keys, conds, seed=52).reshape(len(vals), 1, -1), st)
        out = sks.mul_by_boolean_assign(ct, cb, st, boolean_is_left=left)
        assert recompose(decrypt_blocks(p, keys, out.to_blocks(st))) == [v * k for v, k in zip(vals, conds)], left


def test_profile_ranges_bracket_the_rounds_of_an_addition():
    """TFHE_HIP_PROFILE=1: the radix layer pushes a named range around every round, keyswitch, bootstrap and carry propagation
    (csrc/profile.h; the reference: PUSH_RANGE("apply lut") / ("scatter") / ("gather"), cuda/src/integer/integer.cuh:874,958,981
    over tfhe-cuda-common/cuda/include/helper_profile.cuh).  In its own interpreter (the switch is read once); without the
    switch nothing is pushed.  The host-emulation build has no roctx library to hand the ranges to: they are counted."""
    import os
    import subprocess
    import sys
    import textwrap
    from .harness import EMU_LIB, build_emu
    build_emu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        from tests.test_radix_integer import setup, encrypt_radix, decrypt_blocks, recompose
        p, keys, st, sks, igpu = setup("emu")
        from tfhe_rs_amd import ffi
        lib = ffi.default_library()
        before = lib.hip_backend_profile_ranges()
        a, b, L = [1234567], [7654321], 12
        ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, a, L, 21), st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(encrypt_radix(p, keys, b, L, 22), st)
        sks.add_assign(ca, cb, st)
        assert recompose(decrypt_blocks(p, keys, ca.to_blocks(st))) == [(a[0] + b[0]) %% (1 << 24)]
        print("ranges", before, lib.hip_backend_profile_ranges())
        """ % root)
    for switch, expect_some in (("1", True), ("0", False)):
        env = dict(os.environ, TFHE_HIP_BACKEND_LIB=EMU_LIB, TFHE_HIP_PROFILE=switch, TFHE_HIP_PROFILE_QUIET="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        before, after = [int(x) for x in r.stdout.split("ranges")[1].split()]
        # one carry propagation = 1 range + per round (1 + keyswitch + bootstrap): at least 3 rounds for 12 blocks
        assert (after - before >= 10) if expect_some else (after == before == 0), r.stdout
