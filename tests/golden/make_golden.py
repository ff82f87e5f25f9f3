#!/usr/bin/env python3
"""Generates tests/golden/reference_kats.json from the literal test vectors the tfhe-rs
sources hold for the PBS hot path (doc-tests, unit-test tables, fixed constants).

The reference is Rust and cannot be built or imported here, so this script TRANSCRIBES the
literals (each with its file:line) and, when /root/reference is present, verifies that every
transcribed literal still appears in the cited file, so the fixture cannot silently drift.
Run in the build container:  python tests/golden/make_golden.py
The GPU box has no /root/reference: tests only read the committed JSON.
"""
import json
import os
import struct

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def check_literals(path, needles):
    full = os.path.join(REF, path)
    if not os.path.exists(full):
        return "unverified (reference tree absent)"
    text = open(full).read()
    missing = [n for n in needles if n not in text]
    if missing:
        raise SystemExit(f"{path}: literals not found in the reference: {missing}")
    return "verified against the reference tree"


def main():
    kats = {"_about": "literal vectors of the tfhe-rs sources; see make_golden.py", "sources": {}}

    # ---- f64 -> i64 conversion table, tfhe/src/core_crypto/fft_impl/fft64/math/fft/x86.rs:1030-1112
    # target rule stated there (AVX-512 arm): x == 2^63 -> i64::MIN else x.round() as i64
    src = "tfhe/src/core_crypto/fft_impl/fft64/math/fft/x86.rs"
    kats["sources"][src] = check_literals(src, ["37.1242161_f64", "1e-310", "0.9 * 2.0_f64.powi(63)",
                                                "1.1 * 2.0_f64.powi(62)", "i64::MIN"])
    p62, p63 = 2.0 ** 62, 2.0 ** 63
    vals = [-p63, -p63, p63, p63, 0.0, -0.0, 37.1242161, -37.1242161, 0.1, -0.1, 1.0, -1.0, 0.9, -0.9, 2.0, -2.0,
            2.0, -2.0, 1e-310, -1e-310, p62, -p62, 1.1 * p62, 1.1 * -p62, 0.9 * p63, -(0.9 * p63), 0.1 * p63,
            0.1 * -p63]

    def target(x):
        if x == p63:
            return -(1 << 63)
        r = round(x)  # no value above is an exact .5 tie, so half-even == half-away here
        return int(r)
    kats["convert_f64_i64"] = {"cite": src + ":1030-1112",
                               "inputs_f64_bits": [f64_bits(v) for v in vals],
                               "targets_i64": [target(v) for v in vals]}

    # ---- decomposer doc-tests, tfhe/src/core_crypto/commons/math/decomposition/decomposer.rs:139-142,199-217
    src = "tfhe/src/core_crypto/commons/math/decomposition/decomposer.rs"
    kats["sources"][src] = check_literals(src, ["1_340_987_234_u32", "1_341_128_704_u32", "2147483647u32"])
    kats["closest_representable_u32"] = {"cite": src + ":139-142", "base_log": 4, "level": 3,
                                         "input": 1340987234, "output": 1341128704}
    kats["decompose_half_basis_u32"] = {"cite": src + ":199-217", "base_log": 4, "level": 3, "input": 2147483647,
                                        "digit_abs_max": 8, "count": 3}

    # ---- monomial doc-tests, tfhe/src/core_crypto/algorithms/polynomial_algorithms.rs:383-395,450-462,530-545
    src = "tfhe/src/core_crypto/algorithms/polynomial_algorithms.rs"
    kats["sources"][src] = check_literals(src, ["&[3, 255, 254]", "&[254, 253, 1]", "&[28, 71, 45]",
                                                "vec![4_u8, 5, 0]", "vec![7_u8, 9, 0]"])
    kats["monomial_div_u8"] = {"cite": src + ":383-395,530-545", "input": [1, 2, 3], "degree": 2,
                               "output": [3, 255, 254]}
    kats["monomial_mul_u8"] = {"cite": src + ":450-462,596-608", "input": [1, 2, 3], "degree": 2,
                               "output": [254, 253, 1]}
    kats["polynomial_wrapping_mul_u8"] = {"cite": src + ":1052-1066", "lhs": [4, 5, 0], "rhs": [7, 9, 0],
                                          "output": [28, 71, 45]}

    # ---- Goldilocks roots, tfhe-ntt/src/prime64.rs:162-179 (psi^N = -1 mod p is checked by the test)
    src = "tfhe-ntt/src/prime64.rs"
    roots = {"256": 14430643036723656017, "512": 4440654710286119610, "1024": 8816101479115663336,
             "2048": 10974926054405199669, "4096": 1206500561358145487}
    kats["sources"][src] = check_literals(src, [f"{v}_u64" for v in roots.values()])
    kats["goldilocks_roots"] = {"cite": src + ":162-179", "p": 0xFFFFFFFF00000001, "roots": roots}

    # ---- parameter constants, tfhe/src/shortint/parameters/v1_4/classic/tuniform/p_fail_2_minus_128/ks_pbs.rs:28-47
    src = "tfhe/src/shortint/parameters/v1_4/classic/tuniform/p_fail_2_minus_128/ks_pbs.rs"
    kats["sources"][src] = check_literals(src, ["LweDimension(918)", "PolynomialSize(2048)",
                                                "DecompositionBaseLog(23)"])
    kats["param_message_2_carry_2"] = {"cite": src + ":28-47", "n": 918, "k": 1, "N": 2048, "pbs_base_log": 23,
                                       "pbs_level": 1, "ks_base_log": 4, "ks_level": 4}

    # ---- SHA-256 of the reference's golden PBS vectors, apps/test-vectors/checksums.sha256
    src = "apps/test-vectors/checksums.sha256"
    sums = {"toy_params": {}, "valid_params_128": {}}
    full = os.path.join(REF, src)
    if os.path.exists(full):
        for line in open(full):
            h, path = line.split()
            for which in sums:
                if f"/{which}/" in path:
                    sums[which][os.path.basename(path).replace(".cbor", "")] = h
        kats["sources"][src] = "read from the reference tree"
    else:  # keep what is already committed
        old = json.load(open(OUT))
        sums = {"toy_params": old["test_vector_sha256_toy"]["sha256"],
                "valid_params_128": old["test_vector_sha256_valid"]["sha256"]}
        kats["sources"][src] = "unverified (reference tree absent)"
    gen = "generator apps/test-vectors/src/main.rs:121-365 (RAND_SEED 0x74666865"
    kats["test_vector_sha256_toy"] = {
        "cite": src + ":1-36 ; " + gen + ", toy params n=10,k=1,N=256, PBS 24x1, KS 37x1, noise stddev 0)",
        "sha256": sums["toy_params"]}
    kats["test_vector_sha256_valid"] = {
        "cite": src + ":1-36 ; " + gen + ", valid params n=833,k=1,N=2048, PBS 23x1, KS 3x5, Gaussian noise "
                      "3.6158408373309336e-06 / 2.845267479601915e-15; main.rs:17-25)",
        "sha256": sums["valid_params_128"]}

    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1, sort_keys=True)
    print("wrote", OUT)
    for k, v in kats["sources"].items():
        print(" ", k, "->", v)


if __name__ == "__main__":
    main()
