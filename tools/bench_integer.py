#!/usr/bin/env python3
"""Config 5 of BASELINE.json on ONE GPU: FheUint64 (32 blocks of PARAM_MESSAGE_2_CARRY_2) add and mul
over a batch of ciphertexts, through the radix layer of the backend (tfhe_rs_amd/integer_gpu.py).
Prints one JSON line per operation: ops/s, PBS per operation, the equivalent KS-PBS/s, and checks
every result against clear arithmetic (decryption by the oracle, outside the timed region).

The reference measures the same thing in tfhe-benchmark/benches/integer/bench.rs (throughput variant:
many independent ciphertexts in flight); its published numbers are for 8xH100 and
multi-bit parameters, so they are quoted there, not compared here."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle as orc  # noqa: E402  (checker only)
from tests.common import C1, decrypt_big, make_keys  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--ops", default="add,mul")
    ap.add_argument("--params", default="classic", choices=["classic", "multibit_g3", "multibit_g4"],
                    help="classic = PARAM_MESSAGE_2_CARRY_2; multibit_g4 = the reference's GPU default set "
                         "(PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2), the one its published numbers use")
    ap.add_argument("--streams", type=int, default=1, help="streams of the set (all on GPU 0 on a 1-GPU box)")
    args = ap.parse_args()
    from tfhe_rs_amd import core_crypto_gpu as gpu
    from tfhe_rs_amd import integer_gpu as igpu
    from tests.common import C4, C4G4
    p = {"classic": C1, "multibit_g3": C4, "multibit_g4": C4G4}[args.params]
    keys = make_keys(p)
    n_gpus = gpu.get_number_of_gpus()
    st = gpu.CudaStreams([i % n_gpus for i in range(args.streams)])
    ksk = gpu.CudaLweKeyswitchKey.from_lwe_keyswitch_key(keys.ksk, p.big_n, p.n, p.ks_base_log, p.ks_level, st)
    if p.grouping:
        bsk = gpu.CudaLweMultiBitBootstrapKey.from_lwe_multi_bit_bootstrap_key(
            keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.grouping, st)
    else:
        bsk = gpu.CudaLweBootstrapKey.from_lwe_bootstrap_key(keys.bsk, p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, st,
                                                             ms_noise_reduction=True)
    sks = igpu.CudaServerKey(ksk, bsk, 4, 4)
    B, L = args.batch, args.blocks
    mask = (1 << (2 * L)) - 1
    rng = np.random.default_rng(1)
    a = [int.from_bytes(rng.bytes(8), "little") & mask for _ in range(B)]
    b = [int.from_bytes(rng.bytes(8), "little") & mask for _ in range(B)]

    def enc(vals, seed):
        r = orc.Rng(seed)
        out = np.empty((B, L, p.big_n + 1), dtype=np.uint64)
        for i, v in enumerate(vals):
            for j in range(L):
                out[i, j] = orc.lwe_encrypt(r, keys.glwe_sk, (((v >> (2 * j)) & 3) * p.delta) % (1 << 64), p.glwe_noise)
        return out

    ha, hb = enc(a, 5), enc(b, 6)
    for op in args.ops.split(","):
        ca = igpu.CudaUnsignedRadixCiphertext.from_blocks(ha, st)
        cb = igpu.CudaUnsignedRadixCiphertext.from_blocks(hb, st)
        st.synchronize()
        t0 = time.perf_counter()
        if op == "add":
            sks.add_assign(ca, cb, st)
            from tfhe_rs_amd import ffi
            pbs = int(ffi.default_library().hip_integer_propagate_pbs_count(L))
            want = [(x + y) & mask for x, y in zip(a, b)]
        else:
            pbs = sks.mul_assign(ca, cb, st, return_pbs_count=True)
            want = [(x * y) & mask for x, y in zip(a, b)]
        st.synchronize()
        dt = time.perf_counter() - t0
        rows = ca.to_blocks(st)
        check = rng.choice(B, size=min(B, 16), replace=False)
        bad = [int(i) for i in check
               if sum(decrypt_big(p, keys, rows[i, j]) << (2 * j) for j in range(L)) != want[i]]
        assert not bad, f"{op}: wrong results at {bad}"
        print(json.dumps({"op": f"FheUint{2 * L} {op}", "batch": B, "seconds": dt, "ops_per_s": B / dt,
                          "pbs_per_op": pbs, "ks_pbs_per_s": B * pbs / dt, "n_gpus": len(set(st.gpu_indexes)),
                          "streams": len(st),
                          "params": p.name, "includes": "scratch allocation, index uploads, all rounds"}))


if __name__ == "__main__":
    main()
