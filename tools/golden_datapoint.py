#!/usr/bin/env python3
"""One JSON line for bench.py's `extra.reference_gpu_golden`: the reference's GPU golden-value test of the 64-bit bootstrap
(tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden/mod.rs; data captured by the reference on an H100) run on this GPU —
keys, bootstrap keys and inputs regenerated from the test's seed (tests/pbs_golden.py), one batched call of BATCH_SIZE = 264
replicated inputs per golden message and parameter set; reports whether every lane equals lane 0 and the oracle's bits, whether
the outputs decrypt to f(m), and how far they sit from the reference's golden ciphertexts in phase (log2).
   python tools/golden_datapoint.py [--backend hip|emu] [--batch N]"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir)))
from tests import pbs_golden as pg                      # noqa: E402
from tests.harness import Ctx, oracle_pbs               # noqa: E402
from tests.test_pbs_golden import golden_setup          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="hip")
    ap.add_argument("--batch", type=int, default=0, help="0 = the BATCH_SIZE the golden data was captured at (264)")
    ap.add_argument("--sets", default="classical,multi_bit_group_4")
    args = ap.parse_args()
    out = {"source": "tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden (H100 golden ciphertexts); tests/test_pbs_golden.py",
           "gate_log2": 53}
    for which in args.sets.split(","):
        p, keys, lut, inputs, messages, golden, batch_size = golden_setup(which)
        b = args.batch or batch_size
        c = Ctx(args.backend, p, keys, "fft64")
        ref = oracle_pbs(p, keys, "fft64", inputs, lut)
        dist, lanes, bits, dec = [], True, True, True
        for i, m in enumerate(messages):
            o = c.pbs(np.repeat(inputs[i:i + 1], b, axis=0), lut)
            lanes &= bool(np.all(o == o[0]))
            bits &= bool(np.array_equal(o[0], ref[i]))
            ph, gph = pg.phase(o[0], keys.glwe_sk), pg.phase(golden[i], keys.glwe_sk)
            dec &= pg.decode(ph) == pg.f(m) and pg.decode(gph) == pg.f(m)
            dist.append(round(math.log2(max(pg.phase_distance(ph, gph), 1)), 1))
        out[which] = {"params": p.name, "batch": b, "messages": messages, "all_lanes_equal": lanes, "gpu_matches_cpu_bits": bits,
                      "decrypts_like_the_golden": dec, "log2_phase_distance_to_golden": dist,
                      "within_gate": bool(max(dist) < 53), "pbs_kernel_id": int(c.lib.hip_backend_last_pbs_kernel())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
