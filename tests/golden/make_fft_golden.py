#!/usr/bin/env python3
"""Transcribes the reference's golden spectrum of its forward transform into a fixture that travels to the GPU box:
  tfhe/src/core_crypto/gpu/algorithms/test/fft/fft_data/fft16x4x16_golden_v1.rs  ->  tests/golden/fft16x4x16_golden_v1.json
(the f64 bit patterns of the 1024 complex frequencies, natural order, produced by the reference's throughput transform on an
H100 from the deterministic input of gpu/algorithms/test/fft/mod.rs:51-71).  Run in the container that has /root/reference."""
import json
import os
import re

SRC = "/root/reference/tfhe/src/core_crypto/gpu/algorithms/test/fft/fft_data/fft16x4x16_golden_v1.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fft16x4x16_golden_v1.json")

src = open(SRC).read()
n = int(re.search(r"POLYNOMIAL_SIZE: usize = (\d+);", src).group(1))
re_part = src[src.index("EXPECTED_RE"):src.index("EXPECTED_IM")]
im_part = src[src.index("EXPECTED_IM"):]
rec = {"source": SRC.replace("/root/reference/", ""), "polynomial_size": n,
       "expected_re": re.findall(r"0x[0-9a-f]{16}", re_part), "expected_im": re.findall(r"0x[0-9a-f]{16}", im_part)}
assert len(rec["expected_re"]) == n // 2 == len(rec["expected_im"])
json.dump(rec, open(OUT, "w"), indent=0)
print("wrote", OUT, len(rec["expected_re"]), "frequencies")
