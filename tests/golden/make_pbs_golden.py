#!/usr/bin/env python3
"""Transcribes the reference's own GPU golden ciphertexts of the 64-bit programmable bootstrap
(tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden/pbs_golden_data/pbs_golden_v1.rs: CLASSICAL_EXPECTED for
PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 and MULTI_BIT_GROUP_4_EXPECTED for
PARAM_GPU_MULTI_BIT_GROUP_4_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128, one output ciphertext per message of
GOLDEN_MESSAGES = [1, 7, 15], captured on an H100) into tests/golden/pbs_golden_v1.json, together with the
constants of pbs_golden/mod.rs that fix the deterministic generation (GOLDEN_SEED, GOLDEN_MESSAGES, BATCH_SIZE).
The GPU box has no /root/reference: tests only read the committed JSON.
Run in the build container:  python tests/golden/make_pbs_golden.py"""
import json
import os
import re

REF = "/root/reference/tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pbs_golden_v1.json")


def main():
    text = open(os.path.join(REF, "pbs_golden_data", "pbs_golden_v1.rs")).read()
    mod = open(os.path.join(REF, "mod.rs")).read()
    seed = int(re.search(r"const GOLDEN_SEED: u128 = (0x[0-9a-f_]+);", mod).group(1).replace("_", ""), 16)
    messages = [int(x) for x in re.search(r"const GOLDEN_MESSAGES: \[u64; 3\] = \[([0-9, ]+)\];", mod).group(1).split(",")]
    batch = int(re.search(r"const BATCH_SIZE: usize = (\d+);", mod).group(1))
    out = {"_about": "the reference's GPU PBS golden ciphertexts (pbs_golden_v1.rs, H100); see make_pbs_golden.py",
           "source": "tfhe/src/core_crypto/gpu/algorithms/test/pbs_golden/{mod.rs:95-119,pbs_golden_data/pbs_golden_v1.rs}",
           "golden_seed": hex(seed), "golden_messages": messages, "batch_size": batch}
    for name, key in (("CLASSICAL_EXPECTED", "classical"), ("MULTI_BIT_GROUP_4_EXPECTED", "multi_bit_group_4")):
        body = text[text.index(f"pub const {name}"):]
        body = body[:body.index("];\n") + 2] if "];\n" in body else body
        cts = []
        for blk in re.findall(r"&\[\s*((?:0x[0-9a-f]{16},\s*)+)\]", body):
            words = re.findall(r"0x([0-9a-f]{16})", blk)
            assert len(words) == 2049, len(words)
            cts.append("".join(words))          # 2049 big-endian hex words of 16 digits, mask then body
        assert len(cts) == len(messages), (name, len(cts))
        out[key] = cts
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
