#!/usr/bin/env python3
"""GPU box: the headline kernel's time per workgroup against how much of the chip is busy.
  1. tools/bin/cumask_sweep variants/lib_ts.so all  -> gpurun_out/<tag>_cumask_sweep.jsonl (per-workgroup records)
  2. rocprofv3 --pmc passes (counters only) over ONE launch of the same kernel with 4 LWEs per workgroup at
     64 workgroups (batch 256) and 256 workgroups (batch 1024), and with 1 LWE per workgroup at 256 workgroups:
     instruction fetch, instruction cache, vector-L1 and L2 stall counters -> gpurun_out/<tag>_penalty_pmc.json
Usage: python tools/penalty_probe.py [tag] [--no-sweep] [--no-pmc]"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "r05")
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
exe = os.path.join(ROOT, "tools", "bin", "cumask_sweep")
lib = os.path.join(ROOT, "variants", "lib_ts.so")

if "--no-sweep" not in sys.argv:
    with open(os.path.join(out_dir, f"{tag}_cumask_sweep.jsonl"), "w") as f, \
            open(os.path.join(out_dir, f"{tag}_cumask_sweep.err"), "w") as e:
        try:
            r = subprocess.run([exe, lib, "all"], stdout=f, stderr=e, timeout=300)
            print("cumask_sweep exit", r.returncode)
        except subprocess.TimeoutExpired:
            print("cumask_sweep timed out")

PASSES = [
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
    "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA",
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB",
    "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum",
    "TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum",
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum",
    "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum",
]
CASES = {"wg64_of_4lwe": (256, 4), "wg256_of_4lwe": (1024, 4), "wg256_of_1lwe": (256, 1)}

if "--no-pmc" not in sys.argv:
    env = dict(os.environ, TMPDIR="/tmp")
    result = {}
    for case, (batch, pb) in CASES.items():
        sums = {}
        for i, counters in enumerate(PASSES):
            d = os.path.join(out_dir, f"pmc_{tag}_pen_{case}_{i}")
            subprocess.run(["rm", "-rf", d])
            try:
                r = subprocess.run(["rocprofv3", "--pmc", *counters.split(), "-d", d, "--", exe, lib, "pmc", str(batch), str(pb)],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=180)
            except (OSError, subprocess.TimeoutExpired) as e:
                print(f"[{case} pass {i}] rocprofv3 did not run: {e}", file=sys.stderr)
                continue
            if r.returncode != 0:
                print(f"[{case} pass {i}] rocprofv3 failed:\n{r.stderr[-600:]}", file=sys.stderr)
                continue
            for db in glob.glob(d + "/**/*.db", recursive=True):
                cur = sqlite3.connect(db).cursor()
                try:
                    rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                                            "from counters_collection group by kernel_name, counter_name"))
                except sqlite3.Error as e:
                    print("sqlite:", e, file=sys.stderr)
                    continue
                for name, cn, v, n in rows:
                    if "pbs_fft_wave_kernel" in name:
                        sums[cn] = v / max(n, 1)
            subprocess.run(["rm", "-rf", d])
        result[case] = {"batch": batch, "lwes_per_workgroup": pb, "counters_per_launch": sums}
        print(case, json.dumps(sums))
    with open(os.path.join(out_dir, f"{tag}_penalty_pmc.json"), "w") as f:
        json.dump(result, f, indent=1)
