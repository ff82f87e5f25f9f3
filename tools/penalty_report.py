#!/usr/bin/env python3
"""Reads gpurun_out/<tag>_cumask_sweep.jsonl (+ <tag>_penalty_pmc.json) written by tools/penalty_probe.py and prints the
attribution table of VERDICT r04 #1: per configuration the workgroups of the FIRST round (started within 0.5 ms of the
launch), their median CMUX-loop duration, the shader clock they ran at (s_memtime ticks / s_memrealtime 100 MHz ticks)
and the shader CYCLES the loop took.  Usage: python tools/penalty_report.py [tag] > profiles/<tag>_penalty_attribution.txt"""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_cumask_sweep.jsonl"))]

print("# headline kernel pbs_fft_wave_kernel<1,23> (PARAM_MESSAGE_2_CARRY_2, 918 CMUX per LWE), measurement build -DWAVE_PROBE_TS=1")
print("# one row per configuration; 'first round' = workgroups that started within 0.5 ms of the launch")
print("# clock = s_memtime delta / s_memrealtime delta x 100 MHz, per workgroup; Mcycles = s_memtime delta of the CMUX loop")
print(f"{'configuration':34s} {'LWE/wg':>6s} {'wgs':>4s} {'ms/launch':>9s} | {'first':>5s} {'loop ms':>8s} {'clock GHz':>9s} {'Mcycles':>8s} {'(min':>7s} {'max)':>7s}  note")
for r in rows:
    if r["what"] != "pbs":
        continue
    t0 = np.array(r["start_ticks"], float)
    t1 = np.array(r["end_ticks"], float)
    mt = np.array(r["memtime_delta"], float)
    first = t0 < 50000
    d = t1 - t0
    clk = mt / d / 10
    print(f"{r['name']:34s} {r['lwes_per_block']:6d} {r['blocks']:4d} {r['ms_per_launch']:9.3f} | {int(first.sum()):5d} {np.median(d[first]) / 1e5:8.3f} "
          f"{np.median(clk[first]):9.3f} {np.median(mt[first]) / 1e6:8.3f} {mt[first].min() / 1e6:7.3f} {mt[first].max() / 1e6:7.3f}  {r['note']}")
    if r["name"] in ("mask_xcc0_full", "mask_xcc0to3_full", "pbs_xcc0to3_while_fma_xcc4to7", "nat_b1024_pb4"):
        xcc = np.array([h >> 32 for h in r["hwid_xcc"]])
        for x in range(8):
            s = first & (xcc == x)
            if s.sum():
                print(f"{'    XCC ' + str(x):34s} {'':6s} {'':4s} {'':9s} | {int(s.sum()):5d} {np.median(d[s]) / 1e5:8.3f} {np.median(clk[s]):9.3f} {np.median(mt[s]) / 1e6:8.3f}")

print()
print("# f64 FMA load (tools/cumask_sweep.hip fma_kernel: 8 waves per CU, register-only v_fma_f64): shader clock over the launch")
for r in rows:
    if r["what"] != "fma":
        continue
    wg = r["wgs"]
    rt = np.array([x["rt"] for x in wg], float)
    mt = np.array([x["mt"] for x in wg], float)
    clk = np.median(mt / rt / 10, axis=0)
    t = np.cumsum(np.median(rt, axis=0)) / 1e5
    step = max(1, len(t) // 12)
    print(f"{r['name']} ({r['note']}; {len(wg)} workgroups, {t[-1]:.1f} ms)")
    print("   t ms : " + " ".join(f"{x:6.1f}" for x in t[::step]))
    print("   GHz  : " + " ".join(f"{x:6.3f}" for x in clk[::step]))

pmc_path = os.path.join(ROOT, "gpurun_out", f"{tag}_penalty_pmc.json")
if os.path.exists(pmc_path):
    pmc = json.load(open(pmc_path))
    print()
    print("# rocprofv3 --pmc, ONE launch each (counters only, separate passes); per launch unless said otherwise")
    keys = ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
            "SQ_IFETCH", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS",
            "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum",
            "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_TAG_STALL_sum", "TCC_BUSY_sum"]
    cases = list(pmc)
    print(f"{'counter':32s} " + " ".join(f"{c:>16s}" for c in cases))
    for k in keys:
        print(f"{k:32s} " + " ".join(f"{pmc[c]['counters_per_launch'].get(k, float('nan')):16.4g}" for c in cases))
    print("# derived")
    for c in cases:
        v = pmc[c]["counters_per_launch"]
        waves = v["SQ_WAVES"]
        per_wave_cmux = v["SQ_INSTS_VALU"] / waves / 918
        print(f"{c}: cycles per launch (GRBM_GUI_ACTIVE / 8 XCC) {v['GRBM_GUI_ACTIVE'] / 8 / 1e6:.2f} M; VALU instructions per wave and CMUX {per_wave_cmux:.0f}; "
              f"cycles per VALU instruction {4 * v['SQ_ACTIVE_INST_VALU'] / v['SQ_INSTS_VALU']:.2f}; instruction-cache hit rate "
              f"{v['SQC_ICACHE_HITS'] / v['SQC_ICACHE_REQ']:.6f}; fetches per wave and CMUX {v['SQ_IFETCH'] / waves / 918:.0f}; "
              f"mean L2 read latency {v['TCP_TCC_READ_REQ_LATENCY_sum'] / v['TCP_TCC_READ_REQ_sum']:.0f} cycles; "
              f"L2 tag-stall cycles per request {v['TCC_TAG_STALL_sum'] / v['TCC_REQ_sum']:.5f}")
