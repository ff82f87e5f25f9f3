#!/usr/bin/env python3
"""PMC passes over ONE batch-4096 launch of each throughput kernel of the path, written as a record that
bench.py can tie to the build it runs (profiles/pmc_latest.json, stamped with tools/build_id.py).

Run on the GPU box:   python tools/pmc_record.py [fft ntt mb_g3 ...] [--tag r02]
For every target the launch is profiled in separate rocprofv3 --pmc passes (counters only; never combined
with the sys/hip/hsa trace domains), as MI355X_MICROARCH.md prescribes.  HBM bytes per launch =
2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction for 16-byte coalesced reads: FETCH_SIZE tallies
128-byte requests at 64 bytes).  Raw per-counter sums go to gpurun_out/pmc_<tag>_<target>.txt (copy the ones
to be judged into profiles/)."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
sys.path.insert(0, ROOT)
from tools.build_id import source_build_id  # noqa: E402

PASSES = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS",
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM",
    "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum",
]
# target -> (tools/measure_all.py selector that issues exactly one launch, substring of the kernel name)
TARGETS = {
    "fft": ("wave1", "pbs_fft_wave_kernel"),
    "ntt": ("nttsplit4096", "pbs_fft_wave_kernel"),      # config 3's default engine: the split-key f64 form
    "ntt_int": ("ntt4096", "pbs_ntt"),
    "mb_g3": ("mb1", "pbs_"),
    "mb_g4": ("mb4one", "pbs_"),
    "n1024": ("n1024x4096", "pbs_fft_wave3"),
    "ks": ("ks1", "ks_gemm"),
    "ks_onelaunch": ("ks1", "ks_mfma"),
    "mb_keybundle": ("mblat151", "mb_keybundle_2048"),   # latency path, 151 ciphertexts: the keybundle launches
}


# --hbm-only: what bench.py needs per kernel — `traffic` (two passes) and the launch's shader cycles / VALU instruction count
# (one pass: sustained clock = cycles / launch time, VALU-issue roofline of the NTT engines)
HBM_PASSES = ["FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES"]


def provenance():
    """Where and with what the counters were taken (written into the record)."""
    info = {"host": os.uname().nodename}
    try:
        info["rocm"] = open("/opt/rocm/.info/version").read().strip()
    except OSError:
        pass
    try:
        r = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=60).stdout
        names = [l.split(":", 1)[1].strip() for l in r.splitlines() if "Marketing Name" in l]
        gfx = [l.split(":", 1)[1].strip() for l in r.splitlines() if l.strip().startswith("Name:") and "gfx" in l]
        info["isa"] = sorted(set(gfx))
        cpu = ("EPYC", "Xeon", "Processor", "Ryzen", "Core(TM)")
        info["gpus"] = ([n for n in names if "Instinct" in n or "MI3" in n] or
                        [n for n in names if n and not any(c in n for c in cpu)][-1:] or [g for g in info["isa"] if g.startswith("gfx")])
    except Exception as e:  # noqa: BLE001
        info["rocminfo"] = f"unavailable ({e.__class__.__name__})"
    try:
        info["rocprofv3"] = subprocess.run(["rocprofv3", "--version"], capture_output=True, text=True,
                                           timeout=60).stdout.strip().splitlines()[0]
    except Exception:  # noqa: BLE001
        pass
    return info


def main():
    global PASSES
    argv = sys.argv[1:]
    opts = {}
    for name in ("--tag", "--out", "--timeout"):
        if name in argv:
            i = argv.index(name)
            opts[name] = argv[i + 1]
            del argv[i:i + 2]
    if "--extra-passes" in argv:   # ad-hoc counter groups, ';'-separated, in front of the standard ones ("--extra-only": instead of)
        i = argv.index("--extra-passes")
        PASSES = [g.strip() for g in argv[i + 1].split(";") if g.strip()] + ([] if "--extra-only" in argv else PASSES)
        del argv[i:i + 2]
        if "--extra-only" in argv:
            argv.remove("--extra-only")
    if "--hbm-only" in argv:
        argv.remove("--hbm-only")
        PASSES = HBM_PASSES
    tag = opts.get("--tag", "run")
    per_pass_timeout = float(opts.get("--timeout", 600))
    targets = [a for a in argv if not a.startswith("--")] or ["fft", "ntt", "mb_g3", "mb_g4"]
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    record = {"build_id": source_build_id(), "batch": 4096, "kernels": {}, "provenance": provenance(),
              "measured_unix_time": __import__("time").time(),
              "method": "rocprofv3 --pmc, one counter group per pass, one launch per pass; hbm_bytes_per_launch = "
                        "2*FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)"}
    for t in targets:
        sel, needle = TARGETS[t]
        sums, kname = {}, None
        for i, counters in enumerate(PASSES):
            d = os.path.join(out_dir, f"pmc_{tag}_{t}_{i}")
            subprocess.run(["rm", "-rf", d])
            try:
                r = subprocess.run(["rocprofv3", "--pmc", *counters.split(), "-d", d, "--", sys.executable,
                                    os.path.join(ROOT, "tools", "measure_all.py"), sel], cwd="/tmp", env=env,
                                   capture_output=True, text=True, timeout=per_pass_timeout)
            except (OSError, subprocess.TimeoutExpired) as e:
                print(f"[{t} pass {i}] rocprofv3 did not run: {e}", file=sys.stderr)
                continue
            if r.returncode != 0:
                print(f"[{t} pass {i}] rocprofv3 failed:\n{r.stderr[-800:]}", file=sys.stderr)
                continue
            for db in glob.glob(d + "/**/*.db", recursive=True):
                cur = sqlite3.connect(db).cursor()
                try:
                    rows = list(cur.execute(
                        "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                        "from counters_collection group by kernel_name, counter_name"))
                except sqlite3.Error as e:
                    print("sqlite:", e, file=sys.stderr)
                    continue
                for name, cn, v, n in rows:
                    if needle in name and "to_fourier" not in name and "to_ntt" not in name and "to_split" not in name and "planes" not in name:
                        kname = name
                        sums[cn] = v / max(n, 1)   # per launch
        with open(os.path.join(out_dir, f"pmc_{tag}_{t}.txt"), "w") as f:
            f.write(f"# {kname}  build {record['build_id']}  (per launch)\n")
            for k in sorted(sums):
                f.write(f"{k}: {sums[k]:.6g}\n")
        if "FETCH_SIZE" in sums and "WRITE_SIZE" in sums:
            sums["hbm_bytes_per_launch"] = 2 * sums["FETCH_SIZE"] * 1024 + sums["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in sums and "TCC_MISS_sum" in sums:
            sums["l2_hit_rate"] = sums["TCC_HIT_sum"] / max(1.0, sums["TCC_HIT_sum"] + sums["TCC_MISS_sum"])
        sums["kernel"] = kname
        record["kernels"][t] = sums
        print(t, json.dumps(sums))
    out_path = opts.get("--out", os.path.join(out_dir, f"pmc_{tag}.json"))
    with open(out_path, "w") as f:
        json.dump(record, f, indent=1)
    print("wrote", out_path, "- copy to profiles/pmc_latest.json to publish")


if __name__ == "__main__":
    main()
