#!/bin/bash
# Register / spill / scratch table of every kernel of the build (profiles/rNN_kernel_resource_usage.txt)
cd "$(dirname "$0")/.."
echo "Register / spill / scratch figures of the kernels of the build (hipcc -Rpass-analysis=kernel-resource-usage, device-only compile for gfx950)."
echo "Columns: VGPRs, AGPRs, SGPRs, VGPR spills, SGPR spills, scratch bytes per lane, waves per SIMD."
echo "pbs_fft_wave_kernel<L, B, G, SHARE, LIMBS, OCTET>: <1,23,0,false,0,false> headline; <1,23,0,false,4,false> split-key exact engine; <2,15,3,true,..> multi-bit g=3; <1,22,4,false,0,true> multi-bit g=4 (OCTET)."
echo
for f in tfhe_rs_amd/csrc/*.hip; do
  b=$(basename $f)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -c \
    -Rpass-analysis=kernel-resource-usage $f -o /dev/null 2>&1 | python3 -c '
import re, subprocess, sys
b = sys.argv[1]
cur = {}
def flush():
    if "name" in cur:
        dm = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        dm = re.sub(r"\(.*", "", dm)
        print("%-18s %4s %3s %4s  vspill %3s  sspill %3s  scratch %4s  occ %s  %s" % (b, cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("VGPRs Spill"), cur.get("SGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), dm))
for l in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        flush(); cur.clear(); cur["name"] = t.split(":", 1)[1].strip()
    elif ":" in t:
        k, v = t.rsplit(":", 1); cur[k.strip()] = v.strip()
flush()
' $b
done
