#!/bin/bash
# PMC passes over one bench launch of the PBS kernel (separate passes, counters only — never combined
# with the sys/hip/hsa trace domains).  Usage on the GPU box: bash tools/pmc.sh <tag>
# Results: gpurun_out/pmc_<tag>_<pass>/ (sqlite) + gpurun_out/pmc_<tag>.txt (summary)
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
declare -A P
P[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P[b]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"
P[c]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL SQ_WAVES GRBM_GUI_ACTIVE"
P[d]="FETCH_SIZE"
P[e]="WRITE_SIZE"
P[f]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
: > $R/gpurun_out/pmc_$tag.txt
for k in a b c d e f; do
  rm -rf $R/gpurun_out/pmc_${tag}_$k
  rocprofv3 --pmc ${P[$k]} -d $R/gpurun_out/pmc_${tag}_$k -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $R/gpurun_out/pmc_${tag}_$k.log 2>&1
  python - <<PY >> $R/gpurun_out/pmc_$tag.txt
import glob, sqlite3
for db in glob.glob("$R/gpurun_out/pmc_${tag}_$k/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    try:
        for name, cn, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            if "pbs" in name or "keyswitch" in name:
                print(f"$k {name[:40]} {cn}: {v:.4g} over {n} dispatch(es)")
    except Exception as e:
        print("err", e)
PY
done
cat $R/gpurun_out/pmc_$tag.txt
