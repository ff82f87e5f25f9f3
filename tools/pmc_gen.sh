#!/bin/bash
# PMC passes over the generic kernel, N=1024 k=2 (one launch per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
declare -A P
P[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P[b]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU GRBM_GUI_ACTIVE"
P[d]="FETCH_SIZE"
P[f]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"
: > $R/gpurun_out/pmc_gen.txt
for k in a b d f; do
  rm -rf $R/gpurun_out/pmc_gen_$k
  rocprofv3 --pmc ${P[$k]} -d $R/gpurun_out/pmc_gen_$k -- python $R/tools/measure_all.py n1024x > $R/gpurun_out/pmc_gen_$k.log 2>&1
  python - <<PY >> $R/gpurun_out/pmc_gen.txt
import glob, sqlite3
for db in glob.glob("$R/gpurun_out/pmc_gen_$k/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    try:
        for name, cn, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            if "pbs" in name or "wave3" in name:
                print(f"$k {name[:48]} {cn}: {v:.4g} over {n} dispatch(es)")
    except Exception as e:
        print("err", e)
PY
done
cat $R/gpurun_out/pmc_gen.txt
