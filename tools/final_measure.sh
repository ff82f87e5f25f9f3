set -x
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -1 gpurun_out/bench_final.json | cut -c1-400
python tools/measure_all.py > gpurun_out/measure_all_final.jsonl 2>&1
cat gpurun_out/measure_all_final.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/rocprof_final
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/rocprof_final -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_final.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/rocprof_final gpurun_out/rocprof_final_summary.txt | head -20
