cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/rocprof_ks
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/rocprof_ks -- python $GRAFT_REPO_ROOT/tools/measure_all.py ks > $GRAFT_REPO_ROOT/gpurun_out/rocprof_ks.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/rocprof_ks gpurun_out/rocprof_ks_summary.txt | head -8
