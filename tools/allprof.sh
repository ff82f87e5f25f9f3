# rocprofv3 kernel-trace statistics over one pass of every kernel of the path (tools/measure_all.py, batch 4096)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/rocprof_all
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/rocprof_all -- python $GRAFT_REPO_ROOT/tools/measure_all.py ks wave generic ntt mb n1024 ks32 mb4 n8192 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_all.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/rocprof_all gpurun_out/rocprof_all_summary.txt | head -16
