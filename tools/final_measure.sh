# Round-end measurement on the GPU box:  bash tools/final_measure.sh <tag>
# everything lands under gpurun_out/ (copy what is to be judged into profiles/)
tag=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_gputest.log; cat gpurun_out/${tag}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/pmc_record.py fft ntt ntt_int mb_g3 mb_g4 n1024 ks --tag ${tag} > gpurun_out/${tag}_pmc.log 2>&1; tail -1 gpurun_out/${tag}_pmc.log
# bench.py ties `traffic` to the build through this record: publish it before the bench line is taken
cp gpurun_out/pmc_${tag}.json profiles/pmc_latest.json
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err; cut -c1-300 gpurun_out/${tag}_bench.json
# the multi-rank code path of bench.py at N = 1 (torch.distributed over RCCL, one rank)
TFHE_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --steps 3 --no-cpu-baseline --no-extra > gpurun_out/${tag}_bench_dist1.json 2> gpurun_out/${tag}_bench_dist1.err; cut -c1-200 gpurun_out/${tag}_bench_dist1.json; tail -2 gpurun_out/${tag}_bench_dist1.err
# rocprofv3 kernel-trace statistics of the bench command itself
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_rocprof
# headline only (--no-extra): the average duration of the headline kernel must be that of bench.py's timed launches,
# and the radix datapoint under `extra` launches the same kernel at other batch sizes
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_rocprof -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/${tag}_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_rocprof_extra -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --parity-sample 64 > $R/gpurun_out/${tag}_rocprof_extra.log 2>&1
cd $R && python tools/rocprof_summary.py gpurun_out/${tag}_rocprof gpurun_out/${tag}_rocprof_stats.txt | head -14
cd $R && python tools/rocprof_summary.py gpurun_out/${tag}_rocprof_extra gpurun_out/${tag}_rocprof_extra_stats.txt | head -3
python tools/measure_all.py ks ks32 ks1024 wave n1024 mb mb4 mblat ntt ntt_split sweep > gpurun_out/${tag}_measure_all.jsonl 2>&1; cat gpurun_out/${tag}_measure_all.jsonl | cut -c1-260
# the multi-GPU form of bench.py on the one GPU of this box (two shards as two streams: logic check, not a scaling number)
TFHE_BENCH_FAKE_MULTI_GPU=1 python bench.py --gpus 2 --steps 3 --no-pmc > gpurun_out/${tag}_bench_fake2gpu.json 2> gpurun_out/${tag}_bench_fake2gpu.err; cut -c1-200 gpurun_out/${tag}_bench_fake2gpu.json
# the scaling sweep's form of the command (headline only): eight shards, must stay under two minutes of wall time
( time TFHE_BENCH_FAKE_MULTI_GPU=1 python bench.py --gpus 8 --steps 3 --scale-quick > gpurun_out/${tag}_bench_fake8gpu_scale_quick.json 2> gpurun_out/${tag}_bench_fake8gpu_scale_quick.err ) 2>&1 | grep real; cut -c1-300 gpurun_out/${tag}_bench_fake8gpu_scale_quick.json
# ... and eight shards (the driver's largest N): 8 streams of the one GPU, config 5 through both shardings
TFHE_BENCH_FAKE_MULTI_GPU=1 python bench.py --gpus 8 --steps 2 --no-pmc > gpurun_out/${tag}_bench_fake8gpu.json 2> gpurun_out/${tag}_bench_fake8gpu.err; cut -c1-200 gpurun_out/${tag}_bench_fake8gpu.json; tail -2 gpurun_out/${tag}_bench_fake8gpu.err
python tools/latency_integer.py classic > gpurun_out/${tag}_latency_integer.jsonl 2>&1; python tools/latency_integer.py multibit_g4 >> gpurun_out/${tag}_latency_integer.jsonl 2>&1; cut -c1-200 gpurun_out/${tag}_latency_integer.jsonl
