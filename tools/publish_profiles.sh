#!/bin/bash
# copies one tools/final_measure.sh pass from gpurun_out/ into profiles/ under the round's names:
#   bash tools/publish_profiles.sh <tag of the pass, e.g. r05f> <round prefix, e.g. r05>
t=$1; r=$2; g=gpurun_out; p=profiles
cp $g/pmc_$t.json $p/pmc_latest.json
for k in fft ntt ntt_int mb_g3 mb_g4 n1024 ks; do cp $g/pmc_${t}_$k.txt $p/${r}_pmc_$k.txt; done
cp $g/${t}_bench.json $p/${r}_bench.json
cp $g/${t}_bench_dist1.json $p/${r}_bench_dist_1rank.json
cp $g/${t}_bench_fake2gpu.json $p/${r}_bench_fake_2gpu_two_streams_of_one_gpu.json
cp $g/${t}_bench_fake8gpu.json $p/${r}_bench_fake_8gpu_eight_streams_of_one_gpu.json
cp $g/${t}_bench_fake8gpu_scale_quick.json $p/${r}_bench_fake8gpu_scale_quick.json
cp $g/${t}_rocprof_stats.txt $p/${r}_rocprof_stats_bench.txt
cp $g/${t}_rocprof_extra_stats.txt $p/${r}_rocprof_stats_bench_with_extra.txt
cp $g/${t}_measure_all.jsonl $p/${r}_measure_all.jsonl
cp $g/${t}_latency_integer.jsonl $p/${r}_latency_integer_fheuint64.jsonl
cp $g/${t}_gputest.log $p/${r}_gputest.log
bash tools/kernel_usage_all.sh > $p/${r}_kernel_resource_usage.txt 2>/dev/null || true
[ -f $g/${t}_redzone.txt ] && cp $g/${t}_redzone.txt $p/${r}_redzone.txt
[ -f $g/${t}_marker_ranges_multibit_g4.txt ] && cp $g/${t}_marker_ranges_multibit_g4.txt $p/${r}_marker_ranges_multibit_g4.txt
