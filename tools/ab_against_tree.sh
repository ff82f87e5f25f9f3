#!/bin/bash
# GPU box: the throughput kernels of the current build against an earlier tree of this repository (variants/old_tree: `git archive <commit> | tar -x -C variants/old_tree`, built in place), same box,
# interleaved: classic 2_2 (wave), multi-bit g = 3 (mb) and g = 4 (mb4), split-key exact engine (ntt_split), N = 1024
# usage: bash tools/ab_against_tree.sh [rounds] [selectors]   -> gpurun_out/ab_against_tree.txt
out=gpurun_out/ab_against_tree.txt; mkdir -p gpurun_out; : > $out
rounds=${1:-2}; sel=${2:-"wave mb mb4 ntt_split n1024"}
fmt() { grep '"batch": 4096' | sed -e 's/.*"params": "\([^"]*\)".*"ms": \([0-9.]*\).*pbs_per_s": \([0-9.]*\).*/  \1 ms \2 pbs \3/'; }
for r in $(seq 1 $rounds); do
  if [ -d variants/old_tree ]; then echo "== round $r: earlier tree" | tee -a $out; (cd variants/old_tree && python tools/measure_all.py $sel 2>&1 | fmt) | tee -a $out; fi
  echo "== round $r: current build" | tee -a $out; python tools/measure_all.py $sel 2>&1 | fmt | tee -a $out
done
