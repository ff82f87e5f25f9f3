mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_backend_parity.py tests/test_error_behaviour.py -m gpu -q -x -k "arith_hooks or cooperative or (pbs_bit_exact and ntt64_split) or (full_size_ntt and ntt64_split)" 2>&1 | tail -6 ) > gpurun_out/r05e_quick_gputest.log; cat gpurun_out/r05e_quick_gputest.log
timeout 600 bash tools/ab.sh "ntt_split" "oldfold noasm pace2 pace4" 2
cp gpurun_out/ab.txt gpurun_out/r05e_ab_split_fold.txt
