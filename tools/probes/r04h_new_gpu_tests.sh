# GPU tier of the tests added in the third session of round 4 -> gpurun_out/r04h_new_gpu_tests2.log
mkdir -p gpurun_out
(time timeout 170 python -m pytest tests/test_pbs_noise.py tests/test_reference_gpu_tests_cpp.py -m gpu -q -s --durations=6) > gpurun_out/r04h_new_gpu_tests2.log 2>&1
grep -v "^test test_gpu\|^$" gpurun_out/r04h_new_gpu_tests2.log | tail -22
