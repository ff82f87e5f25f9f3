# GPU tier of the restated reference tests (core_crypto and integer) -> gpurun_out/r04h_new_gpu_tests3.log
mkdir -p gpurun_out
(time timeout 110 python -m pytest tests/test_reference_gpu_tests_cpp.py -m gpu -q -s --durations=6 -k "integer or reference_parameter") > gpurun_out/r04h_new_gpu_tests3.log 2>&1
grep -v "^$" gpurun_out/r04h_new_gpu_tests3.log | tail -34
