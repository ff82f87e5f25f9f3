mkdir -p gpurun_out
(time timeout 200 python -m pytest tests/test_reference_gpu_tests_cpp.py tests/test_pbs_golden.py tests/test_c_host.py tests/test_abi_surface.py -m gpu -q -s --durations=6) > gpurun_out/r04h_new_gpu_tests.log 2>&1
tail -25 gpurun_out/r04h_new_gpu_tests.log
