# the reference's keyswitch test (tests/cpp/reference_gpu_tests.cpp) under the three behaviours of cuda_malloc_async /
# cuda_drop, and the runtime alone (pool_probe.hip); writes gpurun_out/r04h_ks_cpp_diag2.txt
mkdir -p gpurun_out
g++ -std=c++17 -O2 -o /tmp/rgt tests/cpp/reference_gpu_tests.cpp tfhe_rs_amd/lib/libtfhe_hip_backend.so oracle/libtfhe_oracle.so -Wl,-rpath,$PWD/tfhe_rs_amd/lib -Wl,-rpath,$PWD/oracle
{
echo "== runtime alone"; timeout 60 tools/probes/pool_probe
for mode in pool_hipfree pool sync; do
  for rep in 1 2; do
    echo "== TFHE_HIP_MALLOC_ASYNC=$mode (run $rep)"; TFHE_HIP_MALLOC_ASYNC=$mode timeout 60 /tmp/rgt reference ks_decrypt_custom_mod
  done
done
} > gpurun_out/r04h_ks_cpp_diag2.txt 2>&1
cat gpurun_out/r04h_ks_cpp_diag2.txt
