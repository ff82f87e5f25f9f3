// pool_probe.hip — does hipFree() of a hipMallocAsync pointer corrupt the pool?  A small live allocation A (an "index array")
// is checked after every round of: allocate B from the pool, fill it, free it with hipFree (mode 0) or with hipFreeAsync +
// device synchronisation (mode 1), allocate C from the pool, memset it.  Prints overlaps of C with A and changes of A.
//   hipcc --offload-arch=gfx950 -O2 -o pool_probe tools/probes/pool_probe.hip && ./pool_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint64_t *p, size_t n, uint64_t v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v + i;
}
int main() {
  for (int mode = 0; mode < 2; ++mode) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t na = 193, nb = 193 * 743;
    std::vector<uint64_t> ha(na), back(na), zeros(nb, 0);
    for (size_t i = 0; i < na; ++i) ha[i] = i;
    int overlaps = 0, changed = 0;
    for (int outer = 0; outer < 20; ++outer) {
      uint64_t *A = nullptr;
      CK(hipMallocAsync((void **)&A, na * 8, s));
      CK(hipMemsetAsync(A, 0, na * 8, s));
      CK(hipMemcpyAsync(A, ha.data(), na * 8, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      for (int round = 0; round < 6; ++round) {
        uint64_t *B = nullptr;
        CK(hipMallocAsync((void **)&B, nb * 8, s));
        CK(hipMemsetAsync(B, 0, nb * 8, s));
        CK(hipMemcpyAsync(B, zeros.data(), nb * 8, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        fill<<<(nb + 255) / 256, 256, 0, s>>>(B, nb, 0x1111000000000000ull);
        CK(hipStreamSynchronize(s));
        if ((char *)B < (char *)(A + na) && (char *)A < (char *)(B + nb)) ++overlaps;
        CK(hipDeviceSynchronize());
        if (mode == 0) CK(hipFree(B));
        else { CK(hipFreeAsync(B, nullptr)); CK(hipDeviceSynchronize()); }
        CK(hipMemcpyAsync(back.data(), A, na * 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (back != ha) ++changed;
      }
      CK(hipDeviceSynchronize());
      if (mode == 0) CK(hipFree(A));
      else { CK(hipFreeAsync(A, nullptr)); CK(hipDeviceSynchronize()); }
    }
    printf("mode %d (%s): %d rounds where a new pool allocation overlapped the live one, %d rounds where the live one changed\n", mode,
           mode == 0 ? "hipFree of pool pointers" : "hipFreeAsync + hipDeviceSynchronize", overlaps, changed);
    CK(hipStreamDestroy(s));
  }
  return 0;
}
