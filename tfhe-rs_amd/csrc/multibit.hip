// placeholder until the multi-bit kernels land
#include "kernels.h"
namespace tfhe_hip {
void launch_pbs_multi_bit(hipStream_t, uint32_t, uint32_t, const MultiBitArgs &, const FftTables &, uint64_t *) {
  HX_PANIC("multi-bit PBS not built");
}
}
