// placeholder until the throughput kernel lands
#include "kernels.h"
namespace tfhe_hip {
bool pbs_fft_wave_supported(uint32_t, uint32_t, uint32_t) { return false; }
void launch_pbs_fft_wave(hipStream_t, const PbsArgs &, const FftTables &) { HX_PANIC("throughput kernel not built"); }
}
