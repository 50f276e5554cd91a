// Device-code-only translation unit for tests/test_isa_audit.py: the two shipped instantiations of the four-wave persistent kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S w4_isa_probe.hip
#include "../../bitdelta_amd/csrc/bd_gemm_w4.h"
template __global__ void bd::delta_gemm_w4_kernel<bd::W4Cfg<bd::DT_BF16, 256, 256, false, false, 1 | 8192>>(const bd::GemmParams);
template __global__ void bd::delta_gemm_w4_kernel<bd::W4Cfg<bd::DT_BF16, 256, 128, true, false, 1 | 8192>>(const bd::GemmParams);
