// Device-code-only translation unit for tests/test_isa_audit.py: the decode-step kernels whose prologues / waits were fixed in round 5.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S decode_isa_probe.hip
#include "../../bitdelta_amd/csrc/bd_gemv_rows.h"      // (pulls in bd_gemv_stream.h and bd_serving.h)
template __global__ void bd::decode_attn_kernel<bd::DT_F16, 4, 2>(const bd::AttnParams);
template __global__ void bd::gemv_stream_kernel<bd::DT_F16, 6, true, 2, 4, 1, 2, 1, 2, 0, 1>(const bd::StreamParams);      // o projection: resident rows
template __global__ void bd::delta_rows_kernel<bd::DT_F16, 2, 4, 2, 0, 4>(const bd::RowsParams);
template __global__ void bd::delta_rows_kernel<bd::DT_F16, 1, 4, 2, 0, 2>(const bd::RowsParams);
