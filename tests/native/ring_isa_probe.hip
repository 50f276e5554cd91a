// ISA probe (device-only compile of two instantiations of the loader / consumer decode kernel; tools/isa_summary.py, tests/test_isa_audit.py)
#include "ab/bd_gemv_ring.h"
template __global__ void bd::gemv_ring_kernel<bd::DT_F16, 6, 1>(const bd::RingParams);
template __global__ void bd::gemv_ring_kernel<bd::DT_BF16, 1, 1>(const bd::RingParams);
