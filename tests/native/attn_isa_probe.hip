// Device-only translation unit for tests/test_isa_audit.py: the two shipped instantiations of the prefill attention kernel.
#include "../../bitdelta_amd/csrc/bd_attn_prefill.h"
template __global__ void bd::prefill_attn_kernel<bd::DT_BF16>(const bd::PrefillAttnParams);
template __global__ void bd::prefill_attn_kernel<bd::DT_F16>(const bd::PrefillAttnParams);
