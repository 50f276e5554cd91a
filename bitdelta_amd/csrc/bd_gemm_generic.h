// Edge-case MFMA kernel: any M, N, strides and alignment, K % 32 == 0 (the reference's own pack constraint,
// bitdelta/binary_gemm_kernel.py:13).  Used when the fast path's requirements (K % 64 == 0, 16-byte aligned rows)
// do not hold.  Same math and operand roles as bd_gemm_mfma.h, natural k order, operands loaded straight from
// global memory element by element (no LDS), one 32x32 output tile per wave, 4 waves (2x2) per block.
#pragma once
#include "bd_gemm_mfma.h"

namespace bd {

template <int DT, bool FUSED, bool OUT_F32>
__global__ void __launch_bounds__(256) delta_gemm_generic_kernel(const GemmParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int m = blockIdx.y * 64 + (wave >> 1) * 32 + l31;     // row this lane supplies for the X operand / owns in D
    const int nb = blockIdx.x * 64 + (wave & 1) * 32;            // first column of this wave's tile
    const int ns = nb + l31;                                     // column this lane supplies for the S / W operand
    const unsigned short* X = (const unsigned short*)p.A + (long long)b * p.sAb + (long long)min(m, p.M - 1) * p.sAm;
    const uint32_t* P = (const uint32_t*)p.P + (long long)b * p.sPb + min(ns, p.N - 1);
    constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;

    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    auto load8 = [&](const unsigned short* src) {
        u32x4_t f;
#pragma unroll
        for (int d = 0; d < 4; ++d) f[d] = (uint32_t)src[2 * d] | ((uint32_t)src[2 * d + 1] << 16);
        return f;
    };

    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const int k = k0 + 8 * h;
        const uint32_t byte = (P[(long long)(k >> 5) * p.N] >> (k & 31)) & 0xffu;
        u32x4_t sf;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            sf[d] = (((byte >> (2 * d)) & 1u) ? POS : NEG) | ((((byte >> (2 * d + 1)) & 1u) ? POS : NEG) << 16);
        acc = mfma32<DT>(sf, load8(X + k), acc);
    }
    if constexpr (FUSED) {
        const float* al = p.alpha + (long long)b * p.sAlb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = min(nb + (r & 3) + 8 * (r >> 2) + 4 * h, p.N - 1);
            acc[r] *= al[n / p.gsz];
        }
        const unsigned short* Wr = (const unsigned short*)p.W + (long long)min(ns, p.N - 1) * p.ldw;
        for (int k0 = 0; k0 < p.K; k0 += 16) {
            const int k = k0 + 8 * h;
            acc = mfma32<DT>(load8(Wr + k), load8(X + k), acc);
        }
    }
    if (m >= p.M) return;
    const float* al = p.alpha ? p.alpha + (long long)b * p.sAlb : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = nb + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n >= p.N) continue;
        const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
        float v = acc[r];
        if (!FUSED && p.accumulate) {
            const float cin = OUT_F32 ? ((const float*)p.C)[off] : half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]);
            v = cin + al[n / p.gsz] * v;
        } else if (!FUSED && p.round_mode == 1) {
            v = round_through_f16(v);
        }
        if constexpr (OUT_F32) ((float*)p.C)[off] = v;
        else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
    }
}

}  // namespace bd
