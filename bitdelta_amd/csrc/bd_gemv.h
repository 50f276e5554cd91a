// Decode path (HBM-bound): few activation rows (R = B*M <= 16), every weight byte read once.
//
//   y[b,m,:] = x[b,m,:] . W^T + alpha[b] * (x[b,m,:] . S_b)        (fused; W optional)
//
// Reference call sites: DiffCompressModule.forward at decode, demo/demo_backend.py:93-98 (M = 1, B = tenants);
// BinaryDiff.forward, bitdelta/diff.py:33-39 with a single token.
//
// Work split: grid = (ceil(N/64) column tiles) x (KS k-slices); every block streams its [64 n] x [k-slice] piece of
//   * the 16-bit base weight W [N,K] (rows are k-contiguous): MFMA 16x16x32 with the activations as the 16-wide
//     second operand -- 16-byte loads straight to VGPRs, no LDS (streamed once, not shared between waves);
//   * each tenant's packed sign words P_b [K/32, N]: lane = output column (256-byte coalesced word rows), the 32 signs
//     of a word are expanded to +-1.0 pairs (2 VALU / pair) and contracted with the wave-uniform activation pairs held
//     in SGPRs by v_dot2c_f32_{bf16,f16} (fp32 accumulate) -- the "sign-flip GEMV" variant of the MFMA path.
// Partial sums of the k-slices go to an fp32 workspace [KS][R][N]; gemv_reduce_kernel sums them and rounds once.
// With KS == 1 the block writes the final result itself.
#pragma once
#include "bd_common.h"

namespace bd {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct GemvParams {
    const unsigned short* X;   // [B, M, K]
    const uint32_t* P;         // [B or 1, K/32, N]
    const unsigned short* W;   // [N, K] or nullptr (delta only)
    const float* alpha;        // fp32 [B or 1, G] or nullptr
    void* C;                   // [B, M, N]
    float* ws;                 // [KS][R][N] fp32 partials (KS > 1)
    int B, M, N, K, R;
    long long sXb, sPb, sCb;
    int sXm, sCm, ldw, sAlb, gsz;
    int KS, kslice;            // k-slice length (multiple of 128)
    int round_mode, accumulate, out_f32;
};

template <int DT> __device__ __forceinline__ float dot2acc(uint32_t a, uint32_t b, float c) {
    if constexpr (DT == DT_BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
    else return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
}
template <int DT> __device__ __forceinline__ f32x4_t mfma16(u32x4_t a, u32x4_t b, f32x4_t c) {
    if constexpr (DT == DT_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <int DT> __device__ __forceinline__ void store_out(const GemvParams& p, int r, int n, float v) {
    const int b = r / p.M, m = r - b * p.M;
    const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
    if (p.accumulate) {
        const float cin = p.out_f32 ? ((const float*)p.C)[off] : half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]);
        v += cin;
    } else if (!p.W && !p.alpha && p.round_mode == 1) {
        v = round_through_f16(v);
    }
    if (p.out_f32) ((float*)p.C)[off] = v;
    else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
}

template <int DT, int RMAX>
__global__ void __launch_bounds__(256) gemv_kernel(const GemvParams p) {
    __shared__ float red[4][RMAX][64];    // delta partials per wave
    __shared__ float bs[RMAX][64];        // base GEMV tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int k_lo = ks * p.kslice, k_hi = min(p.K, k_lo + p.kslice);
    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));

    // ---------------- base part: D[n][r] += W[n][k] x[r][k], one 16-column group per wave ----------------
    if (p.W) {
        const int li = lane & 15, g = lane >> 4;
        const int nw = min(n0 + wave * 16 + li, p.N - 1);
        const unsigned short* wr = p.W + (long long)nw * p.ldw + 8 * g;
        const bool xvalid = li < p.R;
        const int xb = xvalid ? li / p.M : 0, xm = xvalid ? li - xb * p.M : 0;    // idle lanes re-read row 0, then mask
        const unsigned short* xr = p.X + (long long)xb * p.sXb + (long long)xm * p.sXm + 8 * g;
        const uint32_t keep = xvalid ? 0xffffffffu : 0u;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k = k_lo; k < k_hi; k += 32) {
            const u32x4_t wf = *(const u32x4_t*)(wr + k);
            u32x4_t xf = *(const u32x4_t*)(xr + k);
            xf = xf & u32x4_t{keep, keep, keep, keep};
            acc = mfma16<DT>(wf, xf, acc);
        }
        // lane holds r = li, n_local = 4g + reg
        if (li < RMAX) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[li][wave * 16 + 4 * g + e] = acc[e];
        }
    }

    // ---------------- delta part: lane = column, waves interleave the slice's word rows ----------------
    float dacc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) dacc[r] = 0.f;
    const int nc = min(n0 + lane, p.N - 1);
    for (int i = (k_lo >> 5) + wave; i < (k_hi >> 5); i += 4) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < p.R) {
                const int b = r / p.M, m = r - b * p.M;
                const uint32_t w = ~p.P[(long long)b * p.sPb + (long long)i * p.N + nc];
                const uint32_t* xs = (const uint32_t*)(p.X + (long long)b * p.sXb + (long long)m * p.sXm + 32 * i);
                const uint32_t rlo = __builtin_amdgcn_perm(w, w, 0x01000100u), rhi = __builtin_amdgcn_perm(w, w, 0x03020302u);
                float a = dacc[r];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int qq = q & 7;
                    u16x2_t v = __builtin_bit_cast(u16x2_t, q < 8 ? rlo : rhi);
                    u16x2_t sh;
                    sh.x = (unsigned short)(15 - 2 * qq);
                    sh.y = (unsigned short)(14 - 2 * qq);
                    v = v << sh;
                    const uint32_t sd = (__builtin_bit_cast(uint32_t, v) & 0x80008000u) | one2;
                    a = dot2acc<DT>(sd, xs[q], a);
                }
                dacc[r] = a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) red[wave][r][lane] = dacc[r];
    __syncthreads();

    // ---------------- combine: wave w finishes rows r = w, w+4, ... ----------------
    const int n = n0 + lane;
    if (n < p.N) {
        for (int r = wave; r < p.R; r += 4) {
            const int b = r / p.M;
            float d = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
            if (p.alpha) d *= p.alpha[(long long)b * p.sAlb + n / p.gsz];
            if (p.W) d += bs[r][lane];
            if (p.KS == 1) store_out<DT>(p, r, n, d);
            else p.ws[((long long)ks * p.R + r) * p.N + n] = d;
        }
    }
}

template <int DT>
__global__ void __launch_bounds__(256) gemv_reduce_kernel(const GemvParams p) {
    const int n = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (n >= p.N) return;
    float s = 0.f;
    for (int ks = 0; ks < p.KS; ++ks) s += p.ws[((long long)ks * p.R + r) * p.N + n];
    store_out<DT>(p, r, n, s);
}

}  // namespace bd
