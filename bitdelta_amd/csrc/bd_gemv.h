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

// words of RB k32-rows x RMAX activation rows are fetched as one batch (RB*R independent 256-byte loads per wave in flight)
template <int RMAX> struct GemvBatch { static constexpr int RB = (32 / RMAX) < 1 ? 1 : ((32 / RMAX) > 8 ? 8 : (32 / RMAX)); };

// k per slice: the block's activation slice is staged in LDS (RMAX rows x kslice x 2 B), static LDS budget 64 KB
// (measured: 4096-k slices for <= 6 rows cut occupancy to 3 blocks/CU and were 15-20 % slower than 1024-k slices + more k-splits)
constexpr int gemv_kslice_max(int /*rmax*/) { return 1024; }

template <int DT, int RMAX>
__global__ void __launch_bounds__(256) gemv_kernel(const GemvParams p) {
    constexpr int RB = GemvBatch<RMAX>::RB;
    constexpr int XROW = gemv_kslice_max(RMAX) * 2 + 16;    // padded LDS row of the activation slice (bytes)
    __shared__ float red[4][RMAX][64];    // delta partials per wave
    __shared__ float bs[RMAX][64];        // base GEMV tile
    __shared__ __attribute__((aligned(16))) char xs_lds[RMAX * XROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int k_lo = ks * p.kslice, k_hi = min(p.K, k_lo + p.kslice);
    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));

    // ---------------- delta part, stage 1: put the first batch of sign words in flight before anything else ----------------
    const int nc = min(n0 + lane, p.N - 1);
    const int i_lo = (k_lo >> 5) + wave, i_hi = k_hi >> 5;          // this wave's word rows: i_lo, i_lo+4, ...
    auto load_batch = [&](uint32_t (&w)[RB][RMAX], int ib) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = min(ib + 4 * rb, (p.K >> 5) - 1);          // clamped rows are skipped at compute time
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int b = min(r, p.R - 1) / p.M;                  // rows >= R repeat the last row (results discarded)
                w[rb][r] = __builtin_nontemporal_load(&p.P[(long long)b * p.sPb + (long long)i * p.N + nc]);
            }
        }
    };
    uint32_t wcur[RB][RMAX];
    load_batch(wcur, i_lo);

    // ---------------- activation slice -> LDS (once per block; rows >= R repeat the last row) ----------------
    {
        const int kn = k_hi - k_lo;                                   // multiple of 32
        for (int idx = threadIdx.x; idx < RMAX * (kn >> 3); idx += 256) {
            const int r = idx / (kn >> 3), c = idx - r * (kn >> 3);
            const int rr = min(r, p.R - 1), b = rr / p.M, m = rr - b * p.M;
            *(u32x4_t*)(xs_lds + r * XROW + c * 16) =
                *(const u32x4_t*)(p.X + (long long)b * p.sXb + (long long)m * p.sXm + k_lo + c * 8);
        }
    }
    __syncthreads();

    // ---------------- base part: D[n][r] += W[n][k] x[r][k], one 16-column group per wave (weights streamed once) -------
    auto base_part = [&]() {
    if (p.W) {
        const int li = lane & 15, g = lane >> 4;
        const int nw = min(n0 + wave * 16 + li, p.N - 1);
        const unsigned short* wr = p.W + (long long)nw * p.ldw + 8 * g;
        const char* xl = xs_lds + min(li, RMAX - 1) * XROW + 16 * g;  // MFMA column li <-> activation row li (rows >= RMAX unused)
        const uint32_t keep = (li < p.R) ? 0xffffffffu : 0u;
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int k = k_lo;
        for (; k + 256 <= k_hi; k += 256) {                          // 8 x (1 KiB of W per wave) in flight
            u32x4_t wf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wf[u] = __builtin_nontemporal_load((const u32x4_t*)(wr + k + 32 * u));
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const u32x4_t x0 = *(const u32x4_t*)(xl + (k - k_lo + 32 * u) * 2) & u32x4_t{keep, keep, keep, keep};
                const u32x4_t x1 = *(const u32x4_t*)(xl + (k - k_lo + 32 * u + 32) * 2) & u32x4_t{keep, keep, keep, keep};
                acc0 = mfma16<DT>(wf[u], x0, acc0);
                acc1 = mfma16<DT>(wf[u + 1], x1, acc1);
            }
        }
        for (; k < k_hi; k += 32) {
            const u32x4_t wf = *(const u32x4_t*)(wr + k);
            const u32x4_t xf = *(const u32x4_t*)(xl + (k - k_lo) * 2) & u32x4_t{keep, keep, keep, keep};
            acc0 = mfma16<DT>(wf, xf, acc0);
        }
        // lane holds r = li, n_local = 4g + reg
        if (li < RMAX) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[li][wave * 16 + 4 * g + e] = acc0[e] + acc1[e];
        }
    }
    };

    // ---------------- delta part, stage 2: lane = column; signs -> +-1.0 pairs -> v_dot2c with the activation pairs --------
    // Branch-free over the RMAX rows and 4 independent accumulators per row: the 16 dot2 of one word would otherwise be one
    // dependent chain (measured: 5x the HBM time per tenant).  Activation pairs come from LDS as wave-uniform broadcasts.
    float dacc[RMAX][4];
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dacc[r][c] = 0.f;
    // The base part is load-bound and the delta part VALU-bound; blocks alternate the order so that, CU-wide, one half's
    // weight stream overlaps the other half's sign expansion (measured: the two phases otherwise simply add up).
    const bool delta_first = ((blockIdx.x + blockIdx.y) & 1) != 0;
    if (!delta_first) base_part();
    for (int ib = i_lo; ib < i_hi; ib += 4 * RB) {
        uint32_t wnext[RB][RMAX];
        const bool more = ib + 4 * RB < i_hi;
        if (more) load_batch(wnext, ib + 4 * RB);                    // next batch in flight under this batch's VALU work
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = ib + 4 * rb;
            if (i < i_hi) {
                const int koff = (32 * i - k_lo) * 2;
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const uint32_t w = ~wcur[rb][r];
                    const uint32_t rlo = (w & 0xffffu) | ((w << 15) & 0x7fff0000u);     // chunk | (chunk>>1)<<16, see bd_gemm_pp.h
                    const uint32_t rhi = (w >> 16) | ((w >> 1) & 0x7fff0000u);
                    const u32x4_t* xv = (const u32x4_t*)(xs_lds + r * XROW + koff);  // 64 bytes = 16 pairs, same address in every lane
                    u32x4_t x4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) x4[c] = xv[c];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const uint32_t rep = q < 8 ? rlo : rhi;
                        const uint32_t sd = ((rep << (15 - 2 * (q & 7))) & 0x80008000u) | one2;
                        dacc[r][q & 3] = dot2acc<DT>(sd, x4[q >> 2][q & 3], dacc[r][q & 3]);
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < RMAX; ++r) wcur[rb][r] = wnext[rb][r];
        }
    }
    if (delta_first) base_part();
#pragma unroll
    for (int r = 0; r < RMAX; ++r) red[wave][r][lane] = (dacc[r][0] + dacc[r][1]) + (dacc[r][2] + dacc[r][3]);
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
