// Decode path, delta only, ONE MASK PER ROW in the reference layout: the reference's own published kernel shape
//
//   y[b, 0, :] = x[b, 0, :] . S_b            binary_bmm(x [B, 1, K], mask [B, K/32, N])      B >= 2, no base weight, no scale
//
// Reference: bitdelta/binary_gemm_kernel.py:297-335 (binary_bmm; Triton kernel :186-295), benchmarked at M = 1, B in {8, 16},
// N = K in {4096, 8192} in notebooks/binary_gemm_kernel_triton.ipynb:800-1044.
//
// Why its own kernel (profiles/r05_reference_notebook_shapes.txt): gemv_stream_kernel reads reference-layout sign words as one dword per
// lane -- a wave-load covers 4 word rows x 16 columns = four 64-byte segments -- keeps all <= 8 masks of a chunk in every block, and falls
// back to the round-1 split-k kernel above 8 masks; these launches sat at 0.19 - 0.28 of the HBM peak.  With one mask per row nothing is
// shared between the rows except the LUT, so the work is cut the other way:
//   * a block owns a 64-COLUMN super-tile of MC masks (MC = 1, 2, 4: grid = N/64 x ceil(B / MC), ~one block per CU at the published
//     shapes); a lane loads dwordx4 = 4 adjacent columns of its word row, so a wave-load is 4 word rows x 256 contiguous bytes (whole
//     128-byte lines, each consumed by ONE instruction: the nt policy applies) -- 1 KiB per load instruction instead of 256 B;
//   * dword j of the load is the sign operand of 16-column tile j = columns {c0 + 4 i + j}: a fixed permutation of the super-tile's
//     columns, undone for free when the partial sums are written to LDS (row i = 4 g + r of tile j is column 16 g + 4 r + j);
//   * D[col][row] = v_mfma_f32_16x16x32(sign fragment, x fragment) with the MC rows of the chunk as the only non-zero columns of the x
//     operand (row t's result = column t of mask t's accumulators); sign fragment of step s = LUT[byte s] (the streaming kernel's
//     conflict-free 16-copy table); the fragments of the next step -- across stage boundaries too -- are read while this step's
//     4 * MC MFMAs run;
//   * the 4 waves split k, NS stages of (MC sign loads + 4 x loads) in flight per wave, raw buffer loads (out of range = zeros, the
//     stream runs past its end without a branch); partial sums meet in LDS, wave t sums mask t's four partials in wave order and stores
//     64 consecutive columns.
// Bound: one MFMA + one ds_read_b128 per 64 sign bytes = 16 B / clk / CU = ~9.8 TB/s chip-wide with the matrix pipe AND the
// LDS saturated -- the HBM roofline itself; the kernel is measured against 8 TB/s.  Algorithmic bytes: B * N * K / 8 (+ 2 B K + out).
#pragma once
#include "bd_gemv_stream.h"

namespace bd {

struct RowsParams {
    const unsigned short* X;   // [B, K] rows (stride sXb elements)
    const uint32_t* P;         // [B, K/32, N] words (stride sPb words)
    void* C;                   // [B, N] (stride sCb elements)
    int B, N, K;
    long long sXb, sPb, sCb;
    uint32_t x_bytes, p_bytes; // descriptor extents
    int round_mode, out_f32;
};

// ABL (tests/native/rows_bench.hip only; wrong results by construction): 1 = no LUT reads (the sign words themselves are the operand),
// 2 = no MFMAs (the fragments are xor-folded into the accumulators), 4 = the stream is never re-issued (one round of loads)
template <int DT, int MC, int NS, int AUXP, int ABL = 0>
__global__ void __launch_bounds__(256) delta_rows_kernel(const RowsParams p) {
    static_assert(MC == 1 || MC == 2 || MC == 4, "masks per block");
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];     // [64 KiB sign LUT][4 waves x MC x 64 fp32 partial sums]
    float* const red = (float*)(dyn_lds + STREAM_LUT_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int nst = p.N >> 6;
    const int c0 = ((int)blockIdx.x % nst) << 6, t0 = ((int)blockIdx.x / nst) * MC;
    const int nit = p.K >> 7;                                            // 128-k iterations
    const int per = (nit + 3) >> 2;
    const int it_lo = min(wave * per, nit), it_hi = min(it_lo + per, nit);

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(p.P, p.p_bytes);
    // x operand: column li of the MFMA = row t0 + li of the chunk (columns >= MC: zeros, never read back)
    const uint32_t x_off = (li < MC && t0 + li < p.B) ? (uint32_t)((long long)(t0 + li) * p.sXb * 2) : STREAM_OOB;
    uint32_t p_off[MC];
#pragma unroll
    for (int t = 0; t < MC; ++t)
        p_off[t] = t0 + t < p.B ? (uint32_t)(((long long)(t0 + t) * p.sPb + c0 + 4 * li) * 4) : STREAM_OOB;
    const uint32_t row_bytes = (uint32_t)p.N * 4u;

    struct Stage { u32x4_t xf[4]; u32x4_t wd[MC]; };
    auto issue = [&](Stage& st, int it) {
        // past the wave's k range: loads that touch no memory.  The test is on the LANE's word row (a per-lane select): written on the
        // wave-uniform iteration index, hipcc turns it into a branch around the loads and waits vmcnt(0) behind every one of them.
        // (A base offset that is already out of range stays out of range: the added row offset is < 2 GiB, host-checked.)
        const uint32_t irow = (uint32_t)(4 * it + g);
        const bool ok = (int)irow < 4 * it_hi;
        const uint32_t xo = ok ? x_off + irow * 64u : STREAM_OOB;
#pragma unroll
        for (int s = 0; s < 4; ++s) st.xf[s] = buf_load16<0>(rx, xo + 16u * s);
#pragma unroll
        for (int t = 0; t < MC; ++t) st.wd[t] = buf_load16<AUXP>(rp, ok ? p_off[t] + irow * row_bytes : STREAM_OOB);
        __builtin_amdgcn_sched_barrier(0);                               // stages enter the load queue in stream order
    };

    Stage st[NS];
    int ii = it_lo;
#pragma unroll
    for (int u = 0; u < NS; ++u) issue(st[u], ii++);

    {   // sign LUT, 16 copies (entry e of copy c at e * 256 + c * 16): gemv_stream_kernel's table
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
#pragma unroll
        for (int j = 0; j < 4096 / 256; ++j) {
            const int slot = threadIdx.x + 256 * j, ee = slot >> 4;
            u32x4_t w;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = (((ee >> (2 * d)) & 1) ? POS : NEG) | ((((ee >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
            *(u32x4_t*)(dyn_lds + slot * 16) = w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4_t acc[MC][4];
#pragma unroll
    for (int t = 0; t < MC; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t copy_off = (uint32_t)li * 16u;
    auto lut = [&](uint32_t w, int s) -> u32x4_t {                       // byte s -> bits 8..15, copy slot -> bits 0..7 (one v_perm_b32)
        const uint32_t off = __builtin_amdgcn_perm(w, copy_off, 0x0c0c0400u + ((uint32_t)s << 8));
        if constexpr (ABL & 1) return u32x4_t{off, w, off ^ w, w + (uint32_t)s};
        return *(const u32x4_t*)(dyn_lds + off);
    };
    u32x4_t sf[2][MC][4];                                                // sign fragments of two consecutive MFMA steps
    auto read_sf = [&](int buf, const Stage& s_, int step) {
#pragma unroll
        for (int t = 0; t < MC; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) sf[buf][t][j] = lut(s_.wd[t][j], step);
    };

    // whole rounds of NS stages, at least one; no branch around the loads, no exit in the middle (see gemv_stream_kernel's main loop)
    const int cnt = max(it_hi - it_lo, 1);
    int f = 0;
    read_sf(0, st[0], 0);
    do {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) read_sf((s + 1) & 1, st[u], s + 1);
                else read_sf(0, st[(u + 1) % NS], 0);                     // the next stage's first step (re-issued one round ahead)
#pragma unroll
                for (int t = 0; t < MC; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (ABL & 2)
                            acc[t][j] = __builtin_bit_cast(f32x4_t, __builtin_bit_cast(u32x4_t, acc[t][j]) ^ sf[s & 1][t][j] ^ st[u].xf[s]);
                        else acc[t][j] = mfma16<DT>(sf[s & 1][t][j], st[u].xf[s], acc[t][j]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(ABL & 4)) issue(st[u], ii++);
        }
        f += NS;
    } while (f < cnt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the run-ahead (out-of-range) loads of the last round

    // partial sums -> LDS in column order: lane (li = t, g), register r of tile j holds column 16 g + 4 r + j of mask t
#pragma unroll
    for (int t = 0; t < MC; ++t) {
        if (li == t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *(f32x4_t*)&red[(wave * MC + t) * 64 + 16 * g + 4 * r] = f32x4_t{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave < MC && t0 + wave < p.B) {                                  // wave t: mask t's 64 columns, the four partials in wave order
        float v = red[(0 * MC + wave) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += red[(w * MC + wave) * 64 + lane];
        if (p.round_mode == 1) v = round_through_f16(v);                 // (store_out, bd_gemv.h: delta only, no scale)
        const long long off = (long long)(t0 + wave) * p.sCb + c0 + lane;
        if (p.out_f32) ((float*)p.C)[off] = v;
        else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
    }
}

}  // namespace bd
