// Decode path, delta only, reference sign layout, <= 16 activation rows per block: the reference's own published kernel shapes
//
//   y[b, m, :] = x[b, m, :] . S_b      binary_bmm(x [B, M, K], mask [B, K/32, N]), binary_matmul(x [M, K], mask [K/32, N])      M <= 16, no base weight, no scale
//
// Reference: bitdelta/binary_gemm_kernel.py:297-335 (binary_bmm; Triton kernel :186-295) and :153-184 (binary_matmul; kernel :48-151), benchmarked at
// M = 1, B in {1, 8, 16} and at B = 1, M in {1, 16}, N = K in {4096, 8192} in notebooks/binary_gemm_kernel_triton.ipynb:595-1044.
//
// Why its own kernel (profiles/r05_reference_notebook_shapes.txt): gemv_stream_kernel reads reference-layout sign words as one dword per
// lane -- a wave-load covers 4 word rows x 16 columns = four 64-byte segments -- keeps all <= 8 masks of a chunk in every block, falls back to the
// round-1 split-k kernel above 8 masks, and re-reads all 16 activation rows from L2 for every 16-column tile (M = 16 at 8192^2: 128 MB of x for
// 8 MB of sign words); these launches sat at 0.04 - 0.28 of the HBM peak.  Here the work is cut the other way:
//   * a block owns a SUPER-TILE of 16 * CW columns (CW = 4: dwordx4 sign loads, 64 columns; CW = 2: dwordx2, 32 columns -- twice the blocks when
//     N / 64 of them would leave CUs idle) of MC masks x rpm rows <= 16 activation rows: one mask per batch entry with M rows each (MC = 1, 2; the
//     whole batch in one launch, grid = N / (16 CW) x ceil(B / MC)), or ONE mask shared by all B * M <= 16 rows;
//   * a lane loads CW adjacent columns of its word row, so a wave-load is 4 word rows x 64 CW contiguous bytes (whole 128-byte lines, each consumed
//     by ONE instruction: the nt policy applies) -- up to 1 KiB per load instruction instead of 256 B -- and each x fragment feeds CW tiles;
//   * dword j of the load is the sign operand of 16-column tile j = columns {c0 + CW i + j}: a fixed permutation of the super-tile's
//     columns, undone for free when the partial sums are written to LDS (row i = 4 g + r of tile j is column CW (4 g + r) + j);
//   * D[col][row] = v_mfma_f32_16x16x32(sign fragment, x fragment) with the chunk's rows as the only non-zero columns of the x operand (a row's
//     result = its column of ITS mask's accumulators); sign fragment of step s = LUT[byte s] (the streaming kernel's conflict-free 16-copy
//     table); the fragments of the next step -- across stage boundaries too -- are read while this step's CW * MC MFMAs run;
//   * the 4 waves split k, NS stages of (MC sign loads + 4 x loads) in flight per wave, raw buffer loads (out of range = zeros, the
//     stream runs past its end without a branch); partial sums meet in LDS, the chunk's rows go round-robin over the waves, each summing its
//     row's four partials in wave order and storing 16 CW consecutive columns.
// Bound: one MFMA + one ds_read_b128 per 64 sign bytes = 16 B / clk / CU = ~9.8 TB/s chip-wide with the matrix pipe AND the
// LDS saturated -- the HBM roofline itself; measured, the M = 1 per-entry-mask launches sit at the board's POWER cap at ~4 TB/s (15 of the 16
// activation columns of every MFMA are empty).  Algorithmic bytes: masks * N * K / 8 (+ 2 B M K + out).
#pragma once
#include "bd_gemv_stream.h"

namespace bd {

typedef float rows_f32x2_t __attribute__((ext_vector_type(2)));

struct RowsParams {
    const unsigned short* X;   // [B, M, K] rows (strides sXb / sXm elements)
    const uint32_t* P;         // [B or 1, K/32, N] words (stride sPb words; 0 = one mask shared by every row)
    void* C;                   // [B, M, N] (strides sCb / sCm elements)
    int B, M, N, K;
    int rpm;                   // activation rows per mask: M (one mask per batch entry) or B * M (shared mask, one chunk)
    long long sXb, sXm, sPb, sCb, sCm;
    uint32_t x_bytes, p_bytes; // descriptor extents
    int round_mode, out_f32;
};

// MC masks x rpm rows <= 16 activation rows per block (the 16 columns of the MFMA's x operand); CW = sign-word columns per lane: 4 (dwordx4 loads,
// 64-column super-tiles) or 2 (dwordx2, 32-column super-tiles: twice the blocks when N / 64 of them would leave CUs idle).
// ABL (tests/native/rows_bench.hip only; wrong results by construction): 1 = no LUT reads (the sign words themselves are the operand),
// 2 = no MFMAs (the fragments are xor-folded into the accumulators), 4 = the stream is never re-issued (one round of loads)
template <int DT, int MC, int NS, int AUXP, int ABL = 0, int CW = 4>
__global__ void __launch_bounds__(256) delta_rows_kernel(const RowsParams p) {
    static_assert(MC == 1 || MC == 2 || MC == 4, "masks per block");
    static_assert(CW == 2 || CW == 4, "sign-word columns per lane");
    constexpr int SW = 16 * CW;                                          // super-tile width
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];     // [64 KiB sign LUT][4 waves x R rows x SW fp32 partial sums]
    float* const red = (float*)(dyn_lds + STREAM_LUT_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int nst = p.N / SW;
    const int c0 = ((int)blockIdx.x % nst) * SW, t0 = ((int)blockIdx.x / nst) * MC;
    const int nit = p.K >> 7;                                            // 128-k iterations
    const int per = (nit + 3) >> 2;
    const int it_lo = min(wave * per, nit), it_hi = min(it_lo + per, nit);
    const int rows_all = p.B * p.M, R = MC * p.rpm;                      // rows of the launch / of a full chunk (host: R <= 16)
    const int row0 = t0 * p.rpm;                                         // first row of this chunk

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(p.P, p.p_bytes);
    // x operand: column li of the MFMA = row row0 + li of the launch (columns past the chunk: zeros, never read back)
    uint32_t x_off = STREAM_OOB;
    const int my_t = li / p.rpm;                                         // the mask (of this chunk) that row li belongs to
    if (li < R && row0 + li < rows_all) {
        const int gr = row0 + li, b = gr / p.M, m = gr - b * p.M;
        x_off = (uint32_t)(((long long)b * p.sXb + (long long)m * p.sXm) * 2);
    }
    uint32_t p_off[MC];
#pragma unroll
    for (int t = 0; t < MC; ++t)
        p_off[t] = (t0 + t) * p.rpm < rows_all ? (uint32_t)(((long long)(t0 + t) * p.sPb + c0 + CW * li) * 4) : STREAM_OOB;
    const uint32_t row_bytes = (uint32_t)p.N * 4u;

    typedef uint32_t wd_t __attribute__((ext_vector_type(CW)));
    struct Stage { u32x4_t xf[4]; wd_t wd[MC]; };
    auto issue = [&](Stage& st, int it) {
        // past the wave's k range: loads that touch no memory.  The test is on the LANE's word row (a per-lane select): written on the
        // wave-uniform iteration index, hipcc turns it into a branch around the loads and waits vmcnt(0) behind every one of them.
        // (A base offset that is already out of range stays out of range: the added row offset is < 2 GiB, host-checked.)
        const uint32_t irow = (uint32_t)(4 * it + g);
        const bool ok = (int)irow < 4 * it_hi;
        uint32_t xo = ok ? x_off + irow * 64u : STREAM_OOB;
        asm volatile("" : "+v"(xo));                                     // (opaque: keeps the select a v_cndmask -- hipcc otherwise sinks the loads
                                                                         //  into both arms of a divergent branch with s_waitcnt vmcnt(0) between them)
#pragma unroll
        for (int s = 0; s < 4; ++s) st.xf[s] = buf_load16<0>(rx, xo + 16u * s);
#pragma unroll
        for (int t = 0; t < MC; ++t) {
            uint32_t po = ok ? p_off[t] + irow * row_bytes : STREAM_OOB;
            asm volatile("" : "+v"(po));
            if constexpr (CW == 4) st.wd[t] = buf_load16<AUXP>(rp, po);
            else st.wd[t] = __builtin_bit_cast(wd_t, __builtin_amdgcn_raw_buffer_load_b64(rp, (int)po, 0, AUXP));
        }
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

    f32x4_t acc[MC][CW];
#pragma unroll
    for (int t = 0; t < MC; ++t)
#pragma unroll
        for (int j = 0; j < CW; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t copy_off = (uint32_t)li * 16u;
    auto lut = [&](uint32_t w, int s) -> u32x4_t {                       // byte s -> bits 8..15, copy slot -> bits 0..7 (one v_perm_b32)
        const uint32_t off = __builtin_amdgcn_perm(w, copy_off, 0x0c0c0400u + ((uint32_t)s << 8));
        if constexpr (ABL & 1) return u32x4_t{off, w, off ^ w, w + (uint32_t)s};
        return *(const u32x4_t*)(dyn_lds + off);
    };
    u32x4_t sf[2][MC][CW];                                               // sign fragments of two consecutive MFMA steps
    auto read_sf = [&](int buf, const Stage& s_, int step) {
#pragma unroll
        for (int t = 0; t < MC; ++t)
#pragma unroll
            for (int j = 0; j < CW; ++j) sf[buf][t][j] = lut(s_.wd[t][j], step);
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
                    for (int j = 0; j < CW; ++j) {
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

    // partial sums -> LDS in column order: lane (li = row, g), register r of tile j of the row's mask holds column CW * (4 g + r) + j
#pragma unroll
    for (int t = 0; t < MC; ++t) {
        if (my_t == t && li < R) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = &red[(wave * R + li) * SW + CW * (4 * g + r)];
                if constexpr (CW == 4) *(f32x4_t*)dst = f32x4_t{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
                else *(rows_f32x2_t*)dst = rows_f32x2_t{acc[t][0][r], acc[t][1][r]};
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // rows of the chunk round-robin over the waves: the four partials in wave order, SW consecutive columns per row
    for (int r = wave; r < R; r += 4) {
        const int gr = row0 + r;
        if (gr < rows_all && lane < SW) {
            float v = red[r * SW + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += red[(w * R + r) * SW + lane];
            if (p.round_mode == 1) v = round_through_f16(v);             // (store_out, bd_gemv.h: delta only, no scale)
            const int b = gr / p.M, m = gr - b * p.M;
            const long long off = (long long)b * p.sCb + (long long)m * p.sCm + c0 + lane;
            if (p.out_f32) ((float*)p.C)[off] = v;
            else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
        }
    }
}

}  // namespace bd
