// Fused binary-delta Linear in ONE pass over k:  C = X.W^T + alpha * (X.S)   (BinaryDiff.forward, bitdelta/diff.py:33-39;
// DiffCompressModule.forward, demo/demo_backend.py:93-98).
//
// bd_gemm_pf.h's fused mode runs two k loops over one accumulator set (delta loop, acc *= alpha, base loop): X is fetched and
// its fragments are read from LDS twice.  Here every k-tile carries all three operands -- X [BM x 64], W [BN x 64] and the
// [2 x BN] packed sign words -- and each X fragment feeds TWO MFMAs, one against the W fragment and one against the expanded
// sign fragment, into two accumulator sets:
//     accW[m][n] += X.W^T      accS[m][n] += X.S        epilogue: out = accW + alpha[n] * accS   (one rounding)
// A 256 x 128 output tile therefore has the MFMA : LDS-read ratio of the 256 x 256 delta-only tile (32 MFMAs per 20 b128 reads per
// wave per k-tile, vs 16 per 20 in the two-loop form), X is read from L2/HBM once, and the schedule is bd_gemm_pf.h's full-tile
// ping-pong unchanged:  L(kt) = all ds_reads + sign expansion (LUT) + vmcnt wait,  M(kt) = 32 MFMAs with the 7 LDS-DMA pieces of
// tile kt+NS-1 in their shadow; two groups of 4 waves one phase apart; 2 barriers per k-tile.
//
// Ring: NS = 3 slots of 49 KiB (X 32 K + W 16 K + signs 1 K) + the 4 KiB sign LUT = 151 KiB of the 160 KiB LDS.
// Ring safety (NS = 3): tile kt's slot is read in L(kt) (phase 2kt for group 0, 2kt+1 for group 1) and refilled with tile kt+3 in
// M(kt+1) (phases 2kt+3 / 2kt+4).  Tile kt+1 was issued in M(kt-1); each wave waits vmcnt(0) for its own pieces at the end of L(kt)
// (nothing newer is in flight at that point), before the barrier that precedes any read of tile kt+1.
#pragma once
#include "bd_gemm_mfma.h"

namespace bd {

// PAIR = 1 (multi-tenant prefill of short prompts, demo/demo_backend.py:297-299: T tenants x <= 64 rows): a 128-row tile holds TWO batch
// entries -- rows 0..63 = entry 2y, rows 64..127 = entry 2y + 1 of the launch (blockIdx.y = pair index), each with its own sign words
// and scales.  The two waves rows of the 2 x 4 wave layout coincide with the two entries, so every wave still expands ONE mask; what
// the pair shares is the W panel (one W stream per CU serves two tenants: these shapes are bound by how fast a CU can pull W through
// L2) and the LDS-DMA / barrier schedule of a full 128-row tile.  The ring slot carries both entries' [2 x BN] sign words.
template <int DT_, int BM_, int BN_, int NS_, bool OUT_F32_, int OPT_ = 1, int PAIR_ = 0>
struct FxCfg : GemmCfg<DT_, BM_, BN_, 2, 4, NS_, true, OUT_F32_, OPT_> {
    using Base = GemmCfg<DT_, BM_, BN_, 2, 4, NS_, true, OUT_F32_, OPT_>;
    static constexpr int PAIR = PAIR_;
    static constexpr int PB_BYTES = Base::BW_BYTES * (PAIR ? 2 : 1);                     // sign words of a k-tile: one mask, or the pair's two
    static constexpr int PB_PIECES = Base::BW_PIECES * (PAIR ? 2 : 1);
    static constexpr int PB_PW = PB_PIECES >= Base::NW ? PB_PIECES / Base::NW : 1;
    static constexpr int STAGE_X = Base::A_BYTES + Base::W_BYTES + PB_BYTES;
    static constexpr int DPW_X = Base::A_PW + Base::W_PW + PB_PW;
    static_assert(!PAIR || (BM_ == 128 && (PB_PIECES % Base::NW == 0 || Base::NW % PB_PIECES == 0)), "pair mode: 128-row tile = 2 x 64 rows");
    static constexpr int LUT_OFF = NS_ * STAGE_X;
    static constexpr int LDS_BYTES = NS_ * STAGE_X + 4096;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NS_ >= 3 && (NS_ - 2) * DPW_X <= 63, "ring depth / vmcnt field");
    static_assert(2 * Base::TM * Base::TN * 16 <= 128, "two accumulator sets must fit 128 VGPRs");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) delta_gemm_fx_kernel(const GemmParams p) {
    constexpr int DT = Cfg::DT, BM = Cfg::BM, BN = Cfg::BN, NS = Cfg::NS;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int A_BYTES = Cfg::A_BYTES, W_BYTES = Cfg::W_BYTES, STAGE = Cfg::STAGE_X;
    constexpr int A_PW = Cfg::A_PW, BW_PW = Cfg::PB_PW, W_PW = Cfg::W_PW;
    constexpr int BW_OFF = A_BYTES + W_BYTES, LUT_OFF = Cfg::LUT_OFF;
    constexpr bool PAIR = Cfg::PAIR != 0;
    constexpr bool USE_LUT = (Cfg::OPT & 1) != 0;
    static_assert(Cfg::NW == 8 && Cfg::WAVES_M == 2, "full-tile ping-pong: 8 waves, two groups");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int grp = wm;
    const int h = lane >> 5, l31 = lane & 31;

    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // split-k (mid-size M: too few tiles to fill the chip): y = b * ksplit + ks; partials go to an fp32 workspace indexed by y
    const int by = blockIdx.y;
    const int ksp = max(p.ksplit, 1);
    const int bq = by / ksp;                   // batch entry, or pair of entries
    const int ksi = by - bq * ksp;
    const int b = PAIR ? 2 * bq : bq;          // (first) batch entry of the tile
    // pair mode: entry of a tile half (the second entry of the last pair of an odd batch does not exist: its half re-reads the first
    // entry's operands and is not stored)
    const bool has2 = !PAIR || b + 1 < p.nbatch;
    const int nk_all = p.K >> 6;
    const int kt_lo = (int)((long long)ksi * nk_all / ksp), kt_hi = (int)((long long)(ksi + 1) * nk_all / ksp);
    const int nk = kt_hi - kt_lo;

    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));

    // ---- DMA source offsets (per lane, constant over k) and ring-slot destinations
    const char* a_src = p.A + ((long long)b * p.sAb + (long long)m0 * p.sAm) * 2 + (long long)kt_lo * 128;
    const char* w_src = p.W + (long long)n0 * p.ldw * 2 + (long long)kt_lo * 128;
    const char* p_src = (const char*)p.P + ((long long)b * p.sPb + n0) * 4 + (long long)kt_lo * 2 * p.N * 4;
    uint32_t a_voff[A_PW], w_voff[W_PW], bw_voff[BW_PW], a_lds[A_PW], w_lds[W_PW], bw_lds[BW_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int rg = wave * A_PW + i;
        const int r = rg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        if constexpr (PAIR) {
            const int half = (r >> 6) && has2, rr = min(r & 63, p.M - 1);
            a_voff[i] = (uint32_t)(((long long)half * p.sAb + (long long)rr * p.sAm) * 2) + (uint32_t)c * 16u;
        } else {
            const int rr = min(m0 + r, p.M - 1) - m0;
            a_voff[i] = (uint32_t)rr * (uint32_t)p.sAm * 2u + (uint32_t)c * 16u;
        }
        a_lds[i] = rg * 1024;
    }
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int rg = wave * W_PW + i;
        const int r = rg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int rr = min(n0 + r, p.N - 1) - n0;
        w_voff[i] = (uint32_t)rr * (uint32_t)p.ldw * 2u + (uint32_t)c * 16u;
        w_lds[i] = A_BYTES + rg * 1024;
    }
#pragma unroll
    for (int i = 0; i < BW_PW; ++i) {
        const int idx2 = (wave * BW_PW + i) % Cfg::PB_PIECES;
        const int half = idx2 / Cfg::BW_PIECES, idx = idx2 % Cfg::BW_PIECES;            // (half = 0 without PAIR)
        const int hh = idx / (BN / 64), seg = idx % (BN / 64);
        const int nn = min(n0 + seg * 64 + lane, p.N - 1) - n0;
        bw_voff[i] = (uint32_t)hh * (uint32_t)p.N * 4u + (uint32_t)nn * 4u + (uint32_t)((half && has2) ? p.sPb * 4 : 0);
        bw_lds[i] = BW_OFF + half * Cfg::BW_BYTES + hh * BN * 4 + seg * 256;
    }
    const int swz = (l31 >> 1) & 7;
    uint32_t a_rd[4], w_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a_rd[s] = (uint32_t)(wm * WM + l31) * 128u + (uint32_t)(((4 * h + s) ^ swz) * 16);
        w_rd[s] = A_BYTES + (uint32_t)(wn * WN + l31) * 128u + (uint32_t)(((4 * h + s) ^ swz) * 16);
    }
    const uint32_t bw_rd = BW_OFF + (PAIR ? wm * Cfg::BW_BYTES : 0) + h * BN * 4 + (wn * WN + l31) * 4;

    f32x16_t accS[TM][TN], accW[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accS[i][j][r] = 0.f; accW[i][j][r] = 0.f; }

    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto issue = [&](int kt, int slot) {
        const char* as = a_src + (long long)kt * 128;
        const char* ws = w_src + (long long)kt * 128;
        const char* ps = p_src + (long long)kt * 2 * p.N * 4;
        const uint32_t base = lds0 + slot * STAGE;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) dma16(a_voff[i], as, base + a_lds[i]);
#pragma unroll
        for (int i = 0; i < W_PW; ++i) dma16(w_voff[i], ws, base + w_lds[i]);
#pragma unroll
        for (int i = 0; i < BW_PW; ++i) dma4(bw_voff[i], ps, base + bw_lds[i]);
    };
    if constexpr (USE_LUT) {      // LUT[byte] = the 8 (+-1.0) 16-bit values of that byte's signs (see bd_gemm_pf.h)
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
        for (int e = threadIdx.x; e < 256; e += Cfg::NT) {
            u32x4_t v;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                v[d] = (((e >> (2 * d)) & 1) ? POS : NEG) | ((((e >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
            *(u32x4_t*)(smem + LUT_OFF + e * 16) = v;
        }
    }
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(min(t, nk - 1), t);
    wait_vmcnt<(NS - 2) * Cfg::DPW_X>();
    phase_end();                                  // tile 0 resident (and the LUT visible)
    if (grp == 1) phase_end();                    // stagger: group 1 runs one phase behind

    constexpr int NMF = 8 * TM * TN;              // 4 k-steps x (S, W) x TM x TN
    constexpr int NPIECE = A_PW + W_PW + BW_PW;
    constexpr int EVERY = NMF / (NPIECE + 1) > 0 ? NMF / (NPIECE + 1) : 1;
    int slot_c = 0, slot_i = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + slot_c * STAGE;
        u32x4_t xf[4][TM], sf[4][TN], wf[4][TN];
        // ---------------- L(kt)
        uint32_t wraw[TN];
        {
            const uint32_t waddr = lds0 + slot_c * STAGE + bw_rd;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(wraw[j]) : "v"(waddr), "n"(j * 128) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[s][i] = *(const u32x4_t*)(st + a_rd[s] + i * 4096);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * TM > 15 ? 15 : 4 * TM) : "memory");   // the sign words are back
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (USE_LUT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t byte = (wraw[j] >> (8 * s)) & 0xffu;
                    sf[s][j] = *(const u32x4_t*)(smem + LUT_OFF + byte * 16);
                }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const uint32_t w = ~wraw[j];
                const uint32_t lo = (w & 0xffffu) | ((w << 15) & 0x7fff0000u), hi = (w >> 16) | ((w >> 1) & 0x7fff0000u);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int q = (s & 1) * 4 + d;
                        sf[s][j][d] = (((s < 2 ? lo : hi) << (15 - 2 * q)) & 0x80008000u) | one2;
                    }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[s][j] = *(const u32x4_t*)(st + w_rd[s] + j * 4096);
        wait_vmcnt<(NS - 3) * Cfg::DPW_X>();      // own pieces of tile kt+1 landed
        phase_end();
        // ---------------- M(kt)
        {
            const int kt_i = min(kt + NS - 1, nk - 1);
            const char* as = a_src + (long long)kt_i * 128;
            const char* ws = w_src + (long long)kt_i * 128;
            const char* ps = p_src + (long long)kt_i * 2 * p.N * 4;
            const uint32_t base = lds0 + slot_i * STAGE;
#pragma unroll
            for (int t = 0; t < NMF; ++t) {
                const int s = t / (2 * TM * TN), r = t % (2 * TM * TN), j = r / (2 * TM), i = (r % (2 * TM)) >> 1;
                if ((t & 1) == 0) accS[i][j] = mfma32<DT>(sf[s][j], xf[s][i], accS[i][j]);
                else accW[i][j] = mfma32<DT>(wf[s][j], xf[s][i], accW[i][j]);
                const int pc = t / EVERY;
                if (t % EVERY == EVERY - 1 && pc < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, base + a_lds[pc < A_PW ? pc : 0]);
                    else if (pc < A_PW + W_PW)
                        dma16(w_voff[pc >= A_PW && pc - A_PW < W_PW ? pc - A_PW : 0], ws,
                              base + w_lds[pc >= A_PW && pc - A_PW < W_PW ? pc - A_PW : 0]);
                    else
                        dma4(bw_voff[pc >= A_PW + W_PW && pc - A_PW - W_PW < BW_PW ? pc - A_PW - W_PW : 0], ps,
                             base + bw_lds[pc >= A_PW + W_PW && pc - A_PW - W_PW < BW_PW ? pc - A_PW - W_PW : 0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        phase_end();
        slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
        slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
    }
    if (grp == 0) phase_end();
    wait_vmcnt<0>();

    // ---- out = accW + alpha[n] * accS  (fp32), then the shared staged epilogue
    const int b_w = PAIR ? (has2 ? b + wm : b) : b;      // this wave's batch entry
    const float* al = p.alpha + (long long)b_w * p.sAlb;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = min(n0 + wn * WN + j * 32 + 8 * q + 4 * h + e, p.N - 1);
                const float a = al[n / p.gsz];
#pragma unroll
                for (int i = 0; i < TM; ++i) accW[i][j][q * 4 + e] = __builtin_fmaf(a, accS[i][j][q * 4 + e], accW[i][j][q * 4 + e]);
            }
    __builtin_amdgcn_s_barrier();
    if constexpr (PAIR) {
        // rows of this wave = rows 0..63 of ITS entry; output slab = the entry's own (split-k: slab entry * ksplit + ks of the workspace)
        if (wm == 0 || has2) gemm_epilogue<Cfg>(p, accW, smem, m0 - wm * WM, n0, wm, wn, (b + wm) * ksp + ksi, lane, wave);
    } else {
        gemm_epilogue<Cfg>(p, accW, smem, m0, n0, wm, wn, by, lane, wave);
    }
}

}  // namespace bd
