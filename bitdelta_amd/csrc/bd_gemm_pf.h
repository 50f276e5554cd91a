// W1A16 binary-delta GEMM, ping-pong schedule with FULL-TILE phases (successor of the half-tile schedule kept in tests/native/ab/bd_gemm_pp.h).
//
// The half-tile schedule alternates half-tile phases (16 MFMAs) and pays ~130-150 ticks of idle matrix pipe at each of its 4 barrier
// hand-offs per k-tile, plus ~170 ticks in M1 for the expansion fillers.  Here a phase covers a WHOLE k-tile:
//     L(kt): ds_read all 4 k-steps' X fragments and the tile's sign words, expand ALL sign fragments (VALU), wait for tile kt+1
//     M(kt): 4*TM*TN MFMAs back to back, with this wave's LDS-DMA pieces of tile kt+NS-1 in their shadow -- nothing else
// Two barriers per k-tile; group 1 (waves 4-7) runs one phase behind group 0.  Costs 4x the fragment registers (X: 16*TM,
// S: 16*TN VGPRs) on top of the 128 accumulator registers (242 VGPRs at 256x256, no spills).
// Measured in the same process as the half-tile schedule (profiles/r01_pf_vs_pp.txt): +8..11 % at 256x256 and 256x128.
// Fused mode: delta loop -> acc *= alpha -> base loop (X and W tiles by LDS-DMA, 3-slot ring) in the same schedule.
// Rejected here as well (profiles/r01_pp_timeline.txt): handing the pipe over 2/4/8 MFMAs before the phase end -- fewer ticks in
// the traced block but 17 % MORE GPU cycles overall (GRBM_GUI_ACTIVE 241 k vs 205 k) and 1110 vs 1240 TF.
//
// Ring safety (NS >= 4): reads of tile kt's slot happen in L(kt) (phase 2kt for group 0, 2kt+1 for group 1).  Its refill (tile
// kt+NS) is issued in M(kt+1), phases >= 2kt+3.  Tile kt+1 is first read in L(kt+1) (phase 2kt+2); every wave waits
// vmcnt((NS-3)*DPW) for its own pieces of tile kt+1 at the end of L(kt) (phases 2kt / 2kt+1), before the barrier that ends phase 2kt+1.
#pragma once
#include "bd_gemm_mfma.h"

namespace bd {

#ifdef BD_TRACE
__device__ unsigned long long bd_trace_pf[2][8][4];
#define BD_PF_STAMP(id)                                                                                      \
    do {                                                                                                     \
        if (blockIdx.x == 0 && (wave & 3) == 0 && kt >= 16 && kt < 24) {                                     \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
            if (lane == 0) bd_trace_pf[wave >> 2][kt - 16][id] = t_;                                         \
        }                                                                                                    \
    } while (0)
#else
#define BD_PF_STAMP(id) do { } while (0)
#endif


template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) delta_gemm_pf_kernel(const GemmParams p) {
    constexpr int DT = Cfg::DT, BM = Cfg::BM, BN = Cfg::BN, NS = Cfg::NS;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int A_BYTES = Cfg::A_BYTES, STAGE_D = Cfg::STAGE_D, STAGE_B = Cfg::STAGE_B;
    constexpr int A_PW = Cfg::A_PW, BW_PW = Cfg::BW_PW, W_PW = Cfg::W_PW;
    static_assert(Cfg::NW == 8 && Cfg::WAVES_M == 2 && NS >= 4, "full-tile ping-pong: 8 waves, two groups");
    static_assert(!Cfg::FUSED || Cfg::NSB >= 3, "fused base loop needs a 3-slot ring in this schedule (256x128 tile); 256x256 fused "
                                                "stays on the half-tile A/B schedule");

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
    const int b = blockIdx.y;
    const int nk = p.K >> 6;

    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));

    const char* a_src = p.A + ((long long)b * p.sAb + (long long)m0 * p.sAm) * 2;
    const char* p_src = (const char*)p.P + ((long long)b * p.sPb + n0) * 4;
    uint32_t a_voff[A_PW], bw_voff[BW_PW], a_lds[A_PW], bw_lds[BW_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int rg = wave * A_PW + i;
        const int r = rg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int rr = min(m0 + r, p.M - 1) - m0;
        a_voff[i] = (uint32_t)rr * (uint32_t)p.sAm * 2u + (uint32_t)c * 16u;
        a_lds[i] = rg * 1024;
    }
#pragma unroll
    for (int i = 0; i < BW_PW; ++i) {
        const int idx = (wave * BW_PW + i) % Cfg::BW_PIECES;
        const int hh = idx / (BN / 64), seg = idx % (BN / 64);
        const int nn = min(n0 + seg * 64 + lane, p.N - 1) - n0;
        bw_voff[i] = (uint32_t)hh * (uint32_t)p.N * 4u + (uint32_t)nn * 4u;
        bw_lds[i] = A_BYTES + hh * BN * 4 + seg * 256;
    }
    const int swz = (l31 >> 1) & 7;
    uint32_t a_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a_rd[s] = (uint32_t)(wm * WM + l31) * 128u + (uint32_t)(((4 * h + s) ^ swz) * 16);
    const uint32_t bw_rd = A_BYTES + h * BN * 4 + (wn * WN + l31) * 4;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto issue = [&](int kt, int slot) {
        const char* as = a_src + (long long)kt * 128;
        const char* ps = p_src + (long long)kt * 2 * p.N * 4;
        const uint32_t base = lds0 + slot * STAGE_D;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) dma16(a_voff[i], as, base + a_lds[i]);
#pragma unroll
        for (int i = 0; i < BW_PW; ++i) dma4(bw_voff[i], ps, base + bw_lds[i]);
    };
    // OPT & 1: sign expansion by table lookup -- LUT[byte] = the 8 (+-1.0) 16-bit values of that byte's signs (one ds_read_b128 per
    // fragment, 2 VALU for the address) instead of 8 VALU per fragment.  4 KiB behind the ring; built once per block.
    constexpr bool USE_LUT = (Cfg::OPT & 1) != 0;
    constexpr int LUT_OFF = NS * STAGE_D;
    if constexpr (USE_LUT) {
        static_assert(!USE_LUT || (LUT_OFF + 4096 <= 160 * 1024), "LUT does not fit");
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
    wait_vmcnt<(NS - 2) * Cfg::DPW_D>();
    phase_end();                                  // tile 0 resident (and the LUT visible)
    if (grp == 1) phase_end();                    // stagger

    constexpr int NMF = 4 * TM * TN;
    constexpr int NPIECE = A_PW + BW_PW;
    constexpr int EVERY = NMF / (NPIECE + 1) > 0 ? NMF / (NPIECE + 1) : 1;
    int slot_c = 0, slot_i = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + slot_c * STAGE_D;
        u32x4_t xf[4][TM], sf[4][TN];
        // ---------------- L(kt)
        BD_PF_STAMP(0);
        uint32_t wraw[TN];
        {
            const uint32_t waddr = lds0 + slot_c * STAGE_D + bw_rd;
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
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * TM > 15 ? 15 : 4 * TM) : "memory");   // the words are back (counter field is 4 bits)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (USE_LUT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t byte = (wraw[j] >> (8 * s)) & 0xffu;            // v_bfe_u32
                    sf[s][j] = *(const u32x4_t*)(smem + LUT_OFF + byte * 16);      // v_lshl_add + ds_read_b128
                }
        } else
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
        wait_vmcnt<(NS - 3) * Cfg::DPW_D>();      // own pieces of tile kt+1 landed
#ifdef BD_TRACE
        __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
        BD_PF_STAMP(1);
        phase_end();
        BD_PF_STAMP(2);
        // ---------------- M(kt)
        {
            const int kt_i = min(kt + NS - 1, nk - 1);
            const char* as = a_src + (long long)kt_i * 128;
            const char* ps = p_src + (long long)kt_i * 2 * p.N * 4;
            const uint32_t base = lds0 + slot_i * STAGE_D;
            if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < NMF; ++t) {
                const int s = t / (TM * TN), j = (t % (TM * TN)) / TM, i = t % TM;
                acc[i][j] = mfma32<DT>(sf[s][j], xf[s][i], acc[i][j]);
                const int pc = t / EVERY;
                if (t % EVERY == EVERY - 1 && pc < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, base + a_lds[pc < A_PW ? pc : 0]);
                    else dma4(bw_voff[pc >= A_PW && pc - A_PW < BW_PW ? pc - A_PW : 0], ps,
                              base + bw_lds[pc >= A_PW && pc - A_PW < BW_PW ? pc - A_PW : 0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(0);
        }
        BD_PF_STAMP(3);
        phase_end();
        slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
        slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
    }
    if (grp == 0) phase_end();
    wait_vmcnt<0>();

    // =========================== fused: acc *= alpha, then the base loop x.W^T in the same full-tile ping-pong ==============
    if constexpr (Cfg::FUSED) {
        constexpr int NSB = Cfg::NSB;
        const float* al = p.alpha + (long long)b * p.sAlb;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = min(n0 + wn * WN + j * 32 + 8 * q + 4 * h + e, p.N - 1);
                    const float a = al[n / p.gsz];
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j][q * 4 + e] *= a;
                }
        const char* w_src = p.W + (long long)n0 * p.ldw * 2;
        uint32_t w_voff[W_PW], w_lds[W_PW];
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            const int rg = wave * W_PW + i;
            const int r = rg * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const int rr = min(n0 + r, p.N - 1) - n0;
            w_voff[i] = (uint32_t)rr * (uint32_t)p.ldw * 2u + (uint32_t)c * 16u;
            w_lds[i] = A_BYTES + rg * 1024;
        }
        uint32_t w_rd[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            w_rd[s] = A_BYTES + (uint32_t)(wn * WN + l31) * 128u + (uint32_t)(((4 * h + s) ^ swz) * 16);
        auto issue_b = [&](int kt, int slot) {
            const char* as = a_src + (long long)kt * 128;
            const char* ws = w_src + (long long)kt * 128;
            const uint32_t base = lds0 + slot * STAGE_B;
#pragma unroll
            for (int i = 0; i < A_PW; ++i) dma16(a_voff[i], as, base + a_lds[i]);
#pragma unroll
            for (int i = 0; i < W_PW; ++i) dma16(w_voff[i], ws, base + w_lds[i]);
        };
        phase_end();                              // every wave has left the delta ring
#pragma unroll
        for (int t = 0; t < NSB - 1; ++t) issue_b(min(t, nk - 1), t);
        wait_vmcnt<(NSB - 2) * Cfg::DPW_B>();
        phase_end();                              // tile 0 resident
        if (grp == 1) phase_end();
        constexpr int NPB = A_PW + W_PW;
        constexpr int EVB = NMF / NPB > 0 ? NMF / NPB : 1;
        int sb_c = 0, sb_i = NSB - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + sb_c * STAGE_B;
            u32x4_t xf[4][TM], wf[4][TN];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[s][i] = *(const u32x4_t*)(st + a_rd[s] + i * 4096);
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[s][j] = *(const u32x4_t*)(st + w_rd[s] + j * 4096);
            }
            wait_vmcnt<(NSB - 3) * Cfg::DPW_B>();  // own pieces of tile kt+1 landed (issued in M(kt-1))
            phase_end();
            {
                const int kt_i = min(kt + NSB - 1, nk - 1);
                const char* as = a_src + (long long)kt_i * 128;
                const char* ws = w_src + (long long)kt_i * 128;
                const uint32_t base = lds0 + sb_i * STAGE_B;
                if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < NMF; ++t) {
                    const int s = t / (TM * TN), j = (t % (TM * TN)) / TM, i = t % TM;
                    acc[i][j] = mfma32<DT>(wf[s][j], xf[s][i], acc[i][j]);
                    const int pc = t / EVB;
                    if (t % EVB == EVB - 1 && pc < NPB) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, base + a_lds[pc < A_PW ? pc : 0]);
                        else dma16(w_voff[pc >= A_PW && pc - A_PW < W_PW ? pc - A_PW : 0], ws,
                                   base + w_lds[pc >= A_PW && pc - A_PW < W_PW ? pc - A_PW : 0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(0);
            }
            phase_end();
            sb_c = (sb_c + 1 == NSB) ? 0 : sb_c + 1;
            sb_i = (sb_i + 1 == NSB) ? 0 : sb_i + 1;
        }
        if (grp == 0) phase_end();
        wait_vmcnt<0>();
    }

    __builtin_amdgcn_s_barrier();
    gemm_epilogue<Cfg>(p, acc, smem, m0, n0, wm, wn, b, lane, wave);
}

}  // namespace bd
