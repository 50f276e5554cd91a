// W1A16 binary-delta GEMM, "ping-pong" schedule (the shipped large-tile kernel).
//
// Same math, operand roles, LDS image and k-permutation as bd_gemm_mfma.h (read that header first); what changes is
// the time structure.  Profiling the one-barrier-per-k-tile kernel (profiles/r01_*): MFMA pipe 52 % busy, waves 30 % in
// s_waitcnt/s_barrier, because the two waves that share a SIMD hit the barrier, the DMA issue and the LDS latency
// at the same time.  Here the 8 waves form two groups (waves 0-3 / 4-7: one wave of each group per SIMD) that run the
// same program ONE PHASE APART (group 1 executes one extra s_barrier up front, group 0 one at the end):
//
//     phase          group 0                      group 1
//     4kt            LOAD  half 0 of tile kt      MFMA  half 1 of tile kt-1
//     4kt+1          MFMA  half 0                 LOAD  half 0 of tile kt
//     4kt+2          LOAD  half 1                 MFMA  half 0
//     4kt+3          MFMA  half 1                 LOAD  half 1
//
//   LOAD = issue this wave's LDS-DMA pieces of tile kt+NS-1 (half 0 only), ds_read the X fragments of two k-steps,
//          read the sign words (half 0) and expand two steps' worth of +-1 fragments, wait for the tile after next.
//   MFMA = 2 k-steps x TM x TN v_mfma_f32_32x32x16 back to back at raised priority.
// Every phase ends in s_barrier, so on each SIMD one wave always owns the matrix pipe while its partner hides LDS
// latency, VALU sign expansion and DMA issue behind it.
//
// LDS ring safety (NS slots, tile t in slot t % NS): the last read of tile kt-1 is group 1's LOAD of its half 1 in
// phase 4kt-1; refills of that slot are issued in phases >= 4kt (after the barrier).  Tile kt is first read in phase
// 4kt; every wave has executed s_waitcnt vmcnt((NS-2)*DPW) for its own pieces of tile kt in its LOAD-half-1 of tile
// kt-1 (phase <= 4kt-1) and a barrier separates that from the first read.
#pragma once
#include "bd_gemm_mfma.h"

namespace bd {

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) delta_gemm_pp_kernel(const GemmParams p) {
    constexpr int DT = Cfg::DT, BM = Cfg::BM, BN = Cfg::BN, NS = Cfg::NS;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int A_BYTES = Cfg::A_BYTES, STAGE_D = Cfg::STAGE_D, STAGE_B = Cfg::STAGE_B;
    constexpr int A_PW = Cfg::A_PW, W_PW = Cfg::W_PW, BW_PW = Cfg::BW_PW;
    static_assert(Cfg::NW == 8 && Cfg::WAVES_M == 2, "ping-pong kernel: 8 waves, group = M half");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int grp = wm;                       // waves w and w+4 share a SIMD -> one wave of each group per SIMD
    const int h = lane >> 5, l31 = lane & 31;

    if constexpr (Cfg::OPT & 256) return;     // ablation: launch cost only
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
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
        __builtin_amdgcn_sched_barrier(0);                   // nothing sinks below the wait (VALU would lose its LDS-latency cover)
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0) only (vmcnt/expcnt fields saturated): this wave's LDS
                                                             // reads have returned before anyone refills; the builtin (not asm)
                                                             // keeps hipcc's own counter model exact -> counted waits elsewhere
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // 2*TM*TN MFMAs; `between(t)` runs after the t-th one (used to drop LDS-DMA issues into the MFMA shadow: the
    // wave is stalled on the busy matrix pipe there anyway, so the DMA's issue cycles are free).
    auto mfma_phase = [&](const u32x4_t (&sfa)[TN], const u32x4_t (&sfb)[TN], const u32x4_t (&xa)[TM], const u32x4_t (&xb)[TM],
                          auto&& between) {
        if constexpr (!(Cfg::OPT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 2 * TM * TN; ++t) {
            const int hsel = t / (TM * TN), j = (t % (TM * TN)) / TM, i = t % TM;
            acc[i][j] = hsel == 0 ? mfma32<DT>(sfa[j], xa[i], acc[i][j]) : mfma32<DT>(sfb[j], xb[i], acc[i][j]);
            between(t);
        }
        if constexpr (!(Cfg::OPT & 2)) __builtin_amdgcn_s_setprio(0);
    };
    auto nothing = [](int) {};

    // =========================== delta loop ===========================
    {
        auto issue = [&](int kt, int slot) {
            const char* as = a_src + (long long)kt * 128;
            const char* ps = p_src + (long long)kt * 2 * p.N * 4;
            const uint32_t base = lds0 + slot * STAGE_D;
#pragma unroll
            for (int i = 0; i < A_PW; ++i) dma16(a_voff[i], as, base + a_lds[i]);
#pragma unroll
            for (int i = 0; i < BW_PW; ++i) dma4(bw_voff[i], ps, base + bw_lds[i]);
        };
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) issue(min(t, nk - 1), t);
        wait_vmcnt<(NS - 2) * Cfg::DPW_D>();
        phase_end();                              // tile 0 resident
        if (grp == 1) phase_end();                // stagger: group 1 runs one phase behind

        int slot_c = 0, slot_i = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + slot_c * STAGE_D;
            u32x4_t xa[TM], xb[TM], sfa[TN], sfb[TN];
            uint32_t rhi[TN];
            // ---------------- LOAD half 0
            const int kt_i = min(kt + NS - 1, nk - 1);
            if constexpr (Cfg::OPT & 1) issue(kt_i, slot_i);
            // Sign words FIRST and by inline asm with a hand-counted wait: hipcc waits lgkmcnt(0) before their first use
            // (the 2*TM fragment reads issued behind them would then sit idle); lgkmcnt(2*TM) releases the expansion
            // VALU as soon as the words are back, so it overlaps the fragment reads' LDS latency.
            uint32_t wraw[TN];
            {
                const uint32_t waddr = lds0 + slot_c * STAGE_D + bw_rd;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(wraw[j]) : "v"(waddr), "n"(j * 128) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const u32x4_t*)(st + a_rd[0] + i * 4096);
#pragma unroll
            for (int i = 0; i < TM; ++i) xb[i] = *(const u32x4_t*)(st + a_rd[1] + i * 4096);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * TM) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const uint32_t w = ~wraw[j];
                const uint32_t rlo = __builtin_amdgcn_perm(w, w, 0x01000100u);
                rhi[j] = __builtin_amdgcn_perm(w, w, 0x03020302u);
                sfa[j] = expand_signs8(rlo, 0, one2);
                sfb[j] = expand_signs8(rlo, 4, one2);
            }
            phase_end();
            // ---------------- MFMA half 0 (+ this wave's LDS-DMA pieces of tile kt+NS-1, one per few MFMAs)
            {
                const char* as = a_src + (long long)kt_i * 128;
                const char* ps = p_src + (long long)kt_i * 2 * p.N * 4;
                const uint32_t base = lds0 + slot_i * STAGE_D;
                constexpr int NPIECE = A_PW + BW_PW, EVERY = (2 * TM * TN) / (NPIECE + 1) > 0 ? (2 * TM * TN) / (NPIECE + 1) : 1;
                mfma_phase(sfa, sfb, xa, xb, [&](int t) {
                    if constexpr (!(Cfg::OPT & 1)) {
                        const int pc = t / EVERY;
                        if (t % EVERY == EVERY - 1 && pc < NPIECE) {
                            __builtin_amdgcn_sched_barrier(0);     // keep the MFMA / DMA interleave as written
                            if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, base + a_lds[pc < A_PW ? pc : 0]);
                            else dma4(bw_voff[pc - A_PW < BW_PW ? (pc >= A_PW ? pc - A_PW : 0) : 0], ps,
                                      base + bw_lds[pc - A_PW < BW_PW ? (pc >= A_PW ? pc - A_PW : 0) : 0]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
            }
            phase_end();
            // ---------------- LOAD half 1
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const u32x4_t*)(st + a_rd[2] + i * 4096);
#pragma unroll
            for (int i = 0; i < TM; ++i) xb[i] = *(const u32x4_t*)(st + a_rd[3] + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                sfa[j] = expand_signs8(rhi[j], 0, one2);
                sfb[j] = expand_signs8(rhi[j], 4, one2);
            }
            wait_vmcnt<(NS - 2) * Cfg::DPW_D>();  // this wave's pieces of tile kt+1 have landed
            phase_end();
            // ---------------- MFMA half 1
            mfma_phase(sfa, sfb, xa, xb, nothing);
            phase_end();
            slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
            slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
        }
        if (grp == 0) phase_end();                // re-align the groups (equal barrier counts)
        wait_vmcnt<0>();
    }

    // =========================== fused: scale, then base loop (same ping-pong, X and W fragments from LDS) ============
    if constexpr (Cfg::FUSED) {
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
        phase_end();                              // all waves left the delta ring
        issue_b(0, 0);
        wait_vmcnt<0>();
        phase_end();
        if (grp == 1) phase_end();
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * STAGE_B;
            u32x4_t xa[TM], xb[TM], wa[TN], wb[TN];
            const int kt_i = min(kt + 1, nk - 1);
            if constexpr (Cfg::OPT & 1) issue_b(kt_i, (kt + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const u32x4_t*)(st + a_rd[0] + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) wa[j] = *(const u32x4_t*)(st + w_rd[0] + j * 4096);
#pragma unroll
            for (int i = 0; i < TM; ++i) xb[i] = *(const u32x4_t*)(st + a_rd[1] + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[j] = *(const u32x4_t*)(st + w_rd[1] + j * 4096);
            phase_end();
            {
                const char* as = a_src + (long long)kt_i * 128;
                const char* ws = w_src + (long long)kt_i * 128;
                const uint32_t base = lds0 + ((kt + 1) & 1) * STAGE_B;
                constexpr int NPIECE = A_PW + W_PW, EVERY = (2 * TM * TN) / NPIECE > 0 ? (2 * TM * TN) / NPIECE : 1;
                mfma_phase(wa, wb, xa, xb, [&](int t) {
                    if constexpr (!(Cfg::OPT & 1)) {
                        const int pc = t / EVERY;
                        if (t % EVERY == EVERY - 1 && pc < NPIECE) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, base + a_lds[pc < A_PW ? pc : 0]);
                            else dma16(w_voff[pc >= A_PW && pc - A_PW < W_PW ? pc - A_PW : 0], ws,
                                       base + w_lds[pc >= A_PW && pc - A_PW < W_PW ? pc - A_PW : 0]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
            }
            phase_end();
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const u32x4_t*)(st + a_rd[2] + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) wa[j] = *(const u32x4_t*)(st + w_rd[2] + j * 4096);
#pragma unroll
            for (int i = 0; i < TM; ++i) xb[i] = *(const u32x4_t*)(st + a_rd[3] + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[j] = *(const u32x4_t*)(st + w_rd[3] + j * 4096);
            wait_vmcnt<0>();
            phase_end();
            mfma_phase(wa, wb, xa, xb, nothing);
            phase_end();
        }
        if (grp == 0) phase_end();
        wait_vmcnt<0>();
    }

    // =========================== epilogue ===========================
    __builtin_amdgcn_s_barrier();
    if constexpr (Cfg::OPT & 128) {           // ablation: no C stores (accumulators kept alive)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
        return;
    }
    gemm_epilogue<Cfg>(p, acc, smem, m0, n0, wm, wn, b, lane, wave);
}

}  // namespace bd
