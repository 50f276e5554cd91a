// W1A16 binary-delta GEMM, software-pipelined schedule: ONE s_barrier per k-tile.
//
// Same math, LDS image, k-permutation, DMA ring and epilogue as bd_gemm_mfma.h / bd_gemm_pp.h.  Time structure: every wave
// issues its MFMAs continuously; in the shadow of the 2*TM... TM*TN MFMAs of k-step s it (a) ds_reads the X fragments of
// step s+1 into the OTHER register set, (b) expands the sign fragments of step s+1 (one dword = 2 VALU per MFMA) and
// (c) issues its LDS-DMA pieces.  Nothing is consumed in the step it is loaded in, so no wave ever waits on LDS latency,
// and the only block-wide synchronisation is the barrier that hands a ring slot over (placed at the start of step 2).
//
// Ring safety (NS slots): the barrier at step 2 of tile kt is passed only after every wave finished step 3 of tile kt-1,
// i.e. all reads of tile kt-1's slot have been consumed; refills of that slot (tile kt+NS-1) are issued after it.  Before the
// barrier every wave waits vmcnt((NS-3)*DPW) = its own pieces of tile kt+1, whose first read (the sign words) follows it.
// A/B REFERENCE ONLY: compiled into tests/native/bd_harness (-DBD_AB_VARIANTS), never into libbitdelta_hip.so.
#pragma once
#include "../../../bitdelta_amd/csrc/bd_gemm_mfma.h"

namespace bd {

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) delta_gemm_sp_kernel(const GemmParams p) {
    constexpr int DT = Cfg::DT, BM = Cfg::BM, BN = Cfg::BN, NS = Cfg::NS;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int A_BYTES = Cfg::A_BYTES, STAGE_D = Cfg::STAGE_D, STAGE_B = Cfg::STAGE_B;
    constexpr int A_PW = Cfg::A_PW, W_PW = Cfg::W_PW, BW_PW = Cfg::BW_PW;
    static_assert(NS >= 3, "ring depth");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
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

    // one dword (2 signs) of the 8-sign fragment `f[j]` from the replicated 16-bit chunk rep[j]; q0 = 0 or 4
    auto expand_dw = [&](u32x4_t (&f)[TN], const uint32_t (&rep)[TN], int q0, int idx) {
        const int j = idx / 4, d = idx % 4, q = q0 + d;
        u16x2_t v = __builtin_bit_cast(u16x2_t, rep[j]);
        u16x2_t sh;
        sh.x = (unsigned short)(15 - 2 * q);
        sh.y = (unsigned short)(14 - 2 * q);
        v = v << sh;
        f[j][d] = (__builtin_bit_cast(uint32_t, v) & 0x80008000u) | one2;
    };

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
        // prologue: NS-2 tiles in flight beyond tile 0 (the loop's refill of slot NS-1 starts at kt = 0, step 2)
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) issue(min(t, nk - 1), t);
        wait_vmcnt<(NS - 2) * Cfg::DPW_D>();
        __builtin_amdgcn_s_barrier();

        uint32_t rlo[TN], rhi[TN], wnext[TN];
        u32x4_t xe[TM], xo[TM], se[TN], so[TN];       // even / odd step register sets
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const uint32_t w = ~*(const uint32_t*)(smem + bw_rd + j * 128);
            rlo[j] = __builtin_amdgcn_perm(w, w, 0x01000100u);
            rhi[j] = __builtin_amdgcn_perm(w, w, 0x03020302u);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) xe[i] = *(const u32x4_t*)(smem + a_rd[0] + i * 4096);
#pragma unroll
        for (int idx = 0; idx < 4 * TN; ++idx) expand_dw(se, rlo, 0, idx);

        constexpr int NMF = TM * TN;              // MFMAs per step
        constexpr int NPIECE = A_PW + BW_PW;
        int slot_c = 0, slot_i = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + slot_c * STAGE_D;
            const int slot_n = (slot_c + 1 == NS) ? 0 : slot_c + 1;
            const char* stn = smem + slot_n * STAGE_D;
            const int kt_i = min(kt + NS - 1, nk - 1);
            const char* as = a_src + (long long)kt_i * 128;
            const char* ps = p_src + (long long)kt_i * 2 * p.N * 4;
            const uint32_t dbase = lds0 + slot_i * STAGE_D;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4_t(&xc)[TM] = (s & 1) ? xo : xe;
                u32x4_t(&xn)[TM] = (s & 1) ? xe : xo;
                u32x4_t(&sc)[TN] = (s & 1) ? so : se;
                u32x4_t(&sn)[TN] = (s & 1) ? se : so;
                if (s == 2) {
                    wait_vmcnt<(NS - 3) * Cfg::DPW_D>();          // own pieces of tile kt+1 landed
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();                  // slot hand-over (see header)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < TN; ++j) wnext[j] = *(const uint32_t*)(stn + bw_rd + j * 128);
                }
                if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < NMF; ++t) {
                    const int j = t / TM, i = t % TM;
                    acc[i][j] = mfma32<DT>(sc[j], xc[i], acc[i][j]);
                    // (a) next step's X fragments: one ds_read_b128 behind each of the first TM MFMAs
                    if (t < TM) {
                        if (s < 3) xn[t] = *(const u32x4_t*)(st + a_rd[s + 1] + t * 4096);
                        else xn[t] = *(const u32x4_t*)(stn + a_rd[0] + t * 4096);
                    }
                    // (b) next step's sign fragments: 4*TN dwords over NMF MFMAs
                    if (s == 3 && t == 0) {
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) {
                            const uint32_t w = ~wnext[jj];
                            rlo[jj] = __builtin_amdgcn_perm(w, w, 0x01000100u);
                            wnext[jj] = __builtin_amdgcn_perm(w, w, 0x03020302u);     // becomes rhi after this step
                        }
                    }
#pragma unroll
                    for (int e = t * 4 * TN / NMF; e < (t + 1) * 4 * TN / NMF; ++e) {
                        if (s == 0) expand_dw(sn, rlo, 4, e);          // step 1: low chunk, pairs 4..7
                        else if (s == 1) expand_dw(sn, rhi, 0, e);     // step 2: high chunk, pairs 0..3
                        else if (s == 2) expand_dw(sn, rhi, 4, e);     // step 3
                        else expand_dw(sn, rlo, 0, e);                 // next tile's step 0 (rlo already the new tile's)
                    }
                    // (c) LDS-DMA pieces of tile kt+NS-1, after the hand-over barrier: steps 2 and 3
                    if (s >= 2) {
                        const int slot_t = (s - 2) * NMF + t;          // 0 .. 2*NMF-1
                        constexpr int EV = (2 * NMF) / NPIECE > 0 ? (2 * NMF) / NPIECE : 1;
                        const int pc = slot_t / EV;
                        if (slot_t % EV == EV - 1 && pc < NPIECE) {
                            if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, dbase + a_lds[pc < A_PW ? pc : 0]);
                            else dma4(bw_voff[pc >= A_PW && pc - A_PW < BW_PW ? pc - A_PW : 0], ps,
                                      dbase + bw_lds[pc >= A_PW && pc - A_PW < BW_PW ? pc - A_PW : 0]);
                        }
                    }
                    if constexpr (!(Cfg::OPT & 1)) __builtin_amdgcn_sched_barrier(0);    // keep the interleave as written
                }
                if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(0);
                if (s == 3) {
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) rhi[jj] = wnext[jj];
                }
            }
            slot_c = slot_n;
            slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
        }
        wait_vmcnt<0>();
    }

    static_assert(!Cfg::FUSED, "fused base loop: use the ping-pong kernel (software-pipelined base loop not written yet)");

    __builtin_amdgcn_s_barrier();
    gemm_epilogue<Cfg>(p, acc, smem, m0, n0, wm, wn, b, lane, wave);
}

}  // namespace bd
