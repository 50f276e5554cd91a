// W1A16 binary-delta GEMM, "ping-pong" schedule (the shipped large-tile kernel).
//
// Same math, operand roles, LDS image and k-permutation as bd_gemm_mfma.h (read that header first); what changes is
// the time structure.  Profiling the one-barrier-per-k-tile kernel (profiles/r01_*): MFMA pipe 52 % busy, waves 30 % in
// s_waitcnt/s_barrier, because the two waves that share a SIMD hit the barrier, the DMA issue and the LDS latency
// at the same time.  Here the 8 waves form two groups (waves 0-3 / 4-7: one wave of each group per SIMD).  Every wave runs the
// same stream  L0 M0 L1 M1 | L0 M0 ...  per k-tile, where
//   L = ds_read only: X fragments of two k-steps (L1 also: tile kt+1's sign words -> replicated chunks, and the s_waitcnt
//       vmcnt for this wave's LDS-DMA pieces of tile kt+2);
//   M = 2 k-steps x TM x TN v_mfma_f32_32x32x16; in their shadow (one dword = 2 VALU per MFMA) the +-1 expansion of the NEXT
//       half's sign fragments and (M0) this wave's LDS-DMA pieces of tile kt+NS-1;
// Shipped schedule: every wave rendezvous before every phase and group 1 runs one phase behind (strict alternation).
// Experimental (OPT & 4): the groups differ ONLY in where they rendezvous: group 0 executes s_barrier before every L, group 1 before every M.
// Between two consecutive barriers group 0 therefore runs "L then M" while group 1 runs "M then L": on each SIMD one wave owns
// the matrix pipe while its partner hides LDS latency behind it, with ONE rendezvous per half-tile and an unsynchronised
// (bubble-free) role swap in the middle.  History (profiles/r01_pp_timeline.txt, s_memtime stamps): a 4-barrier strict
// alternation lost ~130 cycles of matrix pipe at each of its 4 hand-offs; expansion placed in the L phase made L longer than
// the 512-cycle M phase; a 14-VALU make_reps chain behind M1's first MFMA cost 170 cycles.
//
// LDS ring safety (NS >= 4 slots, tile t in slot t % NS; halves numbered n = 2kt+1, 2kt+2; barrier n = group 0's before L(n) =
// group 1's before M(n)).  Reads of tile kt's slot end with L(2kt+2): group 1 finishes it before barrier 2kt+2, group 0 before
// barrier 2kt+3.  The refill of that slot (tile kt+NS) is issued in M0 of tile kt+1 = M(2kt+3), which both groups enter after
// barrier 2kt+3.  Tile kt+1 is first read (its words) in L(2kt+2), by group 1 right after barrier 2kt+1; every wave waits
// vmcnt((NS-3)*DPW) for its pieces of tile kt+1 in L(2kt) = L1 of tile kt-1, which both groups finish before barrier 2kt+1.
// A/B REFERENCE ONLY: compiled into tests/native/bd_harness (-DBD_AB_VARIANTS), never into libbitdelta_hip.so.
#pragma once
#include "../../../bitdelta_amd/csrc/bd_gemm_mfma.h"

namespace bd {

#ifdef BD_TRACE
// development aid (tests/native/bd_trace): shader-clock stamps of block 0, waves 0 and 4, k-tiles 16..23, 8 stamps per tile
__device__ unsigned long long bd_trace_buf[2][8][8];
#define BD_STAMP(id)                                                                                         \
    do {                                                                                                     \
        if (blockIdx.x == 0 && (wave & 3) == 0 && kt >= 16 && kt < 24) {                                     \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
            if (lane == 0) bd_trace_buf[wave >> 2][kt - 16][id] = t_;                                        \
        }                                                                                                    \
    } while (0)
#else
#define BD_STAMP(id) do { } while (0)
#endif

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
    // Per wave and k-tile kt (see the header for the two-group phase picture):
    //   L0: ds_read X fragments of steps 0,1;  s_waitcnt for this wave's pieces of tile kt+1
    //   M0: 2*TM*TN MFMAs (steps 0,1) with, in their shadow, the expansion of steps 2,3 signs (from the word's high half,
    //       already in registers) and this wave's LDS-DMA pieces of tile kt+NS-1
    //   L1: ds_read X fragments of steps 2,3 and the sign words of tile kt+1
    //   M1: MFMAs of steps 2,3 with the expansion of tile kt+1's steps 0,1 signs in their shadow
    // LOAD phases are ds_read-only (~300 cycles), so they hide entirely behind the partner's 512-cycle MFMA phase.
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
        static_assert(NS >= 4, "ring depth (two tiles must be resident ahead of the reader)");  // NS = 3 is not enough here
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) issue(min(t, nk - 1), t);
        wait_vmcnt<(NS - 3) * Cfg::DPW_D>();      // tiles 0 and 1 resident (this wave's pieces) ...
        phase_end();                              // ... and everybody's

        // rep[j]: 16-bit chunk c of the inverted word in the low half and (c >> 1) in the high half, so ONE 32-bit shift by
        // 15-2q moves sign 2q to bit 15 and sign 2q+1 to bit 31: v_lshlrev_b32 + v_and_or_b32 = 2 VALU per dword.
        auto expand_dw = [&](u32x4_t (&fa)[TN], u32x4_t (&fb)[TN], const uint32_t (&rep)[TN], int idx) {
            const int which = idx / (4 * TN), j = (idx / 4) % TN, d = idx % 4;
            const int q = which * 4 + d;
            const uint32_t r = ((rep[j] << (15 - 2 * q)) & 0x80008000u) | one2;
            if (which == 0) fa[j][d] = r; else fb[j][d] = r;
        };
        auto make_reps = [&](uint32_t w, uint32_t& lo, uint32_t& hi) {      // w = inverted packed word
            lo = (w & 0xffffu) | ((w << 15) & 0x7fff0000u);
            hi = (w >> 16) | ((w >> 1) & 0x7fff0000u);
        };
        uint32_t rlo[TN], rhi[TN], wraw[TN];
        u32x4_t s0a[TN], s0b[TN], s1a[TN], s1b[TN];       // s0*: steps 0,1   s1*: steps 2,3
        {   // bootstrap: tile 0's words -> steps 0,1 fragments
#pragma unroll
            for (int j = 0; j < TN; ++j) make_reps(~*(const uint32_t*)(smem + bw_rd + j * 128), rlo[j], rhi[j]);
#pragma unroll
            for (int idx = 0; idx < 8 * TN; ++idx) expand_dw(s0a, s0b, rlo, idx);
        }
        // Rendezvous: group 0 executes it BEFORE each LOAD, group 1 BEFORE each MFMA phase -> one s_barrier per half-tile
        // for every wave, and the two groups run  L M | L M ...  vs  M L | M L ...  between consecutive barriers.
        auto sync_if = [&](bool on) {
            if (on) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::: "memory");
            }
        };
        // Default: strict alternation -- EVERY wave rendezvous after EVERY phase, group 1 one phase behind (4 barriers per tile).
        // OPT & 4: the asymmetric placement described above (2 barriers per tile).  Measured (profiles/r01_pp_timeline.txt):
        // asymmetric is slower -- with L << M the groups drift into bursting together and both idle through the trailing L.
        constexpr bool ASYM = (Cfg::OPT & 4) != 0;
        const bool g0 = ASYM ? (grp == 0) : true, g1 = ASYM ? (grp != 0) : true;
        if (!ASYM && grp == 1) phase_end();       // stagger: group 1 runs one phase behind

        constexpr int NMF = 2 * TM * TN;          // MFMAs per phase
        constexpr int XG = (Cfg::OPT & 64) ? 2 : ((Cfg::OPT & 512) ? 4 : 1);   // MFMAs per (MFMA..., expansion...) group
        constexpr int NDW = 8 * TN;               // sign dwords to expand per phase
        constexpr int NPIECE = A_PW + BW_PW;
        constexpr int EVERY = NMF / (NPIECE + 1) > 0 ? NMF / (NPIECE + 1) : 1;
        int slot_c = 0, slot_i = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + slot_c * STAGE_D;
            const int slot_n = (slot_c + 1 == NS) ? 0 : slot_c + 1;
            u32x4_t xa[TM], xb[TM];
            // ---------------- L0
            sync_if(g0);
            BD_STAMP(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const u32x4_t*)(st + a_rd[0] + i * 4096);
#pragma unroll
            for (int i = 0; i < TM; ++i) xb[i] = *(const u32x4_t*)(st + a_rd[1] + i * 4096);
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): fragments in registers before the (possible) rendezvous
            BD_STAMP(1);
            sync_if(g1);
            BD_STAMP(2);
            // ---------------- M0: steps 0,1; in the shadow: expand steps 2,3 (from rhi) + this wave's DMA pieces
            {
                const int kt_i = min(kt + NS - 1, nk - 1);
                const char* as = a_src + (long long)kt_i * 128;
                const char* ps = p_src + (long long)kt_i * 2 * p.N * 4;
                const uint32_t base = lds0 + slot_i * STAGE_D;
                if constexpr (!(Cfg::OPT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < NMF; ++t) {
                    const int hsel = t / (TM * TN), j = (t % (TM * TN)) / TM, i = t % TM;
                    acc[i][j] = hsel == 0 ? mfma32<DT>(s0a[j], xa[i], acc[i][j]) : mfma32<DT>(s0b[j], xb[i], acc[i][j]);
                    if constexpr (!(Cfg::OPT & 32)) {       // 32 = timing probe without the expansion (WRONG results)
                        if (t % XG == XG - 1) {
#pragma unroll
                            for (int e = (t + 1 - XG) * NDW / NMF; e < (t + 1) * NDW / NMF; ++e) expand_dw(s1a, s1b, rhi, e);
                        }
                    }
                    {
                        const int pc = t / EVERY;
                        if (t % EVERY == EVERY - 1 && pc < NPIECE) {
                            if (pc < A_PW) dma16(a_voff[pc < A_PW ? pc : 0], as, base + a_lds[pc < A_PW ? pc : 0]);
                            else dma4(bw_voff[pc >= A_PW && pc - A_PW < BW_PW ? pc - A_PW : 0], ps,
                                      base + bw_lds[pc >= A_PW && pc - A_PW < BW_PW ? pc - A_PW : 0]);
                        }
                    }
                    if constexpr (!(Cfg::OPT & 1)) { if (t % XG == XG - 1) __builtin_amdgcn_sched_barrier(0); }   // keep the interleave as written
                }
                if constexpr (!(Cfg::OPT & 2)) __builtin_amdgcn_s_setprio(0);
            }
            BD_STAMP(3);
            // ---------------- L1: steps 2,3 fragments + tile kt+1's sign words (-> replicated chunks for M1's expansion)
            sync_if(g0);
            BD_STAMP(4);
            {
                const uint32_t waddr = lds0 + slot_n * STAGE_D + bw_rd;     // tile kt+1's words (tail: a re-fetched copy)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(wraw[j]) : "v"(waddr), "n"(j * 128) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const u32x4_t*)(st + a_rd[2] + i * 4096);
#pragma unroll
            for (int i = 0; i < TM; ++i) xb[i] = *(const u32x4_t*)(st + a_rd[3] + i * 4096);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * TM) : "memory");      // the words are back
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) make_reps(~wraw[j], rlo[j], rhi[j]);      // rhi's old value died in M0
            wait_vmcnt<(NS - 3) * Cfg::DPW_D>();  // own pieces of tile kt+2 landed (published by the next rendezvous)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            BD_STAMP(5);
            sync_if(g1);
            BD_STAMP(6);
            // ---------------- M1: steps 2,3; in the shadow: expand tile kt+1's steps 0,1 (from rlo)
            {
                if constexpr (!(Cfg::OPT & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < NMF; ++t) {
                    const int hsel = t / (TM * TN), j = (t % (TM * TN)) / TM, i = t % TM;
                    acc[i][j] = hsel == 0 ? mfma32<DT>(s1a[j], xa[i], acc[i][j]) : mfma32<DT>(s1b[j], xb[i], acc[i][j]);
                    if constexpr (!(Cfg::OPT & 32)) {
                        if (t % XG == XG - 1) {
#pragma unroll
                            for (int e = (t + 1 - XG) * NDW / NMF; e < (t + 1) * NDW / NMF; ++e) expand_dw(s0a, s0b, rlo, e);
                        }
                    }
                    if constexpr (!(Cfg::OPT & 1)) { if (t % XG == XG - 1) __builtin_amdgcn_sched_barrier(0); }
                }
                if constexpr (!(Cfg::OPT & 2)) __builtin_amdgcn_s_setprio(0);
            }
            BD_STAMP(7);
            slot_c = slot_n;
            slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
        }
        if (!ASYM && grp == 0) phase_end();       // re-align the groups (equal barrier counts)
        wait_vmcnt<0>();
        phase_end();
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
        constexpr int NSB = Cfg::NSB;
        phase_end();                              // all waves left the delta ring
#pragma unroll
        for (int t = 0; t < NSB - 1; ++t) issue_b(min(t, nk - 1), t);
        wait_vmcnt<(NSB - 2) * Cfg::DPW_B>();
        phase_end();
        if (grp == 1) phase_end();
        int sb_c = 0, sb_i = NSB - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + sb_c * STAGE_B;
            u32x4_t xa[TM], xb[TM], wa[TN], wb[TN];
            const int kt_i = min(kt + NSB - 1, nk - 1);
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
                const uint32_t base = lds0 + sb_i * STAGE_B;
                constexpr int NPIECE = A_PW + W_PW, EVERY = (2 * TM * TN) / NPIECE > 0 ? (2 * TM * TN) / NPIECE : 1;
                mfma_phase(wa, wb, xa, xb, [&](int t) {
                    {
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
            wait_vmcnt<(NSB - 2) * Cfg::DPW_B>();   // own pieces of tile kt+1 landed
            phase_end();
            mfma_phase(wa, wb, xa, xb, nothing);
            phase_end();
            sb_c = (sb_c + 1 == NSB) ? 0 : sb_c + 1;
            sb_i = (sb_i + 1 == NSB) ? 0 : sb_i + 1;
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
