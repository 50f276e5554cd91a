// W1A16 binary-delta GEMM / fused binary-delta Linear, FOUR-WAVE PERSISTENT schedule (one wave per SIMD, whole 512-register file).
//
//   delta-only :  C[b] = X[b] . S[b]                               (binary_bmm, reference bitdelta/binary_gemm_kernel.py:186-335)
//   fused      :  C[b] = X[b] . W^T + alpha[b] * (X[b] . S[b])     (BinaryDiff.forward, bitdelta/diff.py:33-39)
//
// Why another schedule (bd_gemm_pf.h / bd_gemm_fx.h are 8-wave ping-pong): their counters say MFMA-busy 62..69 %, two barrier
// hand-offs of ~220 idle cycles per k-tile, 21 ds_read_b128 per 32 MFMAs per wave.  Here a workgroup is 4 waves, each alone on its
// SIMD with a 128 x 128 (delta) or 128 x 64 x {W, S} (fused) wave tile = 16 accumulators = 256 AGPRs:
//   * MFMAs are issued back to back by ONE in-order stream per SIMD; everything else (X-fragment ds_reads, sign expansion,
//     LDS-DMA issue) is placed in the <= 5 issue slots each v_mfma_f32_32x32x16 leaves free (MI355X_MICROARCH.md, "one wave per
//     SIMD" row).  A k-tile (64 k) is 16 REGIONS of 4 MFMAs (one B-operand fragment x 4 X fragments); region g
//         - multiplies B fragment f = g (k-step s = g/4, fragment b = g%4),
//         - PRODUCES B fragment f+3 (sign expansion by VALU or by LUT read; W fragment ds_read in fused mode),
//         - reads its share of the NEXT k-step's X fragments (regions b = 0, 1 read two each),
//         - issues its share of the LDS-DMA pieces of k-tile kt+2 (regions 0..11).
//   * half the LDS reads per MFMA of the 8-wave tile (16 X reads per 64 MFMAs), ONE s_barrier per k-tile (at region 12, where
//     the read stream crosses into the next ring slot), no partner group.
//   * PERSISTENT: grid = min(tiles, CUs); the (tile, k-tile) pairs of a workgroup form one flat stream, the DMA runs two k-tiles
//     ahead ACROSS tile boundaries, and the epilogue has its own LDS staging area, so the C store of tile t overlaps the first
//     k-tiles of tile t+1 (the 8-wave kernels pay ~11 us of un-overlapped C store + ~4 us of prologue per launch at 4096^3).
// Operand layout, k permutation, LDS image / swizzle: identical to bd_gemm_mfma.h (see its header); the MFMA operand order is the
// natural one here (X fragment first -- see w4_epilogue), so the accumulator holds D[m][n] with lane <-> n.
//
// Ring (NS = 3 slots): k-tile c lives in slot c % 3.  Reads of k-tile c happen between barrier B(c-1) and B(c) (B(c) sits at
// region 12 of k-tile c; regions 12..15 already read k-tile c+1's first fragments).  The refill of slot c % 3 with k-tile c+3 is
// issued in regions 0..11 of k-tile c+1, i.e. after B(c), which every wave passes only after lgkmcnt(0) on its reads of k-tile c.
// The only B fragment produced AFTER B(c) from k-tile c is fragment 15 (region 12): it must not read the slot, so it is always a sign
// fragment made by VALU (or read from the LUT area) -- hence the fused fragment order W0 S0 W1 S1.
// k-tile c+1's pieces were issued during k-tile c-1; each wave waits vmcnt(DPW) (only k-tile c+2's pieces may remain in flight)
// before B(c).
#pragma once
#include "bd_gemm_mfma.h"
#include "bd_serving.h"      // swiglu1 (the SwiGLU epilogue shares bd_srv_swiglu's arithmetic)

namespace bd {

template <int V> struct IC { static constexpr int value = V; };
// compile-time loop: the body sees its index as a constant (register arrays must never be indexed by a runtime value)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

// EPI_ = 1 (fused, 16-bit output only): SwiGLU epilogue over an 8-interleaved gate|up projection -- C has N/2 columns
// PAIR_ = 1 (fused, 128 x 128 tile): the tile's two 64-row wave rows are TWO batch entries of <= 64 rows each (the multi-tenant prefill of short
//   prompts, demo/demo_backend.py:297-299: 6 tenants x 64 rows) -- one W stream per tile serves both, each wave row reads its own entry's X rows
//   and sign words.  Wave tile 64 x 64 x {W, S} = 8 accumulators; a region is one B fragment x TWO X fragments.
template <int DT_, int BM_, int BN_, bool FUSED_, bool OUT_F32_, int OPT_ = 0, int EPI_ = 0, int PAIR_ = 0>
struct W4Cfg {
    static constexpr int DT = DT_, BM = BM_, BN = BN_, NS = 3, OPT = OPT_, EPI = EPI_, PAIR = PAIR_;
    static_assert(EPI_ == 0 || (FUSED_ && !OUT_F32_), "SwiGLU epilogue: fused kernel, 16-bit output");
    static constexpr bool FUSED = FUSED_, OUT_F32 = OUT_F32_;
    static constexpr int WAVES_M = 2, WAVES_N = 2, NW = 4, NT = 256;
    static constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    static constexpr int NB = 4;                                   // B-operand fragments per k-step (delta: 4 S; fused: W0 S0 W1 S1)
    static_assert((TM == 4 || (TM == 2 && FUSED && (EPI_ == 0 || PAIR_))) && (!PAIR || TM == 2) && (FUSED ? TN == 2 : TN == 4), "wave tile: 4 B fragments x 4 row blocks (128-row fused tiles: 2)");
    static constexpr int NENT = PAIR ? 2 : 1;                      // batch entries per tile (one per wave row)
    static constexpr int A_BYTES = BM * 128, W_BYTES = FUSED ? BN * 128 : 0, BW_BYTES = BN * 8 * NENT;
    static constexpr int BW_OFF = A_BYTES + W_BYTES;
    static constexpr int STAGE = A_BYTES + W_BYTES + BW_BYTES;
    static constexpr int A_PW = BM / 8 / NW, W_PW = FUSED ? BN / 8 / NW : 0, BW_PIECES = BN / 32 * NENT, BW_PW = BW_PIECES / NW;
    static_assert(BW_PIECES % NW == 0, "sign-word pieces vs waves");
    static constexpr int DPW = A_PW + W_PW + BW_PW;                // LDS-DMA pieces per wave per k-tile
    static constexpr bool USE_LUT = (OPT & 1) != 0;
    static constexpr int LUT_OFF = NS * STAGE, LUT_BYTES = USE_LUT ? 4096 : 0;
    static constexpr int STG_OFF = LUT_OFF + LUT_BYTES;
    static constexpr int ESZ = OUT_F32 ? 4 : 2;
    // epilogue staging, per wave, one 32-row block x JC 32-column blocks at a time:
    //   16-bit outputs: TRANSPOSED image [n][32 m] (64 B per column n, 8-byte quads XOR-swizzled), written with ds_write_b64 (4 consecutive
    //   m of one column = one accumulator register quad), read back with ds_read_b64_tr_b16 (4 consecutive n of one row per lane);
    //   fp32 outputs: [m][n] rows padded by 16 B, 4-byte writes.
    static constexpr int stg_pw(int jc) { return OUT_F32 ? 32 * (jc * 32 * 4 + 16) : jc * 32 * 64; }
    static constexpr bool fits(int jc) { return STG_OFF + NW * stg_pw(jc) <= 160 * 1024; }
    static constexpr int JC = fits(TN) ? TN : (TN >= 2 && fits(TN / 2)) ? TN / 2 : 1;
    static constexpr bool FAST_EPI = fits(JC);          // fused + fp32 output: only the general form's 2-KiB block fits behind the ring
    static constexpr int STG_ROWB = JC * 32 * ESZ + 16, STG_PW = FAST_EPI ? stg_pw(JC) : 2048;
    static constexpr int LDS_BYTES = STG_OFF + NW * STG_PW;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // OPT & 2048: TRICKLED epilogue (16-bit fast form, no residual).  The finished tile is converted to packed 16-bit values in spare VGPRs
    // (NCH chunks of JC * 4 register pairs: 128 registers delta-only, 64 fused) and its staging writes / transpose reads / global stores are
    // issued one small burst per k-tile INSIDE the next tile's k loop (4 k-tiles per chunk), instead of as one 10-k-cycle burst in which
    // every CU of the chip stores its 128 KB at the same moment (32 MB at HBM write speed, MFMA pipes idle).
    static constexpr bool TRICKLE = (OPT & 2048) != 0 && !OUT_F32 && EPI == 0 && FAST_EPI;
    // (A/B loser of round 5, instantiated by tests/native/w4_bench only: w4_convert / the deferred slots index ONE batch entry and ONE scale
    //  row -- a pair tile's second entry would be stored with the first entry's addressing.  ADVICE r05)
    static_assert(!(TRICKLE && PAIR), "the trickled epilogue does not know pair tiles");
    static constexpr int NCH = TM * (TN / JC);          // epilogue chunks per wave tile (32 rows x JC 32-column blocks each)
    // trickle schedule: a group = 4 k-tiles = 8 SLOTS (two per k-tile); a slot carries ONE 1-KiB global store per wave (the CU's store path
    // takes ~16 B/clk: stores issued back to back block the issuing wave ~260 cycles each), plus the staging traffic of the chunk(s) of
    // the current group.  A chunk has UPC = 2 * JC store units (16-row half x 32-column block).
    static constexpr int UPC = 2 * JC, CPG = 8 / UPC, NGRP = (NCH + CPG - 1) / CPG;
    static_assert(UPC <= 8 && 8 % UPC == 0, "store units per chunk vs slots per group");
    static constexpr int TRICKLE_KT = 4 * (NGRP + 1);   // k-tiles the trickle of one tile takes (stores lag the staging by one group)
    static_assert(DPW <= 24, "piece placement");
};

#ifdef BD_TRACE
// s_memtime stamps of workgroup 0, wave 0 (harness builds only): [tile][0] loop entry, [1] after k-tile 7, [2] after k-tile 39,
// [3] loop exit, [4] epilogue done; bd_trace_w4[15][0] = kernel entry, [15][1] = prologue done
__device__ unsigned long long bd_trace_w4[16][8];
#define BD_W4_STAMP(t, id)                                                                     \
    do {                                                                                       \
        if (blockIdx.x == 0 && wave == 0 && (t) < 15) {                                        \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                        \
            if (lane == 0) bd_trace_w4[(t)][id] = t_;                                          \
        }                                                                                      \
    } while (0)
#else
#define BD_W4_STAMP(t, id) do { } while (0)
#endif

// LDS-DMA without the M0 save / restore of bd_common.h's dma16 (this kernel holds nothing in M0: gfx9+ ds_* and MFMA code
// never needs it): 3 issue slots per piece instead of 5 -- in a one-wave-per-SIMD stream every slot beside an MFMA counts.
template <int OFF> __device__ __forceinline__ void w4_dma16(uint32_t voff, const void* sbase, uint32_t lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base), "n"(OFF) : "memory", "scc");
}
template <int OFF> __device__ __forceinline__ void w4_dma4(uint32_t voff, const void* sbase, uint32_t lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base), "n"(OFF) : "memory", "scc");
}

// Split form: the M0 write and the load go into two different MFMA gaps (no s_nop: at least one MFMA separates them).  Valid only
// because nothing else in this kernel touches M0 between the two statements (audited in the ISA: the only M0 writers are these).
template <int OFF> __device__ __forceinline__ void w4_set_m0(uint32_t lds_base) {
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds_base), "n"(OFF) : "memory", "scc");
}
__device__ __forceinline__ void w4_dma16_m0(uint32_t voff, const void* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void w4_dma4_m0(uint32_t voff, const void* sbase) {
    asm volatile("global_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase) : "memory");
}

// Epilogue of one output tile.  NATURAL MFMA operand order (first = X fragment, second = B fragment):
//     acc[b][i][r] = D[m = m0 + wm*WM + 32i + (r&3) + 8(r>>2) + 4h][n = n0 + wn*WN + 32j + l31]
// (delta: j = b; fused: accW = acc[2j], accS = acc[2j+1], out = accW + alpha[n] * accS in fp32 -- one rounding, as bd_gemm_fx.h).
// The 8-wave kernels put the sign fragment FIRST (accumulator = D[n][m], 8-byte staging writes); measured here on sustained launches
// (profiles/r03_w4_energy.txt) the same kernel with the +-1 fragment in the SECOND slot draws less power per MFMA and -- the chip
// being power-limited at ~1.25 kW under either -- runs 8 % faster at a higher clock.  The price is 2-byte staging writes.
// Each wave transposes through ITS OWN staging area (outside the DMA ring, so the next tile's k-tiles keep landing meanwhile).
//   fast form (aligned C; plain, reference fp16 rounding, or fused + residual): one 32-row block x JC 32-column blocks at a time, values
//   rounded to the output type before staging, whole row segments stored with 16-byte stores;
//   general form (C += alpha * acc, unaligned / odd-sized C): 16 x 32 fp32 blocks staged, then a rolled loop of guarded scalar stores.
// Value transforms are gemm_epilogue's (bd_gemm_mfma.h).
template <class Cfg>
__device__ __forceinline__ void w4_epilogue(const GemmParams& p, f32x16_t (&acc)[4][Cfg::TM], char* stg, int m0, int n0, int wm, int wn,
                                            int b, int lane, int wave, int b_alpha = -1) {
    // OPT & 16384 (round 6 A/B, harness only): per-wave STAGGER of the store burst -- wave w starts its epilogue w x 64 cycles late, so that the four
    // waves' 1-KiB stores reach the CU's store path one every ~75 cycles instead of four at once (profiles/r06_w4_cycles.txt)
    if constexpr (Cfg::OPT & 16384) {
        if (wave == 1) __builtin_amdgcn_s_sleep(1);
        else if (wave == 2) __builtin_amdgcn_s_sleep(2);
        else if (wave == 3) __builtin_amdgcn_s_sleep(3);
    }
    constexpr int DT = Cfg::DT, WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN, JC = Cfg::JC, ESZ = Cfg::ESZ;
    constexpr int ROWB = Cfg::STG_ROWB;
    constexpr bool FUSED = Cfg::FUSED;
    // Per-tile opaque copy of the lane id: everything lane-derived below is recomputed per tile.  Without it the compiler hoists ~40
    // lane-constant row / column / address values out of the persistent tile loop, spills them around the k loop, and every reload's
    // `s_waitcnt vmcnt(0)` serialises the output stores AND drains the LDS-DMA ring (cdna_hip_programming.md, 4-wave pitfalls).
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5, l31 = lane & 31;
    const long long c_b = (long long)b * p.sCb;
    const bool acc_mode = !FUSED && p.accumulate;
    const bool res_mode = FUSED && p.accumulate;
    const bool rm16 = !FUSED && p.round_mode == 1;
    // (b_alpha: split-k pair tiles store to slab b = entry * ksplit + slice but scale with the ENTRY's alpha)
    const float* al = (FUSED || acc_mode) ? p.alpha + (long long)(b_alpha >= 0 ? b_alpha : b) * p.sAlb : nullptr;
    // this lane's column scale per column block.  One scale group (the reference's scalar coeff): a scalar load.  Grouped scales: per-lane
    // loads, retired at once with a wait the compiler's waitcnt pass can see -- a VMEM load it believes pending at the k-loop entry makes
    // it put `s_waitcnt vmcnt(0)` in front of the first reuse of that register INSIDE the k loop, which drains the LDS-DMA ring every
    // k-tile (measured: 2970 instead of 2355 cycles per k-tile on the fused kernel).
    float a_col[TN];
    if constexpr (Cfg::EPI == 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) a_col[j] = 0.f;
    } else if (p.gsz >= p.N) {
        const float a0 = al ? al[0] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) a_col[j] = a0;
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) a_col[j] = al ? al[min(n0 + wn * WN + j * 32 + l31, p.N - 1) / p.gsz] : 0.f;
        __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0)
    }
    const bool fast = Cfg::FAST_EPI && !acc_mode && (p.N % 8 == 0) && (p.sCm % 8 == 0) && (p.sCb % 8 == 0) && (((uintptr_t)p.C & 15) == 0);
    char* buf = stg + wave * Cfg::STG_PW;
    // pair tiles: `b` is this wave row's OWN batch entry and its rows start at 0 (the caller passes m0 = 0)
    if constexpr (Cfg::PAIR) wm = 0;
    const bool inside = (m0 + (Cfg::PAIR ? Cfg::WM : Cfg::BM) <= p.M) && (n0 + Cfg::BN <= p.N);      // wave-uniform: interior tiles store without guards

    auto pack2 = [&](float lo, float hi) -> uint32_t {
        if constexpr (DT == DT_BF16) {
            uint32_t r;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
            return r;
        } else {
            return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16);
        }
    };
    // Accumulator elements are fetched from the AGPR file one at a time, where they are used: left to itself the register allocator copies
    // all 256 to VGPRs at the loop exit and spills ~50 of them (scratch traffic + vmcnt(0) stalls in front of the stores).
    auto rd = [](float x) -> float {
        float r;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(x));
        return r;
    };
    auto value = [&](auto ic, auto jc, auto rc) -> float {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value, r = decltype(rc)::value;
        if constexpr (FUSED) return __builtin_fmaf(a_col[j], rd(acc[2 * j + 1][i][r]), rd(acc[2 * j][i][r]));
        else return rd(acc[j][i][r]);
    };
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA result -> accvgpr_read distance for the last MFMAs of the k loop (asm is not padded)

    if constexpr (Cfg::EPI == 1) {
        // ---- SwiGLU epilogue (fast form only; the host checks alignment).  Output rows of W are interleaved in blocks of 8
        //      ([g0..7 | u0..7 | g8..15 | ...], G = 2 scales: alpha[0] gate, alpha[1] up), so in this accumulator layout (lane <-> column)
        //      lanes 0-7 / 16-23 of a 32-column block hold gate columns and lanes 8-15 / 24-31 the matching up columns: both are rounded
        //      to the output type as the separate Linear would have stored them, the up value crosses 8 lanes by DPP (row_ror:8), and the
        //      gate lanes stage round(silu(g)) * u (swiglu1: bd_srv_swiglu's arithmetic) -- 16 output columns per 32-column block.
        typedef short v4s_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;
        static_assert(TN == 2 && Cfg::STG_PW >= 2048, "one chunk = both column blocks = 32 output columns");
        const float a_g = al[0], a_u = al[1];                        // (two scalar loads + a select: no VMEM load, see a_col above)
        const float a_gu = (l31 & 8) ? a_u : a_g;
        const bool is_gate = !(l31 & 8);
        const int no = (l31 & 7) + ((l31 >> 4) << 3);                   // output column inside the block's 16
        uint32_t wr_off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wr_off[q] = no * 64 + (((2 * q + h) ^ ((no >> 1) & 7)) * 8);
        const int G = lane >> 4, t = lane & 15, js = t >> 2, cs = t & 3;
        uint32_t rd_off[2][2];
#pragma unroll
        for (int B = 0; B < 2; ++B)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
                rd_off[B][mh] = (8 * G + 4 * B + js) * 64 + (((4 * mh + cs) ^ ((4 * G + 2 * B + (js >> 1)) & 7)) * 8);
        const int NO = p.N >> 1;
        static_for<0, TM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<0, 8>([&](auto tc) {
                constexpr int jj = decltype(tc)::value / 4, q = decltype(tc)::value % 4;
                __builtin_amdgcn_sched_barrier(0);
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.f;
                    if (e == 0) v = __builtin_fmaf(a_gu, rd(acc[2 * jj + 1][i][4 * q + 0]), rd(acc[2 * jj][i][4 * q + 0]));
                    if (e == 1) v = __builtin_fmaf(a_gu, rd(acc[2 * jj + 1][i][4 * q + 1]), rd(acc[2 * jj][i][4 * q + 1]));
                    if (e == 2) v = __builtin_fmaf(a_gu, rd(acc[2 * jj + 1][i][4 * q + 2]), rd(acc[2 * jj][i][4 * q + 2]));
                    if (e == 3) v = __builtin_fmaf(a_gu, rd(acc[2 * jj + 1][i][4 * q + 3]), rd(acc[2 * jj][i][4 * q + 3]));
                    const float r16 = round16<DT>(v);
                    const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r16), 0x128, 0xf, 0xf, true));
                    o[e] = swiglu1<DT>(r16, other);
                }
                if (is_gate) *(u32x2_t*)(buf + wr_off[q] + jj * 1024) = u32x2_t{o[0] | (o[1] << 16), o[2] | (o[3] << 16)};
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<0, 2>([&](auto mc) {
                constexpr int mh = decltype(mc)::value;
                const v4s_t ra = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(buf + rd_off[0][mh]));
                const v4s_t rb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(buf + rd_off[1][mh]));
                const u32x2_t a2 = __builtin_bit_cast(u32x2_t, ra), b2 = __builtin_bit_cast(u32x2_t, rb);
                const u32x4_t val = u32x4_t{a2.x, a2.y, b2.x, b2.y};
                const int mm = m0 + wm * WM + i * 32 + 16 * mh + t;
                const int n = ((n0 + wn * WN) >> 1) + 8 * G;
                if (inside || (mm < p.M && n < NO)) *(u32x4_t*)(p.C + (c_b + (long long)mm * p.sCm + n) * 2) = val;
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
        return;
    }
    if constexpr (!Cfg::OUT_F32) if (fast) {
        // ---- 16-bit fast form: transposed staging + hardware transpose read (mapping of ds_read_b64_tr_b16 probed on the device,
        //      tests/native/probes/tr_probe.hip: inside a 16-lane group, output lane t gets element (t & 3) of the quads addressed by source
        //      lanes (t >> 2), 4 + (t >> 2), 8 + (t >> 2), 12 + (t >> 2)).
        //      image: byte(n, mq) = n * 64 + ((mq ^ ((n >> 1) & 7)) * 8), n = column inside the chunk, mq = 4-row quad 0..7
        //      -> the ds_write_b64 groups (16 consecutive n, one mq) and the tr-read groups (4 n x 4 mq) both touch 16 distinct 8-byte slots.
        typedef short v4s_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;
        const int swz = (l31 >> 1) & 7;
        uint32_t wr_off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wr_off[q] = l31 * 64 + (((2 * q + h) ^ swz) * 8);
        const int G = lane >> 4, t = lane & 15, js = t >> 2, cs = t & 3;
        uint32_t rd_off[2][2];
#pragma unroll
        for (int B = 0; B < 2; ++B)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
                rd_off[B][mh] = (8 * G + 4 * B + js) * 64 + (((4 * mh + cs) ^ ((4 * G + 2 * B + (js >> 1)) & 7)) * 8);
        static_for<0, TM*(TN / JC)>([&](auto cc) {
            constexpr int i = decltype(cc)::value / (TN / JC), j0 = (decltype(cc)::value % (TN / JC)) * JC;
            static_for<0, JC * 4>([&](auto tc) {
                constexpr int jj = decltype(tc)::value / 4, q = decltype(tc)::value % 4, j = j0 + jj;
                __builtin_amdgcn_sched_barrier(0);           // keeps the accumulator reads of later quads from being hoisted (VGPR pressure)
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.f;
                v[0] = value(IC<i>{}, IC<j>{}, IC<4 * q + 0>{}); v[1] = value(IC<i>{}, IC<j>{}, IC<4 * q + 1>{});
                v[2] = value(IC<i>{}, IC<j>{}, IC<4 * q + 2>{}); v[3] = value(IC<i>{}, IC<j>{}, IC<4 * q + 3>{});
                if constexpr (!FUSED) { if (rm16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_through_f16(v[e]);
                } }
                *(u32x2_t*)(buf + wr_off[q] + jj * 2048) = u32x2_t{pack2(v[0], v[1]), pack2(v[2], v[3])};
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's image only: no block barrier needed
            auto store_rows = [&](auto guardc) {
                constexpr bool GUARD = decltype(guardc)::value != 0;
                static_for<0, JC * 2>([&](auto uc) {
                    // u fastest: the 64-byte halves of a 128-byte line leave back to back
                    constexpr int u = decltype(uc)::value % JC, mh = decltype(uc)::value / JC;
                    const v4s_t ra = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(buf + rd_off[0][mh] + u * 2048));
                    const v4s_t rb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(buf + rd_off[1][mh] + u * 2048));
                    const u32x2_t a2 = __builtin_bit_cast(u32x2_t, ra), b2 = __builtin_bit_cast(u32x2_t, rb);
                    u32x4_t val = u32x4_t{a2.x, a2.y, b2.x, b2.y};
                    const int mm = m0 + wm * WM + i * 32 + 16 * mh + t;
                    const int n = n0 + wn * WN + (j0 + u) * 32 + 8 * G;
                    if (!GUARD || (mm < p.M && n < p.N)) {
                        u32x4_t* dst = (u32x4_t*)(p.C + (c_b + (long long)mm * p.sCm + n) * ESZ);
                        if (res_mode) {
                            const u32x4_t rsd = *(const u32x4_t*)dst;
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const float lo = half_bits_to_f32<DT>(val[d] & 0xffffu) + half_bits_to_f32<DT>(rsd[d] & 0xffffu);
                                const float hi = half_bits_to_f32<DT>(val[d] >> 16) + half_bits_to_f32<DT>(rsd[d] >> 16);
                                val[d] = pack2(lo, hi);
                            }
                        }
                        // plain stores: a wave-instruction covers 16 rows x 64 bytes, and the nt policy made the L2 forward those half lines
                        // un-merged (WRITE_SIZE 45.8 MB instead of 32.0 MB per 4096^2 output, +1 % time; profiles/r03_w4_store_policy.txt)
                        *dst = val;
                    }
                });
            };
            if (inside) store_rows(IC<0>{}); else store_rows(IC<1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image is rewritten by the next chunk
        });
        return;
    }
    if constexpr (Cfg::OUT_F32 && Cfg::FAST_EPI) if (fast) {
        constexpr int SEG = JC * 32 * ESZ / 16;          // 16-byte pieces per staged row
        constexpr int RPI = 64 / SEG;                    // rows per store instruction
        static_for<0, TM*(TN / JC)>([&](auto cc) {
            constexpr int i = decltype(cc)::value / (TN / JC), j0 = (decltype(cc)::value % (TN / JC)) * JC;
            __builtin_amdgcn_sched_barrier(0);           // keeps the accumulator reads of later blocks from being hoisted (VGPR pressure)
            char* wr = buf + (4 * h) * ROWB + l31 * ESZ;
            static_for<0, JC * 8>([&](auto tc) {
                constexpr int jj = decltype(tc)::value / 8, rp = (decltype(tc)::value % 8) * 2, j = j0 + jj;
                if constexpr (rp == 0 || rp == 8) __builtin_amdgcn_sched_barrier(0);
                float v0 = value(IC<i>{}, IC<j>{}, IC<rp>{}), v1 = value(IC<i>{}, IC<j>{}, IC<rp + 1>{});
                if constexpr (!FUSED) { if (rm16) { v0 = round_through_f16(v0); v1 = round_through_f16(v1); } }
                constexpr int row0 = (rp & 3) + 8 * (rp >> 2);                  // row of register rp within the 32-row block (h = 0); rp + 1 is the next row
                char* d0 = wr + row0 * ROWB + jj * 32 * ESZ;
                if constexpr (Cfg::OUT_F32) {
                    *(float*)d0 = v0; *(float*)(d0 + ROWB) = v1;
                } else {
                    const uint32_t pk = pack2(v0, v1);
                    *(unsigned short*)d0 = (unsigned short)pk;
                    *(unsigned short*)(d0 + ROWB) = (unsigned short)(pk >> 16);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's rows only: no block barrier needed
            auto store_rows = [&](auto guardc) {
                constexpr bool GUARD = decltype(guardc)::value != 0;
#pragma unroll
                for (int r0 = 0; r0 < 32; r0 += RPI) {
                    const int rr = r0 + lane / SEG, sg = lane % SEG;
                    const int mm = m0 + wm * WM + i * 32 + rr;
                    const int n = n0 + wn * WN + j0 * 32 + sg * (16 / ESZ);
                    u32x4_t val = *(const u32x4_t*)(buf + rr * ROWB + sg * 16);
                    if (!GUARD || (mm < p.M && n < p.N)) {
                        u32x4_t* dst = (u32x4_t*)(p.C + (c_b + (long long)mm * p.sCm + n) * ESZ);
                        if (res_mode) {
                            const u32x4_t rsd = *(const u32x4_t*)dst;
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                if constexpr (Cfg::OUT_F32) {
                                    val[d] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, val[d]) + __builtin_bit_cast(float, rsd[d]));
                                } else {
                                    const float lo = half_bits_to_f32<DT>(val[d] & 0xffffu) + half_bits_to_f32<DT>(rsd[d] & 0xffffu);
                                    const float hi = half_bits_to_f32<DT>(val[d] >> 16) + half_bits_to_f32<DT>(rsd[d] >> 16);
                                    val[d] = pack2(lo, hi);
                                }
                            }
                        }
                        __builtin_nontemporal_store(val, dst);
                    }
                }
            };
            if (inside) store_rows(IC<0>{}); else store_rows(IC<1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging rows are rewritten by the next block
        });
        return;
    }
    // ---- general form: 16 rows (register half qh = r >> 3) x 32 columns of fp32 at a time
    constexpr int GROW = 32 * 4;                        // (4-byte accesses along n: conflict-free without padding)
    static_assert(16 * GROW <= Cfg::STG_PW, "general-form staging block");
    static_for<0, TM * TN * 2>([&](auto ixc) {
        constexpr int ix = decltype(ixc)::value, i = ix / (TN * 2), j = (ix / 2) % TN, qh = ix & 1;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 8>([&](auto rc) {
            constexpr int r = 8 * qh + decltype(rc)::value;
            constexpr int row = (r & 3) + 8 * ((r >> 2) & 1);                    // within the 16-row half (h = 0)
            *(float*)(buf + (row + 4 * h) * GROW + l31 * 4) = value(IC<i>{}, IC<j>{}, IC<r>{});
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma nounroll
        for (int t = 0; t < 8; ++t) {
            const int idx = t * 64 + lane, rr = idx >> 5, cc = idx & 31;
            const int mm = m0 + wm * WM + i * 32 + 16 * qh + rr;
            const int n = n0 + wn * WN + j * 32 + cc;
            if (mm < p.M && n < p.N) {
                float v = *(const float*)(buf + rr * GROW + cc * 4);
                const long long off = c_b + (long long)mm * p.sCm + n;
                if (acc_mode) {
                    const float cin = Cfg::OUT_F32 ? ((const float*)p.C)[off] : half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]);
                    v = cin + al[n / p.gsz] * v;
                } else if (rm16) {
                    v = round_through_f16(v);
                } else if (res_mode) {
                    if constexpr (Cfg::OUT_F32) v = ((const float*)p.C)[off] + v;
                    else v = half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]) + half_bits_to_f32<DT>(f32_to_half_bits<DT>(v));
                }
                if constexpr (Cfg::OUT_F32) ((float*)p.C)[off] = v;
                else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
}

// ---- trickled epilogue, part 1 (at the end of a tile's k loop): accumulators -> output type, packed in VGPRs.  Same value transforms, same
//      rounding and the same (chunk, column block, quad) order as w4_epilogue's 16-bit fast form: pk[cc][jj * 4 + q] is exactly the register
//      pair that form stages with one ds_write_b64.
template <class Cfg>
__device__ __forceinline__ void w4_convert(const GemmParams& p, f32x16_t (&acc)[4][Cfg::TM], u32x2_t (&pk)[Cfg::NCH][Cfg::JC * 4], char* stg,
                                           int n0, int wn, int b, int lane, int wave) {
    constexpr int DT = Cfg::DT, WN = Cfg::WN, TN = Cfg::TN, JC = Cfg::JC;
    constexpr bool FUSED = Cfg::FUSED;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, h = lane >> 5;
    char* const wr = stg + wave * Cfg::STG_PW + l31 * 64;      // chunk 0 goes straight into the (idle) staging image: 2 * JC * 4 fewer live registers
    const int swz = (l31 >> 1) & 7;
    const bool rm16 = !FUSED && p.round_mode == 1;
    const float* al = FUSED ? p.alpha + (long long)b * p.sAlb : nullptr;
    float a_col[TN];
    if (p.gsz >= p.N) {
        const float a0 = al ? al[0] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) a_col[j] = a0;
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) a_col[j] = al ? al[min(n0 + wn * WN + j * 32 + l31, p.N - 1) / p.gsz] : 0.f;
        __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): see w4_epilogue
    }
    auto pack2 = [&](float lo, float hi) -> uint32_t {
        if constexpr (DT == DT_BF16) {
            uint32_t r;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
            return r;
        } else {
            return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16);
        }
    };
    auto rd = [](float x) -> float {
        float r;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(x));
        return r;
    };
    auto value = [&](auto ic, auto jc, auto rc) -> float {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value, r = decltype(rc)::value;
        if constexpr (FUSED) return __builtin_fmaf(a_col[j], rd(acc[2 * j + 1][i][r]), rd(acc[2 * j][i][r]));
        else return rd(acc[j][i][r]);
    };
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA result -> accvgpr_read distance for the last MFMAs of the k loop
    static_for<0, Cfg::NCH>([&](auto cc) {
        constexpr int i = decltype(cc)::value / (TN / JC), j0 = (decltype(cc)::value % (TN / JC)) * JC;
        static_for<0, JC * 4>([&](auto tc) {
            constexpr int jj = decltype(tc)::value / 4, q = decltype(tc)::value % 4, j = j0 + jj;
            __builtin_amdgcn_sched_barrier(0);
            float v[4];
            v[0] = value(IC<i>{}, IC<j>{}, IC<4 * q + 0>{}); v[1] = value(IC<i>{}, IC<j>{}, IC<4 * q + 1>{});
            v[2] = value(IC<i>{}, IC<j>{}, IC<4 * q + 2>{}); v[3] = value(IC<i>{}, IC<j>{}, IC<4 * q + 3>{});
            if constexpr (!FUSED) { if (rm16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round_through_f16(v[e]);
            } }
            const u32x2_t pr = u32x2_t{pack2(v[0], v[1]), pack2(v[2], v[3])};
            if constexpr (decltype(cc)::value == 0) *(u32x2_t*)(wr + (((2 * q + h) ^ swz) * 8) + jj * 2048) = pr;
            else pk[decltype(cc)::value][decltype(tc)::value] = pr;
        });
    });
}

// ---- trickled epilogue, part 2: slot S (0..7) of group G of the NEXT tile's k loop (two slots per k-tile, inside the k-tile: see ktile).
//      In group G the chunks [G * CPG, (G + 1) * CPG) are staged (transposed image, as w4_epilogue) and read back with the hardware
//      transpose read, one 16-row x 32-column store unit per slot into rbuf[S]; the unit read in slot S of group G - 1 is stored in slot S
//      of group G, just before rbuf[S] is re-read.  So every slot issues at most ONE global store per wave and a store unit waits a whole
//      group (4 k-tiles) for its LDS reads.  LDS operations of one wave execute in issue order: a chunk's writes, its reads and the next
//      chunk's writes need no waits between them.  The stores are ordinary VMEM operations retired IN ORDER with the LDS-DMA pieces: issued
//      in the middle of a k-tile's pieces they are among the last DPW operations at that k-tile's vmcnt(DPW) (which then covers the first
//      pieces of the look-ahead instead) and must have completed only by the NEXT k-tile's wait, ~1.4 k-tiles later.
template <class Cfg, int S, int G>
__device__ __forceinline__ void w4_slot(const GemmParams& p, const u32x2_t (&pk)[Cfg::NCH][Cfg::JC * 4], u32x4_t (&rbuf)[8], char* stg, int m0,
                                        int n0, int wm, int wn, int b, bool inside, int lane, int wave) {
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TN = Cfg::TN, JC = Cfg::JC, UPC = Cfg::UPC, CPG = Cfg::CPG, NGRP = Cfg::NGRP, NCH = Cfg::NCH;
    constexpr int ci = S / UPC, ui = S % UPC, mh = ui / JC, u = ui % JC;
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;
    // The (slot, group) blocks differ only in WHICH registers they touch; SimplifyCFG would sink their common tails into one block fed by
    // phis of the pk element addresses, which turns pk into an indexed array in scratch.  An asm statement with a distinct immediate at
    // both ends of every block cannot be merged, and nothing is sunk or hoisted across it.
    asm volatile("; trickle slot %0 begin" ::"n"(G * 8 + S) : "memory");
    asm volatile("" : "+v"(lane));                         // lane-derived addresses are recomputed here, never hoisted + spilled
    char* buf = stg + wave * Cfg::STG_PW;
    const int h = lane >> 5, l31 = lane & 31;
    const int Gq = lane >> 4, t = lane & 15, js = t >> 2, cs = t & 3;
    // ---- store of the unit read one group ago
    constexpr int cst = (G - 1) * CPG + ci;                // its chunk
    if constexpr (G >= 1 && cst < NCH && !(Cfg::OPT & 4096)) {
        constexpr int i = cst / (TN / JC), j0 = (cst % (TN / JC)) * JC;
        const int mm = m0 + wm * WM + i * 32 + 16 * mh + t;
        const int n = n0 + wn * WN + (j0 + u) * 32 + 8 * Gq;
        if (inside || (mm < p.M && n < p.N)) *(u32x4_t*)(p.C + ((long long)b * p.sCb + (long long)mm * p.sCm + n) * 2) = rbuf[S];
    }
    // ---- staging of this group's chunk: writes when its first unit comes up, then this slot's transpose reads
    constexpr int cw = G * CPG + ci;
    if constexpr (G < NGRP && cw < NCH) {
        if constexpr (ui == 0 && cw > 0) {                 // (chunk 0 was staged by w4_convert)
            const int swz = (l31 >> 1) & 7;
            static_for<0, JC * 4>([&](auto tcc) {          // (compile-time indices: pk must stay a set of registers, never an array in scratch)
                constexpr int tc = decltype(tcc)::value, jj = tc / 4, q = tc % 4;
                *(u32x2_t*)(buf + l31 * 64 + (((2 * q + h) ^ swz) * 8) + jj * 2048) = pk[cw][tc];
            });
        }
        const uint32_t o0 = (8 * Gq + js) * 64 + (((4 * mh + cs) ^ ((4 * Gq + (js >> 1)) & 7)) * 8);
        const uint32_t o1 = (8 * Gq + 4 + js) * 64 + (((4 * mh + cs) ^ ((4 * Gq + 2 + (js >> 1)) & 7)) * 8);
        const v4s_t ra = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(buf + o0 + u * 2048));
        const v4s_t rb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(buf + o1 + u * 2048));
        const u32x2_t a2 = __builtin_bit_cast(u32x2_t, ra), b2 = __builtin_bit_cast(u32x2_t, rb);
        rbuf[S] = u32x4_t{a2.x, a2.y, b2.x, b2.y};
    }
    asm volatile("; trickle slot %0 end" ::"n"(G * 8 + S) : "memory");
}

template <class Cfg>
__global__ void __launch_bounds__(256) delta_gemm_w4_kernel(const GemmParams p) {
    constexpr int DT = Cfg::DT, BM = Cfg::BM, BN = Cfg::BN, NS = Cfg::NS;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE, BW_OFF = Cfg::BW_OFF;
    constexpr int A_PW = Cfg::A_PW, W_PW = Cfg::W_PW, BW_PW = Cfg::BW_PW, DPW = Cfg::DPW;
    constexpr bool FUSED = Cfg::FUSED, USE_LUT = Cfg::USE_LUT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int ntotal = ntiles * max(p.nbatch, 1);          // the persistent stream runs over (batch entry, tile) pairs
    const int G = (int)gridDim.x;
    // k-tiles of ONE stream element.  Pair tiles with p.ksplit > 1: the batch index of the persistent stream is pair * ksplit + slice, a slice
    // contracts k-tiles [slice * nk, (slice + 1) * nk) and stores its fp32 partial to slab (entry * ksplit + slice) of C (host: nk divides)
    const int nk = Cfg::PAIR ? (p.K >> 6) / p.ksplit : (p.K >> 6);

    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));

    // ---- tile walk of this persistent workgroup: round r covers tiles [r*G, r*G + Gr), XCD-remapped inside the round
    auto tile_of = [&](int r, int& tm, int& tn, int& tb) -> bool {
        const int base = r * G;
        const int Gr = min(G, ntotal - base);
        if ((int)blockIdx.x >= Gr) return false;
        const int lin = base + xcd_remap(blockIdx.x, Gr);
        tb = __builtin_amdgcn_readfirstlane(lin / ntiles);
        tile_coords(p, lin - tb * ntiles, tm, tn);
        return true;
    };

    // ---- fragment read offsets (tile independent)
    const int swz = (l31 >> 1) & 7;
    uint32_t a_rd[4], w_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a_rd[s] = (uint32_t)(wm * WM + l31) * 128u + (uint32_t)(((4 * h + s) ^ swz) * 16);
        w_rd[s] = A_BYTES + (uint32_t)(wn * WN + l31) * 128u + (uint32_t)(((4 * h + s) ^ swz) * 16);
    }
    const uint32_t bw_rd = BW_OFF + (Cfg::PAIR ? wm * BN * 8 : 0) + h * BN * 4 + (wn * WN + l31) * 4;    // (pair: this wave row's entry's words)

    // ---- loader state: the k-tile being fetched (two ahead of the one being multiplied), across tile boundaries.
    //      Running wave-uniform source pointers (advanced by one k-tile per iteration), per-lane offsets fixed per tile.
    int ld_r = 0, ld_kt = 0;
    const char *ld_a = nullptr, *ld_w = nullptr, *ld_p = nullptr;
    const long long p_step = (long long)p.N * 8;                          // 2 word rows per k-tile
    uint32_t a_voff[A_PW], w_voff[W_PW > 0 ? W_PW : 1], bw_voff[BW_PW];
    const uint32_t a_ldsw = lds0 + wave * A_PW * 1024;                    // this wave's pieces inside a ring slot
    const uint32_t w_ldsw = lds0 + A_BYTES + wave * (W_PW > 0 ? W_PW : 1) * 1024;
    uint32_t bw_ldsw[BW_PW];
#pragma unroll
    for (int i = 0; i < BW_PW; ++i) {
        const int idx = wave * BW_PW + i;          // LDS image [entry][word row 0..1][BN] dwords; a piece = 64 columns of one word row
        bw_ldsw[i] = lds0 + BW_OFF + (idx / (BN / 64)) * BN * 4 + (idx % (BN / 64)) * 256;
    }
    auto loader_tile = [&](int r) {
        int tm, tn, b;
        if (!tile_of(r, tm, tn, b)) { if (!tile_of(r - 1 >= 0 ? r - 1 : 0, tm, tn, b)) { tm = 0; tn = 0; b = 0; } }   // past the end: re-fetch (never read)
        const int m0 = tm * BM, n0 = tn * BN;
        // pair tiles: b = pair index; entries 2b and 2b + 1 (the second clamped to the last entry of an odd batch: its rows are computed and
        // dropped); entry offsets ride in the 32-bit per-lane offsets (host: sAb < 2^30, sPb < 2^29)
        const int ksl = Cfg::PAIR ? b % p.ksplit : 0;                       // k slice of this stream element
        const int e0 = Cfg::PAIR ? 2 * (b / p.ksplit) : b;
        const int e1off = Cfg::PAIR ? (min(e0 + 1, p.nent - 1) - e0) : 0;
        ld_a = p.A + ((long long)e0 * p.sAb + (long long)m0 * p.sAm) * 2 + (long long)ksl * nk * 128;
        ld_p = (const char*)p.P + ((long long)e0 * p.sPb + n0) * 4 + (long long)ksl * nk * p_step;
        if constexpr (FUSED) ld_w = p.W + (long long)n0 * p.ldw * 2 + (long long)ksl * nk * 128;
        int ln = lane;
        asm volatile("" : "+v"(ln));                                        // recomputed per tile, not hoisted + spilled
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            const int rg = wave * A_PW + i;
            const int r8 = rg * 8 + (ln >> 3);
            const int c = (ln & 7) ^ ((r8 >> 1) & 7);
            if constexpr (Cfg::PAIR) {
                const int ent = r8 >> 6, rr = min(r8 & 63, p.M - 1);       // LDS rows 0..63 = entry 2b, 64..127 = entry 2b + 1
                a_voff[i] = (uint32_t)(ent * e1off) * (uint32_t)p.sAb * 2u + (uint32_t)rr * (uint32_t)p.sAm * 2u + (uint32_t)c * 16u;
            } else {
                const int rr = min(m0 + r8, p.M - 1) - m0;
                a_voff[i] = (uint32_t)rr * (uint32_t)p.sAm * 2u + (uint32_t)c * 16u;
            }
        }
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < W_PW; ++i) {
                const int rg = wave * W_PW + i;
                const int r8 = rg * 8 + (ln >> 3);
                const int c = (ln & 7) ^ ((r8 >> 1) & 7);
                const int rr = min(n0 + r8, p.N - 1) - n0;
                w_voff[i] = (uint32_t)rr * (uint32_t)p.ldw * 2u + (uint32_t)c * 16u;
            }
        }
#pragma unroll
        for (int i = 0; i < BW_PW; ++i) {
            const int idx = wave * BW_PW + i;
            const int ent = idx / (2 * (BN / 64)), hh = (idx / (BN / 64)) & 1, seg = idx % (BN / 64);
            const int nn = min(n0 + seg * 64 + ln, p.N - 1) - n0;
            bw_voff[i] = (uint32_t)(ent * e1off) * (uint32_t)p.sPb * 4u + (uint32_t)hh * (uint32_t)p.N * 4u + (uint32_t)nn * 4u;
        }
    };
    // piece pc of the loader's current k-tile into the ring slot at byte offset slot_off
    auto dma_piece = [&](auto pcc, uint32_t slot_off) {
        constexpr int pc = decltype(pcc)::value;
        if constexpr (pc < A_PW) {
            w4_dma16<pc * 1024>(a_voff[pc], ld_a, a_ldsw + slot_off);
        } else if constexpr (pc < A_PW + W_PW) {
            constexpr int i = pc - A_PW;
            w4_dma16<i * 1024>(w_voff[i], ld_w, w_ldsw + slot_off);
        } else {
            constexpr int i = pc - A_PW - W_PW;
            w4_dma4<0>(bw_voff[i], ld_p, bw_ldsw[i] + slot_off);
        }
    };
    auto dma_m0 = [&](auto pcc, uint32_t slot_off) {
        constexpr int pc = decltype(pcc)::value;
        if constexpr (pc < A_PW) w4_set_m0<pc * 1024>(a_ldsw + slot_off);
        else if constexpr (pc < A_PW + W_PW) w4_set_m0<(pc - A_PW) * 1024>(w_ldsw + slot_off);
        else w4_set_m0<0>(bw_ldsw[pc - A_PW - W_PW] + slot_off);
    };
    auto dma_go = [&](auto pcc) {
        constexpr int pc = decltype(pcc)::value;
        if constexpr (pc < A_PW) w4_dma16_m0(a_voff[pc], ld_a);
        else if constexpr (pc < A_PW + W_PW) w4_dma16_m0(w_voff[pc - A_PW], ld_w);
        else w4_dma4_m0(bw_voff[pc - A_PW - W_PW], ld_p);
    };
    auto loader_advance = [&]() {
        ld_a += 128; ld_p += p_step;
        if constexpr (FUSED) ld_w += 128;
        if (++ld_kt == nk) { ld_kt = 0; ++ld_r; loader_tile(ld_r); }
    };

    if constexpr (USE_LUT) {
        // OPT & 128 / 256 / 512 (operand-value energy A/B, harness only; results need the row-sum correction x.S = x.(2B) - sum x):
        //   128: bit -> {0x0000, 2.0 = 0x4000}   256: bit -> {0x0000, 1.0}   512: all zero
        constexpr uint32_t ONE = One2<DT>::v & 0xffffu;
        constexpr uint32_t POS = (Cfg::OPT & 512) ? 0u : (Cfg::OPT & 128) ? 0x4000u : ONE;
        constexpr uint32_t NEG = (Cfg::OPT & (128 | 256 | 512)) ? 0u : (ONE | 0x8000u);
        const int e = threadIdx.x;
        u32x4_t v;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            v[d] = (((e >> (2 * d)) & 1) ? POS : NEG) | ((((e >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
        *(u32x4_t*)(smem + Cfg::LUT_OFF + e * 16) = v;
    }

    // ---- register state of the multiply stream
    f32x16_t acc[4][TM];                       // [B fragment b][row block i]; fused: b = 2j (W) / 2j+1 (S)
    u32x4_t xf[2][TM];                         // X fragments of k-step s (parity s & 1)
    u32x4_t bf[4];                             // B fragments, produced 3 regions ahead
    uint32_t sw_lo[TN], sw_hi[TN];             // prepared sign words of the k-tile whose fragments are being produced
    uint32_t sw_raw[TN];

    auto zero_acc = [&]() {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bb][i][r] = 0.f;
    };

    // produce B fragment (k-step sp, fragment bp) from ring slot `st` (byte pointer) -- sign expansion or W read
    auto produce = [&](auto spc, auto bpc, const char* st) {
        constexpr int sp = decltype(spc)::value, bp = decltype(bpc)::value;
        if constexpr (FUSED && !(bp & 1)) {      // fused fragment order W0 S0 W1 S1: the step's LAST fragment is VALU-made (see ring note)
            bf[bp] = *(const u32x4_t*)(st + w_rd[sp] + (bp >> 1) * 4096);
        } else {
            constexpr int j = FUSED ? (bp >> 1) : bp;
            if constexpr (USE_LUT) {
                const uint32_t byte = (sw_raw[j] >> (8 * sp)) & 0xffu;
                bf[bp] = *(const u32x4_t*)(smem + Cfg::LUT_OFF + byte * 16);
            } else {
                const uint32_t src = sp < 2 ? sw_lo[j] : sw_hi[j];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int q = (sp & 1) * 4 + d;
                    bf[bp][d] = ((src << (15 - 2 * q)) & 0x80008000u) | one2;
                }
                asm volatile("" : "+v"(bf[bp]));      // pin: the IR optimiser otherwise sinks the expansion towards its use (3 regions later)
            }
        }
    };
    uint32_t sw_new[TN];                       // raw words of the NEXT k-tile (read at region 12, committed at region 13)
    auto read_words = [&](const char* st) {
#pragma unroll
        for (int j = 0; j < TN; ++j) sw_new[j] = *(const uint32_t*)(st + bw_rd + j * 128);
    };
    // prepared forms of word j: lo = {w[15:0], w[16:1]}, hi = {w[31:16], w[31:17]} of the inverted word, so that one shift puts bit 2q on
    // bit 15 and bit 2q+1 on bit 31 (4 VALU per word: v_not, v_lshrrev, 2 x v_perm)
    auto commit_word = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (USE_LUT) {
            sw_raw[j] = sw_new[j];
        } else {
            const uint32_t w = ~sw_new[j], w1 = w >> 1;
            sw_lo[j] = __builtin_amdgcn_perm(w1, w, 0x05040100u);     // bytes {w.0, w.1, w1.0, w1.1}
            sw_hi[j] = __builtin_amdgcn_perm(w1, w, 0x07060302u);     // bytes {w.2, w.3, w1.2, w1.3}
            asm volatile("" : "+v"(sw_lo[j]), "+v"(sw_hi[j]));
        }
    };

    // One region (4 MFMAs).  g = 4 s + b.  MF = 0: prologue form (no MFMAs, no DMA).
    auto region = [&](auto gc, auto mfc, const char* st_cur, const char* st_nxt, uint32_t slot_ld) {
        constexpr int g = decltype(gc)::value, s = g >> 2, bb = g & 3;
        constexpr bool MF = decltype(mfc)::value != 0;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 12) {
            if constexpr (Cfg::OPT & 8) wait_vmcnt<0>(); else
            wait_vmcnt<DPW>();                          // own pieces of k-tile c+1 have landed
            __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): own reads of k-tile c are done (slot c is refilled after B(c))
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_words(st_nxt);
        }
        // X fragments of the next k-step: regions b = 0, 1 read two each
        if constexpr (bb < 2 && !((Cfg::OPT & 32) && MF)) {
            const char* st = (s == 3) ? st_nxt : st_cur;
#pragma unroll
            for (int ii = 0; ii < TM / 2; ++ii) {                       // (two each; pair tiles: one each)
                const int i = (TM / 2) * bb + ii;
                xf[(s + 1) & 1][i] = *(const u32x4_t*)(st + a_rd[(s + 1) & 3] + i * 4096);
            }
        }
        // commit the next k-tile's words one per region: word j is first needed by fragment 16 + j (region 13 + j); the current
        // k-tile's word j was last used by fragment 12 + j, produced at region 9 + j
        if constexpr (g >= 13) { if constexpr (g - 13 < TN) commit_word(IC<(g - 13 < TN ? g - 13 : 0)>{}); }
        if constexpr (g == 0 && TN == 4) commit_word(IC<TN - 1>{});
        // B fragment f + 3 (f + 3 >= 16: first fragments of the next k-tile)
        {
            constexpr int f = g + 3;
            if constexpr (f < 16) produce(IC<(f >> 2) & 3>{}, IC<f & 3>{}, st_cur);
            else produce(IC<((f - 16) >> 2) & 3>{}, IC<(f - 16) & 3>{}, st_nxt);
        }
        if constexpr (MF) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // OPT & 4 (energy A/B only, results transposed inside each 32 x 32 block): the 8-wave kernels' order, sign fragment first
                if constexpr (Cfg::OPT & 4) acc[bb][i] = mfma32<DT>(bf[bb], xf[s & 1][i], acc[bb][i]);
                else if constexpr (Cfg::OPT & 2) acc[bb][i & 2] = mfma32<DT>(xf[s & 1][i], bf[bb], acc[bb][i & 2]);   // energy A/B: dependent pairs
                else if constexpr (Cfg::OPT & 1024) {      // energy A/B (results wrong): X-stationary order -- 4 consecutive MFMAs share the X fragment
                    const int bi = (bb + (i == 3 ? 0 : i)) & 3;
                    acc[i][bb] = mfma32<DT>(xf[s & 1][bb], bf[bi], acc[i][bb]);
                }
                else if constexpr (Cfg::OPT & 16) acc[bb][0] = mfma32<DT>(xf[s & 1][i], bf[bb], acc[bb][0]);          // energy A/B: chains of 4
                else acc[bb][i] = mfma32<DT>(xf[s & 1][i], bf[bb], acc[bb][i]);
            }
            // interleave: 1 MFMA, then its share of the region's reads and VALU
            // VALU per region: 8 (expansion) or 2 (LUT address) + 4 in the regions that commit a word; the MFMA in front of an LDS-DMA
            // piece gets none (the piece is 3 issue slots)
            constexpr int p0 = g < 12 ? g * DPW / 12 : 0, p1 = g < 12 ? (g + 1) * DPW / 12 : 0;
            static_assert(p1 - p0 <= 2, "at most two pieces per region");
            constexpr bool commits = !USE_LUT && (g >= 13 || (g == 0 && TN == 4));
            constexpr bool s_type = !(FUSED && !(((g + 3) & 3) & 1));
            constexpr int nvalu = (s_type ? (USE_LUT ? 2 : 8) : 0) + (commits ? 4 : 0) + 2;
            if constexpr (TM == 2) {
                // pair tiles: two MFMAs per region -- [MFMA, <= 2 DS, half the VALU] twice, the LDS-DMA piece(s) behind the second
                constexpr int vq = (nvalu + 1) / 2;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
                if constexpr (p1 > p0) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma_piece(IC<p0>{}, slot_ld);
                    if constexpr (p1 - p0 > 1) dma_piece(IC<(p0 + 1 < DPW ? p0 + 1 : p0)>{}, slot_ld);
                }
            } else if constexpr ((Cfg::OPT & 64) && p1 - p0 == 1) {
                // split form: [MFMA MFMA] m0 write [MFMA MFMA] load -- each statement alone in its gap
                constexpr int vq = (nvalu + 3) / 4;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
                __builtin_amdgcn_sched_barrier(0);
                dma_m0(IC<p0>{}, slot_ld);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
                __builtin_amdgcn_sched_barrier(0);
                dma_go(IC<p0>{});
            } else {
            constexpr int slots = p1 > p0 ? 3 : 4;
            constexpr int vq = (nvalu + slots - 1) / slots;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (slots == 4) __builtin_amdgcn_sched_group_barrier(0x002, vq, 0);
            if constexpr (p1 > p0 && !(Cfg::OPT & 8)) {
                __builtin_amdgcn_sched_barrier(0);
                dma_piece(IC<p0>{}, slot_ld);
                if constexpr (p1 - p0 > 1) dma_piece(IC<(p0 + 1 < DPW ? p0 + 1 : p0)>{}, slot_ld);
            }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#define BD_W4_ALL_PIECES(slot)                                                                                            \
    do {                                                                                                                  \
        dma_piece(IC<0>{}, slot); dma_piece(IC<(1 < DPW ? 1 : 0)>{}, slot);                                               \
        if constexpr (DPW > 2) dma_piece(IC<(2 < DPW ? 2 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 3) dma_piece(IC<(3 < DPW ? 3 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 4) dma_piece(IC<(4 < DPW ? 4 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 5) dma_piece(IC<(5 < DPW ? 5 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 6) dma_piece(IC<(6 < DPW ? 6 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 7) dma_piece(IC<(7 < DPW ? 7 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 8) dma_piece(IC<(8 < DPW ? 8 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 9) dma_piece(IC<(9 < DPW ? 9 : 0)>{}, slot);                                                  \
        if constexpr (DPW > 10) dma_piece(IC<(10 < DPW ? 10 : 0)>{}, slot);                                               \
        if constexpr (DPW > 11) dma_piece(IC<(11 < DPW ? 11 : 0)>{}, slot);                                               \
        if constexpr (DPW > 12) dma_piece(IC<(12 < DPW ? 12 : 0)>{}, slot);                                               \
        if constexpr (DPW > 13) dma_piece(IC<(13 < DPW ? 13 : 0)>{}, slot);                                               \
        static_assert(DPW >= 2 && DPW <= 14, "piece count");                                                              \
    } while (0)
#define BD_W4_R(G, MFV) region(IC<G>{}, IC<MFV>{}, st_cur, st_nxt, l_slot)

    // ================================================================ prologue
    int tm_c = 0, tn_c = 0, b_c = 0;
    if (!tile_of(0, tm_c, tn_c, b_c)) return;
    BD_W4_STAMP(15 - 1 + 0 * 1, 5);
    loader_tile(0);
    BD_W4_ALL_PIECES(0u); loader_advance();
    BD_W4_ALL_PIECES((uint32_t)STAGE); loader_advance();
    __builtin_amdgcn_sched_barrier(0);
    {
        // regions 12..15 without MFMAs: barrier (k-tile 0 landed, LUT visible), sign words, first X fragments, B fragments 0..2
        const char *st_cur = smem, *st_nxt = smem;
        const uint32_t l_slot = 0;
        BD_W4_R(12, 0); BD_W4_R(13, 0); BD_W4_R(14, 0); BD_W4_R(15, 0);
    }

    // ================================================================ tile loop
    int c_slot = 0;                                   // ring slot of the k-tile being multiplied
    // ---- trickled epilogue state: the PREVIOUS tile's outputs, packed, waiting to be staged + stored inside this tile's k loop
    constexpr bool TRICKLE = Cfg::TRICKLE;
    [[maybe_unused]] u32x2_t pk[Cfg::NCH][Cfg::JC * 4];
    [[maybe_unused]] u32x4_t rbuf[8];
    [[maybe_unused]] bool pk_live = false, pk_inside = false;
    [[maybe_unused]] int pk_m0 = 0, pk_n0 = 0, pk_b = 0;
    [[maybe_unused]] const bool trickle_able = TRICKLE && !p.accumulate && (p.N % 8 == 0) && (p.sCm % 8 == 0) && (p.sCb % 8 == 0) &&
                                               (((uintptr_t)p.C & 15) == 0) && nk >= Cfg::TRICKLE_KT;
    // slot SL (0..7) of group kq of the pending tile; wave-uniform branches only
    auto site = [&](auto slc, int kq) __attribute__((always_inline)) {
        if constexpr (TRICKLE) {
            constexpr int SL = decltype(slc)::value;
            if (pk_live && kq <= Cfg::NGRP) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, Cfg::NGRP + 1>([&](auto gc) {
                    if (kq == decltype(gc)::value)
                        w4_slot<Cfg, SL, decltype(gc)::value>(p, pk, rbuf, smem + Cfg::STG_OFF, pk_m0, pk_n0, wm, wn, pk_b, pk_inside, lane, wave);
                });
                __builtin_amdgcn_sched_barrier(0);
                if (SL == 7 && kq == Cfg::NGRP) pk_live = false;
            }
        }
    };
    // one k-tile of the multiply stream (16 regions) + the loader's step.  U = its index inside a group of four (trickle slots 2U, 2U + 1,
    // placed after regions 2 and 8: inside the window in which the k-tile's LDS-DMA pieces are issued, see w4_slot); U < 0: no slots
    auto ktile = [&](auto uc, int kq) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        const int n_slot = c_slot + 1 == NS ? 0 : c_slot + 1;
        const uint32_t l_slot = (n_slot + 1 == NS ? 0 : n_slot + 1) * STAGE;      // k-tile c+2 goes where k-tile c-1 was
        const char* st_cur = smem + c_slot * STAGE;
        const char* st_nxt = smem + n_slot * STAGE;
        BD_W4_R(0, 1); BD_W4_R(1, 1); BD_W4_R(2, 1);
        if constexpr (U >= 0) site(IC<(U >= 0 ? 2 * U : 0)>{}, kq);
        BD_W4_R(3, 1);
        BD_W4_R(4, 1); BD_W4_R(5, 1); BD_W4_R(6, 1); BD_W4_R(7, 1);
        BD_W4_R(8, 1);
        if constexpr (U >= 0) site(IC<(U >= 0 ? 2 * U + 1 : 0)>{}, kq);
        BD_W4_R(9, 1); BD_W4_R(10, 1); BD_W4_R(11, 1);
        BD_W4_R(12, 1); BD_W4_R(13, 1); BD_W4_R(14, 1); BD_W4_R(15, 1);
        loader_advance();
        c_slot = n_slot;
    };
    for (int r = 0;; ++r) {
        const int m0 = tm_c * BM, n0 = tn_c * BN;
        zero_acc();
        BD_W4_STAMP(r, 0);
        if constexpr (TRICKLE || (Cfg::OPT & 8192)) {      // (OPT & 8192: the k loop unrolled by four on its own, no trickle -- A/B of the loop form)
            // groups of four k-tiles, one trickle phase behind each; then the k-tiles left over (nk % 4).  Both loops have ONE exit: with
            // an exit test after every k-tile the 256 accumulators flow out through four edges and the allocator routes them through scratch.
            const int nk4 = nk >> 2;
            for (int kq = 0; kq < nk4; ++kq) {
                ktile(IC<0>{}, kq); ktile(IC<1>{}, kq); ktile(IC<2>{}, kq); ktile(IC<3>{}, kq);
#ifdef BD_TRACE
                if (kq == 1) BD_W4_STAMP(r, 1);
                if (kq == 9) BD_W4_STAMP(r, 2);
#endif
            }
            for (int kt = nk4 * 4; kt < nk; ++kt) ktile(IC<-1>{}, 0);
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                ktile(IC<-1>{}, 0);
#ifdef BD_TRACE
                if (kt == 7) BD_W4_STAMP(r, 1);
                if (kt == 39) BD_W4_STAMP(r, 2);
#endif
            }
        }
        BD_W4_STAMP(r, 3);
        // ---- epilogue of tile r (the DMA of the next tile's first two k-tiles is already in flight / landed)
        int tm_n = 0, tn_n = 0, b_n = 0;
        const bool more = tile_of(r + 1, tm_n, tn_n, b_n);
        bool deferred = false;
        if constexpr (TRICKLE) {
            if (trickle_able && more) {
                // another tile follows: convert now (the accumulators are about to be reused), stage + store inside its k loop
                w4_convert<Cfg>(p, acc, pk, smem + Cfg::STG_OFF, n0, wn, b_c, lane, wave);
                pk_live = true; pk_m0 = m0; pk_n0 = n0; pk_b = b_c;
                pk_inside = (m0 + BM <= p.M) && (n0 + BN <= p.N);
                deferred = true;
            }
        }
        if (!deferred) {
            if constexpr (Cfg::PAIR) {       // each wave row stores its own entry (the absent partner of an odd batch stores nothing)
                const int ent = 2 * (b_c / p.ksplit) + wm;
                if (ent < p.nent) w4_epilogue<Cfg>(p, acc, smem + Cfg::STG_OFF, 0, n0, wm, wn, ent * p.ksplit + b_c % p.ksplit, lane, wave, ent);
            } else w4_epilogue<Cfg>(p, acc, smem + Cfg::STG_OFF, m0, n0, wm, wn, b_c, lane, wave);
        }
        BD_W4_STAMP(r, 4);
        if (!more) break;
        tm_c = tm_n; tn_c = tn_n; b_c = b_n;
    }
    wait_vmcnt<0>();
}

#undef BD_W4_R
#undef BD_W4_ALL_PIECES

}  // namespace bd
