// Decode path, loader / consumer form (variant 700): ONE launch per Linear, one 5-wave block per CU -- a LOADER wave that streams
// the block's weights, packed sign words and activation slices into an LDS ring with LDS-DMA (global_load_lds, no registers in
// between), and four CONSUMER waves that turn landed ring slots into MFMAs.
//
//   y[b,m,:] = x[b,m,:] . W^T + alpha[b] * (x[b,m,:] . S_b)        R = B*M <= 16 activation rows, packed sign layout
//
// Reference call sites: DiffCompressModule.forward at decode (demo/demo_backend.py:93-98, M = 1, B = tenants; the loop that calls
// it: :245-251) and BinaryDiff.forward (bitdelta/diff.py:33-39) with a few tokens.
//
// Why, next to gemv_stream_kernel (variant 600; profiles/r02_decode_pmc.txt, r04_stream_probe.txt):
//   * in the register-load kernel a wave alternates "issue a stage of buffer loads" and "consume a stage" and the in-flight bytes
//     live in its VGPRs; here the loader does nothing but issue (about 20 cycles per 1-KiB piece), runs as far ahead as the ring
//     allows (NSLOT slots of a whole (16-column tile, 128-k iteration) stage each = 100-120 KiB per CU) and never waits for a
//     consumer's arithmetic; the consumers never issue a vector-memory instruction in the loop;
//   * the weight and sign streams can take the non-temporal policy (every byte is read once by ONE CU): with register loads nt
//     cost 9-35 % because several load instructions shared cache lines; an LDS-DMA piece is a whole 1-KiB run;
//   * work that used to sit in FRONT of the stream -- the RMSNorm of the residual stream (xmode 2), the copy of the activation rows
//     into LDS (xmode 1), the scale table -- now runs on the consumer waves WHILE the loader fills the ring.
//
// Work split (same as variant 600): block b owns the contiguous column range [b*cpb, (b+1)*cpb) and walks it in 16-column MFMA
// tiles; a stage is (tile, it) with it = the 128-k iteration; the block's stages form one flat stream j = tile * nit + it that the
// loader issues in order.  Consumer q takes the stages with it % 4 == q (so every consumer accumulates an interleaved quarter of
// k in ascending order), the four partial tiles meet in LDS at the end of a tile and consumer (tile & 3) sums them in consumer
// order: deterministic, no atomics, no workspace, no second launch.  (The accumulation order differs from variant 600's contiguous
// k quarters, so the two variants agree to fp32 rounding of the partial sums, not bit for bit; each is bit-identical to itself across
// its fused / unfused forms.)
//
// A ring slot holds, lane-linear (what a lane reads back is what the same lane id of the loader asked for):
//   [W: 4 steps x 64 lanes x 16 B][signs: dwordx4 / dword pieces, lane-linear each (RingSigns)][x (xmode 0): nxi pieces x 64 lanes x 16 B]
//   W     lane (li = l & 15, g = l >> 4), step s: the 8 elements k = 128 it + 32 s + 8 g .. + 7 of column c_lo + 16 tile + li
//   signs lane l: its t_pad tenant dwords of the packed layout (byte s of a dword = the 8 signs of step s; bd_gemv_stream.h, PK)
//   x     piece k, lane l: row 4 k + (l >> 4), 16-byte chunk ((l & 15) ^ row) of the row's 256-byte slice of this iteration (the XOR keeps
//         the consumers' fragment reads -- lane (li, g) reads chunk 4 s + g of row li % R -- off each other's banks)
// Synchronisation is flags in LDS, no s_barrier after the prologue: ready[slot] = use count written by the loader once its
// hand-counted s_waitcnt vmcnt says the stage has landed; done[slot] = use count written by the slot's consumer once the slot's
// bytes are in its registers.  vmcnt is a 6-bit counter: at most 63 / (pieces per stage) stages are in flight per loader, the rest
// of the ring holds landed, unconsumed stages.  Every spin is bounded (RING_SPIN_LIMIT polls, then s_trap): a protocol bug aborts
// the launch instead of hanging the GPU.
#pragma once
#include "../../../bitdelta_amd/csrc/bd_gemv_stream.h"

namespace bd {

constexpr int RING_SPIN_LIMIT = 1 << 21;
constexpr int RING_MAX_SLOTS = 32;
constexpr int RING_RED_BYTES = 4 * 64 * 8 * 4;               // [4 consumers][64 lanes][4 base + 4 delta] fp32 (one buffer: a tile is >= 4 stages)
constexpr int RING_LUT_BYTES = 256 * 4 * 16;                 // sign table, 4 copies (SX = 1)
constexpr int RING_ALPHA_MAX = 256;                            // (row, scale group) pairs of one block kept in LDS
constexpr int RING_FLAG_WORDS = 128;
constexpr int RING_LDS_MAX = 160 * 1024;
// flag words (dword indices from flag_off)
constexpr int RF_READY = 0, RF_DONE = 32, RF_TILE_ARR = 64 /* [4] */, RF_TILE_DONE = 72, RF_PRO_A = 76 /* [4] */,
              RF_PRO_B = 80 /* [4] */;

struct RingParams {
    GemvParams g;
    int cpb;                       // columns per block (multiple of 4)
    int tp;                        // dwords per lane per stage of the packed signs (t_pad)
    // W addressing in bytes from g.W: (n >> 4) * w_tile + (n & 15) * w_col + it * w_it + s * w_step + g * 16
    //   tile-major [N/16][K/128][4][16][4][8]: w_tile = nit * 4096, w_col = 64,      w_it = 4096, w_step = 1024
    //   row-major  [N][ldw]:                   w_tile = 32 * ldw,   w_col = 2 * ldw, w_it = 256,  w_step = 64
    uint32_t w_tile, w_col, w_it, w_step;
    // LDS geometry (bytes): ring at 0
    uint32_t slot_bytes, p_off, x_off, nslot;
    uint32_t red_off, alpha_off, flag_off, lut_off, xs_off, xrow;
    int xmode;                     // 0 = x rides the ring, 1 = resident copy made by the consumers, 2 = resident + RMSNorm (X = residual stream)
    int nxi;                       // xmode 0: x pieces per stage = ceil(R / 4)
    int nl;                        // loader waves (1 or 2): the block has 4 + nl waves
    int rot;                       // 1 = block b walks k cyclically from iteration (7 b) % nit: the CUs do not sweep the same DRAM offsets in lockstep
    int sx;                        // sign expansion of the consumers: 0 = VALU (2 ops per +-1 pair), 1 = 4-copy LDS table
    int nt;                        // 1 = non-temporal policy on the weight and sign streams
    int epi;                       // 1 = SwiGLU epilogue over an 8-interleaved gate|up pair (N/2 output columns)
    const unsigned short* nw;      // xmode 2: norm weight [tenants or 1, K]
    long long sNw;
    float eps;
    int jsh;                       // xmode 2: K = 2048 << jsh
    unsigned long long* trace;     // -DBD_RING_TRACE builds only: [5 waves][1024 stages][4] s_memtime stamps of block 0 (else unused)
};


// LDS geometry of a launch (host side; shared by the C ABI and the development tools).  tune: bit 0 nt streams, bit 1 activations ride
// the ring even when a resident copy fits, bit 2 one loader wave, bit 3 table sign expansion, bit 4 per-block rotation of the k walk,
// bits 8..13 cap on the ring slots.
// false: the launch does not fit (K % 128, K < 512, rows, t_pad, or fewer than 8 ring slots).
inline bool ring_plan_geometry(RingParams& rp, int R, int K, int tp, bool w_tiled, long long ldw, bool norm, int tune) {
    if (K % 128 || K < 512 || R < 1 || R > 16) return false;
    if (!(tp == 1 || tp == 2 || tp == 4 || tp == 6 || tp == 8)) return false;
    rp.tp = tp;
    rp.nt = tune & 1;
    rp.nl = (tune & 4) ? 1 : 2;
    rp.sx = (tune & 8) ? 1 : 0;
    rp.rot = (tune & 16) ? 1 : 0;
    rp.jsh = 0;
    while ((2048 << rp.jsh) < K) ++rp.jsh;
    const int nit = K / 128;
    if (w_tiled) { rp.w_tile = (uint32_t)nit * 4096u; rp.w_col = 64u; rp.w_it = 4096u; rp.w_step = 1024u; }
    else { rp.w_tile = 32u * (uint32_t)ldw; rp.w_col = 2u * (uint32_t)ldw; rp.w_it = 256u; rp.w_step = 64u; }
    const uint32_t fixed = RING_RED_BYTES + RING_ALPHA_MAX * 4 + RING_FLAG_WORDS * 4 + (rp.sx ? RING_LUT_BYTES : 0);
    const uint32_t xrow = (uint32_t)K * 2u + 32u;
    const long long xs_bytes = (long long)R * xrow;
    const uint32_t slot_res = 4096u + (tp > 4 ? 2048u : 1024u);                  // resident activations: [W][sign pieces, 1 KiB each]
    auto slots = [&](uint32_t slot_bytes, long long extra) {
        const long long room = (long long)RING_LDS_MAX - fixed - extra;
        int n = room <= 0 ? 0 : (int)(room / slot_bytes);
        if (n > RING_MAX_SLOTS) n = RING_MAX_SLOTS;
        const int cap = (tune >> 8) & 63;
        if (cap && n > cap) n = cap;
        return n & ~3;
    };
    int xmode = 0;
    if (norm) {
        if (slots(slot_res, xs_bytes) < 8) return false;
        xmode = 2;
    } else if (!(tune & 2) && slots(slot_res, xs_bytes) >= 12) {
        xmode = 1;
    }
    rp.xmode = xmode;
    rp.nxi = (R + 3) / 4;
    rp.p_off = 4096u;
    rp.x_off = slot_res;
    rp.slot_bytes = xmode == 0 ? slot_res + 1024u * (uint32_t)rp.nxi : slot_res;
    rp.slot_bytes = (rp.slot_bytes + 255u) & ~255u;
    const int ns = slots(rp.slot_bytes, xmode ? xs_bytes : 0);
    if (ns < 8) return false;
    rp.nslot = (uint32_t)ns;
    rp.red_off = rp.nslot * rp.slot_bytes;
    rp.alpha_off = rp.red_off + RING_RED_BYTES;
    rp.flag_off = rp.alpha_off + RING_ALPHA_MAX * 4;
    rp.lut_off = rp.flag_off + RING_FLAG_WORDS * 4;
    rp.xs_off = rp.lut_off + (rp.sx ? RING_LUT_BYTES : 0);
    rp.xrow = xrow;
    return true;
}
inline unsigned ring_lds_bytes(const RingParams& rp) { return rp.xs_off + (rp.xmode ? (unsigned)rp.g.R * rp.xrow : 0u); }

typedef volatile __attribute__((address_space(3))) uint32_t* ldsflag_t;
typedef __attribute__((address_space(3))) char* ldsptr_t;

template <int NT> __device__ __forceinline__ void ring_dma16(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// four 1-KiB pieces into consecutive 1-KiB LDS runs: one M0 save / restore for the block (the loader's issue rate is what feeds the CU)
template <int NT> __device__ __forceinline__ void ring_dma16x4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5 nt\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5 nt\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5 nt\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_addr) : "memory", "scc");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_addr) : "memory", "scc");
}
template <int NT> __device__ __forceinline__ void ring_dma16x2(uint32_t v0, uint32_t v1, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sbase), "s"(lds_addr) : "memory", "scc");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sbase), "s"(lds_addr) : "memory", "scc");
}

// s_waitcnt vmcnt(n) for a run-time n in 0 .. 63 (the immediate is part of the instruction)
__device__ __forceinline__ void ring_wait_vm(int n) {
    switch (n) {
#define BD_W1(i) case i: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(i) : "memory"); break;
#define BD_W8(b) BD_W1(b) BD_W1(b + 1) BD_W1(b + 2) BD_W1(b + 3) BD_W1(b + 4) BD_W1(b + 5) BD_W1(b + 6) BD_W1(b + 7)
        BD_W8(0) BD_W8(8) BD_W8(16) BD_W8(24) BD_W8(32) BD_W8(40) BD_W8(48) BD_W8(56)
#undef BD_W8
#undef BD_W1
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// bounded spin on an LDS word; gives up with a trap (the launch fails loudly; nothing hangs)
__device__ __forceinline__ void ring_spin_until(ldsflag_t f, uint32_t want) {
    int spin = 0;
#pragma clang loop unroll(disable)
    while (*f != want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spin > RING_SPIN_LIMIT) __builtin_trap();
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ring_spin_until4(ldsflag_t f, uint32_t want) {      // four consecutive words (one per consumer)
    int spin = 0;
#pragma clang loop unroll(disable)
    for (;;) {
        const uint32_t a = f[0], b = f[1], c = f[2], d = f[3];
        if (a == want && b == want && c == want && d == want) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spin > RING_SPIN_LIMIT) __builtin_trap();
    }
    asm volatile("" ::: "memory");
}

// Sign block of a stage for t_pad = NM dwords per lane: [4 lane groups g][16 columns][NM] dwords = 16 NM chunks of 16 bytes, copied
// chunk by chunk in lane order (consecutive lanes read consecutive 16 bytes of a 4-column run: whole cache lines per request group) by
// PI dwordx4 pieces; the last piece's surplus lanes repeat the last chunk (their LDS bytes are never read).  In LDS the block is in its
// natural order, so consumer lane l = 16 g + li finds its NM dwords at l * NM * 4.
template <int NM> struct RingSigns {
    static_assert(NM == 1 || NM == 2 || NM == 4 || NM == 6 || NM == 8, "t_pad");
    static constexpr int CHUNKS = 16 * NM;
    static constexpr int PI = (CHUNKS + 63) / 64;                      // 1 (t_pad <= 4) or 2
};

// Development probe (-DBD_RING_TRACE, tests/native/ring_trace): wave-level s_memtime stamps of block 0 into rp.trace --
//   loader:   trace[0][j][0..2] = { slot free seen, stage issued, stage published }
//   consumer: trace[1 + q][i][0..3] = { ready seen, bytes in registers (slot released), compute done, tile finished (or 0) }
#ifdef BD_RING_TRACE
#define BD_RT(w, idx, slot_, val)                                                                              \
    do {                                                                                                       \
        if (rp.trace && blockIdx.x == 0 && lane == 0 && (idx) < 1024) rp.trace[((w) * 1024 + (idx)) * 4 + (slot_)] = (val); \
    } while (0)
#define BD_RT_NOW() __builtin_amdgcn_s_memtime()
#else
#define BD_RT(w, idx, slot_, val) do {} while (0)
#define BD_RT_NOW() 0ull
#endif

template <int DT, int NM, int NT>
__global__ void __launch_bounds__(384) gemv_ring_kernel(const RingParams rp) {
    using SG = RingSigns<NM>;
    constexpr int PI = SG::PI;
    const GemvParams& p = rp.g;
    extern __shared__ __attribute__((aligned(1024))) char dyn_lds[];
    const ldsptr_t lds = (ldsptr_t)dyn_lds;
    const ldsflag_t flags = (ldsflag_t)(lds + rp.flag_off);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int c_lo = blk * rp.cpb, c_hi = min(p.N, c_lo + rp.cpb);
    const int ntile = (c_hi - c_lo + 15) >> 4;
    const int nit = p.K >> 7;                                           // K % 128 == 0 (host-checked)
    const int total = ntile * nit;
    const int NSLOT = (int)rp.nslot;
    // stage (tile, i) covers the 128-k iteration (i + it0) % nit: with rot, neighbouring blocks start a seventh of... an odd stride
    // apart, so the chip's 256 sequential streams are spread over the DRAM channels instead of sweeping the same offsets together
    const int it0 = rp.rot ? (blk * 7) % nit : 0;

    // The flag words are zeroed by the first loader and every wave meets ONCE, before anything else: from here on nobody waits at a
    // barrier (round 4's first version had the loaders meet the consumers after their prologue loads -- 6 us of idle stream).
    if (wave == 4) {
        flags[lane] = 0u;
        flags[64 + lane] = 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if (wave >= 4) {
        // =================================================================================== LOADERS (rp.nl of them: waves 4, 5)
        // loader lw issues the stages j = lw, lw + nl, ...; nslot % nl == 0, so its slots are j % nslot for its own j only
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - 4, NL = rp.nl;
        const uint32_t ring0 = lds_addr_of(dyn_lds);
        const int OPS = 4 + PI + (rp.xmode == 0 ? rp.nxi : 0);
        const int DMAX = 63 / OPS;                                      // stages whose pieces fit this wave's 6-bit vmcnt
        // ---- per-lane source offsets.  Every piece is lane-linear in LDS (lane l lands at 16 l), so the lane -> source map decides
        // both what the consumers find where and how the requests coalesce: consecutive lanes read consecutive 16-byte chunks.
        //   W piece s:    lane l = 4 r + c -> row r of the tile, chunk c: the 8 elements k = 128 it + 32 s + 8 c .. + 7
        //   sign piece k: lane l -> chunk 64 k + l of the tile's [4 g][16 columns][tp] dword block (RingSigns)
        //   x piece k:    lane l = 16 g + li -> row 4 k + g (clamped), chunk li ^ row of the row's 256-byte slice
        uint32_t xl[4] = {0u, 0u, 0u, 0u};
        if (rp.xmode == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = min(4 * k + g, p.R - 1), b = row / p.M, m = row - b * p.M;
                xl[k] = (uint32_t)(((long long)b * p.sXb + (long long)m * p.sXm) * 2) + (uint32_t)((li ^ row) & 15) * 16u;
            }
        }
        const int wr = lane >> 2, wc = lane & 3;
        int sgrp[PI], scol[PI];
        uint32_t swithin[PI];
#pragma unroll
        for (int k = 0; k < PI; ++k) {
            const int c = min(64 * k + lane, SG::CHUNKS - 1), jj = c % (4 * NM), o = 16 * jj;
            sgrp[k] = c / (4 * NM);
            scol[k] = o / (4 * NM);
            swithin[k] = (uint32_t)(o - scol[k] * 4 * NM);
        }
        const int n_pad = ((p.N + 15) >> 4) << 4;                       // the packed sign tiles (and the tile-major W) are whole 16-column tiles
        uint32_t w_lane = 0, p_lane[PI];
        auto set_tile = [&](int t) {
            const int n = min(c_lo + t * 16 + wr, p.N - 1);
            w_lane = (uint32_t)(n >> 4) * rp.w_tile + (uint32_t)(n & 15) * rp.w_col + (uint32_t)wc * 16u;
#pragma unroll
            for (int k = 0; k < PI; ++k) {
                // packed dword index ((tile_g * nit + it) * 4 + g) * 16 + (n & 15), tp dwords each.  A column past the packed tiles (last
                // tile of the last block) is clamped AND read from its first byte: a chunk that started 16 bytes into the last column
                // would run 8 bytes past the end of the array (memory fault on the N = 6144 launches of the first v3 run)
                const int nv = c_lo + t * 16 + scol[k];
                const int ns = min(nv, n_pad - 1);
                p_lane[k] = (((uint32_t)(ns >> 4) * (uint32_t)nit * 4u + (uint32_t)sgrp[k]) * 16u + (uint32_t)(ns & 15)) * (uint32_t)NM * 4u +
                            (nv > n_pad - 1 ? 0u : swithin[k]);
            }
        };
        const uint32_t p_it = 64u * (uint32_t)NM * 4u;                  // bytes per iteration of the packed signs
        // my stages: count, and the incremental (tile, it, slot) of the next one to issue / to publish
        const int mine = total > lw ? (total - lw + NL - 1) / NL : 0;
        int issued = 0, published = 0;
        int t = 0, it = lw, slot = lw % NSLOT, pslot = lw % NSLOT;
        uint32_t use = 0, puse = 1;                                     // use = j / nslot of the next stage to issue; puse = its j / nslot + 1 of the next to publish
        while (it >= nit) { it -= nit; ++t; }
        set_tile(t);
        // ONE loop with one issue site and one publish site (the instruction cache is shared: every duplicated body costs all the waves)
        int spin = 0;
        uint32_t done_seen = 0;
#pragma clang loop unroll(disable)
        for (;;) {
            bool can = issued < mine && issued - published < DMAX;
            if (can && use > 0) {                                       // the slot's previous stage must have been consumed
                if (done_seen != use) done_seen = flags[RF_DONE + slot];
                can = done_seen == use;
            }
            if (can) {
                BD_RT(lw ? 5 : 0, issued, 0, BD_RT_NOW());
                const uint32_t base = ring0 + (uint32_t)slot * rp.slot_bytes;
                int itr = it + it0;
                if (itr >= nit) itr -= nit;
                const uint32_t wo = w_lane + (uint32_t)itr * rp.w_it;
                ring_dma16x4<NT>(wo, wo + rp.w_step, wo + 2u * rp.w_step, wo + 3u * rp.w_step, p.W, base);
                if constexpr (PI == 2) ring_dma16x2<NT>(p_lane[0] + (uint32_t)itr * p_it, p_lane[PI - 1] + (uint32_t)itr * p_it, p.P, base + rp.p_off);
                else ring_dma16<NT>(p_lane[0] + (uint32_t)itr * p_it, p.P, base + rp.p_off);
                if (rp.xmode == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < rp.nxi) ring_dma16<0>(xl[k] + (uint32_t)itr * 256u, p.X, base + rp.x_off + k * 1024);
                }
                BD_RT(lw ? 5 : 0, issued, 1, BD_RT_NOW());
                ++issued;
                it += NL;
                if (it >= nit) {
                    do { it -= nit; ++t; } while (it >= nit);
                    set_tile(t);
                }
                slot += NL;
                if (slot >= NSLOT) { slot -= NSLOT; ++use; }
                // the done flag of the NEXT slot is read now and looked at in the next round: its latency hides behind the round
                if (use > 0) done_seen = flags[RF_DONE + slot];
                spin = 0;
            } else if (published < issued) {
                // nothing to issue right now (ring full, vmcnt budget used, or all issued): hand over the oldest in-flight stage
                ring_wait_vm((issued - published - 1) * OPS);
                flags[RF_READY + pslot] = puse;
                BD_RT(lw ? 5 : 0, published, 2, BD_RT_NOW());
                ++published;
                pslot += NL;
                if (pslot >= NSLOT) { pslot -= NSLOT; ++puse; }
            } else if (issued >= mine) {
                break;
            } else {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > RING_SPIN_LIMIT) __builtin_trap();
            }
        }
        return;
    }

    // ======================================================================================= CONSUMERS (waves 0..3, threads 0..255)
    const int q = wave;
    // ---- prologue loads (oldest in the wave's queue): scales, activation rows / norm weights
    const int g0 = rp.epi ? 0 : c_lo / p.gsz, ng = rp.epi ? 2 : (c_hi - 1) / p.gsz - g0 + 1;
    const bool al_lds = p.alpha != nullptr && p.R * ng <= RING_ALPHA_MAX;
    float a_pre = 0.f;
    if (al_lds) {
        const int idx = min((int)threadIdx.x, p.R * ng - 1), r = idx / ng, jg = idx - r * ng;
        a_pre = p.alpha[(long long)(r / p.M) * p.sAlb + g0 + jg];
    }
    float* const a_lds = (float*)(dyn_lds + rp.alpha_off);
    float* const red = (float*)(dyn_lds + rp.red_off);

    constexpr int XCH = 16;
    if (rp.xmode == 2) {
        // RMSNorm of the residual stream, rmsnorm_tenant_kernel's arithmetic and thread mapping (bd_serving.h): bit-identical to the
        // separate launch.  M == 1, K = 2048 << jsh, R * K <= 16 * 2048 (host-checked).
        const int jsh = rp.jsh;
        u32x4_t xraw[XCH], graw[XCH];
#pragma unroll
        for (int jj = 0; jj < XCH; ++jj) {
            const int r = jj >> jsh, c = ((int)threadIdx.x + 256 * (jj - (r << jsh))) * 8;
            const bool ok = r < p.R;
            xraw[jj] = ok ? *(const u32x4_t*)(p.X + (long long)r * p.sXb + c) : u32x4_t{0u, 0u, 0u, 0u};
            graw[jj] = ok ? *(const u32x4_t*)(rp.nw + (long long)r * rp.sNw + c) : u32x4_t{0u, 0u, 0u, 0u};
        }
        float* const part = red;                                        // [row][consumer]: idle until the first tile ends
        float ss = 0.f;
#pragma unroll
        for (int jj = 0; jj < XCH; ++jj) {
            ss = sumsq8<DT>(xraw[jj], ss);
            if (((jj + 1) & ((1 << jsh) - 1)) == 0) {
                const float w = wave_sum(ss);
                if (lane == 0) part[(jj >> jsh) * 4 + q] = w;
                ss = 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flags[RF_PRO_A + q] = 1u;
        ring_spin_until4(flags + RF_PRO_A, 1u);
#pragma unroll
        for (int jj = 0; jj < XCH; ++jj) {
            const int r = jj >> jsh, c = ((int)threadIdx.x + 256 * (jj - (r << jsh))) * 8;
            if (r < p.R) {
                const float rs = rms_scale(part[r * 4], part[r * 4 + 1], part[r * 4 + 2], part[r * 4 + 3], p.K, rp.eps);
                *(u32x4_t*)(dyn_lds + rp.xs_off + (uint32_t)r * rp.xrow + (uint32_t)c * 2u) = norm8<DT>(xraw[jj], graw[jj], rs);
            }
        }
    } else if (rp.xmode == 1) {
        // resident copy of the R activation rows: 16-byte chunks, thread-linear over (row, chunk)
        const int cpr = p.K >> 3, nch = p.R * cpr;
#pragma clang loop unroll(disable)
        for (int c0 = 0; c0 < nch; c0 += 256 * 4) {
            u32x4_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256 + (int)threadIdx.x;
                const int r = min(c, nch - 1) / cpr, cc = min(c, nch - 1) - r * cpr, b = r / p.M, m = r - b * p.M;
                v[u] = *(const u32x4_t*)(p.X + (long long)b * p.sXb + (long long)m * p.sXm + cc * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256 + (int)threadIdx.x;
                if (c < nch) {
                    const int r = c / cpr, cc = c - r * cpr;
                    *(u32x4_t*)(dyn_lds + rp.xs_off + (uint32_t)r * rp.xrow + (uint32_t)cc * 16u) = v[u];
                }
            }
        }
    }
    if (rp.sx) {     // sign table, 4 copies: entry e (8 sign bits -> 8 x +-1.0) of copy c at e * 64 + c * 16; slot index = thread-linear
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int sl = (int)threadIdx.x + 256 * jj, ee = sl >> 2;
            u32x4_t w;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = (((ee >> (2 * d)) & 1) ? POS : NEG) | ((((ee >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
            *(u32x4_t*)(dyn_lds + rp.lut_off + sl * 16) = w;
        }
    }
    if (al_lds && (int)threadIdx.x < p.R * ng) a_lds[threadIdx.x] = a_pre;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) flags[RF_PRO_B + q] = 1u;
    ring_spin_until4(flags + RF_PRO_B, 1u);

    // ---- per-lane read offsets
    const int rr = li % p.R;                                            // activation row of MFMA column li (lanes >= R: a duplicate, never stored)
    uint32_t xoff[4];                                                   // + stage base: the 16-byte activation fragment of step s
#pragma unroll
    for (int s = 0; s < 4; ++s)
        xoff[s] = rp.xmode == 0 ? rp.x_off + (uint32_t)rr * 256u + (uint32_t)(((4 * s + g) ^ rr) & 15) * 16u
                                : (uint32_t)rr * rp.xrow + (uint32_t)(4 * s + g) * 16u;
    const uint32_t woff = (uint32_t)li * 64u + (uint32_t)g * 16u;        // row li, chunk g of a 1-KiB step run (the loader copies it linearly)
    uint32_t one2 = One2<DT>::v;
    asm volatile("" : "+v"(one2));                                      // opaque: keeps v_and_or_b32's third operand in a register

    struct Stage { u32x4_t wf[4]; u32x4_t xf[4]; uint32_t wd[NM]; };
    // number of this consumer's stages per tile, and in total
    const int cq = q < nit ? (nit - q + 3) >> 2 : 0;
    const int mine = cq * ntile;

    // own-stage iterator: (tile, it) -> ring slot and use count, advanced incrementally (no divisions in the loop)
    struct Pos { int tile, it, slot; uint32_t use; };
    auto advance = [&](Pos& o) {
        int dj = 4;
        o.it += 4;
        if (o.it >= nit) { dj = nit - (o.it - 4) + q; o.it = q; ++o.tile; }
        o.slot += dj;
#pragma clang loop unroll(disable)
        while (o.slot >= NSLOT) { o.slot -= NSLOT; ++o.use; }
    };
    auto fetch = [&](Stage& st, const Pos& o) {
        ring_spin_until(flags + RF_READY + o.slot, o.use);
        const ldsptr_t sb = lds + (uint32_t)o.slot * rp.slot_bytes;
        int itr = o.it + it0;
        if (itr >= nit) itr -= nit;
        const ldsptr_t xb = rp.xmode == 0 ? sb : lds + rp.xs_off + (uint32_t)itr * 256u;
        {   // this lane's NM tenant dwords: natural order, lane l at l * NM * 4
            const ldsptr_t pp = sb + rp.p_off + (uint32_t)lane * (NM * 4u);
            if constexpr (NM == 1) {
                st.wd[0] = *(const __attribute__((address_space(3))) uint32_t*)pp;
            } else if constexpr (NM == 4 || NM == 8) {
#pragma unroll
                for (int k = 0; k < NM / 4; ++k) {
                    const u32x4_t v = *(const __attribute__((address_space(3))) u32x4_t*)(pp + 16 * k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) st.wd[4 * k + e] = v[e];
                }
            } else {                                                    // 2 or 6: 8-byte aligned pairs
#pragma unroll
                for (int k = 0; k < NM / 2; ++k) {
                    const u32x2_t v = *(const __attribute__((address_space(3))) u32x2_t*)(pp + 8 * k);
                    st.wd[2 * k] = v[0];
                    st.wd[2 * k + 1] = v[1];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) st.wf[s] = *(const __attribute__((address_space(3))) u32x4_t*)(sb + woff + s * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) st.xf[s] = *(const __attribute__((address_space(3))) u32x4_t*)(xb + xoff[s]);
    };

    f32x4_t accB = {0.f, 0.f, 0.f, 0.f}, accD[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const ldsptr_t lut_lane = lds + rp.lut_off + (uint32_t)(lane & 3) * 16u;
    // steps [s0, s1) of a stage: base MFMA + one MFMA per tenant mask on the step's activation fragment.  Sign fragment of (tenant t,
    // step s) = byte s of the tenant's dword: VALU expansion (v_perm_b32 replicates the byte into both halves, then 2 ops per pair of
    // +-1.0, expand_signs8) or the 4-copy LDS table (v_bfe + v_lshl_add + ds_read_b128; ~2-way conflicts on random bytes)
    auto compute = [&](const Stage& st, int s0, int s1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < s0 || s >= s1) continue;
            accB = mfma16<DT>(st.wf[s], st.xf[s], accB);
            if (rp.sx) {
                u32x4_t sf[NM];
#pragma unroll
                for (int t = 0; t < NM; ++t)
                    sf[t] = *(const __attribute__((address_space(3))) u32x4_t*)(lut_lane + (((st.wd[t] >> (8 * s)) & 0xffu) << 6));
#pragma unroll
                for (int t = 0; t < NM; ++t) accD[t] = mfma16<DT>(sf[t], st.xf[s], accD[t]);
            } else {
#pragma unroll
                for (int t = 0; t < NM; ++t) {
                    const uint32_t rep = __builtin_amdgcn_perm(~st.wd[t], 0u, 0x0c000c00u | ((4u + s) << 16) | (4u + s));
                    accD[t] = mfma16<DT>(expand_signs8(rep, 0, one2), st.xf[s], accD[t]);
                }
            }
        }
    };

    auto finish_tile = [&](int tile) {
        const uint32_t visit = (uint32_t)tile;
        ring_spin_until(flags + RF_TILE_DONE, visit);                   // the previous tile has been summed (one buffer)
        float* const rb = red;
        {
            const int b = min(li, p.R - 1) / p.M;
            const int bm = p.sPb == 0 ? 0 : b;
            u32x4_t d = __builtin_bit_cast(u32x4_t, accD[0]);
#pragma unroll
            for (int t = 1; t < NM; ++t) {
                const uint32_t mk = bm == t ? 0xffffffffu : 0u;
                d = (__builtin_bit_cast(u32x4_t, accD[t]) & u32x4_t{mk, mk, mk, mk}) | (d & ~u32x4_t{mk, mk, mk, mk});
            }
            *(f32x4_t*)&rb[(q * 64 + lane) * 8] = accB;
            *(u32x4_t*)&rb[(q * 64 + lane) * 8 + 4] = d;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flags[RF_TILE_ARR + q] = visit + 1u;
        if (q == (tile & 3)) {
            ring_spin_until4(flags + RF_TILE_ARR, visit + 1u);
            if (li < p.R) {
                const int b = li / p.M;
                f32x4_t sb = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {                           // fixed consumer order: deterministic
                    sb += *(const f32x4_t*)&rb[(w * 64 + lane) * 8];
                    sd += *(const f32x4_t*)&rb[(w * 64 + lane) * 8 + 4];
                }
                if (rp.epi) {
                    // tile = [8 gate | 8 up] columns: lane groups 0, 1 hold gate columns 4 g + e, groups 2, 3 the matching up columns
                    const int grp = g >> 1;
                    float a = 1.f;
                    if (al_lds) a = a_lds[li * 2 + grp];
                    else if (p.alpha) a = p.alpha[(long long)b * p.sAlb + grp];
                    const int n_out = ((c_lo + tile * 16) >> 1) + 4 * (g & 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = round16<DT>(scale_then_add(sd[e], a, sb[e]));
                        const float u = __shfl(v, (lane + 32) & 63, 64);
                        if (g < 2) {
                            const long long off = (long long)b * p.sCb + (long long)(li - b * p.M) * p.sCm + n_out + e;
                            ((unsigned short*)p.C)[off] = (unsigned short)swiglu1<DT>(v, u);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = c_lo + tile * 16 + 4 * g + e;
                        if (n < c_hi) {
                            float a = 1.f;
                            if (al_lds) a = a_lds[li * ng + (n / p.gsz - g0)];
                            else if (p.alpha) a = p.alpha[(long long)b * p.sAlb + n / p.gsz];
                            store_out<DT>(p, li, n, scale_then_add(sd[e], a, sb[e]));
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flags[RF_TILE_DONE] = visit + 1u;
        }
        accB = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };

    // One loop, one site of everything (instruction-cache footprint): the reads of stage i + 1 are issued into `nxt` before stage i
    // computes out of `cur`, and its ring slot is handed back as soon as those reads have returned -- after the first MFMA step of stage
    // i, not at the next round (the ring is what bounds the bytes in flight).  The register hand-over is a copy (~40 v_mov per stage).
    // A consumer that owns no iteration (K < 512) still meets every tile.
    Stage cur, nxt;
    Pos pc{0, q, q, 1u}, pn{0, q, q, 1u};                               // stream index of the first own stage = q < 8 <= NSLOT
    if (mine > 0) {
        fetch(nxt, pn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flags[RF_DONE + pn.slot] = pn.use;
    }
    const int rounds = mine > 0 ? mine : ntile;
#pragma clang loop unroll(disable)
    for (int i = 0; i < rounds; ++i) {
        bool last_of_tile = true;
        int tile = i;
        if (mine > 0) {
            cur = nxt;
            pc = pn;
            BD_RT(1 + q, i, 1, BD_RT_NOW());
            const bool more = i + 1 < mine;
            if (more) {
                advance(pn);
                fetch(nxt, pn);
                BD_RT(1 + q, i + 1, 0, BD_RT_NOW());
            }
            compute(cur, 0, 1);
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next stage's bytes are in registers: hand its slot back
                flags[RF_DONE + pn.slot] = pn.use;
            }
            compute(cur, 1, 4);
            BD_RT(1 + q, i, 2, BD_RT_NOW());
            last_of_tile = pc.it + 4 >= nit;
            tile = pc.tile;
        }
        if (last_of_tile) {
            finish_tile(tile);
            BD_RT(1 + q, i, 3, BD_RT_NOW());
        }
    }
}

}  // namespace bd
