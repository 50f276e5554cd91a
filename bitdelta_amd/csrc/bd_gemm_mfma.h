// W1A16 binary-delta GEMM for gfx950 (MI355X): MFMA path.
//
//   delta-only :  C[b] = X[b] . S[b]                     (binary_bmm / binary_matmul,
//                                                          reference bitdelta/binary_gemm_kernel.py:48-335)
//   fused      :  C[b] = X[b] . W^T + alpha[b] * (X[b] . S[b])   (BinaryDiff.forward, bitdelta/diff.py:33-39;
//                                                          DiffCompressModule.forward, demo/demo_backend.py:93-98)
//   S[k,n] = 2*bit(P[k/32,n], k%32) - 1,   P int32 [K/32, N]  (bit j of word i <-> k = 32 i + j, :109-111)
//
// Design (see DESIGN.md "delta_gemm_mfma"):
//  * X tile [BM x 64k] goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) in a NS-deep ring, one s_barrier per
//    k-tile, hand-counted s_waitcnt vmcnt(N) so NS-1 tiles stay in flight across barriers.
//  * the packed sign words of the tile ([2 x BN] int32 = BN*8 bytes per 64 k) ride the same ring by 4-byte LDS-DMA;
//    each lane reads ONE word per 32 output columns per k-tile and expands it in registers into +-1.0 16-bit
//    fragments (2 VALU / dword), so the 1-bit operand never costs LDS bandwidth or a bf16 LDS image.
//  * MFMA 32x32x16 with SWAPPED operands (first = S fragment, second = X fragment): the accumulator then holds
//    D[n][m] with 4 consecutive n per register quad, so C rows are stored as 8/16-byte pieces.
//  * k order inside a 64-wide tile is permuted (lane-half h, step s  <->  k = 32h + 8s + 0..7) identically for both
//    operands, so one 32-bit word feeds the four MFMA steps of its lane with no cross-lane traffic.
//  * fused mode keeps ONE accumulator set: delta loop, acc *= alpha, then the base loop (X and W tiles by LDS-DMA)
//    accumulates x.W^T on top; one rounding at the end.
//
// Fast-path requirements (checked by the host dispatcher, bd_api.hip): K % 64 == 0, X rows 16-byte aligned
// (sAm % 8 == 0, base % 16 == 0), W likewise.  Everything else goes to the generic kernel (bd_gemm_generic.h).
#pragma once
#include "bd_common.h"

namespace bd {

struct GemmParams {
    const char* A;        // X  [B, M, K] 16-bit
    const int32_t* P;     // packed signs [B or 1, K/32, N]
    char* C;              // out [B, M, N]
    const char* W;        // base weight [N, K] 16-bit (fused only)
    const float* alpha;   // fp32 [B or 1, G] (fused / accumulate only)
    int M, N, K;
    int tiles_m, tiles_n;     // tiles covered by THIS launch ...
    int tile_m0, tile_n0;     // ... starting at this tile of the problem (launch-chunking experiment, bd_api.hip launch_tile)
    int ksplit;               // >= 1.  > 1 (bd_gemm_fx.h only): blockIdx.y = b * ksplit + ks; the block contracts k-tiles
                              // [ks*nk/ksplit, (ks+1)*nk/ksplit) and writes its partial to C[blockIdx.y] (an fp32 workspace)
    long long sAb, sPb, sCb;  // batch strides in elements (sPb = 0 broadcasts one mask)
    int sAm, sCm, ldw;        // row strides in elements
    int sAlb, gsz;            // alpha batch stride (0 = broadcast), columns per scale group
    int round_mode;           // 1: fp32 -> fp16 -> out (reference epilogue); 0: fp32 -> out
    int accumulate;           // delta-only: C = C_in + alpha * acc  (adds the delta onto an existing base GEMM result)
    int nbatch;               // bd_gemm_w4.h only (persistent grid over batch x tiles; the other kernels take the batch from blockIdx.y)
    int nent;                 // bd_gemm_w4.h pair tiles only: batch entries of the problem (nbatch = pairs = (nent + 1) / 2)
    int group_m;              // tile order: tiles are walked in groups of group_m tile rows, m fastest inside a group, then n.
                              // 1 = n fastest (an XCD's contiguous run shares X row panels in its L2; right when the other
                              // operand is the tiny packed mask); tiles_m = m fastest (the XCD owns a column slice of W);
                              // in between = a 2-D block per XCD, which minimises X + W bytes per XCD for the fused kernel
};

__device__ __forceinline__ void tile_coords(const GemmParams& p, int wg, int& tile_m, int& tile_n) {
    const int gsz = p.group_m * p.tiles_n;                       // branch-free: keeps the values provably wave-uniform
    const int g = wg / gsz, r = wg - g * gsz;
    const int rows = min(p.group_m, p.tiles_m - g * p.group_m);  // the last group may be short
    const int q = r / rows;
    tile_m = __builtin_amdgcn_readfirstlane(p.tile_m0 + g * p.group_m + (r - q * rows));
    tile_n = __builtin_amdgcn_readfirstlane(p.tile_n0 + q);
}

// OPT bits (tuning switches, measured in DESIGN.md): 1 = sched_group_barrier MFMA/VALU/DS interleave of the delta
// k-step, 2 = s_setprio(1) around the MFMA clusters.
// Ablation bits (timing experiments only -- results are WRONG with any of them set): 4 = no LDS-DMA inside the k loop,
// 8 = no sign expansion (constant fragment), 16 = no X-fragment ds_read inside the loop, 32 = no barrier,
// 64 = no MFMA (fragments kept alive).
template <int DT_, int BM_, int BN_, int WAVES_M_, int WAVES_N_, int NS_, bool FUSED_, bool OUT_F32_, int OPT_ = 0>
struct GemmCfg {
    static constexpr int DT = DT_, BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, NS = NS_, OPT = OPT_;
    static constexpr bool FUSED = FUSED_, OUT_F32 = OUT_F32_;
    static constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    static constexpr int A_BYTES = BM * 128, BW_BYTES = BN * 8, W_BYTES = BN * 128;
    static constexpr int STAGE_D = A_BYTES + BW_BYTES;   // delta loop ring slot
    static constexpr int STAGE_B = A_BYTES + W_BYTES;    // base loop ring slot
    static constexpr int NSB = (3 * STAGE_B <= 160 * 1024) ? 3 : 2;   // base loop ring depth (3 slots fit for BN = 128)
    static constexpr int A_PW = BM / 8 / NW;             // 1-KiB X pieces per wave per k-tile
    static constexpr int W_PW = BN / 8 / NW;             // 1-KiB W pieces per wave per k-tile
    static constexpr int BW_PIECES = BN / 32;            // 256-B sign-word pieces per k-tile ([2 x BN] words)
    static constexpr int BW_PW = BW_PIECES >= NW ? BW_PIECES / NW : 1;   // per wave; narrow tiles: every wave still issues one
                                                         // (waves >= BW_PIECES re-fetch a piece) so the vmcnt count is uniform
    static constexpr int DPW_D = A_PW + BW_PW, DPW_B = A_PW + W_PW;
    static constexpr int LUT_BYTES = 4096;               // bd_gemm_pf.h's sign-expansion table sits behind the delta ring
    static constexpr int LDS_BYTES = (FUSED && NSB * STAGE_B > NS * STAGE_D + LUT_BYTES) ? NSB * STAGE_B : NS * STAGE_D + LUT_BYTES;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
    static_assert(BW_PIECES % NW == 0 || NW % BW_PIECES == 0, "sign-word pieces vs waves");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert((NS - 2) * DPW_D <= 63, "vmcnt field");
};

// Epilogue shared by the tile kernels.  acc[i][j][4q+e] = D[n = n0 + wn*WN + 32j + 8q + 4h + e][m = m0 + wm*WM + 32i + l31]
// (swapped-operand MFMA: 4 consecutive n per register quad).  Fast path: each wave transposes its tile through a private
// LDS buffer (rows padded by 16 B: 2-way write conflicts at most) and stores whole 128-byte row segments -- 8 rows per
// store instruction instead of 32..64 scattered 8/16-byte pieces (the scattered form was store-ISSUE-bound: ~12 us of an
// 18 us fixed cost at 4096^2).  Slow path (N or ldc not a multiple of 8 elements, unaligned C): direct guarded stores.
template <class Cfg>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[Cfg::TM][Cfg::TN], char* smem, int m0, int n0,
                                              int wm, int wn, int b, int lane, int wave) {
    constexpr int DT = Cfg::DT, WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int ESZ = Cfg::OUT_F32 ? 4 : 2;
    constexpr int ROWB = WN * ESZ + 16;                 // padded LDS row
    constexpr int IPP = (TM * 32 * ROWB * Cfg::NW <= Cfg::LDS_BYTES) ? TM : (TM >= 2 && (TM / 2) * 32 * ROWB * Cfg::NW <= Cfg::LDS_BYTES ? TM / 2 : 1);
    constexpr bool STAGE_OK = IPP * 32 * ROWB * Cfg::NW <= Cfg::LDS_BYTES;
    const int h = lane >> 5, l31 = lane & 31;
    const long long c_b = (long long)b * p.sCb;
    const bool acc_mode = !Cfg::FUSED && p.accumulate;
    // fused Linear with the residual connection in its epilogue (prefill: `hidden = residual + proj(x)`): C holds the residual; the
    // Linear's output is rounded to the output type first and the sum is rounded again -- the two roundings of the separate
    // `y = proj(x); hidden = residual + y` it replaces (one pass over [M, N] less).
    const bool res_mode = Cfg::FUSED && p.accumulate;
    const float* al = (acc_mode) ? p.alpha + (long long)b * p.sAlb : nullptr;
    const bool fast = STAGE_OK && (p.N % 8 == 0) && (p.sCm % 8 == 0) && (p.sCb % 8 == 0) && (((uintptr_t)p.C & 15) == 0);

    // value transform shared by both paths
    auto xform = [&](float v, int n, long long off) -> float {
        if (acc_mode) {
            const float cin = Cfg::OUT_F32 ? ((const float*)p.C)[off] : half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]);
            return cin + al[n / p.gsz] * v;
        }
        if (!Cfg::FUSED && p.round_mode == 1) return round_through_f16(v);
        if (res_mode) {
            if constexpr (Cfg::OUT_F32) return ((const float*)p.C)[off] + v;
            else return half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]) + half_bits_to_f32<DT>(f32_to_half_bits<DT>(v));
        }
        return v;
    };

    if (fast) {
        char* buf = smem + wave * (IPP * 32 * ROWB);
        constexpr int SEG = WN * ESZ / 16;               // 16-byte pieces per row
        constexpr int RPI = 64 / SEG;                    // rows covered by one store instruction
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += IPP) {
#pragma unroll
            for (int ii = 0; ii < IPP; ++ii) {
                const int i = i0 + ii;
                const int m = m0 + wm * WM + i * 32 + l31;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = j * 32 + 8 * q + 4 * h;
                        const int n = n0 + wn * WN + nl;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[i][j][q * 4 + e];
                            if (acc_mode || (!Cfg::FUSED && p.round_mode == 1)) {
                                const bool ok = (m < p.M) && (n + e < p.N);
                                v[e] = ok ? xform(v[e], n + e, c_b + (long long)m * p.sCm + n + e) : 0.f;
                            }
                        }
                        char* dst = buf + (ii * 32 + l31) * ROWB + nl * ESZ;
                        if constexpr (Cfg::OUT_F32) {
                            *(f32x4_t*)dst = f32x4_t{v[0], v[1], v[2], v[3]};
                        } else {
                            const uint32_t h0 = f32_to_half_bits<DT>(v[0]), h1 = f32_to_half_bits<DT>(v[1]);
                            const uint32_t h2 = f32_to_half_bits<DT>(v[2]), h3 = f32_to_half_bits<DT>(v[3]);
                            *(u32x2_t*)dst = u32x2_t{h0 | (h1 << 16), h2 | (h3 << 16)};
                        }
                    }
            }
            // this wave's rows only: no block barrier needed, just LDS write->read ordering inside the wave
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r0 = 0; r0 < IPP * 32; r0 += RPI) {
                const int r = r0 + lane / SEG, sg = lane % SEG;
                const int m = m0 + wm * WM + i0 * 32 + r;
                const int n = n0 + wn * WN + sg * (16 / ESZ);
                u32x4_t val = *(const u32x4_t*)(buf + r * ROWB + sg * 16);
                if (m < p.M && n < p.N) {   // N % 8 == 0 -> a 16-byte piece is entirely in or out
                    u32x4_t* dst = (u32x4_t*)(p.C + (c_b + (long long)m * p.sCm + n) * ESZ);
                    if (res_mode) {         // coalesced: the residual piece this lane is about to overwrite
                        const u32x4_t rsd = *(const u32x4_t*)dst;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            if constexpr (Cfg::OUT_F32) {
                                val[d] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, val[d]) + __builtin_bit_cast(float, rsd[d]));
                            } else {
                                const float lo = half_bits_to_f32<DT>(val[d] & 0xffffu) + half_bits_to_f32<DT>(rsd[d] & 0xffffu);
                                const float hi = half_bits_to_f32<DT>(val[d] >> 16) + half_bits_to_f32<DT>(rsd[d] >> 16);
                                val[d] = f32_to_half_bits<DT>(lo) | (f32_to_half_bits<DT>(hi) << 16);
                            }
                        }
                    }
                    if constexpr (Cfg::OPT & 1024) *dst = val;                 // A/B: plain stores
                    else __builtin_nontemporal_store(val, dst);                // C is written once and not re-read by this kernel
                }
            }
            if (i0 + IPP < TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + i * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WN + j * 32 + 8 * q + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) continue;
                    const long long off = c_b + (long long)m * p.sCm + n + e;
                    const float v = xform(acc[i][j][q * 4 + e], n + e, off);
                    if constexpr (Cfg::OUT_F32) ((float*)p.C)[off] = v;
                    else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
                }
            }
    }
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) delta_gemm_kernel(const GemmParams p) {
    constexpr int DT = Cfg::DT, BM = Cfg::BM, BN = Cfg::BN, NS = Cfg::NS;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int A_BYTES = Cfg::A_BYTES, STAGE_D = Cfg::STAGE_D, STAGE_B = Cfg::STAGE_B;
    constexpr int A_PW = Cfg::A_PW, W_PW = Cfg::W_PW, BW_PW = Cfg::BW_PW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int h = lane >> 5, l31 = lane & 31;

    // ---- tile mapping: XCD-aware, n fastest inside an XCD's run so neighbours share the X row panel in L2
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int b = blockIdx.y;
    const int nk = p.K >> 6;

    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));   // opaque VGPR -> v_and_or_b32 gets selected

    // ---- DMA source offsets (per lane, constant over k)
    const char* a_src = p.A + ((long long)b * p.sAb + (long long)m0 * p.sAm) * 2;
    const char* p_src = (const char*)p.P + ((long long)b * p.sPb + n0) * 4;
    uint32_t a_voff[A_PW], bw_voff[BW_PW];
    uint32_t a_lds[A_PW], bw_lds[BW_PW];   // relative to the ring slot
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int rg = wave * A_PW + i;
        const int r = rg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);                 // source chunk for physical chunk lane&7
        const int rr = min(m0 + r, p.M - 1) - m0;                   // clamp rows past M (results discarded)
        a_voff[i] = (uint32_t)rr * (uint32_t)p.sAm * 2u + (uint32_t)c * 16u;
        a_lds[i] = rg * 1024;
    }
#pragma unroll
    for (int i = 0; i < BW_PW; ++i) {
        const int idx = (wave * BW_PW + i) % Cfg::BW_PIECES;
        const int hh = idx / (BN / 64), seg = idx % (BN / 64);
        const int nn = min(n0 + seg * 64 + lane, p.N - 1) - n0;     // clamp columns past N
        bw_voff[i] = (uint32_t)hh * (uint32_t)p.N * 4u + (uint32_t)nn * 4u;
        bw_lds[i] = A_BYTES + hh * BN * 4 + seg * 256;
    }

    // ---- fragment read offsets
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

        int slot_c = 0, slot_i = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (!(Cfg::OPT & 4)) wait_vmcnt<(NS - 2) * Cfg::DPW_D>();   // this wave's pieces of tile kt have landed
            if constexpr (!(Cfg::OPT & 32)) __builtin_amdgcn_s_barrier();         // ... everyone's have; everyone is done reading tile kt-1
            if constexpr (!(Cfg::OPT & 4))
                issue(min(kt + NS - 1, nk - 1), slot_i);   // refill the slot tile kt-1 used (tail re-reads the last tile: keeps the count fixed)

            const char* st = smem + slot_c * STAGE_D;
            uint32_t rlo[TN], rhi[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const uint32_t w = ~*(const uint32_t*)(st + bw_rd + j * 128);
                rlo[j] = __builtin_amdgcn_perm(w, w, 0x01000100u);
                rhi[j] = __builtin_amdgcn_perm(w, w, 0x03020302u);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4_t xf[TM], sf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (Cfg::OPT & 16) {
                        xf[i] = u32x4_t{one2, one2 + (uint32_t)i, one2, one2};
                        asm volatile("" : "+v"(xf[i]));
                    } else {
                        xf[i] = *(const u32x4_t*)(st + a_rd[s] + i * 4096);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (Cfg::OPT & 8) {
                        sf[j] = u32x4_t{rlo[j], rhi[j], one2, one2};
                        asm volatile("" : "+v"(sf[j]));
                    } else {
                        sf[j] = expand_signs8(s < 2 ? rlo[j] : rhi[j], (s & 1) * 4, one2);
                    }
                }
                if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if constexpr (Cfg::OPT & 64) {
                            asm volatile("" ::"v"(sf[j]), "v"(xf[i]));
                        } else {
                            acc[i][j] = mfma32<DT>(sf[j], xf[i], acc[i][j]);
                        }
                    }
                if constexpr (Cfg::OPT & 2) __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (Cfg::OPT & 1) {
                // per MFMA: ~2 sign-expansion VALU and (first half) one fragment ds_read in its shadow
#pragma unroll
                for (int g = 0; g < 4 * TM * TN; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, (8 * TN + TM * TN - 1) / (TM * TN), 0);  // VALU share
                    if (g % TN == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                }
            }
            slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
            slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
        }
        wait_vmcnt<0>();                // drain the tail re-reads before LDS is reused / the wave ends
    }

    // =========================== fused: scale, then base loop ===========================
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
        __builtin_amdgcn_s_barrier();   // all waves left the delta ring
#pragma unroll
        for (int t = 0; t < NSB - 1; ++t) issue_b(min(t, nk - 1), t);
        int sb_c = 0, sb_i = NSB - 1;
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<(NSB - 2) * Cfg::DPW_B>();
            __builtin_amdgcn_s_barrier();
            issue_b(min(kt + NSB - 1, nk - 1), sb_i);
            const char* st = smem + sb_c * STAGE_B;
            sb_c = (sb_c + 1 == NSB) ? 0 : sb_c + 1;
            sb_i = (sb_i + 1 == NSB) ? 0 : sb_i + 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4_t xf[TM], wf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[i] = *(const u32x4_t*)(st + a_rd[s] + i * 4096);
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[j] = *(const u32x4_t*)(st + w_rd[s] + j * 4096);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = mfma32<DT>(wf[j], xf[i], acc[i][j]);
            }
        }
        wait_vmcnt<0>();
    }

    // =========================== epilogue ===========================
    __builtin_amdgcn_s_barrier();      // every wave is out of the rings: LDS is reused as the C staging buffer
    gemm_epilogue<Cfg>(p, acc, smem, m0, n0, wm, wn, b, lane, wave);
}

}  // namespace bd
