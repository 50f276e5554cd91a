// Prefill attention, one-wave-per-SIMD form (round 6; the 8-waves-per-CU form of round 3 is bd_attn_prefill.h and keeps the short prompts).
//     O[b, s, h, :] = softmax_k( Q[b, s, h, :] . K[b, k, h / G, :] * scale  +  causal / left-padding mask ) . V[b, k, h / G, :]
// head_dim = 128, 16-bit I/O, fp32 online softmax: the caller glue between the fused q|k|v Linear and the o projection (HF SDPA over the
// left-padded tenant batch of demo/demo_backend.py:262-275; bench_model.DecoderLayer at configs[1]).
//
// What was wrong with the round-3 kernel (63 us at S = 2048, 32 heads, causal = 23 % of the matrix peak): 32 query rows per wave means every MFMA
// needs a fresh 1-KiB operand fragment from LDS, and the softmax's VALU work (as long as the MFMA work at head_dim 128) only overlapped with
// the OTHER workgroup's MFMAs -- measured: not at all (3750 cycles per 32-MFMA tile per wave with two waves per SIMD, 3000 with one).
// This kernel:
//   * one 4-wave workgroup per CU, one wave per SIMD, the whole register file: a wave owns 64 query rows = TWO 32-row blocks, every K / V
//     fragment read from LDS feeds two MFMAs (half the LDS bytes per MFMA);
//   * causal balance WITHOUT co-resident workgroups: the two blocks of a wave are MIRRORED -- block a and block NB - 1 - a of the NB = S / 32
//     blocks of the sequence -- so every wave does the same number of (32-row x 64-key) tile halves whatever a is; workgroup g holds
//     a = 4 g .. 4 g + 3 (nearly equal key ranges: the K / V tiles in LDS are shared); S / 256 workgroups per (batch, head);
//   * inside ONE wave the two blocks are STAGGERED so that VALU and matrix work overlap without a partner wave: S^T = K . Q^T of both blocks
//     (shared K fragments), softmax of the light block, then the O^T += V^T . P^T MFMAs of the light block INTERLEAVED with the softmax VALU
//     slices of the heavy block (independent instruction streams, pinned group by group with sched_barrier), then the heavy block's PV with the
//     LDS writes of tile j + 1 among them; two LDS buffers, ONE barrier per tile.  (A cross-tile pipeline -- QK^T of tile j + 1 under the softmax
//     of tile j -- was written first and needs S^T twice: with the query fragments, P and the staging registers it does not fit the 256 arch
//     VGPRs, and hipcc answered with 500 scratch spills; the staggered form needs no second S^T.)
//   * same fragment algebra as the round-3 kernel (both products transposed, a lane owns one query row per block, P never leaves the lane,
//     V^T through ds_read_b64_tr_b16), same LDS images (K rows 272 B, V rows 320 B: conflict-free).
// Variants of the tile body are compile-time: which blocks are still active for the softmax / PV of tile j (both, or the heavy one only) and
// for the QK^T of tile j + 1, and whether tile j needs element masks (a block's diagonal tile, the first tile of a left-padded sequence).
#pragma once
#include "../../../bitdelta_amd/csrc/bd_attn_prefill.h"

namespace bd {

template <int V> struct AIC { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void a_sfor(F&& f) {
    if constexpr (I < N) { f(AIC<I>{}); a_sfor<I + 1, N>(f); }
}

constexpr int PREFILL_ATTN64_BUF = 64 * 272 + 64 * 320;
constexpr int PREFILL_ATTN64_LDS = 2 * PREFILL_ATTN64_BUF;
#ifndef BD_ATTN64_Q_AGPR
#define BD_ATTN64_Q_AGPR 0
#endif
constexpr int Q_IN_AGPR = BD_ATTN64_Q_AGPR;
#ifndef BD_ATTN64_ABL
#define BD_ATTN64_ABL 0
#endif
constexpr int ATTN64_ABL = BD_ATTN64_ABL;
#ifdef BD_ATTN64_TRACE
// s_memtime stamps of workgroup 0 / wave 0 (harness builds only): [tile slot][0] step entry, [1] QK^T done, [2] softmax of the first block done,
// [3] first PV done, [4] step done, [5] after the barrier
__device__ unsigned long long g_attn64_trace[64][8];
#define BD_A64_STAMP(slot, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_attn64_trace[(slot) & 63][i] = __builtin_readcyclecounter(); } while (0)
#else
#define BD_A64_STAMP(slot, i) do { } while (0)
#endif            // timing ablations (harness only): 1 no softmax, 2 no PV, 4 no QK^T, 8 no staging       // query blocks whose fragments are pinned to AGPRs

template <int DT> struct Attn64 {
    static constexpr int HD = 128, KVB = 64, KROW = 272, VROW = 320;
    static constexpr int K_BYTES = KVB * KROW, BUF = PREFILL_ATTN64_BUF;
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;

    static __device__ __forceinline__ float other_half(float x) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
        return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi_) {
        if constexpr (DT == DT_BF16) {
            uint32_t r;      // (s_nop: a transcendental result needs one wait state before a VALU read; the hazard pass does not look inside asm)
            asm("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi_));
            return r;
        } else {
            return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi_) << 16);
        }
    }

    struct State {
        f32x16_t o[2][4];          // O^T accumulators: block h, 32-wide d block
        u32x4_t q[2][8];           // query fragments (B operands): block h, d = 16 s + 8 hi .. + 7
        float m[2], l[2];          // running max (raw score units) / running sum (this lane half's keys)
        int qrow[2];               // this lane's query row of block h
    };

    // One tile step.  AC = blocks active at tile j (bit 0 light, bit 1 heavy: 3 or 2).  kb / vb = K / V image of tile j (lane offsets applied).
    // `stage(AIC<ev>)` is called at fixed points so that the caller can place the staging of tile j + 1: ev 0 = after QK^T, 100 + g = group g of the
    // light block's PV, 200 + g = group g of the heavy block's PV.
    template <int AC, class Stage>
    static __device__ __forceinline__ void step(State& st, const char* kb, const char* vb, int kv0, int ks, int causal, float c, int hi, bool mask,
                                                Stage&& stage) {
        const float NEG_INF = -__builtin_inff();
        BD_A64_STAMP(kv0 >> 6, 0);
        f32x16_t sc[2][2];
        u32x4_t pf[2][4];
        float mc[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f}, psum[2] = {0.f, 0.f}, pmx[2][4];
        // ---------------- S^T = K . Q^T: fragment (s, t) of K feeds every active block; fragments read one step ahead ----------------
        {
            // K fragments are read TWO steps ahead (a ring of three): one step of two MFMAs is 64 cycles of matrix pipe, an LDS read returns in
            // ~130-200 -- at one step ahead every step waited for its fragment (3300 cycles per tile for 1024 cycles of MFMA, measured)
            auto kaddr = [&](int f) __attribute__((always_inline)) { return kb + (32 * (f & 1) * KROW + 32 * (f >> 1)); };
            u32x4_t kf[3];
            kf[0] = *(const u32x4_t*)kaddr(0);
            kf[1] = *(const u32x4_t*)kaddr(1);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[h][t][r] = 0.f;
            a_sfor<0, 16>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value, s = f >> 1, t = f & 1;    // the two key blocks alternate: no back-to-back dependent MFMAs
                if constexpr (f + 2 < 16) kf[(f + 2) % 3] = *(const u32x4_t*)kaddr(f + 2);
                if constexpr ((AC & 1) && !(ATTN64_ABL & 4)) sc[0][t] = mfma32<DT>(kf[f % 3], st.q[0][s], sc[0][t]);
                if constexpr ((AC & 2) && !(ATTN64_ABL & 4)) sc[1][t] = mfma32<DT>(kf[f % 3], st.q[1][s], sc[1][t]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        BD_A64_STAMP(kv0 >> 6, 1);
        stage(AIC<0>{});
        if (mask) {                   // element masks of an edge tile (a block's diagonal tile, the first tile of a left-padded sequence): rare
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!((AC >> h) & 1)) continue;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key_abs = kv0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool dead = (causal && key_abs > st.qrow[h]) || key_abs < ks;
                        sc[h][t][r] = dead ? NEG_INF : sc[h][t][r];
                    }
            }
        }
        // ---------------- softmax of block h in 22 slices: 0..3 partial maxima, 4 finalise, 5..20 exponentials, 21 the running sum ----------------
        auto slice = [&](auto hc, auto ic) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value, i = decltype(ic)::value;
            if constexpr (ATTN64_ABL & 1) {
                if constexpr (i >= 5 && i < 21) { constexpr int e = i - 5; pf[h][2 * (e >> 3) + ((e >> 2) & 1)][e & 3] = __builtin_bit_cast(uint32_t, sc[h][e >> 3][(e & 7) * 2]); }
            } else if constexpr (i < 4) {
                constexpr int t = i >> 1, r0 = 8 * (i & 1);
                const float a = fmaxf(fmaxf(sc[h][t][r0], sc[h][t][r0 + 1]), sc[h][t][r0 + 2]);
                const float b = fmaxf(fmaxf(sc[h][t][r0 + 3], sc[h][t][r0 + 4]), sc[h][t][r0 + 5]);
                pmx[h][i] = fmaxf(fmaxf(a, b), fmaxf(sc[h][t][r0 + 6], sc[h][t][r0 + 7]));
            } else if constexpr (i == 4) {
                float mx = fmaxf(fmaxf(pmx[h][0], pmx[h][1]), fmaxf(pmx[h][2], pmx[h][3]));
                mx = fmaxf(mx, other_half(mx));
                const float m_new = fmaxf(st.m[h], mx);
                const float m_safe = m_new == NEG_INF ? 0.f : m_new;             // a row with no valid key so far: every p = exp2(-inf) = 0
                mc[h] = m_safe * c;
                alpha[h] = __builtin_amdgcn_exp2f((st.m[h] - m_safe) * c);
                st.m[h] = m_new;
            } else if constexpr (i < 21) {
                constexpr int e = i - 5, t = e >> 3, kk = (e >> 2) & 1, w = e & 3;
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[h][t][8 * kk + 2 * w], c, -mc[h]));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[h][t][8 * kk + 2 * w + 1], c, -mc[h]));
                psum[h] += p0 + p1;
                pf[h][2 * t + kk][w] = pack2(p0, p1);
            } else {
                st.l[h] = st.l[h] * alpha[h] + psum[h];
            }
        };
        auto rescale = [&](auto hc) __attribute__((always_inline)) {             // rare after the first tiles: the running maximum settles
            constexpr int h = decltype(hc)::value;
            if (__any(alpha[h] != 1.f)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {                                  // one accumulator at a time, fenced (register pressure)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.o[h][dt][r] *= alpha[h];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto vfrag = [&](int g) __attribute__((always_inline)) -> u32x4_t {      // g = 4 ks4 + dt (compile-time at every call site)
            const char* a0 = vb + (16 * (g >> 2) * VROW + 64 * (g & 3));
            const v4s_t ra = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(a0));
            const v4s_t rb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(a0 + 8 * VROW));
            const u32x2_t a2 = __builtin_bit_cast(u32x2_t, ra), b2 = __builtin_bit_cast(u32x2_t, rb);
            return u32x4_t{a2.x, a2.y, b2.x, b2.y};
        };
        // O^T(h) += V^T . P^T(h): 16 MFMAs, V fragments read one step ahead; `extra(g)` rides in group g's shadow
        auto pv = [&](auto hc, auto&& extra) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            u32x4_t vf[3];                                                       // read two steps ahead
            vf[0] = vfrag(0);
            vf[1] = vfrag(1);
            a_sfor<0, 16>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g + 2 < 16) vf[(g + 2) % 3] = vfrag(g + 2);
                if constexpr (!(ATTN64_ABL & 2)) st.o[h][g & 3] = mfma32<DT>(vf[g % 3], pf[h][g >> 2], st.o[h][g & 3]);
                else if constexpr (g < 4) st.o[h][g & 3][0] += __builtin_bit_cast(float, vf[g % 3][0] ^ pf[h][g >> 2][0]);
                extra(gc);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto ev1 = [&](auto gc) __attribute__((always_inline)) { stage(AIC<200 + decltype(gc)::value>{}); };
        if constexpr (AC == 3) {
            a_sfor<0, 22>([&](auto ic) __attribute__((always_inline)) { slice(AIC<0>{}, ic); });
            rescale(AIC<0>{});
            BD_A64_STAMP(kv0 >> 6, 2);
            pv(AIC<0>{}, [&](auto gc) __attribute__((always_inline)) {           // the heavy block's softmax under the light block's MFMAs
                constexpr int g = decltype(gc)::value;
                a_sfor<g * 22 / 16, (g + 1) * 22 / 16>([&](auto ic) __attribute__((always_inline)) { slice(AIC<1>{}, ic); });
                stage(AIC<100 + g>{});
            });
            BD_A64_STAMP(kv0 >> 6, 3);
            rescale(AIC<1>{});
            pv(AIC<1>{}, ev1);
            BD_A64_STAMP(kv0 >> 6, 4);
        } else {
            a_sfor<0, 22>([&](auto ic) __attribute__((always_inline)) { slice(AIC<1>{}, ic); });
            rescale(AIC<1>{});
            pv(AIC<1>{}, ev1);
        }
    }
};

template <int DT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) prefill_attn64_kernel(const PrefillAttnParams p) {
    using A = Attn64<DT>;
    constexpr int HD = A::HD, KVB = A::KVB, KROW = A::KROW, VROW = A::VROW, K_BYTES = A::K_BYTES, BUF = A::BUF;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 3 * BUF = 111 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int NB = p.S >> 5, np = NB >> 1;                           // 32-row blocks of the sequence (S % 64 == 0: even), mirrored pairs
    const int wgph = (np + 3) >> 2;                                  // workgroups per (batch, head)
    const int HB = p.H * p.B;
    // block -> (head, pair group): the workgroups of consecutive heads (a GQA group shares its K / V) on one XCD (block id % 8)
    int hb, g;
    if ((HB & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        hb = xcd * (HB >> 3) + slot / wgph;
        g = slot % wgph;
    } else {
        hb = (int)blockIdx.x / wgph;
        g = (int)blockIdx.x % wgph;
    }
    const int h = hb % p.H, b = hb / p.H;
    const int kvh = h / (p.H / p.KVH);
    const int a = 4 * g + wave;
    const bool active = a < np;
    const int blk[2] = {active ? a : 0, active ? NB - 1 - a : 0};
    const int ks = p.kv_start ? min(max(p.kv_start[b], 0), p.S) : 0;
    const unsigned short* qp = p.q + (long long)b * p.sqb + (long long)h * HD;
    const unsigned short* kp = p.k + (long long)b * p.skb + (long long)kvh * HD;
    const unsigned short* vp = p.v + (long long)b * p.svb + (long long)kvh * HD;
    const float NEG_INF = -__builtin_inff();

    typename A::State st;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        st.qrow[hh] = blk[hh] * 32 + l31;
        st.m[hh] = NEG_INF;
        st.l[hh] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            st.q[hh][s] = *(const u32x4_t*)(qp + (long long)st.qrow[hh] * p.sqs + 16 * s + 8 * hi);
            // The query fragments are MFMA B operands only, for the whole kernel: pin them into the ACCUMULATOR half of the register file (MFMA reads
            // A / B operands from either half on gfx950).  Left to the allocator they compete with S^T, P and the staging registers for the 256
            // arch VGPRs, lose, and are reloaded from scratch in front of every MFMA (read in the ISA).
            if (Q_IN_AGPR > hh) asm volatile("" : "+a"(st.q[hh][s]));
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[hh][dt][r] = 0.f;
    }
    // tile limits (inclusive): of this wave's blocks, and of the workgroup (= its wave 0's heavy block)
    const int last_tile = (p.S >> 6) - 1;
    const int lim[2] = {p.causal ? blk[0] >> 1 : last_tile, p.causal ? blk[1] >> 1 : last_tile};
    const int j_lo = ks / KVB, j_hi = p.causal ? (NB - 1 - 4 * g) >> 1 : last_tile;

    // register staging: thread -> 16-byte chunk (tid & 15) of keys (tid >> 4) + 16 i.  K and V have their own four registers and, in the dual-block
    // step, DISJOINT lifetimes (K: loaded after QK^T, written under the light block's PV; V: loaded there, written under the heavy block's PV), so
    // the staging costs 16 registers, not 32, where the arch-VGPR file is full.
    // Raw BUFFER loads: descriptor (SGPRs) + ONE per-thread 32-bit offset + a scalar offset per load.  Written with global loads, hipcc kept eight
    // 64-bit per-thread addresses alive across the tile loop and spilled them -- and a scratch reload's s_waitcnt vmcnt(0) in the loop also waits
    // for the tile loads just issued: a full memory round trip per tile, twice (2300 of the 9400 cycles per tile of the first version).
    // For the same reason every lane-derived constant is RECOMPUTED per tile behind an opaque copy of the thread id (LaneK): hoisted to kernel
    // entry they are live across the whole loop nest, and the ones that lose the allocation are reloaded from scratch inside it.
    u32x4_t kst[4], vst[4];
    const uint32_t k_step = 16u * (uint32_t)p.sks * 2u, v_step = 16u * (uint32_t)p.svs * 2u;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(kp), (short)0, (int)((uint32_t)p.S * (uint32_t)p.sks * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(vp), (short)0, (int)((uint32_t)p.S * (uint32_t)p.svs * 2u), 0x00020000);
    struct LaneK { uint32_t k_lane, v_lane, k_thr, v_thr, wk, wv; int hi; };
    auto lane_consts = [&]() __attribute__((always_inline)) -> LaneK {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                                  // opaque: not hoisted out of the tile loop
        const int ln = t & 63, l31_ = ln & 31, hi_ = ln >> 5, t16 = ln & 15, G = ln >> 4, key = t >> 4, ch = t & 15;
        LaneK c;
        c.hi = hi_;
        c.k_lane = (uint32_t)(l31_ * KROW + 16 * hi_);               // key row l31 (+ 32 t), chunk 2 s + hi
        // transposing-read lane constants: group G (G >> 1 = hi, G & 1 = 16-column half), source lane t16: key row t16 >> 2, quad t16 & 3
        c.v_lane = (uint32_t)((4 * hi_ + (t16 >> 2)) * VROW + (16 * (G & 1) + 4 * (t16 & 3)) * 2);
        c.k_thr = (uint32_t)key * (uint32_t)p.sks * 2u + (uint32_t)ch * 16u;
        c.v_thr = (uint32_t)key * (uint32_t)p.svs * 2u + (uint32_t)ch * 16u;
        c.wk = (uint32_t)(key * KROW + ch * 16);
        c.wv = (uint32_t)(K_BYTES + key * VROW + ch * 16);
        return c;
    };
    auto gload_k = [&](const LaneK& c, int j) __attribute__((always_inline)) {
        const uint32_t tile = (uint32_t)j * (uint32_t)KVB * (uint32_t)p.sks * 2u;    // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) kst[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)c.k_thr, (int)(tile + (uint32_t)i * k_step), 0));
    };
    auto gload_v = [&](const LaneK& c, int j) __attribute__((always_inline)) {
        const uint32_t tile = (uint32_t)j * (uint32_t)KVB * (uint32_t)p.svs * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) vst[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)c.v_thr, (int)(tile + (uint32_t)i * v_step), 0));
    };
    auto lwrite_k = [&](const LaneK& c, char* base, int i) __attribute__((always_inline)) { *(u32x4_t*)(base + c.wk + 16 * i * KROW) = kst[i]; };
    auto lwrite_v = [&](const LaneK& c, char* base, int i) __attribute__((always_inline)) { *(u32x4_t*)(base + c.wv + 16 * i * VROW) = vst[i]; };
    auto stage_all = [&](const LaneK& c, int j, char* base) __attribute__((always_inline)) {
        gload_k(c, j);
        gload_v(c, j);
#pragma unroll
        for (int i = 0; i < 4; ++i) lwrite_k(c, base, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) lwrite_v(c, base, i);
    };

    auto needs_mask = [&](int t) -> bool { return (p.causal && (t == lim[0] || t == lim[1])) || ks > t * KVB; };

    if (j_lo <= j_hi) {
        stage_all(lane_consts(), j_lo, lds);
        __builtin_amdgcn_s_waitcnt(0x0f70);                          // vmcnt(0): nothing pending at the loop entry (see bd_attn_prefill.h)
        __syncthreads();
        // Three consecutive loops instead of one loop with a variant switch: tiles where both blocks are active, then the heavy block alone, then
        // (waves that finish before their workgroup) staging only.  With the two step variants inlined into ONE loop body the register allocator
        // spilled 126 VGPRs at their merge; each loop now holds a single variant (16 spills in a one-variant build).
        auto tile = [&](auto acc, int j) __attribute__((always_inline)) {
            constexpr int AC = decltype(acc)::value;
            const int bc = (j - j_lo) & 1;
            const bool more = !(ATTN64_ABL & 8) && j + 1 <= j_hi;
            char* wbase = lds + (bc ^ 1) * BUF;
            const LaneK c = lane_consts();
            if constexpr (AC != 0) {
                // The light block's query fragments are pinned to arch VGPRs once per tile (an empty asm with a "v" constraint).  The accumulator half of
                // the register file is exactly full (O 128 + S^T 64 + the heavy block's 64 query registers, which hipcc parks there by itself); without
                // the pin it also tried to park these and reloaded three of them from SCRATCH in front of their MFMAs -- s_waitcnt vmcnt(0) in the
                // matrix stream, every tile (read in the ISA).
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) asm volatile("" : "+v"(st.q[0][s_]));
                const char* kb = lds + (uint32_t)(bc * BUF) + c.k_lane;
                const char* vb = lds + (uint32_t)(bc * BUF) + c.v_lane + K_BYTES;
                auto stage = [&](auto evc) __attribute__((always_inline)) {
                    constexpr int ev = decltype(evc)::value;
                    if (!more) return;
                    if constexpr (AC == 3) {
                        if constexpr (ev == 0) gload_k(c, j + 1);
                        else if constexpr (ev >= 108 && ev < 112) lwrite_k(c, wbase, ev - 108);
                        else if constexpr (ev == 112) gload_v(c, j + 1);
                        else if constexpr (ev >= 212 && ev < 216) lwrite_v(c, wbase, ev - 212);
                    } else {
                        if constexpr (ev == 0) { gload_k(c, j + 1); gload_v(c, j + 1); }
                        else if constexpr (ev >= 208 && ev < 212) lwrite_k(c, wbase, ev - 208);
                        else if constexpr (ev >= 212 && ev < 216) lwrite_v(c, wbase, ev - 212);
                    }
                };
                A::template step<AC>(st, kb, vb, j * KVB, ks, p.causal, p.c, c.hi, needs_mask(j), stage);
            } else if (more) {                                        // an idle wave still stages its share of the next tile
                stage_all(c, j + 1, wbase);
            }
            __syncthreads();
            BD_A64_STAMP(j, 5);
        };
        const int e3 = active ? min(lim[0], j_hi) : j_lo - 1, e2 = active ? min(lim[1], j_hi) : j_lo - 1;
        int j = j_lo;
        for (; j <= e3; ++j) tile(AIC<3>{}, j);
        for (; j <= e2; ++j) tile(AIC<2>{}, j);
        for (; j <= j_hi; ++j) tile(AIC<0>{}, j);
    }

    // ---- epilogue: 1 / l per lane, each block's [32 rows][128] image through LDS (row pitch 272 B), whole 256-byte rows out
    if (!active) return;                                             // (no barrier below)
    constexpr int OROW = 272;
    unsigned short* op = p.o + (long long)b * p.sob + (long long)h * HD;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const float l_tot = st.l[hh] + A::other_half(st.l[hh]);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        char* ob = lds + (wave * 2 + hh) * (32 * OROW);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = 32 * dt + 8 * rq + 4 * hi;
                *(u32x2_t*)(ob + l31 * OROW + d * 2) = u32x2_t{A::pack2(st.o[hh][dt][4 * rq] * inv, st.o[hh][dt][4 * rq + 1] * inv),
                                                              A::pack2(st.o[hh][dt][4 * rq + 2] * inv, st.o[hh][dt][4 * rq + 3] * inv)};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this wave's own image: no block barrier
        const int row0 = blk[hh] * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (lane >> 4), ch = lane & 15;
            const u32x4_t val = *(const u32x4_t*)(ob + row * OROW + ch * 16);
            *(u32x4_t*)(op + (long long)(row0 + row) * p.sos + ch * 8) = val;
        }
    }
}

}  // namespace bd
