// Prefill attention of the serving loop / the prefill step (caller glue of the Linear hot path, like bd_serving.h):
//     O[b, s, h, :] = softmax_k( Q[b, s, h, :] . K[b, k, h / G, :] * scale  +  causal / left-padding mask ) . V[b, k, h / G, :]
// head_dim = 128, 16-bit I/O, fp32 online softmax -- what F.scaled_dot_product_attention(is_causal=True) computes for the tenant batch of
// demo/demo_backend.py's prefill (HF attention with the left-padded attention_mask, :262-275), without materialising [S, S] scores.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); wave w owns 32 rows.  K / V tiles of 64 keys are register-staged
// (global -> VGPR while the previous tile computes -> LDS), double-buffered, one barrier per tile; two workgroups per CU (74 KiB of LDS
// each, 2 waves per SIMD) so that one workgroup's softmax VALU work runs under the other's MFMAs.
//   S^T = K . Q^T   (v_mfma 32x32x16, A = K fragment, B = Q fragment): D column = query (lane & 31), D rows = keys -> a lane holds 16 of
//                   the 32 scores of ITS query row per 32-key tile; the row maximum is 31 v_max + one exchange with lane ^ 32.
//   O^T = V^T . P^T (A = V^T fragment through ds_read_b64_tr_b16, B = P fragment): again D column = query, so the per-row rescale
//                   exp2(m_old - m_new) and the final 1 / l are per-LANE scalars, and the P fragment is exactly the lane's own
//                   exponentiated scores packed in register order: the MFMA k index (hi, e) <-> key 16 kk + 8 (e >> 2) + 4 hi + (e & 3)
//                   is a permutation the V fragment reads through (two 4-key transposing reads), so no cross-lane traffic for P at all.
// LDS images: K [64 keys][272 B]: rows 16 bytes apart modulo the bank span, so the b128 fragment reads of 16 keys are conflict-free and
// every read address is one lane constant + an immediate; V [64 keys][320 B]: rows 64 bytes apart modulo the 256-byte bank span, so the 16 quads (4 keys x 32 bytes) of a
// transposing read group and its neighbour group cover all banks once.
// Causal: tiles past the diagonal of the workgroup are never loaded, a wave skips the tiles past ITS diagonal, and only tiles that
// straddle it are masked.  Heavy (late) query blocks are dispatched first.
#pragma once
#include "bd_common.h"

namespace bd {

constexpr int PREFILL_ATTN_LDS = 2 * (64 * 272 + 64 * 320);

struct PrefillAttnParams {
    const unsigned short* q;       // [B, S, H, 128] through strides (elements): batch, sequence; head h at + 128 h
    const unsigned short* k;       // [B, S, KVH, 128]
    const unsigned short* v;
    unsigned short* o;             // [B, S, H, 128]
    long long sqb, sqs, skb, sks, svb, svs, sob, sos;
    const int* kv_start;           // [B] first valid key of each sequence (left padding) or null
    int B, S, H, KVH, nqb;         // S % 64 == 0; nqb = ceil(S / 128)
    float c;                       // softmax scale * log2(e)
    int causal;
};

template <int DT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) prefill_attn_kernel(const PrefillAttnParams p) {
    constexpr int HD = 128, QW = 32, QB = 128, KVB = 64;
    constexpr int KROW = 272, VROW = 320;
    constexpr int K_BYTES = KVB * KROW, V_BYTES = KVB * VROW, BUF = K_BYTES + V_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 * BUF = 72 KiB (dynamic: above the static limit)
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int per = p.H * p.B;
    // causal: query block i costs i + 1 tile pairs.  The heavy half goes out first, heaviest first, then the light half lightest first, so
    // that the two workgroups a CU holds (dispatch fills the CUs round-robin twice) add up to the same work everywhere
    const int rank = (int)blockIdx.x / per, nheavy = (p.nqb + 1) / 2;
    const int qb = rank < nheavy ? p.nqb - 1 - rank : rank - nheavy;
    const int hb = (int)blockIdx.x % per;
    const int h = hb % p.H, b = hb / p.H;
    const int kvh = h / (p.H / p.KVH);
    const int Q0 = qb * QB, qw0 = Q0 + wave * QW;
    const int ks = p.kv_start ? min(max(p.kv_start[b], 0), p.S) : 0;      // device data the host cannot validate: clamp into [0, S]
    const unsigned short* qp = p.q + (long long)b * p.sqb + (long long)h * HD;
    const unsigned short* kp = p.k + (long long)b * p.skb + (long long)kvh * HD;
    const unsigned short* vp = p.v + (long long)b * p.svb + (long long)kvh * HD;
    const float NEG_INF = -__builtin_inff();

    // this lane's query row as 8 B-operand fragments (d = 16 s + 8 hi .. + 7)
    u32x4_t qf[8];
    {
        const int qrow = min(qw0 + l31, p.S - 1);
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *(const u32x4_t*)(qp + (long long)qrow * p.sqs + 16 * s + 8 * hi);
    }
    const int last_q = min(Q0 + QB, p.S) - 1;
    const int j_lo = ks / KVB, j_hi = (p.causal ? last_q : p.S - 1) / KVB;        // inclusive tile range of the workgroup

    // register staging: thread -> 16-byte chunk (tid & 15) of keys (tid >> 4) + 16 i
    u32x4_t kst[4], vst[4];
    const int ld_key = tid >> 4, ld_ch = tid & 15;
    auto gload = [&](int j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long key = (long long)j * KVB + ld_key + 16 * i;
            kst[i] = *(const u32x4_t*)(kp + key * p.sks + ld_ch * 8);
            vst[i] = *(const u32x4_t*)(vp + key * p.svs + ld_ch * 8);
        }
    };
    auto lwrite = [&](int buf) {
        char* base = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = ld_key + 16 * i;
            *(u32x4_t*)(base + key * KROW + ld_ch * 16) = kst[i];
            *(u32x4_t*)(base + K_BYTES + key * VROW + ld_ch * 16) = vst[i];
        }
    };

    f32x16_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;

    // transposing-read lane constants: group G = lane >> 4 (G >> 1 = hi, G & 1 = 16-column half), source lane t16: key row t16 >> 2, quad t16 & 3
    const int t16 = lane & 15, G = lane >> 4;
    const uint32_t v_lane = (uint32_t)((4 * hi + (t16 >> 2)) * VROW + (16 * (G & 1) + 4 * (t16 & 3)) * 2);
    const uint32_t k_lane = (uint32_t)(l31 * KROW + 16 * hi);         // key row l31 (+ 32 t), chunk 2 s + hi

    // value of lane ^ 32: v_permlane32_swap exchanges the upper half of one register with the lower half of the other (no LDS round trip)
    auto other_half = [](float x) -> float {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
        return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
    };
    auto pack2 = [&](float lo, float hi_) -> uint32_t {
        if constexpr (DT == DT_BF16) {
            // s_nop: the operands come straight from v_exp_f32, and a transcendental result needs one wait state before a VALU
            // instruction reads it -- the compiler's hazard pass does not look inside inline assembly
            uint32_t r;
            asm("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi_));
            return r;
        } else {
            return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi_) << 16);
        }
    };

    auto compute = [&](int buf, int j) {
        const int kv0 = j * KVB;
        if (p.causal && kv0 > qw0 + QW - 1) return;                        // past this wave's diagonal (wave-uniform)
        const char* kb = lds + (uint32_t)(buf * BUF) + k_lane;             // + immediates: no per-read address arithmetic
        const char* vb = lds + (uint32_t)(buf * BUF + K_BYTES) + v_lane;
        f32x16_t sacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
        // the two 32-key accumulators alternate: a dependent MFMA chain on one accumulator leaves the matrix pipe idle between links
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4_t kf = *(const u32x4_t*)(kb + (32 * t * KROW + 32 * s));
                sacc[t] = mfma32<DT>(kf, qf[s], sacc[t]);
            }
        // masks: keys past the query (causal) or before the first valid key (left padding); only on tiles that straddle either edge
        const bool edge = (p.causal && kv0 + KVB - 1 > qw0) || ks > kv0;
        if (edge) {
            const int q_abs = qw0 + l31;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key_abs = kv0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if ((p.causal && key_abs > q_abs) || key_abs < ks) sacc[t][r] = NEG_INF;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[t][r]);
        mx = fmaxf(mx, other_half(mx));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == NEG_INF ? 0.f : m_new;               // a row with no valid key so far: every p = exp2(-inf) = 0
        const float mc = m_safe * p.c;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * p.c);
        m_run = m_new;
        float psum = 0.f;
        u32x4_t pf[4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][8 * kk + 2 * w], p.c, -mc));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][8 * kk + 2 * w + 1], p.c, -mc));
                    psum += p0 + p1;
                    pf[2 * t + kk][w] = pack2(p0, p1);
                }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.f)) {                                         // the running maximum settles after a few tiles
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        // O^T += V^T . P^T : step ks4 = (t, kk) covers keys 32 t + 16 kk + {4 hi + 0..3, 8 + 4 hi + 0..3} of this lane half
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* a0 = vb + (16 * ks4 * VROW + 64 * dt);
                const v4s_t ra = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(a0));
                const v4s_t rb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(a0 + 8 * VROW));
                const u32x2_t a2 = __builtin_bit_cast(u32x2_t, ra), b2 = __builtin_bit_cast(u32x2_t, rb);
                oacc[dt] = mfma32<DT>(u32x4_t{a2.x, a2.y, b2.x, b2.y}, pf[ks4], oacc[dt]);
            }
        }
    };

    if (j_lo <= j_hi) {
        gload(j_lo);
        lwrite(0);
        // every load so far (the query fragments too) has landed BEFORE the loop: left pending, the compiler's wait-count pass carries them
        // across the back edge and makes each QK^T MFMA of every tile wait for the loads of the NEXT tile issued just above it
        __builtin_amdgcn_s_waitcnt(0x0f70);                                // vmcnt(0)
        __syncthreads();
        for (int j = j_lo; j <= j_hi; ++j) {
            const int cur = (j - j_lo) & 1;
            if (j < j_hi) gload(j + 1);
            compute(cur, j);
            if (j < j_hi) lwrite(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: 1 / l per lane, this wave's [32 rows][128] image through LDS (row pitch 272 B), whole 256-byte rows out
    const float l_tot = l_run + other_half(l_run);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    constexpr int OROW = 272;
    char* ob = lds + wave * (QW * OROW);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int d = 32 * dt + 8 * rq + 4 * hi;
            *(u32x2_t*)(ob + l31 * OROW + d * 2) = u32x2_t{pack2(oacc[dt][4 * rq] * inv, oacc[dt][4 * rq + 1] * inv),
                                                          pack2(oacc[dt][4 * rq + 2] * inv, oacc[dt][4 * rq + 3] * inv)};
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // this wave's own image: no block barrier
    unsigned short* op = p.o + (long long)b * p.sob + (long long)h * HD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4), ch = lane & 15;
        const u32x4_t val = *(const u32x4_t*)(ob + row * OROW + ch * 16);
        if (qw0 + row < p.S) *(u32x4_t*)(op + (long long)(qw0 + row) * p.sos + ch * 8) = val;
    }
}

}  // namespace bd
