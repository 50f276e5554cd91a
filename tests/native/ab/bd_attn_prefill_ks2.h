// Round 6 A/B loser, kept for reproduction (profiles/r06_prefill_attention.txt): the KEY-SPLIT form of the prefill attention kernel.
// Included by tests/native/attn_bench.hip under -DKS2; not part of the shipped library.
#pragma once
#include "../../../bitdelta_amd/csrc/bd_attn_prefill.h"

namespace bd {

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Key-split form (round 6): ONE 8-wave workgroup per CU; waves 0-3 and 4-7 own the SAME four 32-row query slices but the lower / upper 32 keys of
// every 64-key tile.  Why: the launch is bound by its LONGEST workgroup -- the last query block of a causal 2048-token sequence walks 32 tiles one
// after the other at ~3000 cycles per tile per wave, and its co-resident partner (a light block: the two workgroups of a CU were paired heavy + light)
// is gone after 2 tiles, so the heavy one runs alone at single-wave efficiency for 30 of them (63 us measured against ~25 us of matrix + VALU work
// per CU).  Splitting the KEYS of a tile over two waves of the same workgroup halves the serial chain per tile (8 + 8 MFMAs and 16 scores per lane
// per wave instead of 16 + 16 and 32), the two waves of a SIMD overlap each other's softmax and MFMAs for the WHOLE walk, and with one workgroup per
// CU the second half of the grid (the light blocks) is dispatched as the heavy ones retire: longest-first scheduling instead of a fixed pairing.
// Each wave keeps its own online-softmax state over its half of the keys; the two halves meet ONCE, at the end, through LDS (standard split-KV
// merge: O = O0 e^(m0 - m) + O1 e^(m1 - m), l likewise).  Same fragment algebra, LDS images and masks as prefill_attn_kernel above.
template <int DT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) prefill_attn_ks2_kernel(const PrefillAttnParams p) {
    constexpr int HD = 128, QW = 32, QB = 128, KVB = 64;
    constexpr int KROW = 272, VROW = 320;
    constexpr int K_BYTES = KVB * KROW, V_BYTES = KVB * VROW, BUF = K_BYTES + V_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 * BUF = 72 KiB
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int rw = wave & 3, kh = wave >> 2;                          // query slice, key half
    const int per = p.H * p.B;
    const int qb = p.nqb - 1 - (int)blockIdx.x / per;                 // heaviest query blocks first; the light ones fill in as CUs free up
    const int hb = (int)blockIdx.x % per;
    const int h = hb % p.H, b = hb / p.H;
    const int kvh = h / (p.H / p.KVH);
    const int Q0 = qb * QB, qw0 = Q0 + rw * QW;
    const int ks = p.kv_start ? min(max(p.kv_start[b], 0), p.S) : 0;
    const unsigned short* qp = p.q + (long long)b * p.sqb + (long long)h * HD;
    const unsigned short* kp = p.k + (long long)b * p.skb + (long long)kvh * HD;
    const unsigned short* vp = p.v + (long long)b * p.svb + (long long)kvh * HD;
    const float NEG_INF = -__builtin_inff();

    u32x4_t qf[8];
    {
        const int qrow = min(qw0 + l31, p.S - 1);
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *(const u32x4_t*)(qp + (long long)qrow * p.sqs + 16 * s + 8 * hi);
    }
    const int last_q = min(Q0 + QB, p.S) - 1;
    const int j_lo = ks / KVB, j_hi = (p.causal ? last_q : p.S - 1) / KVB;

    // register staging: thread -> 16-byte chunk (tid & 15) of keys (tid >> 4) + 32 i
    u32x4_t kst[2], vst[2];
    const int ld_key = tid >> 4, ld_ch = tid & 15;
    auto gload = [&](int j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long key = (long long)j * KVB + ld_key + 32 * i;
            kst[i] = *(const u32x4_t*)(kp + key * p.sks + ld_ch * 8);
            vst[i] = *(const u32x4_t*)(vp + key * p.svs + ld_ch * 8);
        }
    };
    auto lwrite = [&](int buf) {
        char* base = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = ld_key + 32 * i;
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

    const int t16 = lane & 15, G = lane >> 4;
    const uint32_t v_lane = (uint32_t)((4 * hi + (t16 >> 2)) * VROW + (16 * (G & 1) + 4 * (t16 & 3)) * 2) + (uint32_t)(32 * kh * VROW);
    const uint32_t k_lane = (uint32_t)((l31 + 32 * kh) * KROW + 16 * hi);         // key row 32 kh + l31, chunk 2 s + hi

    auto other_half = [](float x) -> float {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
        return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
    };
    auto pack2 = [&](float lo, float hi_) -> uint32_t {
        if constexpr (DT == DT_BF16) {
            uint32_t r;
            asm("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi_));
            return r;
        } else {
            return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi_) << 16);
        }
    };

    auto compute = [&](int buf, int j) {
        const int kv0 = j * KVB + 32 * kh;                                 // this wave's 32 keys of the tile
        if (p.causal && kv0 > qw0 + QW - 1) return;                        // past this wave's diagonal (wave-uniform)
        if (kv0 + 31 < ks) return;                                         // entirely left padding
        const char* kb = lds + (uint32_t)(buf * BUF) + k_lane;
        const char* vb = lds + (uint32_t)(buf * BUF + K_BYTES) + v_lane;
        f32x16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        // (one accumulator: the eight MFMAs form a dependent chain, but the partner wave of the SIMD fills the gaps)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const u32x4_t kf = *(const u32x4_t*)(kb + 32 * s);
            sacc = mfma32<DT>(kf, qf[s], sacc);
        }
        const bool edge = (p.causal && kv0 + 31 > qw0) || ks > kv0;
        if (edge) {
            const int q_abs = qw0 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key_abs = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if ((p.causal && key_abs > q_abs) || key_abs < ks) sacc[r] = NEG_INF;
            }
        }
        float mx = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
        mx = fmaxf(mx, other_half(mx));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == NEG_INF ? 0.f : m_new;
        const float mc = m_safe * p.c;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * p.c);
        m_run = m_new;
        float psum = 0.f;
        u32x4_t pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[8 * kk + 2 * w], p.c, -mc));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[8 * kk + 2 * w + 1], p.c, -mc));
                psum += p0 + p1;
                pf[kk][w] = pack2(p0, p1);
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* a0 = vb + (16 * kk * VROW + 64 * dt);
                const v4s_t ra = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(a0));
                const v4s_t rb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(a0 + 8 * VROW));
                const u32x2_t a2 = __builtin_bit_cast(u32x2_t, ra), b2 = __builtin_bit_cast(u32x2_t, rb);
                oacc[dt] = mfma32<DT>(u32x4_t{a2.x, a2.y, b2.x, b2.y}, pf[kk], oacc[dt]);
            }
        }
    };

    if (j_lo <= j_hi) {
        gload(j_lo);
        lwrite(0);
        __builtin_amdgcn_s_waitcnt(0x0f70);                                // vmcnt(0): see prefill_attn_kernel
        __syncthreads();
        for (int j = j_lo; j <= j_hi; ++j) {
            const int cur = (j - j_lo) & 1;
            if (j < j_hi) gload(j + 1);
            compute(cur, j);
            if (j < j_hi) lwrite(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- merge of the two key halves (once per workgroup): waves 4-7 hand (m, l, O) to waves 0-3 through LDS
    float l_tot = l_run + other_half(l_run);
    float* mo = (float*)lds;                                               // [4 slices][4 dt][16 r][64 lanes] + [4][2][64]: 66 KB <= 72 KB
    float* ml = mo + 4 * 4 * 16 * 64;
    if (kh == 1) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mo[((rw * 4 + dt) * 16 + r) * 64 + lane] = oacc[dt][r];
        ml[(rw * 2 + 0) * 64 + lane] = m_run;
        ml[(rw * 2 + 1) * 64 + lane] = l_tot;
    }
    __syncthreads();
    if (kh == 0) {
        const float m1 = ml[(rw * 2 + 0) * 64 + lane], l1 = ml[(rw * 2 + 1) * 64 + lane];
        const float m = fmaxf(m_run, m1);
        const float ms = m == NEG_INF ? 0.f : m;
        const float a0 = __builtin_amdgcn_exp2f((m_run - ms) * p.c), a1 = __builtin_amdgcn_exp2f((m1 - ms) * p.c);
        l_tot = l_tot * a0 + l1 * a1;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * a0 + mo[((rw * 4 + dt) * 16 + r) * 64 + lane] * a1;
    }
    __syncthreads();                                                       // the merge area is reused by the output images below
    if (kh == 1) return;

    // ---- epilogue (waves 0-3): 1 / l per lane, this wave's [32 rows][128] image through LDS (row pitch 272 B), whole 256-byte rows out
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    constexpr int OROW = 272;
    char* ob = lds + rw * (QW * OROW);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int d = 32 * dt + 8 * rq + 4 * hi;
            *(u32x2_t*)(ob + l31 * OROW + d * 2) = u32x2_t{pack2(oacc[dt][4 * rq] * inv, oacc[dt][4 * rq + 1] * inv),
                                                          pack2(oacc[dt][4 * rq + 2] * inv, oacc[dt][4 * rq + 3] * inv)};
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned short* op = p.o + (long long)b * p.sob + (long long)h * HD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4), ch = lane & 15;
        const u32x4_t val = *(const u32x4_t*)(ob + row * OROW + ch * 16);
        if (qw0 + row < p.S) *(u32x4_t*)(op + (long long)(qw0 + row) * p.sos + ch * 8) = val;
    }
}


}  // namespace bd
