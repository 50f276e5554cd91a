// Decode step, persistent form: the Linears of a decoder layer that follow each other without an attention in between --
//     [o + residual] -> [RMSNorm -> gate|up -> SwiGLU] -> [down + residual] -> [RMSNorm -> q|k|v of the NEXT layer]
// -- as ONE launch of one block per CU, with a grid barrier between the phases.  (Reference call sites: the four
// DiffCompressModule.forward calls of a decoder layer, demo/demo_backend.py:93-98, and the HF glue between them.)
//
// Why (profiles/r02_decode_step.txt): the hipGraph replay of the step is the SUM of its kernel durations, and every one of the 161
// launches carries ~4-5 us in which HBM does nothing -- dispatch, table build, first-stage latency at one end; last-tile reduce,
// drain and the kernel boundary at the other.  A grid barrier alone costs about as much as it saves.  It pays only if the weight
// stream does not stop at the barrier: while a block waits for the other blocks' outputs, the first NS stages of the NEXT phase's
// weights must already be in flight.
//
// The one hardware fact that shapes this file: on gfx9-family parts vector loads and stores share ONE in-order counter (vmcnt).
// A wave that holds prefetched weight stages cannot learn that its output stores have completed, and cannot read the barrier
// counter, without first waiting for every prefetched stage to land -- which serialises exactly what should overlap.  So the four
// waves of a block are not symmetric at a phase boundary:
//   * waves 0-2 turn the run-ahead loads of their LAST round (which the single-phase kernel aims out of range) at the next phase's
//     first NS stages, finish, and sleep in s_barrier (a hardware barrier: no counter involved);
//   * wave 3 -- the OWNER -- reduces and stores EVERY tile of the block (so one s_waitcnt vmcnt(0) of its own covers all of the
//     block's outputs), does not run ahead, and after its last store: waits for the stores, THEN issues its own first NS stages of
//     the next phase, arrives at the grid barrier (a store to its own flag word), and polls.  Its first poll result returns when its
//     prefetch has landed, which is about when the barrier completes; then it joins the s_barrier and the block goes on.
// Outputs are written with agent-scope relaxed atomics (they go to the coherence point; the XCDs' L2s are not coherent for plain
// stores).  Activations are READ with the default cache policy: a plain phase re-reads its fragments in every stage of every wave
// and a norm phase's 96 KB of rows are wanted by all 32 blocks of an XCD, so both must be served by L2.  That is safe because each
// such buffer is first touched by this launch after the barrier behind its producer, and no 128-byte line of it is written from
// two XCDs (host-checked for the one buffer whose block ranges are not line-aligned by construction).  The residual stream
// ping-pongs between buffers (out = res + ...), so no phase reads a line that an earlier phase of the same launch already pulled
// into its L2.  Same arithmetic, same order as the separate launches (gemv_stream_kernel XL / EPI): results are bit-identical to
// them (tests/test_gpu_serving.py).
//
// STATUS (profiles/r02_decode_step.txt, r02_decode_chain_timeline.txt): correct and deterministic, but 157 us per Mistral layer
// against 144.5 us for the four launches it replaces -- the loops run at the streaming ceiling, and a phase boundary (stores
// acknowledged -> flag visible -> poll -> activation reload queued behind the landing prefetch -> RMSNorm) is 8-10 us of dependent
// round trips of which the 4 prefetched stages cover 4.4.  Opt-in (TenantDecoder.persistent); DESIGN.md section 8.1 lists what
// would make it pay.
#pragma once
#include "bd_gemv_stream.h"

namespace bd {

constexpr int CHAIN_MAX_PHASES = 4;
constexpr int CHAIN_SPIN_LIMIT = 1 << 21;      // polls of a grid barrier before giving up (sets sync[1]; never hangs the GPU)
constexpr int CHAIN_FLAG_WORD0 = 16;           // arrival flags start 64 bytes into the sync area
constexpr int CHAIN_SYNC_BYTES = 64 + 4 * 1024;   // epoch / error words + one flag per block (<= 1024 blocks)

struct ChainPhase {
    const unsigned short* X;      // activations [R rows] (row stride sX elements); XL phases: the un-normalised residual stream
    const unsigned short* W;      // base weight [N, K] (row stride ldw)
    const uint32_t* P;            // packed decode signs [ceil(N/16)][ceil(K/128)][4][16][tp]
    const float* alpha;           // fp32 [R, G] (row stride sAl; 0 broadcasts)
    unsigned short* C;            // output [R, N] (EPI: [R, N/2]), row stride sC
    const unsigned short* Rsd;    // residual [R, N] (row stride sR) added in the epilogue, or nullptr
    const unsigned short* nw;     // XL: norm weight [R or 1, K] (row stride sNw; 0 broadcasts)
    int N, K, ldw, cpb, gsz;
    int sX, sC, sR, sAl, sNw;
    uint32_t x_bytes, w_bytes, p_bytes, n_bytes;
    float eps;
    int jsh;                      // XL: K = 2048 << jsh
};

struct ChainParams {
    ChainPhase ph[CHAIN_MAX_PHASES];     // [0] o (plain + residual), [1] gate|up (XL + SwiGLU), [2] down (plain + residual), [3] q|k|v (XL)
    int nph;                             // 3 or 4
    int R;                               // tenants = activation rows (one token each)
    uint32_t tp;                         // dwords per (tile, iteration, lane group, column) of the sign packs
    uint32_t xs_off, xrow;               // LDS: normalised activation rows (XL phases)
    unsigned long long* trace;           // BD_CHAIN_TRACE builds only (else unused)
    unsigned* sync;                      // [0] epoch, [1] error, [CHAIN_FLAG_WORD0 + block] arrival flags; zero-filled ONCE by the host
};

__device__ __forceinline__ void chain_store4(unsigned short* dst, uint32_t lo, uint32_t hi) {
    __hip_atomic_store((unsigned long long*)dst, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long chain_load4(const unsigned short* src) {
    return __hip_atomic_load((unsigned long long*)const_cast<unsigned short*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int AUX_SC1 = 16;       // raw buffer load cache policy bit 4 on gfx940+: agent-coherent (bypasses this XCD's stale L2 lines)

// Development probe (build with -DBD_CHAIN_TRACE): the owner wave of every block stamps s_memtime at 6 points of every phase into
// trace[block][phase][6] = { phase entry, first compute, loop end, stores acknowledged, flag published, barrier passed }.
#ifdef BD_CHAIN_TRACE
#define BD_CT(slot)                                                                                                   \
    do {                                                                                                              \
        if (owner && cp.trace) {                                                                                      \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
            if (lane == 0) cp.trace[((long long)blockIdx.x * CHAIN_MAX_PHASES + I) * 6 + (slot)] = t_;                 \
        }                                                                                                             \
    } while (0)
#else
#define BD_CT(slot) do {} while (0)
#endif

template <int NM> struct ChainStage { u32x4_t xn[4]; u32x4_t wf[4]; uint32_t wd[NM]; };

struct ChainGeom {                 // one wave's view of one phase
    int c_lo, c_hi, ntile, nit, it_lo, it_hi;
};
__device__ __forceinline__ ChainGeom chain_geom(const ChainPhase& q, int blk, int wave) {
    ChainGeom gm;
    gm.c_lo = min(blk * q.cpb, q.N);
    gm.c_hi = min(q.N, gm.c_lo + q.cpb);
    gm.ntile = (gm.c_hi - gm.c_lo + 15) >> 4;
    gm.nit = ((q.K >> 5) + 3) >> 2;
    const int per = (gm.nit + 3) / 4;
    gm.it_lo = min(wave * per, gm.nit);
    gm.it_hi = min(gm.it_lo + per, gm.nit);
    return gm;
}

// KIND: 0 plain, 1 RMSNorm prologue (XL), 2 RMSNorm prologue + SwiGLU epilogue.  PRE: the first NS (W, signs) stages of this phase are
// already in st[] (issued by the previous phase).  I = phase index.
template <int DT, int NM, int NS, int KIND, bool PRE, int I>
__device__ __forceinline__ void chain_phase(const ChainParams& cp, ChainStage<NM> (&st)[NS], char* dyn_lds, int lane, int wave, int blk,
                                            unsigned& epoch) {
    constexpr bool XL = KIND >= 1, EPI = KIND == 2;
    const ChainPhase& q = cp.ph[I];
    // (a RUN-TIME test also for the last possible phase, where it is always false: with a compile-time `false` the next-phase offsets
    // fold to constants and hipcc turns the offset selects of `issue` into branches around the loads)
    const bool has_next = I + 1 < cp.nph;
    const ChainPhase& qn = cp.ph[I + 1 < CHAIN_MAX_PHASES ? I + 1 : I];
    const int li = lane & 15, g = lane >> 4;
    const bool owner = wave == 3;
    const int R = cp.R;
    float* const red = (float*)(dyn_lds + STREAM_LUT_BYTES);
    float* const a_lds = (float*)(dyn_lds + STREAM_LUT_BYTES + stream_red_bytes(4));
    const ChainGeom gm = chain_geom(q, blk, wave), gn = chain_geom(qn, blk, wave);
    const int c_lo = gm.c_lo, c_hi = gm.c_hi, ntile = gm.ntile, nit = gm.nit, it_lo = gm.it_lo, it_hi = gm.it_hi;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(q.X, q.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(q.W, q.w_bytes), rp = make_rsrc(q.P, q.p_bytes);
    const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(qn.W, has_next ? qn.w_bytes : 0u), rp2 = make_rsrc(qn.P, has_next ? qn.p_bytes : 0u);
    const uint32_t x_off = li < R ? (uint32_t)((long long)li * q.sX * 2) : STREAM_OOB;

    BD_CT(0);
    // ---- scales of this block's columns -> LDS table (first load of the phase)
    const int g0 = (EPI || ntile == 0) ? 0 : c_lo / q.gsz, ng = EPI ? 2 : (ntile > 0 ? (c_hi - 1) / q.gsz - g0 + 1 : 1);
    const bool al_lds = R * ng <= 256;
    float a_pre = 0.f;
    if (al_lds) {
        const int idx = min((int)threadIdx.x, R * ng - 1), r = idx / ng, j = idx - r * ng;
        a_pre = q.alpha[(long long)r * q.sAl + g0 + j];
    }

    // ---- XL: raw rows + norm weights (rmsnorm_tenant_kernel's thread mapping); the rows were written by the previous phase
    constexpr int XCH = 16;
    [[maybe_unused]] u32x4_t xraw[XL ? XCH : 1], graw[XL ? XCH : 1];
    [[maybe_unused]] const int jsh = q.jsh;
    if constexpr (XL) {
        const __amdgpu_buffer_rsrc_t rn = make_rsrc(q.nw, q.n_bytes);
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int r = j >> jsh, c = ((int)threadIdx.x + 256 * (j - (r << jsh))) * 8;
            const bool ok = r < R;
            // (default cache policy, like the plain phases' fragments: with sc1 every one of the 256 blocks fetched the same 96 KB from
            //  the memory side -- 8-10 us from phase entry to the first MFMA instead of ~3, measured with the BD_CHAIN_TRACE probe)
            xraw[j] = buf_load16<0>(rx, ok ? (uint32_t)(((long long)r * q.sX + c) * 2) : STREAM_OOB);
            graw[j] = buf_load16<0>(rn, ok ? (uint32_t)(((long long)r * q.sNw + c) * 2) : STREAM_OOB);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // one stage = (tile, iteration) in the natural k order: 4 x 16 B of the lane's W row (+ of its x row), the tenants' sign dwords.
    // `nxt` (uniform): aim the (W, signs) loads at stage `un` of the NEXT phase instead (waves 0-2; the owner aims out of range)
    auto issue = [&](ChainStage<NM>& s_, int tile, int it, bool nxt, int un) {
        const int n1 = c_lo + tile * 16 + li;
        const bool t_ok = tile < ntile, col1 = n1 < c_hi;
        const int k1 = 128 * it + 8 * g;
        const int it2 = gn.it_lo + un;
        const int n2 = gn.c_lo + li;
        const bool col2 = has_next && !owner && gn.ntile > 0 && n2 < gn.c_hi && it2 < gn.it_hi;
        const int k2 = 128 * it2 + 8 * g;
        const __amdgpu_buffer_rsrc_t rws = nxt ? rw2 : rw, rps = nxt ? rp2 : rp;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok1 = t_ok && (k1 + 32 * s < q.K);
            // (default cache policy: every wave of the chip re-reads these fragments tile after tile -- they must hit L2.  Safe because
            //  this launch's first touch of the buffer comes after the barrier behind its producer, and no line of it is shared
            //  between XCDs: host-checked, bd_decode_chain)
            if constexpr (!XL) s_.xn[s] = buf_load16<0>(rx, (ok1 && !nxt) ? x_off + (uint32_t)(k1 + 32 * s) * 2u : STREAM_OOB);
            const uint32_t o1 = (ok1 && col1) ? (uint32_t)n1 * (uint32_t)q.ldw * 2u + (uint32_t)(k1 + 32 * s) * 2u : STREAM_OOB;
            const uint32_t o2 = (col2 && (k2 + 32 * s < qn.K)) ? (uint32_t)n2 * (uint32_t)qn.ldw * 2u + (uint32_t)(k2 + 32 * s) * 2u : STREAM_OOB;
            s_.wf[s] = buf_load16<0>(rws, nxt ? o2 : o1);
        }
        const bool pok1 = t_ok && it < nit && col1;
        const uint32_t p1 = pok1 ? ((((uint32_t)(n1 >> 4) * (uint32_t)nit + (uint32_t)it) * 4u + (uint32_t)g) * 16u + (uint32_t)(n1 & 15)) * cp.tp * 4u : STREAM_OOB;
        const uint32_t p2 = col2 ? ((((uint32_t)(n2 >> 4) * (uint32_t)gn.nit + (uint32_t)it2) * 4u + (uint32_t)g) * 16u + (uint32_t)(n2 & 15)) * cp.tp * 4u : STREAM_OOB;
        const uint32_t po = nxt ? p2 : p1;
        if constexpr (NM == 1) {
            s_.wd[0] = buf_load4<0>(rps, po);
        } else if constexpr (NM == 2) {
            const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rps, (int)po, 0, 0));
            s_.wd[0] = v[0]; s_.wd[1] = v[1];
        } else {
            const u32x4_t v = buf_load16<0>(rps, po);
#pragma unroll
            for (int t = 0; t < 4 && t < NM; ++t) s_.wd[t] = v[t];
            if constexpr (NM == 6) {
                const u32x2_t v2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rps, (int)(po == STREAM_OOB ? STREAM_OOB : po + 16u), 0, 0));
                s_.wd[4] = v2[0]; s_.wd[5] = v2[1];
            } else if constexpr (NM == 8) {
                const u32x4_t v2 = buf_load16<0>(rps, po == STREAM_OOB ? STREAM_OOB : po + 16u);
#pragma unroll
                for (int t = 0; t < 4; ++t) s_.wd[4 + t] = v2[t];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    int ti = 0, ii = it_lo;                                              // next stage to issue
    auto advance = [&](int& t, int& i) { if (++i >= it_hi) { i = it_lo; ++t; } };
    if constexpr (!PRE) {
#pragma unroll
        for (int u = 0; u < NS; ++u) { issue(st[u], ti, ii, false, 0); advance(ti, ii); }
    } else {
        // the (W, signs) parts of the first NS stages are in flight or landed; plain phases add their activation fragments now
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if constexpr (!XL) {
                const int k1 = 128 * ii + 8 * g;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    st[u].xn[s] = buf_load16<0>(rx, (ti < ntile && k1 + 32 * s < q.K) ? x_off + (uint32_t)(k1 + 32 * s) * 2u : STREAM_OOB);
                __builtin_amdgcn_sched_barrier(0);
            }
            advance(ti, ii);
        }
    }

    if constexpr (XL) {
        float* const part = red;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            ss = sumsq8<DT>(xraw[j], ss);
            if (((j + 1) & ((1 << jsh) - 1)) == 0) {
                const float w = wave_sum(ss);
                if (lane == 0) part[(j >> jsh) * 4 + wave] = w;
                ss = 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int r = j >> jsh, c = ((int)threadIdx.x + 256 * (j - (r << jsh))) * 8;
            if (r < R) {
                const float rs = rms_scale(part[r * 4], part[r * 4 + 1], part[r * 4 + 2], part[r * 4 + 3], q.K, q.eps);
                *(u32x4_t*)(dyn_lds + cp.xs_off + (uint32_t)r * cp.xrow + (uint32_t)c * 2u) = norm8<DT>(xraw[j], graw[j], rs);
            }
        }
    }
    if (al_lds && (int)threadIdx.x < R * ng) a_lds[threadIdx.x] = a_pre;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    BD_CT(1);
    f32x4_t accB = {0.f, 0.f, 0.f, 0.f}, accD[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t copy_off = (uint32_t)li * 16u;
    auto lut = [&](uint32_t w, int s) -> u32x4_t {
        const uint32_t off = __builtin_amdgcn_perm(w, copy_off, 0x0c0c0400u + ((uint32_t)s << 8));
        return *(const u32x4_t*)(dyn_lds + off);
    };
    [[maybe_unused]] const uint32_t xl_base = XL ? cp.xs_off + (uint32_t)min(li, R - 1) * cp.xrow + (uint32_t)g * 16u : 0u;
    [[maybe_unused]] u32x4_t xq[XL ? 2 : 1][XL ? 4 : 1];
    auto read_xq = [&](int set, int it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            xq[XL ? set : 0][XL ? s : 0] = *(const u32x4_t*)(dyn_lds + xl_base + (uint32_t)min(it, nit - 1) * 256u + 64u * s);
    };
    auto compute = [&](const ChainStage<NM>& cur, [[maybe_unused]] int par, [[maybe_unused]] int it_next) {
        if constexpr (XL) read_xq(par ^ 1, it_next);
        u32x4_t sf[2][NM];
#pragma unroll
        for (int t = 0; t < NM; ++t) sf[0][t] = lut(cur.wd[t], 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
#pragma unroll
                for (int t = 0; t < NM; ++t) sf[(s + 1) & 1][t] = lut(cur.wd[t], s + 1);
            }
            const u32x4_t xv = XL ? xq[XL ? par : 0][XL ? s : 0] : cur.xn[s];
            accB = mfma16<DT>(cur.wf[s], xv, accB);
#pragma unroll
            for (int t = 0; t < NM; ++t) accD[t] = mfma16<DT>(sf[s & 1][t], xv, accD[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // end of a tile: the 4 partial tiles meet in LDS; the OWNER wave sums them in wave order and stores (every tile: see the header)
    auto finish_tile = [&](int tile) {
        float* const rb = red + (tile & 1) * (4 * 64 * 8);
        {
            const int bm = min(li, R - 1);
            u32x4_t d = __builtin_bit_cast(u32x4_t, accD[0]);
#pragma unroll
            for (int t = 1; t < NM; ++t) {
                const uint32_t mk = bm == t ? 0xffffffffu : 0u;
                d = (__builtin_bit_cast(u32x4_t, accD[t]) & u32x4_t{mk, mk, mk, mk}) | (d & ~u32x4_t{mk, mk, mk, mk});
            }
            *(f32x4_t*)&rb[(wave * 64 + lane) * 8] = accB;
            *(u32x4_t*)&rb[(wave * 64 + lane) * 8 + 4] = d;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (owner && li < R) {
            f32x4_t sb = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                sb += *(const f32x4_t*)&rb[(w * 64 + lane) * 8];
                sd += *(const f32x4_t*)&rb[(w * 64 + lane) * 8 + 4];
            }
            if constexpr (EPI) {
                const int grp = g >> 1;
                float a = 1.f;
                if (al_lds) a = a_lds[li * 2 + grp];
                else a = q.alpha[(long long)li * q.sAl + grp];
                const int n_out = ((c_lo + tile * 16) >> 1) + 4 * (g & 1);
                uint32_t o16[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = round16<DT>(scale_then_add(sd[e], a, sb[e]));
                    const float u = __shfl(v, (lane + 32) & 63, 64);
                    o16[e] = swiglu1<DT>(v, u);
                }
                if (g < 2) chain_store4(q.C + (long long)li * q.sC + n_out, o16[0] | (o16[1] << 16), o16[2] | (o16[3] << 16));
            } else {
                const int n = c_lo + tile * 16 + 4 * g;                 // 4 consecutive columns per lane (N % 16 == 0: all in range)
                float rsd[4] = {0.f, 0.f, 0.f, 0.f};
                if (q.Rsd) {
                    const unsigned long long rv = chain_load4(q.Rsd + (long long)li * q.sR + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rsd[e] = half_bits_to_f32<DT>((uint32_t)(rv >> (16 * e)) & 0xffffu);
                }
                uint32_t o16[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = 1.f;
                    if (al_lds) a = a_lds[li * ng + ((n + e) / q.gsz - g0)];
                    else a = q.alpha[(long long)li * q.sAl + (n + e) / q.gsz];
                    float v = scale_then_add(sd[e], a, sb[e]);
                    if (q.Rsd) v += rsd[e];
                    o16[e] = f32_to_half_bits<DT>(v);
                }
                if (n < c_hi) chain_store4(q.C + (long long)li * q.sC + n, o16[0] | (o16[1] << 16), o16[2] | (o16[3] << 16));
            }
        }
        accB = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };

    // main loop: whole rounds of NS stages (gemv_stream_kernel's structure); the issues of the LAST round are all past the end of
    // this phase's stream: they fetch the next phase's stages 0 .. NS-1 instead (waves 0-2)
    const int cntb = max(it_hi - it_lo, 1);
    const int total = ntile * cntb;
    int tc = 0, ic = it_lo;
    int f = 0;
    static_assert(NS % 2 == 0, "stage parity selects the activation fragment set");
    if constexpr (XL) read_xq(0, it_lo);
    // The first round is peeled: at a PRE entry the activation fragments of the first NS stages are younger than ALL of their
    // prefetched (W, signs) parts; if that entry state flowed into the loop header, hipcc's waitcnt pass would merge it with the
    // steady state and wait as if every round looked like the entry (vmcnt(12) instead of vmcnt(30): one stage in flight, not three).
    auto round = [&](bool last) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            compute(st[u], u & 1, ic + 1 >= it_hi ? it_lo : ic + 1);
            __builtin_amdgcn_sched_barrier(0);
            issue(st[u], ti, ii, last, u);
            advance(ti, ii);
            if (++ic >= it_hi && tc < ntile) {
                finish_tile(tc);
                ic = it_lo;
                ++tc;
            }
        }
    };
    round(NS >= total);
    f = NS;
    while (f < total) {
        round(f + NS >= total);
        f += NS;
    }

    // ---- phase boundary
    BD_CT(2);
    if (has_next) {
        if (owner) {
            __builtin_amdgcn_s_waitcnt(0x0f70);                          // vmcnt(0): every output store of this block is acknowledged
            BD_CT(3);
#pragma unroll
            for (int u = 0; u < NS; ++u) {                               // its own first NS stages of the next phase, only now
                ChainStage<NM>& s_ = st[u];
                const int it2 = gn.it_lo + u, n2 = gn.c_lo + li, k2 = 128 * it2 + 8 * g;
                const bool col2 = gn.ntile > 0 && n2 < gn.c_hi && it2 < gn.it_hi;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    s_.wf[s] = buf_load16<0>(rw2, (col2 && (k2 + 32 * s < qn.K)) ? (uint32_t)n2 * (uint32_t)qn.ldw * 2u + (uint32_t)(k2 + 32 * s) * 2u : STREAM_OOB);
                const uint32_t po = col2 ? ((((uint32_t)(n2 >> 4) * (uint32_t)gn.nit + (uint32_t)it2) * 4u + (uint32_t)g) * 16u + (uint32_t)(n2 & 15)) * cp.tp * 4u : STREAM_OOB;
                if constexpr (NM == 1) {
                    s_.wd[0] = buf_load4<0>(rp2, po);
                } else if constexpr (NM == 2) {
                    const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rp2, (int)po, 0, 0));
                    s_.wd[0] = v[0]; s_.wd[1] = v[1];
                } else {
                    const u32x4_t v = buf_load16<0>(rp2, po);
#pragma unroll
                    for (int t = 0; t < 4 && t < NM; ++t) s_.wd[t] = v[t];
                    if constexpr (NM == 6) {
                        const u32x2_t v2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rp2, (int)(po == STREAM_OOB ? STREAM_OOB : po + 16u), 0, 0));
                        s_.wd[4] = v2[0]; s_.wd[5] = v2[1];
                    } else if constexpr (NM == 8) {
                        const u32x4_t v2 = buf_load16<0>(rp2, po == STREAM_OOB ? STREAM_OOB : po + 16u);
#pragma unroll
                        for (int t = 0; t < 4; ++t) s_.wd[4 + t] = v2[t];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // grid barrier I: arrive, then poll (the poll's result cannot return before the loads above have landed: that is the overlap).
            // No read-modify-write: 256 blocks adding to one counter are serialised at the coherence point (measured: ~12 us per
            // barrier).  Every block has its own flag word and publishes "epoch * 4 + I + 1" (the epoch word is bumped once per launch,
            // so the flags never have to be reset and never have to be zero between launches); a poll is ONE 1-KiB coherent load per
            // 256 flags by the owner wave and a wave-wide AND.
            const unsigned target = epoch * 4u + (unsigned)I + 1u;
            unsigned* const flags = cp.sync + CHAIN_FLAG_WORD0;
            if (lane == 0) __hip_atomic_store(&flags[blockIdx.x], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            BD_CT(4);
            const __amdgpu_buffer_rsrc_t rf = make_rsrc(flags, gridDim.x * 4u);
            int spins = 0;
            for (;;) {
                bool ok = true;
                for (unsigned base = 0; base < gridDim.x; base += 256) {
                    const u32x4_t v = buf_load16<AUX_SC1>(rf, (base + 4u * (unsigned)lane) * 4u);      // past the grid: zeros
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ok = ok && (base + 4u * (unsigned)lane + e >= gridDim.x || (int)(v[e] - target) >= 0);
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > CHAIN_SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(&cp.sync[1], 1u + (unsigned)I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            BD_CT(5);
        }
        __builtin_amdgcn_s_barrier();                                    // waves 0-2 sleep here until the owner has seen the barrier
        asm volatile("" ::: "memory");
    }
}

template <int DT, int NM, int NS>
__global__ void __launch_bounds__(256) decode_chain_kernel(const ChainParams cp) {
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];     // [64 KiB sign LUT][16 KiB reduction][2 KiB scales][R activation rows]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    {   // sign LUT, 16 copies (gemv_stream_kernel), built ONCE for all phases
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int slot = threadIdx.x + 256 * j, ee = slot >> 4;
            u32x4_t w;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = (((ee >> (2 * d)) & 1) ? POS : NEG) | ((((ee >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
            *(u32x4_t*)(dyn_lds + slot * 16) = w;
        }
    }
    ChainStage<NM> st[NS];
    // the launch's epoch: the OLDEST load of every wave (reading it at the first barrier would put it behind the prefetched stages
    // and delay this block's arrival by their latency)
    unsigned epoch = __hip_atomic_load(&cp.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_phase<DT, NM, NS, 0, false, 0>(cp, st, dyn_lds, lane, wave, blk, epoch);      // o + residual
    chain_phase<DT, NM, NS, 2, true, 1>(cp, st, dyn_lds, lane, wave, blk, epoch);       // RMSNorm -> gate|up -> SwiGLU
    chain_phase<DT, NM, NS, 0, true, 2>(cp, st, dyn_lds, lane, wave, blk, epoch);       // down + residual
    if (cp.nph > 3) chain_phase<DT, NM, NS, 1, true, 3>(cp, st, dyn_lds, lane, wave, blk, epoch);   // RMSNorm -> q|k|v of the next layer
    // exit: block 0 starts the next epoch.  (Every block read the epoch word before it arrived at barrier 0, which block 0 has passed;
    // the next launch reads it after this kernel has completed.)
    if (blockIdx.x == 0 && threadIdx.x == 192)
        __hip_atomic_store(&cp.sync[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace bd
