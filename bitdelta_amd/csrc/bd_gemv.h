// Decode path (HBM-bound): few activation rows (R = B*M <= 16), every weight byte read once.
//
//   y[b,m,:] = x[b,m,:] . W^T + alpha[b] * (x[b,m,:] . S_b)        (fused; W optional)
//
// Reference call sites: DiffCompressModule.forward at decode, demo/demo_backend.py:93-98 (M = 1, B = tenants);
// BinaryDiff.forward, bitdelta/diff.py:33-39 with a single token.
//
// Work split: grid = (ceil(N/64) column tiles) x (KS k-slices); every block streams its [64 n] x [k-slice] piece of
//   * the 16-bit base weight W [N,K] (rows are k-contiguous): MFMA 16x16x32 with the activations as the 16-wide
//     second operand -- 16-byte loads straight to VGPRs, no LDS (streamed once, not shared between waves);
//   * each tenant's packed sign words P_b [K/32, N]: lane = output column (256-byte coalesced word rows), the 32 signs
//     of a word are expanded to +-1.0 pairs (2 VALU / pair) and contracted with the wave-uniform activation pairs held
//     in SGPRs by v_dot2c_f32_{bf16,f16} (fp32 accumulate) -- the "sign-flip GEMV" variant of the MFMA path.
// Partial sums of the k-slices go to an fp32 workspace [KS][R][N]; the last block to arrive at a column tile's ticket sums them
// in k-slice order and rounds once (gemv_ticket_reduce; gemv_reduce_kernel is the two-launch fallback / A-B reference).
// With KS == 1 the block writes the final result itself.
#pragma once
#include "bd_common.h"

namespace bd {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct GemvParams {
    const unsigned short* X;   // [B, M, K]
    const uint32_t* P;         // [B or 1, K/32, N]
    const unsigned short* W;   // [N, K] or nullptr (delta only)
    const float* alpha;        // fp32 [B or 1, G] or nullptr
    void* C;                   // [B, M, N]
    float* ws;                 // [KS][R][N] fp32 partials (KS > 1)
    uint32_t* tickets;         // one arrival counter per 64-column tile (KS > 1, in-launch reduction); nullptr -> gemv_reduce_kernel
    int B, M, N, K, R;
    long long sXb, sPb, sCb;
    int sXm, sCm, ldw, sAlb, gsz;
    int KS, kslice;            // k-slice length (multiple of 128)
    int round_mode, accumulate, out_f32;
};

template <int DT> __device__ __forceinline__ float dot2acc(uint32_t a, uint32_t b, float c) {
    if constexpr (DT == DT_BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
    else return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
}
template <int DT> __device__ __forceinline__ f32x4_t mfma16(u32x4_t a, u32x4_t b, f32x4_t c) {
    if constexpr (DT == DT_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <int DT> __device__ __forceinline__ void store_out(const GemvParams& p, int r, int n, float v) {
    const int b = r / p.M, m = r - b * p.M;
    const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
    if (p.accumulate) {
        const float cin = p.out_f32 ? ((const float*)p.C)[off] : half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]);
        v += cin;
    } else if (!p.W && !p.alpha && p.round_mode == 1) {
        v = round_through_f16(v);
    }
    if (p.out_f32) ((float*)p.C)[off] = v;
    else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
}

// the same store with the residual value (what `accumulate` would read from C) already in a register: the streaming kernel fetches it at
// kernel start -- a load issued in the epilogue is a full memory round trip with nothing left to hide it
template <int DT> __device__ __forceinline__ void store_out_residual(const GemvParams& p, int r, int n, float v, float cin) {
    const int b = r / p.M, m = r - b * p.M;
    const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
    v += cin;
    if (p.out_f32) ((float*)p.C)[off] = v;
    else ((unsigned short*)p.C)[off] = (unsigned short)f32_to_half_bits<DT>(v);
}

// The two 16-bit stores above, returning the value AS STORED (rounded to the output type): the RMSNorm hand-off of the streaming decode
// kernel sums its square (bd_gemv_stream.h, StreamParams::ssq_out).  16-bit outputs only (host-checked).
template <int DT> __device__ __forceinline__ float store_out_ret(const GemvParams& p, int r, int n, float v) {
    const int b = r / p.M, m = r - b * p.M;
    const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
    if (p.accumulate) v += half_bits_to_f32<DT>(((const unsigned short*)p.C)[off]);
    const uint32_t bits = f32_to_half_bits<DT>(v);
    ((unsigned short*)p.C)[off] = (unsigned short)bits;
    return half_bits_to_f32<DT>(bits & 0xffffu);
}
template <int DT> __device__ __forceinline__ float store_out_residual_ret(const GemvParams& p, int r, int n, float v, float cin) {
    const int b = r / p.M, m = r - b * p.M;
    const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
    const uint32_t bits = f32_to_half_bits<DT>(v + cin);
    ((unsigned short*)p.C)[off] = (unsigned short)bits;
    return half_bits_to_f32<DT>(bits & 0xffffu);
}

// In-launch split-k reduction.  Every block has stored its fp32 partial tile to ws[ks][r][n]; the block that arrives LAST at the
// tile's ticket sums the KS partials in k-slice order (so the result does not depend on which block that is), rounds once, stores,
// and puts the ticket back to 0 for the next launch.
// Coherence: the XCDs' L2s are not coherent with each other for ordinary stores, and a real agent-scope release/acquire fence
// costs an L2 write-back + invalidate per block (measured: 45-250 us per launch instead of 8-48).  So the shared words -- partials
// and tickets -- are only ever touched by agent-scope RELAXED atomics (sc1 stores/loads that go to the coherence point), and the
// ordering is explicit: s_waitcnt vmcnt(0) (every partial store of this wave acknowledged) -> s_barrier (... of this block) ->
// ticket increment -> (last block only) partial loads, which are issued after the increment has returned.
// Contract (include/bitdelta_hip.h): the ticket area of the workspace is zero when the launch is enqueued.
constexpr int GEMV_TICKET_BYTES = 64 * 1024;      // fixed-size area at the start of the workspace: 16384 tiles (N <= 1 Mi columns)
__device__ __forceinline__ void gemv_store_partial(float* dst, float v) {
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int DT>
__device__ __forceinline__ void gemv_ticket_reduce(const GemvParams& p, int n0, int lane, int wave) {
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): this wave's partial stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&p.tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old >= (uint32_t)p.KS) __builtin_trap();  // dirty workspace: fail loudly instead of returning garbage
        s_last = (old == (uint32_t)p.KS - 1u);
    }
    __syncthreads();
    if (!s_last) return;
    const int n = n0 + lane;
    if (n < p.N) {
        for (int r = wave; r < p.R; r += 4) {
            float s = 0.f;
            for (int k0 = 0; k0 < p.KS; k0 += 8) {            // 8 independent loads in flight, summed in k order
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = __hip_atomic_load(&p.ws[((long long)min(k0 + j, p.KS - 1) * p.R + r) * p.N + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (k0 + j < p.KS) ? v[j] : 0.f;
            }
            store_out<DT>(p, r, n, s);
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(&p.tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// words of RB k32-rows x RMAX activation rows are fetched as one batch (RB*R independent 256-byte loads per wave in flight)
template <int RMAX> struct GemvBatch { static constexpr int RB = (32 / RMAX) < 1 ? 1 : ((32 / RMAX) > 8 ? 8 : (32 / RMAX)); };

// k per slice: the block's activation slice is staged in LDS (RMAX rows x kslice x 2 B), static LDS budget 64 KB
// (measured: 4096-k slices for <= 6 rows cut occupancy to 3 blocks/CU and were 15-20 % slower than 1024-k slices + more k-splits)
constexpr int gemv_kslice_max(int /*rmax*/) { return 1024; }

// SPEC = wave-specialised form (fused launches): 8 waves per block, waves 0-3 stream the base weight (MFMA, load-bound), waves 4-7
// expand and contract the sign words (VALU-bound).  One wave of each kind shares a SIMD, so the weight stream and the sign work
// overlap instead of adding up (the single-role kernel measured floor + base + delta: a wave busy expanding signs is not issuing
// weight loads, and in-order vmcnt stops one wave from keeping both streams in flight).
template <int DT, int RMAX, bool SPEC = false>
__global__ void __launch_bounds__(SPEC ? 512 : 256) gemv_kernel(const GemvParams p) {
    constexpr int RB = GemvBatch<RMAX>::RB;
    constexpr int XROW = gemv_kslice_max(RMAX) * 2 + 16;    // padded LDS row of the activation slice (bytes)
    __shared__ float red[4][RMAX][64];    // delta partials per wave
    __shared__ float bs[RMAX][64];        // base GEMV tile
    __shared__ __attribute__((aligned(16))) char xs_lds[RMAX * XROW];
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_id & 3;                                   // column group (base role) / word-row phase (delta role)
    const bool do_base = SPEC ? wave_id < 4 : true, do_delta = SPEC ? wave_id >= 4 : true;
    constexpr int NTHR = SPEC ? 512 : 256;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int k_lo = ks * p.kslice, k_hi = min(p.K, k_lo + p.kslice);
    uint32_t one2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(one2) : "n"(One2<DT>::v));

    // ---------------- delta part, stage 1: put the first batch of sign words in flight before anything else ----------------
    const int nc = min(n0 + lane, p.N - 1);
    const int i_lo = (k_lo >> 5) + wave, i_hi = k_hi >> 5;          // this wave's word rows: i_lo, i_lo+4, ...
    auto load_batch = [&](uint32_t (&w)[RB][RMAX], int ib) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = min(ib + 4 * rb, (p.K >> 5) - 1);          // clamped rows are skipped at compute time
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int b = min(r, p.R - 1) / p.M;                  // rows >= R repeat the last row (results discarded)
                w[rb][r] = __builtin_nontemporal_load(&p.P[(long long)b * p.sPb + (long long)i * p.N + nc]);
            }
        }
    };
    uint32_t wcur[RB][RMAX];
    if (do_delta) load_batch(wcur, i_lo);

    // ---------------- activation slice -> LDS (once per block; rows >= R repeat the last row) ----------------
    {
        const int kn = k_hi - k_lo;                                   // multiple of 32
        for (int idx = threadIdx.x; idx < RMAX * (kn >> 3); idx += NTHR) {
            const int r = idx / (kn >> 3), c = idx - r * (kn >> 3);
            const int rr = min(r, p.R - 1), b = rr / p.M, m = rr - b * p.M;
            *(u32x4_t*)(xs_lds + r * XROW + c * 16) =
                *(const u32x4_t*)(p.X + (long long)b * p.sXb + (long long)m * p.sXm + k_lo + c * 8);
        }
    }
    __syncthreads();

    // ---------------- base part: D[n][r] += W[n][k] x[r][k], one 16-column group per wave (weights streamed once) -------
    auto base_part = [&]() {
    if (p.W && do_base) {
        const int li = lane & 15, g = lane >> 4;
        const int nw = min(n0 + wave * 16 + li, p.N - 1);
        const unsigned short* wr = p.W + (long long)nw * p.ldw + 8 * g;
        const char* xl = xs_lds + min(li, RMAX - 1) * XROW + 16 * g;  // MFMA column li <-> activation row li (rows >= RMAX unused)
        const uint32_t keep = (li < p.R) ? 0xffffffffu : 0u;
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int k = k_lo;
        for (; k + 256 <= k_hi; k += 256) {                          // 8 x (1 KiB of W per wave) in flight
            u32x4_t wf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wf[u] = __builtin_nontemporal_load((const u32x4_t*)(wr + k + 32 * u));
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const u32x4_t x0 = *(const u32x4_t*)(xl + (k - k_lo + 32 * u) * 2) & u32x4_t{keep, keep, keep, keep};
                const u32x4_t x1 = *(const u32x4_t*)(xl + (k - k_lo + 32 * u + 32) * 2) & u32x4_t{keep, keep, keep, keep};
                acc0 = mfma16<DT>(wf[u], x0, acc0);
                acc1 = mfma16<DT>(wf[u + 1], x1, acc1);
            }
        }
        for (; k < k_hi; k += 32) {
            const u32x4_t wf = *(const u32x4_t*)(wr + k);
            const u32x4_t xf = *(const u32x4_t*)(xl + (k - k_lo) * 2) & u32x4_t{keep, keep, keep, keep};
            acc0 = mfma16<DT>(wf, xf, acc0);
        }
        // lane holds r = li, n_local = 4g + reg
        if (li < RMAX) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[li][wave * 16 + 4 * g + e] = acc0[e] + acc1[e];
        }
    }
    };

    // ---------------- delta part, stage 2: lane = column; signs -> +-1.0 pairs -> v_dot2c with the activation pairs --------
    // Branch-free over the RMAX rows and 4 independent accumulators per row: the 16 dot2 of one word would otherwise be one
    // dependent chain (measured: 5x the HBM time per tenant).  Activation pairs come from LDS as wave-uniform broadcasts.
    float dacc[RMAX][4];
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dacc[r][c] = 0.f;
    // The base part is load-bound and the delta part VALU-bound; blocks alternate the order so that, CU-wide, one half's
    // weight stream overlaps the other half's sign expansion (measured: the two phases otherwise simply add up).
    const bool delta_first = SPEC ? false : ((blockIdx.x + blockIdx.y) & 1) != 0;
    if (!delta_first) base_part();
    if (do_delta)
    for (int ib = i_lo; ib < i_hi; ib += 4 * RB) {
        uint32_t wnext[RB][RMAX];
        const bool more = ib + 4 * RB < i_hi;
        if (more) load_batch(wnext, ib + 4 * RB);                    // next batch in flight under this batch's VALU work
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = ib + 4 * rb;
            if (i < i_hi) {
                const int koff = (32 * i - k_lo) * 2;
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const uint32_t w = ~wcur[rb][r];
                    const uint32_t rlo = (w & 0xffffu) | ((w << 15) & 0x7fff0000u);     // chunk | (chunk>>1)<<16, the same trick as the MFMA tile kernels
                    const uint32_t rhi = (w >> 16) | ((w >> 1) & 0x7fff0000u);
                    const u32x4_t* xv = (const u32x4_t*)(xs_lds + r * XROW + koff);  // 64 bytes = 16 pairs, same address in every lane
                    u32x4_t x4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) x4[c] = xv[c];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const uint32_t rep = q < 8 ? rlo : rhi;
                        const uint32_t sd = ((rep << (15 - 2 * (q & 7))) & 0x80008000u) | one2;
                        dacc[r][q & 3] = dot2acc<DT>(sd, x4[q >> 2][q & 3], dacc[r][q & 3]);
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < RMAX; ++r) wcur[rb][r] = wnext[rb][r];
        }
    }
    if (delta_first) base_part();
    if (do_delta) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) red[wave][r][lane] = (dacc[r][0] + dacc[r][1]) + (dacc[r][2] + dacc[r][3]);
    }
    __syncthreads();

    // ---------------- combine: wave w finishes rows r = w, w+4, ... ----------------
    const int n = n0 + lane;
    if (n < p.N && wave_id < 4) {
        for (int r = wave; r < p.R; r += 4) {
            const int b = r / p.M;
            float d = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
            if (p.alpha) d *= p.alpha[(long long)b * p.sAlb + n / p.gsz];
            if (p.W) d += bs[r][lane];
            if (p.KS == 1) store_out<DT>(p, r, n, d);
            else gemv_store_partial(&p.ws[((long long)ks * p.R + r) * p.N + n], d);
        }
    }
    if (p.KS > 1 && p.tickets) gemv_ticket_reduce<DT>(p, n0, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------------
// MFMA form of the decode kernel (the shipped one; gemv_kernel above is kept as the A/B reference, variant 300).
//
// gemv_kernel spends 3 VALU per sign pair (2 to expand, 1 v_dot2c): at 6 tenants the delta part alone took longer than streaming
// the base weight.  Here BOTH parts run on the matrix pipe with one shared activation fragment per k-step:
//   * wave = 16 output columns x the block's k-slice; lane (c = l & 15, g = l >> 4);
//   * an iteration covers 4 word rows (128 k): lane group g owns word row i+g, MFMA step s (of 4) covers the k-octets
//     {32(i+g) + 8s .. +7 : g = 0..3} for every operand -- so a lane loads ONE sign word per mask per iteration (4 rows x 64 B
//     per wave-load) and 64 contiguous bytes of its W row (4 x 16 B), with no cross-lane exchange;
//   * sign fragment of step s = LUT[byte s of the word] (256 x 16 B table in LDS, one ds_read_b128; v_bfe + v_lshl_add for the
//     address) -> 0.25 VALU per sign instead of 1.5;
//   * D[col][row] = v_mfma_f32_16x16x32(W or S fragment, x fragment): base accumulator + one accumulator per distinct mask.
//     With per-tenant masks only the rows of that tenant are kept from its accumulator (the other 16-M columns of D are wasted
//     matrix-pipe work, which is idle here anyway: <= 68 MFMAs per 8 KB of HBM bytes per wave).
// Rows >= R of the activation tile are never staged: they only feed D columns that are never stored.
template <int DT, int NM, bool HASW, int LC>
__global__ void __launch_bounds__(256) gemv_mfma_kernel(const GemvParams p) {
    // LC = copies of the sign LUT.  LC = 16: entry e of copy c lives at e*256 + c*16 and lane l reads copy l & 15, so the 16 lanes of
    // every ds_read_b128 service group hit 16 different 16-byte slots of the 256-byte bank row whatever their bytes are
    // (conflict-free, 4 LDS cycles per read = 128 signs/clk/CU).  With one copy, random bytes collide ~3-way (measured 0.92 us per
    // tenant per 4096^2 mask against 0.35 us of HBM time).  64 KiB of LDS: used when there are >= 3 masks to expand.
    constexpr int XROW = gemv_kslice_max(16) * 2 + 16;
    constexpr int LUT_BYTES = 256 * 16 * LC;
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];     // [LUT][R rows x XROW bytes of activations]
    __shared__ float outs[16][65];
    char* const xs_lds = dyn_lds + LUT_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int k_lo = ks * p.kslice, k_hi = min(p.K, k_lo + p.kslice);
    const int i0 = k_lo >> 5, i_hi = k_hi >> 5, i_last = (p.K >> 5) - 1;
    const int nit = (i_hi - i0 + 3) >> 2;
    const int nmask = p.sPb == 0 ? 1 : p.B;                       // distinct masks (a stride-0 mask is shared by every row)

    const int nw = min(n0 + wave * 16 + li, p.N - 1);
    const unsigned short* wr = HASW ? p.W + (long long)nw * p.ldw : nullptr;
    const uint32_t* pw[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) pw[t] = p.P + (long long)min(t, nmask - 1) * p.sPb + nw;

    struct Stage { u32x4_t wf[4]; uint32_t wd[NM]; };
    auto load_iter = [&](Stage& st, int it) {
        const int irow = min(i0 + 4 * it + g, i_last);              // rows past the slice are zeroed through the x fragment
#pragma unroll
        for (int t = 0; t < NM; ++t) st.wd[t] = __builtin_nontemporal_load(pw[t] + (long long)irow * p.N);
        if constexpr (HASW) {
#pragma unroll
            for (int s = 0; s < 4; ++s) st.wf[s] = __builtin_nontemporal_load((const u32x4_t*)(wr + irow * 32 + 8 * s));
        }
    };
    // Two register stages alternate (loop unrolled by 2, static indices): the next 128 k of W rows and sign words are in flight
    // under the current iteration's MFMAs and nothing is ever copied between stages.
    Stage st[2];
    load_iter(st[0], 0);

    {   // sign LUT: entry e = the 8 (+-1.0) 16-bit values of byte e (bit j <-> k offset j)
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
        const int e = threadIdx.x;
        u32x4_t v;
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = (((e >> (2 * d)) & 1) ? POS : NEG) | ((((e >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
#pragma unroll
        for (int c = 0; c < LC; ++c) *(u32x4_t*)(dyn_lds + e * 16 * LC + c * 16) = v;
    }
    {   // activation slice -> LDS (rows < R only)
        const int kn = k_hi - k_lo;                                   // multiple of 32
        for (int idx = threadIdx.x; idx < p.R * (kn >> 3); idx += 256) {
            const int r = idx / (kn >> 3), c = idx - r * (kn >> 3);
            const int b = r / p.M, m = r - b * p.M;
            *(u32x4_t*)(xs_lds + r * XROW + c * 16) =
                *(const u32x4_t*)(p.X + (long long)b * p.sXb + (long long)m * p.sXm + k_lo + c * 8);
        }
    }
    __syncthreads();

    f32x4_t accB = {0.f, 0.f, 0.f, 0.f}, accD[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* xl = xs_lds + min(li, p.R - 1) * XROW + 64 * g;                     // (32 (4 it + g) + 8 s) * 2 bytes = 256 it + 64 g + 16 s
    const uint32_t copy_off = LC == 16 ? (uint32_t)li * 16u : 0u;

    auto compute = [&](const Stage& cur, int it) {
        const uint32_t keep = (i0 + 4 * it + g < i_hi) ? 0xffffffffu : 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4_t xf = *(const u32x4_t*)(xl + it * 256 + 16 * s) & u32x4_t{keep, keep, keep, keep};
            if constexpr (HASW) accB = mfma16<DT>(cur.wf[s], xf, accB);
#pragma unroll
            for (int t = 0; t < NM; ++t) {
                uint32_t off;
                if constexpr (LC == 16) off = __builtin_amdgcn_perm(cur.wd[t], copy_off, 0x0c0c0400u + ((uint32_t)s << 8));   // byte s -> bits 8..15, copy slot -> bits 0..7
                else off = ((cur.wd[t] >> (8 * s)) & 0xffu) * 16u;
                const u32x4_t sf = *(const u32x4_t*)(dyn_lds + off);
                accD[t] = mfma16<DT>(sf, xf, accD[t]);
            }
        }
    };
    // One iteration ahead.  (Measured and rejected: two iterations ahead with unconditional, clamped loads -- the compiler's
    // s_waitcnt pass needs straight-line loads to keep two stages in flight, and at the usual 4 iterations per wave (512-k slices)
    // the re-fetched tail doubled the load instructions: T=1 10.0 -> 14.2 us, gate T=6 48.4 -> 58.1 us.)
    for (int it = 0; it < nit; it += 2) {
        if (it + 1 < nit) load_iter(st[1], it + 1);
        compute(st[0], it);
        if (it + 1 < nit) {
            if (it + 2 < nit) load_iter(st[0], it + 2);
            compute(st[1], it + 1);
        }
    }

    // lane (li, g) holds D[col = 4g + e][row = li]: pick the accumulator of this row's mask, scale, add the base, stage in LDS
    if (li < p.R) {
        const int b = li / p.M;
        const int bm = p.sPb == 0 ? 0 : b;
        f32x4_t d = accD[0];
#pragma unroll
        for (int t = 1; t < NM; ++t)
            if (bm == t) d = accD[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = d[e];
            if (p.alpha) v *= p.alpha[(long long)b * p.sAlb + min(n0 + wave * 16 + 4 * g + e, p.N - 1) / p.gsz];
            if constexpr (HASW) v += accB[e];
            outs[li][wave * 16 + 4 * g + e] = v;
        }
    }
    __syncthreads();
    const int n = n0 + lane;
    if (n < p.N) {
        for (int r = wave; r < p.R; r += 4) {
            const float v = outs[r][lane];
            if (p.KS == 1) store_out<DT>(p, r, n, v);
            else gemv_store_partial(&p.ws[((long long)ks * p.R + r) * p.N + n], v);
        }
    }
    if (p.KS > 1 && p.tickets) gemv_ticket_reduce<DT>(p, n0, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------------
// No-split-k form of the MFMA decode kernel: ONE launch per Linear whenever the activation rows fit in LDS.
//
// gemv_kernel / gemv_mfma_kernel need ~512 blocks, so they slice k across blocks and pay a second launch (4.6 us of an 18 us
// Linear) to sum the slices.  Here a block owns 16 output columns and ALL of k (or one of KS large slices when R*K*2 bytes of
// activations exceed the LDS): its 8 waves split the k range, each running gemv_mfma_kernel's iteration (lane group g <-> word
// row i+g, sign fragments from the LUT, base + per-mask accumulators on the matrix pipe), and the 8 partial tiles are summed
// through LDS in wave order.  N/16 blocks fill the chip from N = 4096 up; column groups are XCD-remapped so the two 16-column
// blocks that share every 128-byte line of sign words run on the same XCD.
template <int DT, int NM, bool HASW, int LC, int PER>
__global__ void __launch_bounds__(512) gemv_col16_kernel(const GemvParams p) {
    constexpr int NW = 8;
    // LC = copies of the sign LUT.  LC = 16 (used when 64 KiB more LDS is free): entry e of copy c sits at e*256 + c*16 and lane l
    // reads copy l & 15, so the 16 lanes of every ds_read_b128 service group hit 16 different 16-byte slots of the 256-byte bank
    // row whatever their bytes are -- conflict-free (random bytes on ONE table collide ~3-way, the measured 0.65 us per tenant per
    // 4096^2 mask on top of the HBM time).  Here the table is built once per block of 8 waves under the first loads' latency.
    constexpr int LUT_BYTES = 4096 * LC;
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];     // [LUT][R rows x xrow bytes of activations]
    __shared__ float red[NW][64][8];                                     // per wave, per lane: 4 base + 4 delta partials
    char* const xs_lds = dyn_lds + LUT_BYTES;
    const int xrow = p.kslice * 2 + 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int n0 = xcd_remap(blockIdx.x, gridDim.x) * 16, ks = blockIdx.y;
    const int k_lo = ks * p.kslice, k_hi = min(p.K, k_lo + p.kslice);
    const int i0 = k_lo >> 5, i_hi = k_hi >> 5, i_last = (p.K >> 5) - 1;
    const int nit_blk = (i_hi - i0 + 3) >> 2;                            // 128-k iterations in this block's slice
    const int per = (nit_blk + NW - 1) / NW;
    const int it_lo = min(wave * per, nit_blk), it_hi = min(it_lo + per, nit_blk);
    const int nmask = p.sPb == 0 ? 1 : p.B;

    const int nw = min(n0 + li, p.N - 1);
    const unsigned short* wr = HASW ? p.W + (long long)nw * p.ldw : nullptr;
    const uint32_t* pw[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) pw[t] = p.P + (long long)min(t, nmask - 1) * p.sPb + nw;

    struct Stage { u32x4_t wf[4]; uint32_t wd[NM]; };
    auto load_iter = [&](Stage& st, int it) {
        const int irow = min(i0 + 4 * it + g, i_last);
#pragma unroll
        for (int t = 0; t < NM; ++t) st.wd[t] = __builtin_nontemporal_load(pw[t] + (long long)irow * p.N);
        if constexpr (HASW) {
#pragma unroll
            for (int s = 0; s < 4; ++s) st.wf[s] = __builtin_nontemporal_load((const u32x4_t*)(wr + irow * 32 + 8 * s));
        }
    };
    // PER = 4 (every wave owns exactly 4 iterations: K = 4096 in one slice), straight-line: iterations 0-1 are in flight while the
    // activations are staged; iterations 2-3 are issued right after the barrier and land under the MFMAs of 0-1.  (vmcnt retires in
    // order, so anything issued BEFORE the staging loads must land before the staged data can be used: issuing all four up front
    // would serialise "everything arrives" -> "all compute".)  With one iteration of prefetch the base stream and the delta work of a
    // wave ran back to back (floor + base + delta; profiles/r01_decode_kernels.txt).  PER = 0: generic loop, one iteration ahead.
    constexpr int NST = PER == 4 ? 4 : 2;
    Stage st[NST];
    if constexpr (PER == 4) {
        load_iter(st[0], it_lo);
        load_iter(st[1], it_lo + 1);
    } else {
        if (it_lo < it_hi) load_iter(st[0], it_lo);
    }

    if (threadIdx.x < 256 || LC == 16) {   // sign LUT (LC = 16: both halves of the block write 8 copies each): entry e = the 8 (+-1.0) 16-bit values of byte e (bit j <-> k offset j)
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
        const int e = threadIdx.x & 255, half = threadIdx.x >> 8;
        u32x4_t v;
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = (((e >> (2 * d)) & 1) ? POS : NEG) | ((((e >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
        if constexpr (LC == 16) {
            // slot index = 16 e + c is linear in the thread id, so every ds_write_b128 stores 64 consecutive 16-byte slots
            // (one entry per 16 lanes): conflict-free.  (Thread = entry, 16 strided copies each, was 8-way conflicted: +1.5 us.)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int slot = threadIdx.x + 512 * j, ee = slot >> 4;
                u32x4_t w;
#pragma unroll
                for (int d = 0; d < 4; ++d) w[d] = (((ee >> (2 * d)) & 1) ? POS : NEG) | ((((ee >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
                *(u32x4_t*)(dyn_lds + slot * 16) = w;
            }
        } else {
            *(u32x4_t*)(dyn_lds + e * 16) = v;
        }
    }
    {   // activation slice -> LDS (rows < R only)
        const int kn = k_hi - k_lo;                                   // multiple of 32
        for (int idx = threadIdx.x; idx < p.R * (kn >> 3); idx += 512) {
            const int r = idx / (kn >> 3), c = idx - r * (kn >> 3);
            const int b = r / p.M, m = r - b * p.M;
            *(u32x4_t*)(xs_lds + r * xrow + c * 16) =
                *(const u32x4_t*)(p.X + (long long)b * p.sXb + (long long)m * p.sXm + k_lo + c * 8);
        }
    }
    __syncthreads();

    f32x4_t accB = {0.f, 0.f, 0.f, 0.f}, accD[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* xl = xs_lds + min(li, p.R - 1) * xrow + 64 * g;
    const uint32_t copy_off = (uint32_t)li * 16u;

    auto compute = [&](const Stage& cur, int it) {
        const uint32_t keep = (i0 + 4 * it + g < i_hi) ? 0xffffffffu : 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4_t xf = *(const u32x4_t*)(xl + it * 256 + 16 * s) & u32x4_t{keep, keep, keep, keep};
            if constexpr (HASW) accB = mfma16<DT>(cur.wf[s], xf, accB);
#pragma unroll
            for (int t = 0; t < NM; ++t) {
                uint32_t off;
                if constexpr (LC == 16) off = __builtin_amdgcn_perm(cur.wd[t], copy_off, 0x0c0c0400u + ((uint32_t)s << 8));   // byte s -> bits 8..15, copy slot -> bits 0..7
                else off = ((cur.wd[t] >> (8 * s)) & 0xffu) * 16u;
                const u32x4_t sf = *(const u32x4_t*)(dyn_lds + off);
                accD[t] = mfma16<DT>(sf, xf, accD[t]);
            }
        }
    };
    if constexpr (PER == 4) {
        load_iter(st[2], it_lo + 2);
        load_iter(st[3], it_lo + 3);
#pragma unroll
        for (int u = 0; u < 4; ++u) compute(st[u], it_lo + u);
    } else {
        for (int it = it_lo; it < it_hi; it += 2) {
            if (it + 1 < it_hi) load_iter(st[1], it + 1);
            compute(st[0], it);
            if (it + 1 < it_hi) {
                if (it + 2 < it_hi) load_iter(st[0], it + 2);
                compute(st[1], it + 1);
            }
        }
    }

    // lane (li, g) holds D[col = 4g + e][row = li] of this wave's k range: pick the accumulator of the row's mask, park both parts
    {
        const int b = min(li, p.R - 1) / p.M;
        const int bm = p.sPb == 0 ? 0 : b;
        f32x4_t d = accD[0];
#pragma unroll
        for (int t = 1; t < NM; ++t)
            if (bm == t) d = accD[t];
        *(f32x4_t*)&red[wave][lane][0] = accB;
        *(f32x4_t*)&red[wave][lane][4] = d;
    }
    __syncthreads();
    if (wave == 0 && li < p.R) {
        const int b = li / p.M;
        f32x4_t sb = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; ++w) {                                 // fixed wave order: deterministic
            sb += *(const f32x4_t*)&red[w][lane][0];
            sd += *(const f32x4_t*)&red[w][lane][4];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * g + e;
            if (n < p.N) {
                float v = sd[e];
                if (p.alpha) v *= p.alpha[(long long)b * p.sAlb + n / p.gsz];
                if constexpr (HASW) v += sb[e];
                if (p.KS == 1) store_out<DT>(p, li, n, v);
                else gemv_store_partial(&p.ws[((long long)ks * p.R + li) * p.N + n], v);
            }
        }
    }
}

template <int DT>
__global__ void __launch_bounds__(256) gemv_reduce_kernel(const GemvParams p) {
    const int n = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (n >= p.N) return;
    float s = 0.f;
    for (int k0 = 0; k0 < p.KS; k0 += 8) {                    // 8 independent loads in flight (a dependent chain of KS L2 round
        float v[8];                                           // trips made this kernel 4.6 us), summed in k order
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p.ws[((long long)min(k0 + j, p.KS - 1) * p.R + r) * p.N + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (k0 + j < p.KS) ? v[j] : 0.f;
    }
    store_out<DT>(p, r, n, s);
}

// Sum of the split-k partials of the fused tile kernel (bd_gemm_fx.h, GemmParams::ksplit): ws [B][KS][M][N] fp32 -> C [B][M][N].
// One thread per 4 consecutive columns; partials are read as float4 (N % 4 == 0 is guaranteed by the fast-path checks).
template <int DT>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, void* C, int B, int KS, int M, int N,
                                                             long long sCb, int sCm, int out_f32, int accumulate = 0) {
    const long long q4 = (long long)blockIdx.x * 256 + threadIdx.x;        // index of a float4 inside one [M][N] slab
    const int b = blockIdx.y;
    const long long per = (long long)M * N / 4;
    if (q4 >= per) return;
    const long long e = q4 * 4;
    const int m = (int)(e / N), n = (int)(e - (long long)m * N);
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KS; k0 += 4) {
        f32x4_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *(const f32x4_t*)(ws + ((long long)b * KS + min(k0 + j, KS - 1)) * M * N + e);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < KS) s += v[j];
    }
    const long long off = (long long)b * sCb + (long long)m * sCm + n;
    // accumulate: C holds the residual (`hidden = residual + proj(x)`): the Linear's output is rounded to the output type first and the
    // sum is rounded again -- the same two roundings as the tile kernels' residual epilogue (gemm_epilogue, res_mode)
    if (out_f32) {
        if (accumulate) s += *(const f32x4_t*)((const float*)C + off);
        *(f32x4_t*)((float*)C + off) = s;
    } else {
        uint32_t h0 = f32_to_half_bits<DT>(s[0]), h1 = f32_to_half_bits<DT>(s[1]);
        uint32_t h2 = f32_to_half_bits<DT>(s[2]), h3 = f32_to_half_bits<DT>(s[3]);
        if (accumulate) {
            const u32x2_t r = *(const u32x2_t*)((const unsigned short*)C + off);
            h0 = f32_to_half_bits<DT>(half_bits_to_f32<DT>(h0) + half_bits_to_f32<DT>(r[0] & 0xffffu));
            h1 = f32_to_half_bits<DT>(half_bits_to_f32<DT>(h1) + half_bits_to_f32<DT>(r[0] >> 16));
            h2 = f32_to_half_bits<DT>(half_bits_to_f32<DT>(h2) + half_bits_to_f32<DT>(r[1] & 0xffffu));
            h3 = f32_to_half_bits<DT>(half_bits_to_f32<DT>(h3) + half_bits_to_f32<DT>(r[1] >> 16));
        }
        *(u32x2_t*)((unsigned short*)C + off) = u32x2_t{h0 | (h1 << 16), h2 | (h3 << 16)};
    }
}

}  // namespace bd
