// Decode path, streaming form: ONE launch per Linear, one 8-wave block per CU, every weight byte in flight from the first cycle.
//
//   y[b,m,:] = x[b,m,:] . W^T + alpha[b] * (x[b,m,:] . S_b)        R = B*M <= 16 activation rows, W optional
//
// Reference call sites: DiffCompressModule.forward at decode (demo/demo_backend.py:93-98, M = 1, B = tenants) and
// BinaryDiff.forward (bitdelta/diff.py:33-39) with a few tokens.
//
// Why a third decode kernel (profiles/r01_decode_kernels.txt): gemv_kernel / gemv_mfma_kernel slice k over ~512 blocks and pay a
// second launch for the reduction; gemv_col16_kernel removed that launch but a wave still ran "stage x -> barrier -> load -> compute"
// back to back (4 short iterations per wave, one stage of prefetch), so the base stream and the sign work added up instead of
// overlapping, and a launch cost 4.5 us before the first useful byte.  Here:
//   * grid = one block of 8 waves per CU; block b owns the contiguous column range [b*cpb, (b+1)*cpb) (cpb ~ N/256, so every CU
//     streams the same number of bytes whatever N is: 24 columns of a fused q+k+v, 112 of a fused gate+up) and walks it in
//     16-column MFMA tiles; its 8 waves split k; a wave's (tile, k-iteration) pairs form ONE flat stream that is prefetched NS
//     stages deep ACROSS tile boundaries, so HBM never sees a per-tile ramp;
//   * nothing is staged before the weight stream starts: the activation fragments are loaded straight from global memory (L2) into
//     the MFMA operand layout as part of each stage (R rows x 64 B per load; the whole activation block is <= 0.5 MB and L2-resident),
//     so there is no "x -> LDS -> barrier" in front of the first weight load and no LDS budget that depends on K;
//   * every load is a raw buffer load: out-of-range rows / columns / k-groups return zero in hardware, which (a) removes every clamp
//     and keep-mask from the loop and (b) lets the stream run past its end with loads that touch no memory, so the loop body is
//     straight-line and hipcc's own s_waitcnt vmcnt(N) counting keeps NS-1 stages in flight while one is consumed;
//   * an iteration covers 4 word rows (128 k): lane group g = l >> 4 owns word row 4*it + g and MFMA step s covers the k-octets
//     {32 (4 it + g) + 8 s .. + 7} for every operand (W, x, signs), so a lane loads ONE sign word per mask and 64 contiguous bytes
//     of its W row and of its x row per iteration, with no cross-lane exchange (same operand mapping as gemv_mfma_kernel);
//   * sign fragment of step s = LUT[byte s of the word]: 256 entries x 16 B, 16 copies (entry e of copy c at e*256 + c*16, lane
//     reads copy l & 15) so the 16 lanes of every ds_read_b128 service group hit 16 different 16-byte slots: conflict-free whatever
//     the bytes are (one copy: ~3-way conflicts on random bytes = the measured 0.65 us per tenant per 4096^2 mask).  64 KiB, built
//     once per block while the first stages are in flight;
//   * D[col][row] = v_mfma_f32_16x16x32(W or S fragment, x fragment): one base accumulator + one per distinct mask; at the end of
//     a tile the 8 waves' partial tiles meet in LDS (double-buffered, one s_barrier per tile, loads stay in flight across it) and
//     wave (tile & 7) sums them in wave order -- deterministic, no atomics, no workspace, no second launch.
// Algorithmic bytes per launch: 2*N*K (W) + nmask*N*K/8 (signs) + 2*R*K (x) + out; HBM roofline.
#pragma once
#include "bd_gemv.h"
#include "bd_serving.h"

namespace bd {

typedef uint32_t bufrsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
// aux: 0 = default cache policy, 2 = nt (streamed once by ONE CU: the weight and sign streams)
template <int AUX> __device__ __forceinline__ u32x4_t buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, AUX));
}
template <int AUX> __device__ __forceinline__ uint32_t buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, AUX);
}

constexpr int STREAM_LUT_BYTES = 65536;
constexpr int STREAM_RED_BYTES = 2 * 8 * 64 * 8 * 4;          // [2 buffers][<= 8 waves][64 lanes][4 base + 4 delta] fp32
constexpr int STREAM_ALPHA_MAX = 512;                          // (row, scale group) pairs of one block kept in LDS
constexpr int STREAM_LDS_BYTES = STREAM_LUT_BYTES + STREAM_RED_BYTES + STREAM_ALPHA_MAX * 4;
constexpr int stream_red_bytes(int nw) { return 2 * nw * 64 * 8 * 4; }                            // what an nw-wave block uses of it
constexpr int STREAM_XS_OFF = STREAM_LUT_BYTES + stream_red_bytes(4) + STREAM_ALPHA_MAX * 4;      // XL kernels (4 waves): activation rows
// FG = 1 ("fine grid", round 6): ONE 16-column tile per block, blocks small enough for TWO per CU (<= 80 KB of LDS, <= 256 VGPRs), so a
//   launch whose tile count sits between one and two per CU (Mistral's fused q|k|v: 384 tiles on 256 CUs) runs as ONE round of single-tile
//   blocks instead of blocks that walk 1.5 tiles (the half-empty second tile cost a full tile's stages: 16 dependent stages per wave for
//   what is 12 stages of bytes).  What makes the footprint fit is the NIBBLE sign table: 16 entries x 8 bytes (4 sign values), entry e of
//   lane copy c = l & 31 at e * 256 + c * 8 -- the copy index selects the bank pair, the entry the LDS row, so the 32 lanes of a
//   ds_read_b64 service group touch 64 distinct banks whatever the nibbles are: conflict-free in 4 KiB instead of 64 KiB; a fragment is
//   two ds_read_b64 (same LDS cycles as one ds_read_b128) and four VALU instead of one.
// FG = 2 (round 6): TWO-PASS resident rows for launches whose activation rows do not fit LDS at once -- the down projection of a multi-tenant
//   step (6 x 14336 x 2 B = 172 KB).  One block per CU (512 registers), ONE tile per block, the nibble table (which is what makes room: 4 KiB
//   instead of 64), and each wave's contiguous k quarter cut in two halves: LDS holds, per row, the first halves of the four quarters, then --
//   after one pair of barriers in the middle of the stream -- the second halves.  The rows of BOTH passes are fetched at kernel start (the second
//   set waits in 96 registers), so the switch needs no memory wait and the weight prefetch runs straight through it.  A stage is then 4 W + 2
//   sign loads instead of 10 loads (no per-stage activation loads), as in the single-pass form.  Every wave accumulates exactly the k
//   iterations, in exactly the order, of the one-pass forms: bit-identical to them.
constexpr int STREAM_FG_LUT_BYTES = 4096;
constexpr int stream_lut_bytes(int fg) { return fg ? STREAM_FG_LUT_BYTES : 65536; }
constexpr int STREAM_FG_XS_OFF = STREAM_FG_LUT_BYTES + 2 * 4 * 64 * 8 * 4 + 512 * 4;      // FG kernels (4 waves): activation rows
constexpr int STREAM_FG_LDS_MAX = 80 * 1024;                   // two blocks per CU
constexpr uint32_t STREAM_OOB = 0x80000000u;                   // a byte offset that is out of range for every descriptor (extents are < 2 GiB:
                                                               // checked on the host) and cannot wrap when an immediate offset is added

// alpha * delta + base with the product rounded on its own: the same two roundings in every epilogue form (left to the compiler,
// one form was contracted to an fma and the other -- scale applied inside a branch -- was not: 1 ulp apart before the final rounding)
__device__ __forceinline__ float scale_then_add(float d, float a, float b) {
#pragma clang fp contract(off)
    const float m = d * a;
    return m + b;
}

#ifdef BD_STREAM_TRACE
// s_memtime stamps (harness builds only: tests/native/stream_tl.hip): [block][0] kernel entry, [1] prologue loads issued, [2] first barrier passed,
// [3] first stage consumed, [4] main loop done, [5] kernel exit; wave 0 of every block
__device__ unsigned long long g_stream_trace[1024][8];
#define BD_ST_STAMP(i) do { if (threadIdx.x == 0) g_stream_trace[blockIdx.x & 1023][i] = __builtin_readcyclecounter(); } while (0)
#else
#define BD_ST_STAMP(i) do { } while (0)
#endif
struct StreamParams {
    GemvParams g;
    int cpb;                 // columns per block (multiple of 4)
    uint32_t x_bytes, w_bytes, p_bytes;     // descriptor extents (bytes from the base pointers)
    // per-tenant dense weights (bd_tenant_linear: NM = 0, gridDim.y = tenants): element offsets of tenant blockIdx.y
    long long sXt, sWt, sCt;
    // sign-word addressing, in words:  word(row i, column n) = P[(n >> 4) * pts + i * prs + (n & 15)]
    //   reference layout [K/32, N]:            pts = 16,          prs = N
    //   tile-major layout [N/16, K/32, 16]:    pts = 16 * K/32,   prs = 16   (a 16-column tile's words are one contiguous run over k)
    uint32_t pts, prs;
    // packed decode layout (PK kernels): tenants interleaved, `tp` dwords per (tile, iteration, lane group, column) -- see PK below
    uint32_t tp;
    // fused RMSNorm prologue (XL kernels): X is the residual stream; the block normalises the R rows into LDS rows of `xrow` bytes
    // at LDS offset `xs_off` and reads its activation fragments from there.  nw = norm weight [tenants or 1, K], stride sNw elements
    const unsigned short* nw;
    long long sNw;
    uint32_t n_bytes, xs_off, xrow;
    int jsh;                 // K = 2048 << jsh
    float eps;
    int no_res_prefetch;     // A/B switch (bd_set_stream_tuning bit 10): 1 = the residual is read in the epilogue, as before round 4
    // RMSNorm HAND-OFF between two launches of a decoder layer (round 5; removes the stand-alone rmsnorm launch without making every block
    // re-reduce the rows):
    //   ssq_out (producer: o / down with the residual epilogue, 16-bit output): the wave that finishes a 16-column tile also writes
    //     sum over the tile's 16 columns of (stored value)^2 per row to ssq_out[tile][16 rows]  (tile = n / 16; fixed summation order);
    //   ssq_in  (consumer: XL = 3): X is the producer's pre-multiplied copy round(x (.) norm_w); the block copies it into LDS, sums the K/16
    //     partials of each row in a fixed order, and applies rsqrt(mean + eps) as a per-row SCALAR to its accumulators in the epilogue:
    //     W . (nw (.) x) * rs  instead of  W . (nw (.) round(x * rs)) -- the same value up to the position of one rounding.
    const float* ssq_in;
    float* ssq_out;
    // ... and the producer can also leave the PRE-MULTIPLIED copy behind: xw_out[row][n] = round(stored value * nw_next[row][n]) with nw_next
    // the weight of the RMSNorm that follows (same strides as C; nw_next [tenants or 1, N], stride sNwNext elements).  The consumer (XL = 3
    // with nw == nullptr) then reads exactly the bytes the resident-row form reads today -- no norm-weight rows, no multiply -- and only
    // scales its accumulators.
    const unsigned short* nw_next;
    long long sNwNext;
    unsigned short* xw_out;
    // ... and scale the sums of squares it writes (round 6): with nw_next = nw / s for a power of two s >= max |nw| (the host divides
    // once; exact), xw_out = round16(x . nw / s) can never exceed |x| -- no fp16 overflow on massive-activation rows whatever the norm weight --
    // and ssq_scale = 1 / s^2 together with eps / s^2 on the consumer makes its row scalar s . rsqrt(mean(x^2) + eps): the same product, exactly.
    float ssq_scale;
};

// NW = waves per block (8: two per SIMD, 256 VGPRs each; 4: one per SIMD, the whole register file, deeper prefetch).
// WNAT = 1: the base weight and its activation fragments use the NATURAL k order -- MFMA step s of lane group g covers
//   k = 128 it + 32 s + 8 g .. + 7, so one W load instruction reads 16 rows x 64 contiguous bytes (every 64-byte sector fetched by
//   exactly one instruction); the sign operand keeps the word-row order (a lane's word is 32 consecutive k), which needs its own
//   activation fragments: x is loaded in both orders (L2 hits).  WNAT = 0: one activation fragment set, W in the word-row order
//   (each W instruction touches 64 sectors and uses 16 bytes of each; the 4 instructions of a stage complete them).
// AUX = cache policy of the streams: bit 1 (value 2) = nt on the base-weight loads, bit 2 (value 4) = nt on the sign loads.  Measured
//   (profiles/r02_decode_stream_ab.txt): with the word-row order nt costs 35 % of the pure weight stream (3.4 vs 5.2 TB/s at 235 MB): the
//   4 load instructions of a stage each use 16 bytes of the same 64-byte sectors, and a non-temporal line does not stay in L1 for the
//   next one.  With the TILE-MAJOR weight (WT) every load instruction reads its own contiguous 1-KiB run, and the pure-stream probe
//   (profiles/r04_stream_probe.txt) has nt register loads at 6.56 TB/s against 5.66 default; the packed sign dwords of a lane still
//   span two load instructions, so the sign stream keeps the default policy.
// PK = 1: PACKED sign layout, everything in the natural k order with ONE activation fragment set.  The serving side repacks a tenant
//   set's masks once (binary_gemm_kernel.pack_decode_masks) into  P[tile n/16][iteration k/128][lane group g][column n%16][tenant t]
//   dwords whose byte s holds the 8 signs of k = 128 it + 32 s + 8 g .. + 7 (a 4 x 4 byte transpose of the 4 word rows of an
//   iteration), so that (a) byte s of a lane's dword is exactly the sign fragment of natural MFMA step s, (b) W and x are read with
//   the sector-friendly natural pattern (16 rows x 64 contiguous bytes per instruction), (c) the NM tenants' dwords of a lane are
//   adjacent: one or two wide loads instead of NM dword loads.  Why: the PMC passes of the word-row kernel
//   (profiles/r02_decode_pmc.txt) show its waves 37 % issue-stalled and 35 % busy -- one wave per SIMD, and 14 load instructions per
//   5.5 KB stage that each touch 32-64 cache lines -- not 70 % waiting on memory as an HBM-bound kernel should be.
// XL = 1 (packed layout, 4-wave blocks): FUSED RMSNorm.  X is the un-normalised residual stream h [R rows, K]; every block computes
//   w[tenant] * round16(h * rsqrt(mean(h^2) + eps)) for all R rows itself (R*K*2 bytes from L2 -- 48 KB for 6 tenants of a 4096-wide
//   model -- while its first weight stages are in flight), keeps the result in LDS and reads its activation fragments from there.
//   Same arithmetic, same order as rmsnorm_tenant_kernel (bd_serving.h: shared helpers), so the result is bit-identical to
//   "rmsnorm launch, then Linear launch"; what disappears is a 4 us launch + gap in front of two of the four Linears of a layer.
// XL = 2 (packed layout, 4-wave blocks): the same resident-activation machinery WITHOUT the norm -- the R activation rows are copied
//   into LDS once (R * K * 2 bytes from L2, behind the first weight stages) and every stage reads its fragments from there.  A stage
//   is then 4 W loads + the sign loads: no per-stage activation loads (4 of the 10 load instructions of a 6-tenant stage, a quarter of
//   the bytes through the texture path), which is what lets NS = 8 stages fit the 6-bit vmcnt counter: twice the weight bytes in flight.
// XL = 3 (packed layout, 4-wave blocks): RMSNorm by HAND-OFF (see StreamParams::ssq_in): XL = 2's resident rows -- X is the copy of the
//   residual stream the PRODUCING launch pre-multiplied by the norm weight (xw_out) -- and the row's 1/rms, from the partial sums of squares
//   that launch left behind, applied in the epilogue.  No per-block reduction over the rows, no extra barrier, nothing to normalise or
//   multiply element by element; 16 registers more than XL = 2.  (A form that multiplied raw rows by the norm weight itself -- 64 more
//   registers of norm-weight chunks, 96 KB per block from L2 -- ran the kernel at 256 VGPRs and lost what the removed launch gave.)
// EPI = 1 (packed layout): SwiGLU epilogue for a fused gate|up projection whose output rows are interleaved in blocks of 8
//   ([g0..7 | u0..7 | g8..15 | ...]): a 16-column tile holds 8 gate and the 8 matching up columns, the reducing wave rounds both to
//   16 bits (what the separate Linear would have stored), and stores round16(silu(g)) * u -- N/2 output columns.  Scale group of
//   a column = gate (0) or up (1).  Bit-identical to "Linear launch, then swiglu launch".
// WT = 1 (packed layout): the base weight is TILE-MAJOR too -- W'[n/16][k/128][s][n%16][g][8] with k = 128 it + 32 s + 8 g + e (the
//   serving side repacks it once, binary_gemm_kernel.tile_weight): the four load instructions of a stage read four consecutive
//   1-KiB runs of ONE contiguous 4-KiB block, and a wave's consecutive stages consecutive blocks, instead of 16 rows 2K bytes apart.
template <int DT, int NM, bool HASW, int NS, int NW = 4, int WNAT = 0, int AUX = 0, int PK = 0, int XL = 0, int EPI = 0, int WT = 0, int FG = 0>
__global__ void __launch_bounds__(64 * NW, FG == 1 ? 2 : 1) gemv_stream_kernel(const StreamParams sp) {
    static_assert(!WT || (PK && HASW), "tile-major W: packed layout");
    static_assert(!FG || (PK && WT && NW == 4 && (XL == 2 || XL == 3)), "fine grid: resident-row forms, packed layout, tile-major W, 256-thread blocks");
    static_assert(FG != 2 || XL == 2, "two-pass rows: the plain resident-row form");
    constexpr int LUTB = stream_lut_bytes(FG);
    static_assert(!PK || (WNAT == 1 && NM > 0), "packed layout = natural order, with a sign operand");
    static_assert(!(XL || EPI) || (PK && (NW == 4 || (XL == 2 && NW == 8))), "fused prologue / epilogue: packed layout, 256-thread blocks (resident rows: 512 too)");
    static_assert(XL >= 0 && XL <= 3, "activation forms");
    static_assert(!XL || NS % 2 == 0, "stage parity selects the activation fragment set");
    constexpr int AUXW = AUX & 2, AUXP = (AUX & 4) ? 2 : 0;
    BD_ST_STAMP(0);
    GemvParams p = sp.g;
    if constexpr (XL != 0) p.M = 1;         // the resident / normalised-row forms are decode launches with one row per tenant (host-checked): a constant
                                            // lets the compiler fold the ~8 run-time divisions by M out of the prologue (row -> (tenant, row) maps)
    if constexpr (NM == 0) {       // blockIdx.y = tenant: its own activation rows, weight matrix and output rows
        p.X += (long long)blockIdx.y * sp.sXt;
        p.W += (long long)blockIdx.y * sp.sWt;
        p.C = (char*)p.C + (long long)blockIdx.y * sp.sCt * (p.out_f32 ? 4 : 2);
    }
    constexpr int NMA = NM > 0 ? NM : 1;      // array extent (no zero-length arrays)
    constexpr bool XP = !PK && (NM > 0 || !(HASW && WNAT));   // activation fragments in word-row order (sign operand; W too unless WNAT)
    constexpr bool XN = PK || (HASW && WNAT);                  // activation fragments in natural order (W operand; signs too when PK)
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];     // [64 KiB sign LUT][32 KiB reduction buffers]
    float* const red = (float*)(dyn_lds + LUTB);
    float* const a_lds = (float*)(dyn_lds + LUTB + stream_red_bytes(NW));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);                    // neighbouring column ranges share sign-word lines: same XCD
    const int c_lo = blk * sp.cpb, c_hi = min(p.N, c_lo + sp.cpb);
    const int ntile = (c_hi - c_lo + 15) >> 4;
    const int nrow = p.K >> 5;                                           // word rows
    const int nit = (nrow + 3) >> 2;                                     // 128-k iterations over all of k
    const int per = (nit + NW - 1) / NW;
    const int it_lo = min(wave * per, nit), it_hi = min(it_lo + per, nit);
    // (a wave whose k range is empty -- K < 1024 -- still walks one all-zero stage per tile, so that every wave meets every barrier)
    const int nmask = p.sPb == 0 ? 1 : p.B;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.X, sp.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(HASW ? (const void*)p.W : (const void*)p.X, HASW ? sp.w_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(NM > 0 ? (const void*)p.P : (const void*)p.X, NM > 0 ? sp.p_bytes : 0u);

    // activation row of MFMA column li (rows >= R: out of range -> zeros; their D columns are never stored)
    uint32_t x_off;
    {
        const int b = li / p.M, m = li - b * p.M;
        x_off = li < p.R ? (uint32_t)(((long long)b * p.sXb + (long long)m * p.sXm) * 2) : STREAM_OOB;
    }

    // The scales this block can need -- (row, scale group) for the groups its column range touches -- are fetched by the first threads
    // BEFORE the stream starts: the oldest entry of the wave's in-order load queue, so waiting for it never drains a weight load.
    // (A load issued at the end of a tile would be the youngest: s_waitcnt vmcnt(0), the whole prefetch lost once per tile.)
    const int g0 = EPI ? 0 : c_lo / p.gsz, ng = EPI ? 2 : (c_hi - 1) / p.gsz - g0 + 1;
    constexpr int ALPHA_CAP = XL == 3 ? STREAM_ALPHA_MAX - 64 : STREAM_ALPHA_MAX;      // (XL = 3 keeps its row sums in the last 64 slots)
    const bool al_lds = p.alpha != nullptr && p.R * ng <= (ALPHA_CAP < 64 * NW ? ALPHA_CAP : 64 * NW);
    float a_pre = 0.f;
    if (al_lds) {
        const int idx = min((int)threadIdx.x, p.R * ng - 1), r = idx / ng, j = idx - r * ng;
        a_pre = p.alpha[(long long)(r / p.M) * p.sAlb + g0 + j];
    }

    // Residual epilogue (`accumulate`: o / down of a decoder layer add onto the residual stream): the values this lane will add in the
    // FIRST tile its wave reduces (tile = wave index) are fetched now, for the same reason -- in the epilogue the load would be a full,
    // exposed memory round trip (the residual was written by the previous launch, usually from another XCD) and the youngest entry of
    // the queue.  Later tiles of the same wave (wide residual launches) fall back to the load in store_out.
    // (Not in the norm-prologue form: no caller adds a residual there, and that form's instantiations stay the code the round-3/4 test
    // matrix has run against.)
    constexpr bool CPRE = EPI == 0 && XL != 1;
    [[maybe_unused]] uint32_t c_pre[4] = {0u, 0u, 0u, 0u};               // raw bits: converted where they are used, so nothing waits here
    const bool c_pre_ok = CPRE && p.accumulate && wave < ntile && !sp.no_res_prefetch;
    if (CPRE && c_pre_ok && li < p.R) {
        const int b = li / p.M, m = li - b * p.M;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = min(c_lo + wave * 16 + 4 * g + e, p.N - 1);
            const long long off = (long long)b * p.sCb + (long long)m * p.sCm + n;
            c_pre[e] = p.out_f32 ? ((const uint32_t*)p.C)[off] : (uint32_t)((const unsigned short*)p.C)[off];
        }
    }

    // hand-off producer with a pre-multiplied copy: the next norm's weights for the columns this lane stores in its first tile, fetched now
    // for the same reason as c_pre
    [[maybe_unused]] uint32_t nw_pre[4] = {0u, 0u, 0u, 0u};
    if constexpr (CPRE) {
        if (sp.xw_out && wave < ntile && li < p.R) {
            const int b = li / p.M;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = min(c_lo + wave * 16 + 4 * g + e, p.N - 1);
                nw_pre[e] = (uint32_t)sp.nw_next[(long long)b * sp.sNwNext + n];
            }
        }
    }

    // XL: the R raw rows (and their norm weights) are the OLDEST loads of the wave -- like the scales above, consuming them never
    // waits for a weight stage.  Thread t owns the 16-byte chunks c = 8 t + 2048 i of every row: rmsnorm_tenant_kernel's mapping.
    constexpr int XCH = FG == 2 ? 24 : NW == 8 ? 8 : 16;                 // chunks per thread: R * K <= 16 * 2048 (FG = 2: R * K / 2 <= 24 * 2048; host-checked)
    constexpr int XNT = 64 * NW;                                         // threads that share the copy
    [[maybe_unused]] u32x4_t xraw[XL ? XCH : 1], graw[XL == 1 ? XCH : 1];
    [[maybe_unused]] u32x4_t xraw1[FG == 2 ? XCH : 1];                   // FG = 2: the second pass's rows, fetched now, written to LDS at the switch
    // XL = 3: this thread's share of the producer's partial sums of squares, rows 0..7 (host-checked: R <= 8); tiles t, t + 256, ...
    // Raw loads only, NO arithmetic here: a sum inside a loop makes the compiler wait for each load where it is issued -- a full memory
    // round trip (the partials were just written by another launch, from other XCDs) in front of the first weight load: +5 us on the
    // q|k|v launch (rocprofv3 kernel trace, profiles/r05_decode_step.txt).  Buffer loads: tiles past K / 16 read as zero, no branch.
    constexpr int SSQ_T = 2;                                             // tiles per thread: K <= 16 * 256 * 2 (host-checked)
    [[maybe_unused]] u32x4_t ssq_raw[XL == 3 ? SSQ_T : 1][2];
    if constexpr (XL == 3) {
        const __amdgpu_buffer_rsrc_t rq = make_rsrc(sp.ssq_in, (uint32_t)(p.K >> 4) * 64u);
#pragma unroll
        for (int i = 0; i < SSQ_T; ++i) {
            const uint32_t off = ((uint32_t)threadIdx.x + (uint32_t)XNT * i) * 64u;
            ssq_raw[i][0] = buf_load16<0>(rq, off);
            ssq_raw[i][1] = p.R > 4 ? buf_load16<0>(rq, off + 16u) : u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    [[maybe_unused]] const int jsh = sp.jsh;                             // log2(chunks per row per thread): K = 2048 << jsh (host-checked)
    if constexpr (XL) {
        const __amdgpu_buffer_rsrc_t rn = make_rsrc(sp.nw, sp.n_bytes);
        // XL = 2 / 3: chunk q = thread + XNT j of the flat [R][K / 8] array.  (row, chunk) advance INCREMENTALLY -- one division here instead of one per
        // chunk: written as r = q / kc inside the loop, the 16 run-time divisions put ~870 instructions (~1.7 us on a one-wave-per-SIMD kernel) in
        // front of the first weight load of the launch (read in the ISA, round 5)
        [[maybe_unused]] const int kc = p.K >> 3;
        [[maybe_unused]] int xr_ = (int)threadIdx.x / kc, xc_ = (int)threadIdx.x - xr_ * kc;
        if constexpr (FG == 2) {
            // chunk q = thread + 256 j of the flat [R][4 wave quarters][K / 64 chunks of one half quarter]; pass 1 = the same chunk, K / 8 elements on
            const int khc = p.K >> 6;                                    // 16-byte chunks of a half quarter (>= 128: host-checked)
            int off = (int)threadIdx.x, seg = 0, r = 0;
            while (off >= khc) { off -= khc; ++seg; }                    // (thread < 256 <= 2 khc: at most two steps)
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const bool ok = r < p.R;
                const uint32_t e = ok ? (uint32_t)(((long long)r * p.sXb + (long long)seg * (p.K >> 2) + off * 8) * 2) : STREAM_OOB;
                xraw[j] = buf_load16<0>(rx, e);
                off += XNT;
#pragma unroll
                for (int w_ = 0; w_ < 2; ++w_) {
                    const bool wrap = off >= khc;
                    off -= wrap ? khc : 0;
                    seg += wrap ? 1 : 0;
                }
                const bool wr = seg >= 4;
                seg -= wr ? 4 : 0;
                r += wr ? 1 : 0;
            }
        } else
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            // XL = 1: K = 2048 << jsh, a thread's chunks of one row are consecutive j (the row sums need that).  XL = 2: any K % 8 == 0 --
            // chunk q = thread + 256 j of the flat [R][K / 8] array (the same chunks as the line above when K is a power of two)
            int r, c;
            if constexpr (XL == 2 || XL == 3) {
                r = xr_; c = xc_ * 8;
                xc_ += XNT;
#pragma unroll
                for (int w_ = 0; w_ < XNT / 128; ++w_) {                 // (K >= 1024, host-checked: at most XNT / 128 wraps; selects, no branch)
                    const bool wrap = xc_ >= kc;
                    xc_ -= wrap ? kc : 0;
                    xr_ += wrap ? 1 : 0;
                }
            }
            else { r = j >> jsh; c = ((int)threadIdx.x + 256 * (j - (r << jsh))) * 8; }      // M == 1 (host-checked): row = tenant
            const bool ok = r < p.R;
            xraw[j] = buf_load16<0>(rx, ok ? (uint32_t)(((long long)r * p.sXb + c) * 2) : STREAM_OOB);
            if constexpr (XL == 1) graw[j] = buf_load16<0>(rn, ok ? (uint32_t)(((long long)r * sp.sNw + c) * 2) : STREAM_OOB);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    struct Stage { u32x4_t xf[XP ? 4 : 1]; u32x4_t xn[(XN && !XL) ? 4 : 1]; u32x4_t wf[4]; uint32_t wd[NMA]; };
    // one stage = (tile, iteration): 4 x 16 B of the lane's x row, 4 x 16 B of its W row, one sign word per mask
    auto issue = [&](Stage& st, int tile, int it) {
        const int irow = 4 * it + g;                                     // this lane group's word row
        const bool krow_ok = tile < ntile && irow < nrow;
        const int n = c_lo + tile * 16 + li;
        const bool col_ok = n < c_hi;
        if constexpr (XP) {
            const uint32_t xo = krow_ok ? x_off + (uint32_t)irow * 64u : STREAM_OOB;
#pragma unroll
            for (int s = 0; s < 4; ++s) st.xf[s] = buf_load16<0>(rx, xo + 16u * s);
        }
        if constexpr (XN) {
            // natural order: step s, lane group g -> k = 128 it + 32 s + 8 g; a k-octet past K is out of range by itself
            // (the row extents of x and W end at K), except that x rows are contiguous: guard the octet explicitly
            const int k0 = 128 * it + 8 * g;
            const bool it_ok = tile < ntile;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool ok = it_ok && (k0 + 32 * s < p.K);
                if constexpr (!XL) st.xn[s] = buf_load16<0>(rx, ok ? x_off + (uint32_t)(k0 + 32 * s) * 2u : STREAM_OOB);
                if constexpr (HASW && WT)
                    st.wf[s] = buf_load16<AUXW>(rw, (it_ok && it < nit && col_ok)
                                                       ? ((uint32_t)(n >> 4) * (uint32_t)nit + (uint32_t)it) * 4096u + (uint32_t)s * 1024u +
                                                             (uint32_t)(n & 15) * 64u + (uint32_t)g * 16u
                                                       : STREAM_OOB);
                else if constexpr (HASW)
                    st.wf[s] = buf_load16<AUXW>(rw, (ok && col_ok) ? (uint32_t)n * (uint32_t)p.ldw * 2u + (uint32_t)(k0 + 32 * s) * 2u : STREAM_OOB);
            }
        } else if constexpr (HASW) {
            const uint32_t wo = (krow_ok && col_ok) ? (uint32_t)n * (uint32_t)p.ldw * 2u + (uint32_t)irow * 64u : STREAM_OOB;
#pragma unroll
            for (int s = 0; s < 4; ++s) st.wf[s] = buf_load16<AUXW>(rw, wo + 16u * s);
        }
        if constexpr (PK) {
            // dword index ((tile_g * nit + it) * 4 + g) * 16 + (n & 15), tp dwords each; iterations past K are zero padding in the pack
            const bool ok = tile < ntile && it < nit && col_ok;
            const uint32_t po = ok ? ((((uint32_t)(n >> 4) * (uint32_t)nit + (uint32_t)it) * 4u + (uint32_t)g) * 16u + (uint32_t)(n & 15)) * sp.tp * 4u
                                   : STREAM_OOB;
            if constexpr (NM == 1) {
                st.wd[0] = buf_load4<AUXP>(rp, po);
            } else if constexpr (NM == 2) {
                const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rp, (int)po, 0, AUXP));
                st.wd[0] = v[0]; st.wd[1] = v[1];
            } else {
                // NM / 4 full 16-byte loads of the lane's adjacent tenant dwords (+ one 8-byte load when NM % 4 == 2: t_pad = 6)
#pragma unroll
                for (int q4 = 0; q4 < NM / 4; ++q4) {
                    const u32x4_t v = buf_load16<AUXP>(rp, po == STREAM_OOB ? STREAM_OOB : po + 16u * q4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) st.wd[4 * q4 + t] = v[t];
                }
                if constexpr (NM % 4 == 2) {
                    const u32x2_t v2 = __builtin_bit_cast(
                        u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rp, (int)(po == STREAM_OOB ? STREAM_OOB : po + 16u * (NM / 4)), 0, AUXP));
                    st.wd[NM - 2] = v2[0]; st.wd[NM - 1] = v2[1];
                }
            }
        } else {
            [[maybe_unused]] const uint32_t po =
                (krow_ok && col_ok) ? ((uint32_t)(n >> 4) * sp.pts + (uint32_t)irow * sp.prs + (uint32_t)(n & 15)) * 4u : STREAM_OOB;
#pragma unroll
            for (int t = 0; t < NM; ++t) {
                const uint32_t tb = (uint32_t)(min(t, nmask - 1)) * (uint32_t)p.sPb * 4u;
                st.wd[t] = buf_load4<AUXP>(rp, po == STREAM_OOB ? STREAM_OOB : po + tb);
            }
        }
        // the stages must enter the load queue in stream order: without this fence hipcc clusters the loads of ALL prologue stages
        // by descriptor (every x, then every W, then every sign word), and stage 0 cannot be consumed before nearly all of them land
        __builtin_amdgcn_sched_barrier(0);
    };

    // flat (tile, iteration) stream of this wave
    int ti = 0, ii = it_lo;                                              // next stage to issue
    auto advance = [&](int& t, int& i) { if (++i >= it_hi) { i = it_lo; ++t; } };
    Stage st[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        issue(st[u], ti, ii);
        advance(ti, ii);
    }
    if constexpr (FG == 2) {
        // the SECOND pass's rows go out behind the prologue's weight stages: they are not needed before the middle of the stream, and in front of
        // the weight loads (returns are in issue order) their 24 L2 round trips per thread delayed the first weight byte of the launch
        const __amdgpu_buffer_rsrc_t rx1 = make_rsrc(p.X, sp.x_bytes);
        const int khc = p.K >> 6;
        int off = (int)threadIdx.x, seg = 0, r = 0;
        while (off >= khc) { off -= khc; ++seg; }
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            xraw1[j] = buf_load16<0>(rx1, r < p.R ? (uint32_t)(((long long)r * p.sXb + (long long)seg * (p.K >> 2) + (p.K >> 3) + off * 8) * 2) : STREAM_OOB);
            off += XNT;
#pragma unroll
            for (int w_ = 0; w_ < 2; ++w_) {
                const bool wrap = off >= khc;
                off -= wrap ? khc : 0;
                seg += wrap ? 1 : 0;
            }
            const bool wr = seg >= 4;
            seg -= wr ? 4 : 0;
            r += wr ? 1 : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    BD_ST_STAMP(1);
    if constexpr (NM > 0 && FG) {   // nibble sign table: thread t writes entry t >> 4 for the lane copies 2 (t & 15), 2 (t & 15) + 1 (one ds_write_b128)
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
        const int ee = (int)threadIdx.x >> 4;
        u32x4_t w;
#pragma unroll
        for (int d = 0; d < 2; ++d) w[d] = w[d + 2] = (((ee >> (2 * d)) & 1) ? POS : NEG) | ((((ee >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
        *(u32x4_t*)(dyn_lds + threadIdx.x * 16) = w;
    } else if constexpr (NM > 0) {   // sign LUT, 16 copies: slot index = 16 e + c is linear in the thread id -> every ds_write_b128 stores 64 consecutive slots
        constexpr uint32_t POS = One2<DT>::v & 0xffffu, NEG = POS | 0x8000u;
#pragma unroll
        for (int j = 0; j < 4096 / (64 * NW); ++j) {
            const int slot = threadIdx.x + 64 * NW * j, ee = slot >> 4;
            u32x4_t w;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = (((ee >> (2 * d)) & 1) ? POS : NEG) | ((((ee >> (2 * d + 1)) & 1) ? POS : NEG) << 16);
            *(u32x4_t*)(dyn_lds + slot * 16) = w;
        }
    }
    // FG = 2: (row, quarter, chunk) -> LDS byte offset of this thread's chunk j; the same positions serve both passes
    [[maybe_unused]] auto rows_to_lds2 = [&](const u32x4_t (&src)[FG == 2 ? XCH : 1]) {
        if constexpr (FG == 2) {
            const int khc = p.K >> 6;
            int off = (int)threadIdx.x, seg = 0, r = 0;
            while (off >= khc) { off -= khc; ++seg; }
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                if (r < p.R) *(u32x4_t*)(dyn_lds + sp.xs_off + (uint32_t)r * sp.xrow + (uint32_t)(seg * khc + off) * 16u) = src[j];
                off += XNT;
#pragma unroll
                for (int w_ = 0; w_ < 2; ++w_) {
                    const bool wrap = off >= khc;
                    off -= wrap ? khc : 0;
                    seg += wrap ? 1 : 0;
                }
                const bool wr = seg >= 4;
                seg -= wr ? 4 : 0;
                r += wr ? 1 : 0;
            }
        }
    };
    if constexpr (FG == 2) rows_to_lds2(xraw);
    else if constexpr (XL == 2 || XL == 3) {           // rows -> LDS (same thread mapping as the norm form)
        const int kc = p.K >> 3;
        int r = (int)threadIdx.x / kc, cq = (int)threadIdx.x - r * kc;       // (incremental, as in the load loop above)
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int c = cq * 8;
            const int r_ = r;
            cq += XNT;
#pragma unroll
            for (int w_ = 0; w_ < XNT / 128; ++w_) {
                const bool wrap = cq >= kc;
                cq -= wrap ? kc : 0;
                r += wrap ? 1 : 0;
            }
            if (r_ < p.R) {
                *(u32x4_t*)(dyn_lds + sp.xs_off + (uint32_t)r_ * sp.xrow + (uint32_t)c * 2u) = xraw[j];   // (XL = 3: X is the producer's xw_out)
            }
        }
    }
    if constexpr (XL == 3) {
        // per-row sums of the producer's partials: thread -> wave (fixed butterfly) -> 4 wave partials in LDS; the epilogue adds the four
        // in wave order (rms_scale).  The slots sit at the end of the scale area (host: R * scale groups <= STREAM_ALPHA_MAX - 64).
        float* const rpart = a_lds + (STREAM_ALPHA_MAX - 64);
        f32x4_t ssq_lo = {0.f, 0.f, 0.f, 0.f}, ssq_hi = {0.f, 0.f, 0.f, 0.f};     // tiles t, t + 256, ... in this fixed order
#pragma unroll
        for (int i = 0; i < SSQ_T; ++i) {
            ssq_lo += __builtin_bit_cast(f32x4_t, ssq_raw[i][0]);
            ssq_hi += __builtin_bit_cast(f32x4_t, ssq_raw[i][1]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < p.R) {                                    // (wave-uniform)
                const float w = wave_sum(r < 4 ? ssq_lo[r & 3] : ssq_hi[r & 3]);
                if (lane == 0) rpart[r * 4 + wave] = w;
            }
        }
    }
    if constexpr (XL == 1) {
        float* const part = red;                  // [row][wave] partial sums (the reduction buffers are idle until the first tile ends)
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            ss = sumsq8<DT>(xraw[j], ss);
            if (((j + 1) & ((1 << jsh) - 1)) == 0) {          // row complete (uniform)
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
            if (r < p.R) {
                const float rs = rms_scale(part[r * 4], part[r * 4 + 1], part[r * 4 + 2], part[r * 4 + 3], p.K, sp.eps);
                *(u32x4_t*)(dyn_lds + sp.xs_off + (uint32_t)r * sp.xrow + (uint32_t)c * 2u) = norm8<DT>(xraw[j], graw[j], rs);
            }
        }
    }
    if (al_lds && (int)threadIdx.x < p.R * ng) a_lds[threadIdx.x] = a_pre;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    BD_ST_STAMP(2);
    f32x4_t accB = {0.f, 0.f, 0.f, 0.f}, accD[NMA];
#pragma unroll
    for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t copy_off = (uint32_t)li * 16u;

    // LUT address of step s of word w: byte s -> bits 8..15, copy slot -> bits 0..7, i.e. byte * 256 + (l & 15) * 16  (one v_perm_b32)
    [[maybe_unused]] const uint32_t copy8 = (uint32_t)(lane & 31) * 8u;
    auto lut = [&](uint32_t w, int s) -> u32x4_t {
        if constexpr (FG) {                 // nibble table: low nibble -> sign values 0..3 (dwords 0, 1), high nibble -> 4..7 (dwords 2, 3)
            const uint32_t lo = (__builtin_amdgcn_ubfe(w, 8u * s, 4u) << 8) | copy8, hi = (__builtin_amdgcn_ubfe(w, 8u * s + 4u, 4u) << 8) | copy8;
            const u32x2_t a = *(const u32x2_t*)(dyn_lds + lo), b = *(const u32x2_t*)(dyn_lds + hi);
            return u32x4_t{a[0], a[1], b[0], b[1]};
        } else {
            const uint32_t off = __builtin_amdgcn_perm(w, copy_off, 0x0c0c0400u + ((uint32_t)s << 8));
            return *(const u32x4_t*)(dyn_lds + off);
        }
    };
    // The sign fragments are read one MFMA step ahead and the steps are fenced: left alone, hipcc hoists all 4*NM ds_read_b128 of a
    // stage to its top (16*NM VGPRs of fragments in flight; with NS stages of loads in registers that spilled the in-flight load
    // destinations to scratch, and scratch traffic shares vmcnt with the weight stream).  One step of NM + 1 MFMAs (>= 110 cycles of
    // matrix pipe) covers the LDS latency of the next step's reads.
    // XL: activation fragment of step s of iteration `it` = 16 bytes at k = 128 it + 32 s + 8 g of LDS row min(li, R-1); the rows are
    // 2K + 16 bytes apart, so the 16 lanes of a ds_read_b128 service group (same g) hit 16 different 16-byte slots
    [[maybe_unused]] const uint32_t xl_base = XL ? sp.xs_off + (uint32_t)min(li, p.R - 1) * sp.xrow + (uint32_t)g * 16u : 0u;
    auto lds16 = [&](uint32_t off) -> u32x4_t { return *(const u32x4_t*)(dyn_lds + off); };
    // XL: the fragments of the NEXT stage are read while this one computes (two register sets, stage parity): with one wave per SIMD
    // an LDS round trip at the top of every stage would be exposed
    [[maybe_unused]] u32x4_t xq[XL ? 2 : 1][XL ? 4 : 1];
    auto read_xq = [&](int set, int it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t idx = (uint32_t)min(it, nit - 1);
            if constexpr (FG == 2) {                      // position inside the resident half: quarter `wave`, iteration modulo the half quarter
                const int rel = min(max(it - it_lo, 0), per - 1), hp = per >> 1;
                idx = (uint32_t)(wave * hp + (rel >= hp ? rel - hp : rel));
            }
            xq[XL ? set : 0][XL ? s : 0] = lds16(xl_base + idx * 256u + 64u * s);   // (run-ahead stages: any row)
        }
    };
    auto compute = [&](const Stage& cur, [[maybe_unused]] int par, [[maybe_unused]] int it_next) {
        constexpr bool SFDB = !(WNAT && NW == 8);
        if constexpr (XL) read_xq(par ^ 1, it_next);       // sign fragments double-buffered across MFMA steps (not when the second
                                                         // activation fragment set already fills the 256-VGPR budget)
        u32x4_t sf[SFDB ? 2 : 1][NMA];
        if constexpr (SFDB) {
#pragma unroll
            for (int t = 0; t < NM; ++t) sf[0][t] = lut(cur.wd[t], 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (SFDB) {
                if (s < 3) {
#pragma unroll
                    for (int t = 0; t < NM; ++t) sf[(s + 1) & 1][t] = lut(cur.wd[t], s + 1);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NM; ++t) sf[0][t] = lut(cur.wd[t], s);
            }
            const u32x4_t xw = XL ? xq[XL ? par : 0][XL ? s : 0] : (XN ? cur.xn[XL ? 0 : s] : cur.xf[XP ? s : 0]);
            const u32x4_t xs_ = XL ? xq[XL ? par : 0][XL ? s : 0] : (PK ? cur.xn[XL ? 0 : s] : cur.xf[XP ? s : 0]);
            if constexpr (HASW) accB = mfma16<DT>(cur.wf[s], xw, accB);
#pragma unroll
            for (int t = 0; t < NM; ++t) accD[t] = mfma16<DT>(sf[SFDB ? (s & 1) : 0][t], xs_, accD[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // end of a tile: the 8 partial tiles meet in LDS; wave (tile & 7) sums them in wave order, scales, adds the base, stores
    auto finish_tile = [&](int tile) {
        float* const rb = red + (tile & 1) * (NW * 64 * 8);
        {
            const int b = min(li, p.R - 1) / p.M;
            const int bm = p.sPb == 0 ? 0 : b;
            // this row's accumulator, picked with bit masks (a chain of `if (bm == t) d = accD[t]` is turned into a scratch array
            // indexed by bm, and scratch traffic shares vmcnt with the weight stream)
            u32x4_t d = NM > 0 ? __builtin_bit_cast(u32x4_t, accD[0]) : u32x4_t{0u, 0u, 0u, 0u};
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
        if (wave == (tile & (NW - 1)) && li < p.R) {
            const int b = li / p.M;
            [[maybe_unused]] float rs_row = 1.f;             // XL = 3: this row's rsqrt(mean(x^2) + eps), from the producer's partial sums
            if constexpr (XL == 3) {
                const float* rp_ = a_lds + (STREAM_ALPHA_MAX - 64) + li * 4;
                rs_row = rms_scale(rp_[0], rp_[1], rp_[2], rp_[3], p.K, sp.eps);
            }
            f32x4_t sb = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) {                               // fixed wave order: deterministic
                sb += *(const f32x4_t*)&rb[(w * 64 + lane) * 8];
                sd += *(const f32x4_t*)&rb[(w * 64 + lane) * 8 + 4];
            }
            if constexpr (EPI == 1) {
                // tile = [8 gate | 8 up] columns (host: N % 16 == 0, cpb % 16 == 0): lane groups 0,1 hold gate columns 4g + e, groups
                // 2,3 the matching up columns; both are rounded to 16 bits as the separate Linear would have stored them
                const int grp = g >> 1;
                float a = 1.f;
                if (al_lds) a = a_lds[li * 2 + grp];
                else if (p.alpha) a = p.alpha[(long long)b * p.sAlb + grp];
                const int n_out = ((c_lo + tile * 16) >> 1) + 4 * (g & 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = round16<DT>(XL == 3 ? scale_then_add(sd[e], a, HASW ? sb[e] : 0.f) * rs_row
                                                        : scale_then_add(sd[e], a, HASW ? sb[e] : 0.f));
                    const float u = __shfl(v, (lane + 32) & 63, 64);
                    if (g < 2) {
                        const long long off = (long long)b * p.sCb + (long long)(li - b * p.M) * p.sCm + n_out + e;
                        ((unsigned short*)p.C)[off] = (unsigned short)swiglu1<DT>(v, u);
                    }
                }
            } else {
                float sq = 0.f;                              // producer side of the hand-off: sum of squares of what this lane stores
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = c_lo + tile * 16 + 4 * g + e;
                    if (n < c_hi) {
                        float a = 1.f;
                        if (al_lds) a = a_lds[li * ng + (n / p.gsz - g0)];
                        else if (p.alpha) a = p.alpha[(long long)b * p.sAlb + n / p.gsz];
                        float val = scale_then_add(sd[e], a, HASW ? sb[e] : 0.f);
                        if constexpr (XL == 3) val *= rs_row;
                        if constexpr (CPRE) {
                            if (sp.ssq_out) {                // (wave-uniform; the hand-off forms return what they stored)
                                const float st_ = (c_pre_ok && tile == wave)
                                    ? store_out_residual_ret<DT>(p, li, n, val, half_bits_to_f32<DT>(c_pre[e]))
                                    : store_out_ret<DT>(p, li, n, val);
                                sq = __builtin_fmaf(st_, st_, sq);
                                if (sp.xw_out) {
                                    const float nwv = half_bits_to_f32<DT>((tile == wave) ? nw_pre[e] : (uint32_t)sp.nw_next[(long long)b * sp.sNwNext + n]);
                                    sp.xw_out[(long long)b * p.sCb + (long long)(li - b * p.M) * p.sCm + n] = (unsigned short)f32_to_half_bits<DT>(st_ * nwv);
                                }
                            } else if (c_pre_ok && tile == wave)
                                store_out_residual<DT>(p, li, n, val, p.out_f32 ? __builtin_bit_cast(float, c_pre[e]) : half_bits_to_f32<DT>(c_pre[e]));
                            else store_out<DT>(p, li, n, val);
                        } else {
                            store_out<DT>(p, li, n, val);
                        }
                    }
                }
                if constexpr (CPRE) {
                    if (sp.ssq_out) {
                        // the row's 16 columns live in lanes li, li + 16, li + 32, li + 48: two fixed exchange steps, lane group 0 writes
                        sq += __shfl_xor(sq, 16, 64);
                        sq += __shfl_xor(sq, 32, 64);
                        if (g == 0) sp.ssq_out[(long long)((c_lo >> 4) + tile) * 16 + li] = sq * sp.ssq_scale;
                    }
                }
            }
        }
        accB = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NM; ++t) accD[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };

    // Main loop: whole rounds of NS stages, at least one round.  Its body has no branch around its loads (past the end of the stream
    // they are out of range and touch no memory) and no exit in the middle, so hipcc's waitcnt pass sees one straight-line stream per
    // round and waits with vmcnt((NS-1) * loads-per-stage).  (A conditional issue, or an exit test between the stages -- which the
    // structuriser turns into an edge back to the loop header -- merges "just re-issued" into the header state and every wait of
    // the first stage becomes vmcnt(0); a separate tail after a possibly zero-trip loop makes the register allocator keep the
    // prologue's load destinations alive across the loop and spill them, i.e. wait for them, before the first round.  Both were
    // read in the ISA, not guessed.)  The last round is padded with all-zero stages: <= NS-1 stages of MFMAs on zeros per wave.
    const int cntb = max(it_hi - it_lo, 1);
    const int total = ntile * cntb;
    int tc = 0, ic = it_lo;                                              // stage being consumed
    int f = 0;
    static_assert(!XL || NS % 2 == 0, "stage parity selects the activation fragment set");
    if constexpr (XL) read_xq(0, it_lo);
    do {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            compute(st[u], u & 1, ic + 1 >= it_hi ? it_lo : ic + 1);
#ifdef BD_STREAM_TRACE
            if (f == 0 && u == 0) BD_ST_STAMP(3);
#endif
            __builtin_amdgcn_sched_barrier(0);
            issue(st[u], ti, ii);
            advance(ti, ii);
            if constexpr (FG == 2) {
                // the switch: every wave has consumed the first half of its quarter (equal counts: host-checked), nobody reads the first-half
                // rows any more -> overwrite them with the second halves (registers since kernel start: no memory wait), and re-read the
                // activation fragments of the next stage, which the step above prefetched from the old rows
                if (ic + 1 - it_lo == (per >> 1) && tc == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    rows_to_lds2(xraw1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    read_xq((u & 1) ^ 1, ic + 1);
                }
            }
            if (++ic >= it_hi && tc < ntile) {
                finish_tile(tc);
                ic = it_lo;
                ++tc;
            }
        }
        f += NS;
    } while (f < total);
    BD_ST_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the run-ahead (out-of-range) loads of the last round
    BD_ST_STAMP(5);
}

}  // namespace bd
