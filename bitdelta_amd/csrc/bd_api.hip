// C ABI of libbitdelta_hip.so (see include/bitdelta_hip.h): argument checking + kernel dispatch.
// No torch types.  State: the test / tuning overrides (bd_set_*) are THREAD-LOCAL, so a thread that forces a kernel family does not
// change what other threads launch; the per-DEVICE caches (CU count, "max dynamic LDS already raised for this kernel") are keyed
// by the current device id -- the reference's demo runs one process over several GPUs (demo/demo_backend.py:23-25).
#include "../../include/bitdelta_hip.h"
#include "../../include/bitdelta_hip_test.h"
#include "bd_bits.h"
#include "bd_gemm_generic.h"
#include "bd_gemm_mfma.h"
#ifdef BD_AB_VARIANTS      // rejected schedules kept as A/B references: harness-only build (tests/native/Makefile)
#include "../../tests/native/ab/bd_gemm_pp.h"
#endif
#include "bd_gemm_pf.h"
#include "bd_gemm_fx.h"
#include "bd_gemm_w4.h"
#include "bd_gemv.h"
#include "bd_gemv_stream.h"
#include "bd_gemv_rows.h"
#ifdef BD_AB_VARIANTS
#include "../../tests/native/ab/bd_gemv_ring.h"     // LDS-DMA loader / consumer decode kernel: lost its A/B (profiles/r04_decode_ring_ab.txt)
#endif
#include "bd_serving.h"
#include "bd_attn_prefill.h"
#include <algorithm>
#include <cstdlib>
#include <atomic>

using namespace bd;

// Environment override of the shape -> variant table (build notes): BD_GEMM_VARIANT=<n> is the initial value of every thread's
// forced variant (exactly bd_set_gemm_variant(n), which still overrides it); BD_TAIL_SPLIT=0 switches the tail split off.  Read once per thread.
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
static thread_local int g_forced_variant = env_int("BD_GEMM_VARIANT", -1);
static thread_local int g_tail_split = env_int("BD_TAIL_SPLIT", 1) ? 1 : 0;          // A/B hook (bd_set_tail_split)
// delta_rows_kernel A/B hook (environment only): bits 1-2 = masks per block forced to 1 / 2 / 4 (value 1 / 2 / 3), bit 4 = never chosen automatically,
// bit 5 = 64-column super-tiles even when they leave CUs idle, bit 6 = 32-column super-tiles always
static thread_local int g_rows_tune = env_int("BD_ROWS_TUNE", 0);
static thread_local int g_norm_rows_min = env_int("BD_NORM_ROWS_MIN", 64);           // bd_srv_rmsnorm: rows from which the wave-per-row kernel runs (A/B hook)
static thread_local int g_attn_depth = env_int("BD_ATTN_DEPTH", 0);                 // decode attention K / V ring depth: 0 = by cache length, 2, 4 (A/B hook)
static thread_local int g_rows_shared_min = env_int("BD_ROWS_SHARED_MIN", 1);       // shared mask: rows from which the kernel is chosen automatically
static thread_local int g_forced_group_m = 0;        // 0 = automatic tile order
static thread_local int g_col16_small_lut = -1;      // sign LUT of the 16-column decode kernel: -1 auto, 1 = single 4-KiB table, 0 = 16-copy conflict-free
                                        // table whenever it fits.  Auto = 16 copies for delta-only launches (-16..18 % at 6-8 masks;
                                        // fused: -2 %, and -15 % WORSE on 14336x4096 where it drops to one block per CU)
static thread_local int g_col16_no_per4 = 0;         // A/B hook: 1 = the 16-column decode kernel always uses its generic one-ahead loop
static thread_local int g_launch_chunking = 0;       // 1 = multi-round tile problems are issued as single-round launches (measured: no gain; off)
static thread_local int g_gemv_wave_spec = 1;        // 1 = fused launches of the VALU decode kernel use its wave-specialised form
static thread_local int g_gemv_two_launch = 1;       // 1 (default) = split-k partials are summed by gemv_reduce_kernel; 0 = in-launch tickets
                                        // (measured slower: the last-arriver tail is serial inside every tile; bd_gemv.h)
static thread_local int g_gemv_target_blocks = 512;
static thread_local int g_stream_tune = 0;            // A/B hook (harness build): bit 0 = natural-order W, bit 1 = nt cache policy, bit 2 = 8-wave blocks, bit 3 = deeper prefetch
static thread_local int g_decode_engine = -1;         // (harness builds) packed-layout decode launches: -1 / 0 = streaming register-load kernel
                                                      // (variant 600), 1 = LDS-DMA loader / consumer kernel (variant 700, tests/native/ab/)
static thread_local int g_ring_tune = -1;             // variant 700 knobs, -1 = defaults: bit 0 = nt weight / sign streams, bit 1 = activations ride the
                                                      // ring even when a resident copy fits, bit 2 = ONE loader wave (default two), bits 8..13 = cap on
                                                      // the ring slots (0 = none)
static thread_local int t_last_variant = -1;
static thread_local int t_last_decode_form = 0;    // 1 = the last streaming decode launch took the fine-grid form

extern "C" int bd_version(void) { return 1; }
extern "C" int bd_set_gemm_variant(int v) { g_forced_variant = v; return BD_OK; }
extern "C" int bd_last_gemm_variant(void) { return t_last_variant; }
extern "C" int bd_set_tail_split(int v) { g_tail_split = v ? 1 : 0; return BD_OK; }
extern "C" int bd_set_tile_group_m(int g) { g_forced_group_m = g; return BD_OK; }
extern "C" int bd_set_decode_two_launch(int on) { g_gemv_two_launch = on ? 1 : 0; return BD_OK; }
extern "C" int bd_set_decode_wave_spec(int on) { g_gemv_wave_spec = on ? 1 : 0; return BD_OK; }
extern "C" int bd_set_launch_chunking(int on) { g_launch_chunking = on ? 1 : 0; return BD_OK; }
extern "C" int bd_set_decode_generic_loop(int on) { g_col16_no_per4 = on ? 1 : 0; return BD_OK; }
extern "C" int bd_set_stream_tuning(int flags) { g_stream_tune = flags; return BD_OK; }
extern "C" int bd_last_decode_form(void) { return t_last_decode_form; }
extern "C" int bd_set_decode_engine(int engine) { g_decode_engine = engine < 0 ? -1 : (engine ? 1 : 0); return BD_OK; }
extern "C" int bd_set_ring_tuning(int flags) { g_ring_tune = flags; return BD_OK; }
extern "C" int bd_set_decode_small_lut(int mode) { g_col16_small_lut = mode < 0 ? -1 : (mode ? 1 : 0); return BD_OK; }

extern "C" const char* bd_error_string(int code) {
    switch (code) {
        case BD_OK: return "ok";
        case BD_E_K_NOT_MULTIPLE: return "K must be divisible by n_bits";
        case BD_E_BAD_NBITS: return "n_bits must be 8, 16, 32 or 64";
        case BD_E_BAD_GROUPS: return "scale groups must divide N";
        case BD_E_BAD_DTYPE: return "unsupported dtype";
        case BD_E_BAD_SHAPE: return "bad shape / stride / forced variant not applicable";
        case BD_E_WORKSPACE: return "workspace missing or too small";
        case BD_E_LAUNCH: return "kernel launch failed";
        case BD_E_NULL: return "null pointer";
        default: return "unknown error";
    }
}

static inline int launch_status() { return hipGetLastError() == hipSuccess ? BD_OK : BD_E_LAUNCH; }

// ------------------------------------------------------------------ pack / unpack
extern "C" int bd_pack(const void* bits, int64_t batch, int64_t K, int64_t N, int64_t s_b, int64_t s_k, int64_t s_n,
                       void* out, int n_bits, void* stream) {
    if (n_bits != 8 && n_bits != 16 && n_bits != 32 && n_bits != 64) return BD_E_BAD_NBITS;
    if (batch < 0 || K < 0 || N < 0) return BD_E_BAD_SHAPE;
    if (K % n_bits) return BD_E_K_NOT_MULTIPLE;
    if (batch == 0 || K == 0 || N == 0) return BD_OK;
    if (!bits || !out) return BD_E_NULL;
    const int64_t KW = K / n_bits;
    if (KW > 65535 * 8LL || batch > 65535) return BD_E_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* b8 = (const uint8_t*)bits;
    if (n_bits == 32 && s_k == 1 && s_n != 1) {
        dim3 grid((unsigned)((N + 63) / 64), (unsigned)((KW + 7) / 8), (unsigned)batch);
        hipLaunchKernelGGL(pack_kmajor_kernel, grid, dim3(256), 0, st, b8, (uint32_t*)out, (long long)KW, (long long)N,
                           (long long)s_b, (long long)s_n);
        return launch_status();
    }
    if (KW > 65535) return BD_E_BAD_SHAPE;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)KW, (unsigned)batch);
#define BD_PACK(T) hipLaunchKernelGGL(pack_kernel<T>, grid, dim3(256), 0, st, b8, (T*)out, (long long)KW, (long long)N, \
                                      (long long)s_b, (long long)s_k, (long long)s_n)
    if (n_bits == 8) BD_PACK(uint8_t);
    else if (n_bits == 16) BD_PACK(uint16_t);
    else if (n_bits == 32) BD_PACK(uint32_t);
    else BD_PACK(uint64_t);
#undef BD_PACK
    return launch_status();
}

extern "C" int bd_unpack(const void* words, int64_t batch, int64_t KW, int64_t N, void* out_bits, int n_bits, void* stream) {
    if (n_bits != 8 && n_bits != 16 && n_bits != 32 && n_bits != 64) return BD_E_BAD_NBITS;
    if (batch < 0 || KW < 0 || N < 0 || KW > 65535 || batch > 65535) return BD_E_BAD_SHAPE;
    if (batch == 0 || KW == 0 || N == 0) return BD_OK;
    if (!words || !out_bits) return BD_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((N + 1023) / 1024), (unsigned)KW, (unsigned)batch);
#define BD_UNPACK(T) hipLaunchKernelGGL(unpack_kernel<T>, grid, dim3(256), 0, st, (const T*)words, (uint8_t*)out_bits, \
                                        (long long)KW, (long long)N)
    if (n_bits == 8) BD_UNPACK(uint8_t);
    else if (n_bits == 16) BD_UNPACK(uint16_t);
    else if (n_bits == 32) BD_UNPACK(uint32_t);
    else BD_UNPACK(uint64_t);
#undef BD_UNPACK
    return launch_status();
}

// ------------------------------------------------------------------ GEMM dispatch
namespace {

struct Problem {
    const void *A, *W;
    const int32_t* P;
    const float* alpha;
    void* C;
    int B, M, N, K;
    int64_t sAb, sAm, sPb, sCb, sCm, ldw, sAlb;
    int G, dtype, out_dtype, round_mode, accumulate;
    int mask_tiled;           // 0: P is [B or 1, K/32, N] (reference layout); 1: tile-major [B or 1, ceil(N/16), K/32, 16];
                              // 2: packed decode layout [ceil(N/16), ceil(K/128), 4, 16, t_pad] (decode kernel only, see bd_gemv_stream.h)
    int w_tiled;              // layout 2 only: W is the tile-major decode copy [N/16][K/128][4 steps][16 rows][4 groups][8] (ldw passed as 0)
    int t_pad;                // layout 2: dwords per (tile, iteration, lane group, column) = tenants padded to 1 / 2 / 4 / 6 / 8
    const void* norm_w;       // layout 2 only: fused RMSNorm prologue (A is the un-normalised residual stream); [B or 1, K], stride sNw
    int64_t sNw;
    float eps;
    int epilogue;             // layout 2 only: 1 = SwiGLU over an 8-interleaved gate|up projection (C has N/2 columns)
    const float* ssq_in;      // layout 2 only: RMSNorm hand-off, consumer side (with norm_w): per-row partial sums of squares [K/16][16]
    float* ssq_out;           // layout 2 only: RMSNorm hand-off, producer side: this launch writes its output's partial sums [N/16][16]
    const void* nw_next;      // ... producer with xw_out: the NEXT RMSNorm's weight [B or 1, N] (stride sNwNext) and the pre-multiplied copy
    int64_t sNwNext;
    void* xw_out;
    const void* pn_w;         // bd_binary_linear_residual_norm: the RMSNorm that FOLLOWS this Linear -- weight [B or 1, N] (stride s_pnw), output pn_h
    int64_t s_pnw;            // [B, M, N] (strides sHb / sHm).  A split-k reduce launch takes it along (splitk_reduce_norm_kernel) and sets
    float pn_eps;             // t_post_norm_done; otherwise the entry point launches the norm after the Linear.
    void* pn_h;
    int64_t sHb, sHm;
    void* ws;
    int64_t ws_bytes;
    hipStream_t st;
};
static thread_local int t_post_norm_done = 0;

constexpr int MAX_DEVICES = 64;
inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}
inline int num_cus() {
    static std::atomic<int> cus[MAX_DEVICES];     // 0 = not queried yet; benign race: idempotent
    const int dev = current_device();
    if (dev >= MAX_DEVICES) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: raise it once per (kernel, device).
// `done` is one bit per device id, owned by the call site (one static per kernel instantiation).
inline bool ensure_dyn_lds(const void* kern, int bytes, std::atomic<uint64_t>& done) {
    const int dev = current_device();
    const uint64_t bit = dev < MAX_DEVICES ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return true;
}

constexpr int GEMV_MAX_M = 16, GEMV_MAX_R = 16;   // decode kernels: <= 16 activation rows per launch; larger batches are chunked

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

inline bool fast_ok(const Problem& q) {
    bool ok = (q.K % 64 == 0) && aligned16(q.A) && (q.sAm % 8 == 0) && (q.sAb % 8 == 0);
    if (q.W) ok = ok && aligned16(q.W) && (q.ldw % 8 == 0);
    // 32-bit DMA offsets inside a block: rows*stride*2 bytes must fit
    ok = ok && ((int64_t)256 * q.sAm * 2 < (1LL << 31)) && ((int64_t)q.N * 8 + 1024 < (1LL << 31));
    if (q.W) ok = ok && ((int64_t)256 * q.ldw * 2 < (1LL << 31));
    return ok;
}
inline bool gemv_ok(const Problem& q) {
    bool ok = (q.M <= GEMV_MAX_M) && ((int64_t)q.B * q.M <= 4 * GEMV_MAX_R) && (q.K % 32 == 0) && aligned16(q.A) &&   // <= 4 chunks
              (q.sAm % 8 == 0) && (q.sAb % 8 == 0);
    if (q.W) ok = ok && aligned16(q.W) && (q.ldw % 8 == 0);
    return ok;
}

inline int gemv_rmax(int R) { return R <= 1 ? 1 : R <= 2 ? 2 : R <= 3 ? 3 : R <= 4 ? 4 : R <= 6 ? 6 : R <= 8 ? 8 : R <= 12 ? 12 : 16; }

inline void gemv_split(const Problem& q, int& KS, int& kslice) {
    const int tiles_n = (q.N + 63) / 64;
    int want = (g_gemv_target_blocks + tiles_n - 1) / tiles_n;   // fat blocks keep many loads in flight each
    if (g_forced_variant > 200 && g_forced_variant <= 264) want = g_forced_variant - 200;     // test hook: 200 + KS
    if (g_forced_variant > 300 && g_forced_variant <= 364) want = g_forced_variant - 300;
    if (g_forced_variant > 400 && g_forced_variant <= 464) want = g_forced_variant - 400;
    int maxks = q.K / 512;                                // at least 512 k per slice
    if (maxks < 1) maxks = 1;
    KS = want < 1 ? 1 : (want > maxks ? maxks : want);
    kslice = ((q.K + KS - 1) / KS + 127) / 128 * 128;
    const int smax = gemv_kslice_max(16);
    if (kslice > smax) kslice = smax;                     // the activation slice lives in LDS
    KS = (q.K + kslice - 1) / kslice;
}

template <int DT, int RMAX>
int launch_gemv_valu(const Problem& q, const GemvParams& gp) {       // A/B reference (variant 300): VALU sign-flip form
    dim3 grid((unsigned)((q.N + 63) / 64), (unsigned)gp.KS);
    // fused launches: the wave-specialised instantiation (4 weight-streaming waves + 4 sign waves per block)
    // (-6..13 % up to 8 rows; at 16 tenants the sign work dominates and halving the sign waves per CU costs 22 %)
    if (q.W && g_gemv_wave_spec && gp.R <= 8) hipLaunchKernelGGL((gemv_kernel<DT, RMAX, true>), grid, dim3(512), 0, q.st, gp);
    else hipLaunchKernelGGL((gemv_kernel<DT, RMAX, false>), grid, dim3(256), 0, q.st, gp);
    return BD_OK;
}

template <int DT, int NM>
int launch_gemv_mfma(const Problem& q, const GemvParams& gp) {
    dim3 grid((unsigned)((q.N + 63) / 64), (unsigned)gp.KS);
    // LC = 16 (conflict-free 16-copy sign LUT, 64 KiB) measured SLOWER than the single 4-KiB table (T=6, 4096^2: 24.3 vs 17.9 us):
    // building 64 KiB per block and 2 blocks/CU cost more than the ~3-way conflicts of random bytes on one table.
    constexpr int LC = 1;
    const unsigned lds = 256u * 16u * LC + (unsigned)gp.R * (gemv_kslice_max(16) * 2 + 16);
    auto kw = gemv_mfma_kernel<DT, NM, true, LC>;
    auto kd = gemv_mfma_kernel<DT, NM, false, LC>;
    static std::atomic<uint64_t> done_w{0}, done_d{0};
    if (!ensure_dyn_lds((const void*)(q.W ? kw : kd), 256 * 16 * LC + 16 * 2064, q.W ? done_w : done_d)) return BD_E_LAUNCH;
    if (q.W) hipLaunchKernelGGL(kw, grid, dim3(256), lds, q.st, gp);
    else hipLaunchKernelGGL(kd, grid, dim3(256), lds, q.st, gp);
    return BD_OK;
}

template <int DT>
int launch_gemv_chunk(const Problem& q, bool valu_form);

// ---- no-split-k decode kernel (gemv_col16_kernel): 16 columns x all of k per block, 8 waves split k, LDS reduction
constexpr int COL16_LDS_BUDGET = 160 * 1024 - 16 * 1024 /* red */ - 4096 /* LUT */ - 1024 /* slack */;
inline void col16_split(int R, int K, int forced_ks, int& KS, int& kslice) {
    int kmax = (COL16_LDS_BUDGET / (R * 2) - 8) / 128 * 128;      // k per block whose R activation rows fit in LDS
    if (kmax < 128) kmax = 128;
    KS = (K + kmax - 1) / kmax;
    if (forced_ks > KS) KS = forced_ks;
    kslice = ((K + KS - 1) / KS + 127) / 128 * 128;
    KS = (K + kslice - 1) / kslice;
}

template <int DT, int NM, int LC, int PER>
int launch_gemv_col16_lc(const Problem& q, const GemvParams& gp) {
    dim3 grid((unsigned)((q.N + 15) / 16), (unsigned)gp.KS);
    const unsigned lds = 4096u * LC + (unsigned)gp.R * (unsigned)(gp.kslice * 2 + 16);
    auto kw = gemv_col16_kernel<DT, NM, true, LC, PER>;
    auto kd = gemv_col16_kernel<DT, NM, false, LC, PER>;
    static std::atomic<uint64_t> done_w{0}, done_d{0};
    if (!ensure_dyn_lds((const void*)(q.W ? kw : kd), 160 * 1024 - 16 * 1024 - 256, q.W ? done_w : done_d)) return BD_E_LAUNCH;
    if (q.W) hipLaunchKernelGGL(kw, grid, dim3(512), lds, q.st, gp);
    else hipLaunchKernelGGL(kd, grid, dim3(512), lds, q.st, gp);
    return BD_OK;
}

template <int DT, int NM, int LC>
int launch_gemv_col16_per(const Problem& q, const GemvParams& gp) {
    // every wave owns exactly 4 whole iterations (one slice of K = 4096): the straight-line, everything-in-flight instantiation
    // (measured, `bd_harness dec500`: -3..7 % for delta-only launches, but +7..13 % SLOWER for fused ones -> delta-only only)
    const bool per4 = gp.KS == 1 && q.K == 4096 && !q.W && !g_col16_no_per4;
    return per4 ? launch_gemv_col16_lc<DT, NM, LC, 4>(q, gp) : launch_gemv_col16_lc<DT, NM, LC, 0>(q, gp);
}

template <int DT, int NM>
int launch_gemv_col16(const Problem& q, const GemvParams& gp) {
    // 16-copy conflict-free sign LUT when it fits next to the activation rows and there are masks enough to expand
    const bool want = g_col16_small_lut < 0 ? (q.W == nullptr) : (g_col16_small_lut == 0);
    const bool big = NM >= 2 && want && 65536 + (int64_t)gp.R * (gp.kslice * 2 + 16) <= 160 * 1024 - 16 * 1024 - 256 &&
                     (g_col16_small_lut == 0 || (q.N + 15) / 16 <= num_cus());      // auto: only when one block per CU covers N
    if constexpr (NM >= 2) { if (big) return launch_gemv_col16_per<DT, NM, 16>(q, gp); }
    return launch_gemv_col16_per<DT, NM, 1>(q, gp);
}

template <int DT>
int launch_gemv_col16_chunk(const Problem& q) {
    GemvParams gp;
    gp.X = (const unsigned short*)q.A;
    gp.P = (const uint32_t*)q.P;
    gp.W = (const unsigned short*)q.W;
    gp.alpha = q.alpha;
    gp.C = q.C;
    gp.B = q.B; gp.M = q.M; gp.N = q.N; gp.K = q.K; gp.R = q.B * q.M;
    gp.sXb = q.sAb; gp.sPb = q.sPb; gp.sCb = q.sCb;
    gp.sXm = (int)q.sAm; gp.sCm = (int)q.sCm; gp.ldw = (int)q.ldw; gp.sAlb = (int)q.sAlb; gp.gsz = q.N / q.G;
    const int forced = (g_forced_variant > 500 && g_forced_variant <= 564) ? g_forced_variant - 500 : 0;
    col16_split(gp.R, q.K, forced, gp.KS, gp.kslice);
    gp.round_mode = q.round_mode; gp.accumulate = q.accumulate; gp.out_f32 = (q.out_dtype == BD_F32);
    gp.tickets = nullptr;
    gp.ws = nullptr;
    if (gp.KS > 1) {
        const int64_t need = GEMV_TICKET_BYTES + (int64_t)gp.KS * gp.R * q.N * 4;
        if (!q.ws || q.ws_bytes < need) return BD_E_WORKSPACE;
        gp.ws = (float*)((char*)q.ws + GEMV_TICKET_BYTES);
    }
    const int nm = q.sPb == 0 ? 1 : q.B;
    if (nm <= 1) launch_gemv_col16<DT, 1>(q, gp);
    else if (nm <= 2) launch_gemv_col16<DT, 2>(q, gp);
    else if (nm <= 3) launch_gemv_col16<DT, 3>(q, gp);
    else if (nm <= 4) launch_gemv_col16<DT, 4>(q, gp);
    else if (nm <= 6) launch_gemv_col16<DT, 6>(q, gp);
    else if (nm <= 8) launch_gemv_col16<DT, 8>(q, gp);
    else if (nm <= 12) launch_gemv_col16<DT, 12>(q, gp);
    else launch_gemv_col16<DT, 16>(q, gp);
    if (gp.KS > 1) {
        dim3 g2((unsigned)((q.N + 255) / 256), (unsigned)gp.R);
        hipLaunchKernelGGL((gemv_reduce_kernel<DT>), g2, dim3(256), 0, q.st, gp);
    }
    return launch_status();
}

// ---- streaming decode kernel (gemv_stream_kernel, variant 600): one 8-wave block per CU, contiguous column range per block
constexpr int STREAM_MIN_N = 512;
constexpr int STREAM_WT_NT_DEFAULT = 1;           // (-3.6 % on the 6-tenant Mistral decode step, same process: profiles/r04_decode_ab.txt)
constexpr int STREAM_FG_DEFAULT = 1;              // fine grid (single-tile blocks, two per CU) where the tile count is between one and two per CU
constexpr int STREAM_XRES_DEFAULT = 1;            // activation rows resident in LDS + deeper prefetch (XL = 2) where the rule below says so
// (STREAM_WT_NT_DEFAULT: non-temporal policy on the tile-major weight loads of the streaming kernel)
inline bool stream_ok(const Problem& q, int rows, int nmask) {
    if (rows > GEMV_MAX_R || nmask > (q.mask_tiled == 2 ? 16 : 8) || q.N < STREAM_MIN_N) return false;
    // 32-bit buffer offsets with an out-of-range sentinel at 2 GiB: every extent must stay below it
    const int64_t lim = (1ll << 31) - 64;
    const int64_t xb = ((int64_t)(q.B - 1) * q.sAb + (int64_t)(q.M - 1) * q.sAm + q.K) * 2;
    const int64_t wb = q.W ? ((int64_t)(q.N - 1) * q.ldw + q.K) * 2 : 0;
    const int64_t pb = q.mask_tiled == 2 ? (int64_t)((q.N + 15) / 16) * ((q.K + 127) / 128) * 4 * 16 * q.t_pad * 4
                                         : ((int64_t)(nmask - 1) * q.sPb + (int64_t)(q.K / 32) * ((q.N + 15) / 16 * 16)) * 4;
    return xb > 0 && xb < lim && wb < lim && pb < lim && q.sAb >= 0 && q.sAm >= 0;
}

constexpr int STREAM_LDS_MAX = 160 * 1024;        // LDS of a gfx950 CU: the fused-norm kernels add R activation rows to STREAM_LDS_BYTES
template <int DT, int NM, bool HASW, int NS, int NW = 8, int WNAT = 0, int AUX = 2, int PK = 0, int XL = 0, int EPI = 0, int WT = 0, int FG = 0>
int launch_stream_inst(const StreamParams& sp, dim3 grid, hipStream_t st) {
    auto kern = gemv_stream_kernel<DT, NM, HASW, NS, NW, WNAT, AUX, PK, XL, EPI, WT, FG>;
    static std::atomic<uint64_t> lds_done{0};
    constexpr int base = FG ? STREAM_FG_XS_OFF : STREAM_LDS_BYTES;
    const int lds = XL ? std::max((int)(sp.xs_off + (uint32_t)sp.g.R * sp.xrow), base) : base;
    if (lds > (FG == 1 ? STREAM_FG_LDS_MAX : STREAM_LDS_MAX)) return BD_E_BAD_SHAPE;
    if (!ensure_dyn_lds((const void*)kern, FG == 1 ? STREAM_FG_LDS_MAX : XL ? STREAM_LDS_MAX : STREAM_LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, sp);
    return BD_OK;
}

// Shipped configuration (profiles/r02_decode_stream_ab.txt): 4-wave blocks (one wave per SIMD, the whole register file: NS4 stages
// of loads in flight per wave), word-row order for W, default cache policy.  Harness builds (-DBD_AB_VARIANTS) add the A/B matrix
// selected by bd_set_stream_tuning: bit 0 natural-order W, bit 1 nt cache policy, bit 2 8-wave blocks (two waves per SIMD), bit 3 the
// deeper of two prefetch depths.
template <int DT, int NM, bool HASW, int NS4>
int launch_stream_tuned(const StreamParams& sp, dim3 grid, hipStream_t st) {
#ifdef BD_AB_VARIANTS
    if constexpr (DT == DT_F16 && HASW && (NM == 0 || NM == 1 || NM == 6)) {
        constexpr int A8 = NM == 6 ? 2 : 4, B8 = NM == 6 ? 3 : 6;          // 8-wave blocks: base / deeper
        constexpr int A4 = NM == 6 ? 4 : 8, B4 = NM == 6 ? 6 : 12;         // 4-wave blocks
        switch (g_stream_tune & 15) {
#define BD_T(code, NS, NW, WN, AX) case code: return launch_stream_inst<DT, NM, HASW, NS, NW, WN, AX>(sp, grid, st)
            BD_T(0, A4, 4, 0, 0); BD_T(1, A4, 4, 1, 0); BD_T(2, A4, 4, 0, 6); BD_T(3, A4, 4, 1, 6);
            BD_T(4, A8, 8, 0, 0); BD_T(5, A8, 8, 1, 0); BD_T(6, A8, 8, 0, 6); BD_T(7, A8, 8, 1, 6);
            BD_T(8, B4, 4, 0, 0); BD_T(9, B4, 4, 1, 0); BD_T(10, B4, 4, 0, 6); BD_T(11, B4, 4, 1, 6);
            BD_T(12, B8, 8, 0, 0); BD_T(13, B8, 8, 1, 0); BD_T(14, B8, 8, 0, 6); BD_T(15, B8, 8, 1, 6);
#undef BD_T
        }
    }
#endif
    return launch_stream_inst<DT, NM, HASW, NS4, 4, 0, 0>(sp, grid, st);
}

#ifdef BD_AB_VARIANTS
// ---- loader / consumer decode kernel (gemv_ring_kernel, variant 700): packed sign layout, fused Linear, K % 128 == 0
constexpr int DECODE_ENGINE_DEFAULT = 0;          // what packed-layout decode launches run when nobody called bd_set_decode_engine
constexpr int RING_TUNE_DEFAULT = 1;              // nt streams, resident activations when they fit, as many slots as fit
inline bool ring_wanted() {
    if (g_forced_variant == 700) return true;
    if (g_forced_variant >= 0) return false;
    return (g_decode_engine < 0 ? DECODE_ENGINE_DEFAULT : g_decode_engine) == 1;
}
// fills the LDS geometry of `rp`; false when the launch does not fit the kernel's envelope (the caller falls back to variant 600,
// or answers BD_E_BAD_SHAPE when 700 was forced)
inline bool ring_plan(const Problem& q, RingParams& rp) {
    const int tune = g_ring_tune < 0 ? RING_TUNE_DEFAULT : g_ring_tune;
    if (!q.W || q.mask_tiled != 2) return false;
    rp.epi = q.epilogue == 1;
    rp.nw = (const unsigned short*)q.norm_w; rp.sNw = q.sNw; rp.eps = q.eps;
    return ring_plan_geometry(rp, q.B * q.M, q.K, q.t_pad, q.w_tiled != 0, q.ldw, q.norm_w != nullptr, tune);
}
template <int DT, int NM, int NT>
int launch_ring_inst2(const RingParams& rp, unsigned grid, hipStream_t st) {
    auto kern = gemv_ring_kernel<DT, NM, NT>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, RING_LDS_MAX, lds_done)) return BD_E_LAUNCH;
    const unsigned lds = ring_lds_bytes(rp);
    if (lds > (unsigned)RING_LDS_MAX) return BD_E_BAD_SHAPE;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (4 + rp.nl)), lds, st, rp);
    return BD_OK;
}
template <int DT, int NM>
int launch_ring_inst(const RingParams& rp, unsigned grid, hipStream_t st) {
    return rp.nt ? launch_ring_inst2<DT, NM, 1>(rp, grid, st) : launch_ring_inst2<DT, NM, 0>(rp, grid, st);
}

#endif

template <int DT>
int launch_gemv_stream_chunk(const Problem& q) {
    StreamParams sp{};
    t_last_decode_form = 0;
    GemvParams& gp = sp.g;
    gp.X = (const unsigned short*)q.A;
    gp.P = (const uint32_t*)q.P;
    gp.W = (const unsigned short*)q.W;
    gp.alpha = q.alpha;
    gp.C = q.C;
    gp.ws = nullptr; gp.tickets = nullptr;
    gp.B = q.B; gp.M = q.M; gp.N = q.N; gp.K = q.K; gp.R = q.B * q.M;
    gp.sXb = q.sAb; gp.sPb = q.sPb; gp.sCb = q.sCb;
    gp.sXm = (int)q.sAm; gp.sCm = (int)q.sCm; gp.ldw = (int)q.ldw; gp.sAlb = (int)q.sAlb; gp.gsz = q.N / q.G;
    gp.KS = 1; gp.kslice = q.K;
    gp.round_mode = q.round_mode; gp.accumulate = q.accumulate; gp.out_f32 = (q.out_dtype == BD_F32);
    const int nmask = q.sPb == 0 ? 1 : q.B;
    // columns per block: the whole chip streams one launch, so every CU gets ~N / CUs columns (rounded up to whole sign-word
    // quads); a few more blocks than CUs would serialise a second, nearly empty round
    const int cus = num_cus();
    int cpb = (q.N + cus - 1) / cus;
    cpb = (cpb + 3) & ~3;
    if (cpb < 4) cpb = 4;
    if (g_forced_variant > 600 && g_forced_variant <= 664) cpb = 4 * (g_forced_variant - 600);     // test hook: 600 + cpb/4
    sp.cpb = cpb;
    sp.pts = q.mask_tiled == 1 ? 16u * (uint32_t)(q.K / 32) : 16u;
    sp.prs = q.mask_tiled == 1 ? 16u : (uint32_t)q.N;
    sp.tp = (uint32_t)q.t_pad;
    sp.nw = (const unsigned short*)q.norm_w; sp.sNw = q.sNw; sp.eps = q.eps;
    sp.ssq_in = q.ssq_in; sp.ssq_out = q.ssq_out;
    sp.ssq_scale = (q.ssq_out && q.eps > 0.f) ? q.eps : 1.f;      // producer launches: the `eps` slot of the entry point carries the factor (header)
    sp.nw_next = (const unsigned short*)q.nw_next; sp.sNwNext = q.sNwNext; sp.xw_out = (unsigned short*)q.xw_out;
    sp.no_res_prefetch = (g_stream_tune & 1024) ? 1 : 0;
    sp.n_bytes = q.norm_w ? (uint32_t)(((int64_t)(q.B - 1) * q.sNw + q.K) * 2) : 0u;
    sp.xs_off = (uint32_t)STREAM_XS_OFF; sp.xrow = (uint32_t)q.K * 2u + 16u;
    sp.jsh = 0;
    while ((2048 << sp.jsh) < q.K) ++sp.jsh;
    if (q.epilogue == 1 || q.ssq_out) { cpb = (cpb + 15) & ~15; sp.cpb = cpb; }      // whole [8 gate | 8 up] tiles / whole hand-off tiles per block
    const unsigned grid = (unsigned)((q.N + cpb - 1) / cpb);
    sp.x_bytes = (uint32_t)(((int64_t)(q.B - 1) * q.sAb + (int64_t)(q.M - 1) * q.sAm + q.K) * 2);
    sp.w_bytes = q.W ? (uint32_t)(((int64_t)(q.N - 1) * q.ldw + q.K) * 2) : 0u;
    if (q.w_tiled)       // tile-major W: [N/16][K/128] blocks of 4 KiB
        sp.w_bytes = (uint32_t)((int64_t)((q.N + 15) / 16) * ((q.K + 127) / 128) * 4096);
    sp.p_bytes = (uint32_t)(((int64_t)(nmask - 1) * q.sPb + (int64_t)(q.K / 32) * (q.mask_tiled ? (q.N + 15) / 16 * 16 : q.N)) * 4);
    int rc;
    if (q.mask_tiled == 2) {      // packed layout: all tenants of the call in one chunk, interleaved; extent from the pack's own geometry
        sp.p_bytes = (uint32_t)((int64_t)((q.N + 15) / 16) * ((q.K + 127) / 128) * 4 * 16 * q.t_pad * 4);
#ifdef BD_AB_VARIANTS
        if (ring_wanted()) {      // loader / consumer kernel (variant 700); launches outside its envelope stay on the streaming kernel
            RingParams rp{};
            rp.g = gp; rp.cpb = cpb;
            if (ring_plan(q, rp)) {
                int rrc;
                switch (q.t_pad) {
                    case 1: rrc = launch_ring_inst<DT, 1>(rp, grid, q.st); break;
                    case 2: rrc = launch_ring_inst<DT, 2>(rp, grid, q.st); break;
                    case 4: rrc = launch_ring_inst<DT, 4>(rp, grid, q.st); break;
                    case 6: rrc = launch_ring_inst<DT, 6>(rp, grid, q.st); break;
                    default: rrc = launch_ring_inst<DT, 8>(rp, grid, q.st); break;
                }
                if (rrc != BD_OK) return rrc;
                t_last_variant = 700;
                return launch_status();
            }
            if (g_forced_variant == 700) return BD_E_BAD_SHAPE;
        }
        // kernel kinds: plain | RMSNorm prologue (XL) | XL + SwiGLU epilogue | SwiGLU epilogue only; each with row-major or tile-major W (WT)
#else
        if (g_forced_variant == 700) return BD_E_BAD_SHAPE;      // harness-only kernel (round 4 A/B loser): never silently replaced
#endif
        // tile-major W: optional nt policy on the weight loads (STREAM_WT_NT_DEFAULT, or bd_set_stream_tuning bit 4 = on / bit 5 = off)
        const bool wnt = (g_stream_tune & 16) ? true : (g_stream_tune & 32) ? false : (STREAM_WT_NT_DEFAULT != 0);
        // ... and optionally the activation rows resident in LDS with a deeper weight prefetch (XL = 2; bd_set_stream_tuning bit 6 = on,
        // bit 7 = off): plain and SwiGLU launches with M = 1 and R * K <= 32768
        // (any K: tile-major W already needs K % 128 == 0; the norm prologue keeps its power-of-two rule, this form does not need it)
        // Every wave must own at least one 128-k iteration: a wave with an empty k range still walks one padded stage per tile, whose
        // zero sign words expand to -1 fragments -- harmless against the zero activation fragments of the per-stage form, NOT against
        // fragments read from the resident rows (K = 1152: 9 iterations over 4 waves leave the last wave empty).
        const int nit_x = (q.K + 127) / 128;
        const bool xres_ok = q.w_tiled && (!q.norm_w || q.ssq_in) && q.M == 1 && q.t_pad <= 8 && q.K >= 1024 && 3 * ((nit_x + 3) / 4) < nit_x &&
                             (int64_t)q.B * q.K <= 16 * 2048 &&
                             (int64_t)STREAM_XS_OFF + (int64_t)q.B * (2 * (int64_t)q.K + 16) <= STREAM_LDS_MAX;
        // Fine grid (FG, round 6; gemv_stream_kernel): a resident-row launch whose 16-column tiles number between one and two per CU --
        // Mistral's fused q|k|v: 384 tiles -- runs as ONE round of single-tile blocks, two per CU (<= 80 KB of LDS with the nibble sign table,
        // <= 256 VGPRs), instead of 256 blocks that walk 1.5 tiles each.  bd_set_stream_tuning: 256 = never, 512 = every eligible launch
        // with at most two tiles per CU (one tile per CU included: the o projection).
        const int fg_tiles = q.N / 16;
        const bool fg_ok = xres_ok && q.N % 16 == 0 && q.t_pad <= 6 && fg_tiles <= 2 * cus &&
                           (int64_t)STREAM_FG_XS_OFF + (int64_t)q.B * (2 * (int64_t)q.K + 16) <= STREAM_FG_LDS_MAX;
        const bool fg = fg_ok && !(g_stream_tune & 256) && ((g_stream_tune & 512) ? true : (STREAM_FG_DEFAULT != 0 && fg_tiles > cus));
        const unsigned fg_grid = (unsigned)fg_tiles;
        if (fg) { sp.cpb = 16; sp.xs_off = (uint32_t)STREAM_FG_XS_OFF; }
        t_last_decode_form = fg ? 1 : 0;
        // Two-pass resident rows (FG = 2, round 6; gemv_stream_kernel): rows that do not fit LDS at once -- the down projection of a multi-tenant
        // step, 6 x 14336 -- with ONE tile per block, every wave's k quarter cut in two halves.  Bit-identical to the per-stage-load form and
        // SLOWER (down 33.0 -> 37.1 us at 2 stages of prefetch, 40.0 at 4; step +4 ... +6 %: profiles/r06_decode_step.txt), so like every A/B
        // loser it exists in harness builds only (-DBD_AB_VARIANTS; bd_set_stream_tuning 8192 = on, + 16384 = 2 stages instead of 4).
#ifdef BD_AB_VARIANTS
        const bool fg2 = (g_stream_tune & 8192) && q.w_tiled && !q.norm_w && !q.ssq_in && q.epilogue == 0 && q.M == 1 && q.t_pad <= 8 && q.N % 16 == 0 &&
                         fg_tiles <= cus && q.K % 1024 == 0 && q.K >= 8192 && (int64_t)q.B * q.K > 16 * 2048 && (int64_t)q.B * (q.K / 2) <= 24 * 2048 &&
                         (int64_t)STREAM_FG_XS_OFF + (int64_t)q.B * ((int64_t)q.K + 16) <= STREAM_LDS_MAX;
        if (fg2) {
            sp.cpb = 16; sp.xs_off = (uint32_t)STREAM_FG_XS_OFF; sp.xrow = (uint32_t)q.K + 16u;
            t_last_decode_form = 2;
#define BD_X2(NM) rc = (g_stream_tune & 16384) ? launch_stream_inst<DT, NM, true, 2, 4, 1, 2, 1, 2, 0, 1, 2>(sp, dim3(fg_grid), q.st) \
                                              : launch_stream_inst<DT, NM, true, 4, 4, 1, 2, 1, 2, 0, 1, 2>(sp, dim3(fg_grid), q.st)
            switch (q.t_pad) {
                case 4: BD_X2(4); break;
                case 6: BD_X2(6); break;
                default: return BD_E_BAD_SHAPE;
            }
#undef BD_X2
            if (rc != BD_OK) return rc;
            return launch_status();
        }
#endif
        if (q.ssq_in) {
            // RMSNorm by hand-off (XL = 3): the resident-row form with the rows pre-multiplied by the norm weight and the row scale in the
            // epilogue; same envelope as the resident rows, nothing else implements it
            if (!xres_ok) return BD_E_BAD_SHAPE;
#define BD_XH(NM) rc = fg ? (q.epilogue == 1 ? launch_stream_inst<DT, NM, true, 2, 4, 1, 2, 1, 3, 1, 1, 1>(sp, dim3(fg_grid), q.st)   \
                                             : launch_stream_inst<DT, NM, true, 2, 4, 1, 2, 1, 3, 0, 1, 1>(sp, dim3(fg_grid), q.st))  \
                     : q.epilogue == 1 ? launch_stream_inst<DT, NM, true, 2, 4, 1, 2, 1, 3, 1, 1>(sp, dim3(grid), q.st)   \
                                       : launch_stream_inst<DT, NM, true, 2, 4, 1, 2, 1, 3, 0, 1>(sp, dim3(grid), q.st)
            switch (q.t_pad) {
                case 1: BD_XH(1); break;
                case 2: BD_XH(2); break;
                case 4: BD_XH(4); break;
                case 6: BD_XH(6); break;
                case 8: rc = q.epilogue == 1 ? launch_stream_inst<DT, 8, true, 2, 4, 1, 2, 1, 3, 1, 1>(sp, dim3(grid), q.st)
                                             : launch_stream_inst<DT, 8, true, 2, 4, 1, 2, 1, 3, 0, 1>(sp, dim3(grid), q.st); break;
                default: return BD_E_BAD_SHAPE;
            }
#undef BD_XH
            if (rc != BD_OK) return rc;
            return launch_status();
        }
        // Default: wherever it applies.  (Until the prefetch depths came down to 2 stages this form lost on short launches -- at 6 stages the
        // 4096 x 4096 o projection of a 6-tenant step was +25 % -- and was dispatched by size; at 2 stages it wins on every eligible launch:
        // 6 tenants 4.805 -> 4.770 ms per step with o included, profiles/r04_decode_step_ab.txt.)
        const bool xres_auto = STREAM_XRES_DEFAULT != 0;
        const bool xres = xres_ok && ((g_stream_tune & 64) ? true : (g_stream_tune & 128) ? false : xres_auto);
        if (xres) {
#define BD_XR(NM, NS8) rc = fg ? (q.epilogue == 1 ? launch_stream_inst<DT, NM, true, NS8, 4, 1, 2, 1, 2, 1, 1, 1>(sp, dim3(fg_grid), q.st)   \
                                                  : launch_stream_inst<DT, NM, true, NS8, 4, 1, 2, 1, 2, 0, 1, 1>(sp, dim3(fg_grid), q.st))  \
                          : q.epilogue == 1 ? launch_stream_inst<DT, NM, true, NS8, 4, 1, 2, 1, 2, 1, 1>(sp, dim3(grid), q.st)   \
                                            : launch_stream_inst<DT, NM, true, NS8, 4, 1, 2, 1, 2, 0, 1>(sp, dim3(grid), q.st)
            // (A/B, not shipped: the short 6-tenant launches -- q|k|v, o -- on 8-wave blocks, half the stages per wave, are 1 % faster on the
            //  step, 4.688 -> 4.639 ms, but 8 partial sums in another order are no longer bit-identical to every other form of the Linear:
            //  profiles/r04_decode_step_ab.txt.  The kernel template still takes NW = 8 with XL = 2.)
            switch (q.t_pad) {
                // Prefetch depth: TWO stages.  Same-process A/Bs of the whole step late in round 4 (profiles/r04_decode_step_ab.txt): 8 -> 6 -> 4
                // -> 2 stages each made the step faster (1 tenant: 3.57 -> 3.32 ms over the whole sequence of changes; 6 tenants: -0.3 %
                // for this form alone) -- the memory system is saturated by far fewer loads in flight than the register file can hold, and
                // beyond that point a deeper queue only adds latency (returns are in issue order).  (The parity of NS selects the
                // activation fragment set, so 2 is the minimum.)
                case 1: BD_XR(1, 2); break;
                case 2: BD_XR(2, 2); break;
                case 4: BD_XR(4, 2); break;
                case 6: BD_XR(6, 2); break;
                case 8: rc = q.epilogue == 1 ? launch_stream_inst<DT, 8, true, 2, 4, 1, 2, 1, 2, 1, 1>(sp, dim3(grid), q.st)
                                             : launch_stream_inst<DT, 8, true, 2, 4, 1, 2, 1, 2, 0, 1>(sp, dim3(grid), q.st); break;
                default: return BD_E_BAD_SHAPE;
            }
#undef BD_XR
            if (rc != BD_OK) return rc;
            return launch_status();
        }
// (tile-major weight: the norm-prologue forms run 2 stages deep -- 6 tenants: gate|up with its norm at 2 instead of 4 stages is -1 % on
//  the whole step; the plain / SwiGLU-only forms NS4 stages)
#define BD_PKW(NM, NS4, AX) (q.norm_w ? (q.epilogue == 1 ? launch_stream_inst<DT, NM, true, 2, 4, 1, AX, 1, 1, 1, 1>(sp, dim3(grid), q.st)   \
                                                         : launch_stream_inst<DT, NM, true, 2, 4, 1, AX, 1, 1, 0, 1>(sp, dim3(grid), q.st)) \
                                      : q.epilogue == 1 ? launch_stream_inst<DT, NM, true, NS4, 4, 1, AX, 1, 0, 1, 1>(sp, dim3(grid), q.st)   \
                                                        : launch_stream_inst<DT, NM, true, NS4, 4, 1, AX, 1, 0, 0, 1>(sp, dim3(grid), q.st))
#define BD_PK(NM, NS4) rc = q.w_tiled                                                                                                   \
                     ? (wnt ? BD_PKW(NM, NS4, 2) : BD_PKW(NM, NS4, 0))                                                                  \
                     : q.norm_w ? (q.epilogue == 1 ? launch_stream_inst<DT, NM, true, 2, 4, 1, 0, 1, 1, 1>(sp, dim3(grid), q.st)   \
                                                        : launch_stream_inst<DT, NM, true, 2, 4, 1, 0, 1, 1, 0>(sp, dim3(grid), q.st))  \
                     : q.epilogue == 1 ? launch_stream_inst<DT, NM, true, NS4, 4, 1, 0, 1, 0, 1>(sp, dim3(grid), q.st)            \
                     : q.W ? launch_stream_inst<DT, NM, true, NS4, 4, 1, 0, 1>(sp, dim3(grid), q.st)                              \
                           : launch_stream_inst<DT, NM, false, NS4, 4, 1, 0, 1>(sp, dim3(grid), q.st)
        switch (q.t_pad) {
            // depth of the plain / SwiGLU-only forms: 1 tenant 3 stages (8 until late in round 4: -6 % on the single-delta step over 8 -> 6 -> 4
            // -> 3), 2 tenants 4 (6: +3 %), 4 and more 4 (3 measured +0.6 % at 6 tenants)
            case 1: BD_PK(1, 3); break;
            case 2: BD_PK(2, 4); break;
            case 4: BD_PK(4, 4); break;
            case 6: BD_PK(6, 4); break;      // (NS 6 with nt weight loads measured 5-14 % slower: tools/ab_decode_depth.py)
            case 8: BD_PK(8, 4); break;
            // 9 .. 16 tenants in ONE launch (the reference publishes B = 16: notebooks/binary_gemm_kernel_triton.ipynb:759): plain and SwiGLU
            // launches; the fused-norm / resident-row forms end at 8 rows of K = 4096 anyway
#define BD_PKL(NM) rc = (q.norm_w || !q.W) ? BD_E_BAD_SHAPE                                                                             \
                     : q.w_tiled ? (q.epilogue == 1 ? (wnt ? launch_stream_inst<DT, NM, true, 4, 4, 1, 2, 1, 0, 1, 1>(sp, dim3(grid), q.st)    \
                                                            : launch_stream_inst<DT, NM, true, 4, 4, 1, 0, 1, 0, 1, 1>(sp, dim3(grid), q.st))  \
                                                    : (wnt ? launch_stream_inst<DT, NM, true, 4, 4, 1, 2, 1, 0, 0, 1>(sp, dim3(grid), q.st)    \
                                                            : launch_stream_inst<DT, NM, true, 4, 4, 1, 0, 1, 0, 0, 1>(sp, dim3(grid), q.st))) \
                     : q.epilogue == 1 ? launch_stream_inst<DT, NM, true, 4, 4, 1, 0, 1, 0, 1>(sp, dim3(grid), q.st)                     \
                                       : launch_stream_inst<DT, NM, true, 4, 4, 1, 0, 1>(sp, dim3(grid), q.st)
            case 12: BD_PKL(12); break;
            case 16: BD_PKL(16); break;
#undef BD_PKL
            default: return BD_E_BAD_SHAPE;
        }
#undef BD_PK
#undef BD_PKW
        if (rc != BD_OK) return rc;
        return launch_status();
    }
    // NS = stages of loads in flight per wave; bounded by the 256-VGPR budget of a 2-waves-per-SIMD block (hipcc spills beyond)
#define BD_STREAM(NM, NS4) rc = q.W ? launch_stream_tuned<DT, NM, true, NS4>(sp, dim3(grid), q.st) \
                                     : launch_stream_tuned<DT, NM, false, NS4>(sp, dim3(grid), q.st)
    if (nmask <= 1) BD_STREAM(1, 8);
    else if (nmask <= 2) BD_STREAM(2, 6);
    else if (nmask <= 3) BD_STREAM(3, 4);
    else if (nmask <= 4) BD_STREAM(4, 4);
    else if (nmask <= 6) BD_STREAM(6, 4);
    else BD_STREAM(8, 4);
#undef BD_STREAM
    if (rc != BD_OK) return rc;
    return launch_status();
}

// batches of more than 16 activation rows run as consecutive launches over chunks of floor(16 / M) batch entries (each chunk streams
// the base weight again; still far cheaper than M = 1 tiles of the MFMA tile kernels, which re-read it once per batch entry)
// ---- delta only, <= 16 activation rows per block, reference layout (delta_rows_kernel, bd_gemv_rows.h): the reference's published
//      binary_bmm (M = 1, one mask per batch entry) and binary_matmul (one mask, M <= 16) decode shapes
struct RowsPlan { int rpm, mc, cw; unsigned grid; };
inline bool rows_plan(const Problem& q, RowsPlan& pl) {
    if (q.W || q.alpha || q.accumulate || q.mask_tiled != 0 || q.M < 1 || q.M > 16 || q.sPb < 0) return false;
    if (q.N % 32 || q.K % 128 || q.sPb % 2 || q.sAb % 8 || q.sAm % 8 || q.sAb < 0 || q.sAm < 0 || !aligned16(q.A) || !aligned16(q.P)) return false;
    const int64_t lim = (1ll << 31) - 64;                       // 32-bit buffer offsets, out-of-range sentinel at 2 GiB
    const int nmask = q.sPb == 0 ? 1 : q.B;
    const int64_t xb = ((int64_t)(q.B - 1) * q.sAb + (int64_t)(q.M - 1) * q.sAm + q.K) * 2;
    const int64_t pb = ((int64_t)(nmask - 1) * q.sPb + (int64_t)(q.K / 32) * q.N) * 4;
    if (!(xb > 0 && xb < lim && pb > 0 && pb < lim)) return false;
    const int64_t cus = num_cus();
    if (q.sPb == 0) {                                            // one mask shared by every row: one chunk of B * M rows
        if ((int64_t)q.B * q.M > 16) return false;
        pl.rpm = q.B * q.M; pl.mc = 1;
    } else {
        pl.rpm = q.M;
        // two masks per block when that still gives every CU TWO blocks (<= 220 VGPRs, 66 KB of LDS at M = 1: they are co-resident and overlap
        // each other's start and end), one mask per block otherwise.  tools/bench_rows.py, us per launch at 1 / 2 / 4 masks per block:
        // B = 16, 4096^2 14.6 / 13.3 / 13.8; B = 8, 8192^2 23.1 / 18.8 / 21.6; B = 16, 8192^2 43.2 / 35.9 / 40.2; B = 8, 4096^2 8.2 / 8.7 / 12.9;
        // B = 4, 4096^2 6.1 / 7.9 / 12.1  (four masks per block -- 336 VGPRs, one block per CU -- stays behind the 804 test hook)
        pl.mc = (2 * q.M <= 16 && (int64_t)(q.N / 64) * ((q.B + 1) / 2) >= 2 * cus) ? 2 : 1;
    }
    if (g_forced_variant > 800 && g_forced_variant <= 804 && q.sPb != 0) pl.mc = g_forced_variant == 801 ? 1 : g_forced_variant == 802 ? 2 : 4;   // test hook
    { const int forced = (g_rows_tune >> 1) & 3; if (forced && q.sPb != 0) pl.mc = 1 << (forced - 1); }
    if (pl.mc * pl.rpm > 16) return false;
    const int64_t nch = q.sPb == 0 ? 1 : (q.B + pl.mc - 1) / pl.mc;
    // 64-column super-tiles (dwordx4 sign loads) when they give every CU a block, 32-column ones (dwordx2) otherwise
    pl.cw = (q.N % 64 == 0 && ((int64_t)(q.N / 64) * nch >= cus || (g_rows_tune & 32))) ? 4 : 2;
    if (g_rows_tune & 64) pl.cw = 2;
    if (pl.cw == 4 && (q.sPb % 4)) return false;
    if (pl.mc == 4 && pl.cw == 2) pl.cw = 4;                     // (the 804 hook keeps its one instantiation)
    if (pl.cw == 4 && q.N % 64) return false;
    pl.grid = (unsigned)((q.N / (16 * pl.cw)) * nch);
    return true;
}
// automatic choice: launches with enough blocks to occupy at least half of the chip; per-entry masks from 4 rows on, a shared mask always
// (tools/bench_rows.py, 800 against the streaming kernel, us: M = 1, one mask 5.4 vs 7.1 at 4096^2 and 7.5 vs 13.2 at 8192^2; M = 16 7.8 vs 9.4 and
// 12.2 vs 21.3; B = 2, M = 8 at 8192^2 10.1 vs 23.9 -- the streaming kernel re-reads all 16 activation rows for every 16-column tile)
inline bool rows_auto(const Problem& q) {
    RowsPlan pl;
    if ((g_rows_tune & 16) || !rows_plan(q, pl)) return false;
    const bool shared = q.sPb == 0 || q.B == 1;
    if (shared ? ((int64_t)q.B * q.M < g_rows_shared_min) : ((int64_t)q.B * q.M < 4)) return false;
    return (int64_t)pl.grid * 2 >= num_cus();
}
template <int DT, int MC, int CW>
int launch_rows_inst(const RowsParams& rp, int R, unsigned grid, hipStream_t st) {
    // 4 stages of loads per wave, nt policy on the sign loads (default policy and 2 stages measured within +-4 % of it, mixed signs:
    // tests/native/rows_bench.hip keeps those instantiations)
    auto kern = delta_rows_kernel<DT, MC, 4, 2, 0, CW>;
    static std::atomic<uint64_t> lds_done{0};
    const int lds = STREAM_LUT_BYTES + 4 * R * 16 * CW * 4;
    if (!ensure_dyn_lds((const void*)kern, STREAM_LUT_BYTES + 4 * 16 * 16 * CW * 4, lds_done)) return BD_E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, rp);
    return BD_OK;
}
template <int DT>
int launch_rows(const Problem& q) {
    RowsPlan pl;
    if (!rows_plan(q, pl)) return BD_E_BAD_SHAPE;
    RowsParams rp{};
    rp.X = (const unsigned short*)q.A; rp.P = (const uint32_t*)q.P; rp.C = q.C;
    rp.B = q.B; rp.M = q.M; rp.N = q.N; rp.K = q.K; rp.rpm = pl.rpm;
    rp.sXb = q.sAb; rp.sXm = q.sAm; rp.sPb = q.sPb; rp.sCb = q.sCb; rp.sCm = q.sCm;
    const int nmask = q.sPb == 0 ? 1 : q.B;
    rp.x_bytes = (uint32_t)(((int64_t)(q.B - 1) * q.sAb + (int64_t)(q.M - 1) * q.sAm + q.K) * 2);
    rp.p_bytes = (uint32_t)(((int64_t)(nmask - 1) * q.sPb + (int64_t)(q.K / 32) * q.N) * 4);
    rp.round_mode = q.round_mode; rp.out_f32 = (q.out_dtype == BD_F32);
    const int R = pl.mc * pl.rpm;
    int rc;
    if (pl.mc == 4) rc = launch_rows_inst<DT, 4, 4>(rp, R, pl.grid, q.st);
    else if (pl.mc == 2) rc = pl.cw == 4 ? launch_rows_inst<DT, 2, 4>(rp, R, pl.grid, q.st) : launch_rows_inst<DT, 2, 2>(rp, R, pl.grid, q.st);
    else rc = pl.cw == 4 ? launch_rows_inst<DT, 1, 4>(rp, R, pl.grid, q.st) : launch_rows_inst<DT, 1, 2>(rp, R, pl.grid, q.st);
    if (rc != BD_OK) return rc;
    t_last_variant = 800;
    return launch_status();
}

template <int DT>
int launch_gemv(const Problem& q, bool valu_form, bool col16 = false, bool stream = false) {
    const int cb = GEMV_MAX_R / q.M;
    auto one = [&](const Problem& c) {
        return stream ? launch_gemv_stream_chunk<DT>(c) : col16 ? launch_gemv_col16_chunk<DT>(c) : launch_gemv_chunk<DT>(c, valu_form);
    };
    if (q.B <= cb) return one(q);
    const int esz = q.out_dtype == BD_F32 ? 4 : 2;
    for (int b0 = 0; b0 < q.B; b0 += cb) {
        Problem c = q;
        c.B = q.B - b0 < cb ? q.B - b0 : cb;
        c.A = (const char*)q.A + (int64_t)b0 * q.sAb * 2;
        c.P = q.P + (int64_t)b0 * q.sPb;
        c.C = (char*)q.C + (int64_t)b0 * q.sCb * esz;
        if (q.alpha) c.alpha = q.alpha + (int64_t)b0 * q.sAlb;
        const int rc = one(c);
        if (rc != BD_OK) return rc;
    }
    return BD_OK;
}

template <int DT>
int launch_gemv_chunk(const Problem& q, bool valu_form) {
    GemvParams gp;
    gp.X = (const unsigned short*)q.A;
    gp.P = (const uint32_t*)q.P;
    gp.W = (const unsigned short*)q.W;
    gp.alpha = q.alpha;
    gp.C = q.C;
    gp.ws = (float*)q.ws;
    gp.B = q.B; gp.M = q.M; gp.N = q.N; gp.K = q.K; gp.R = q.B * q.M;
    gp.sXb = q.sAb; gp.sPb = q.sPb; gp.sCb = q.sCb;
    gp.sXm = (int)q.sAm; gp.sCm = (int)q.sCm; gp.ldw = (int)q.ldw; gp.sAlb = (int)q.sAlb; gp.gsz = q.N / q.G;
    gemv_split(q, gp.KS, gp.kslice);
    gp.round_mode = q.round_mode; gp.accumulate = q.accumulate; gp.out_f32 = (q.out_dtype == BD_F32);
    gp.tickets = nullptr;
    if (gp.KS > 1) {
        const int64_t need = GEMV_TICKET_BYTES + (int64_t)gp.KS * gp.R * q.N * 4;
        if (!q.ws || q.ws_bytes < need) return BD_E_WORKSPACE;
        gp.ws = (float*)((char*)q.ws + GEMV_TICKET_BYTES);
        // in-launch reduction unless the tile count exceeds the ticket area or the two-launch form is forced (g_gemv_two_launch)
        if ((q.N + 63) / 64 <= GEMV_TICKET_BYTES / 4 && !g_gemv_two_launch) gp.tickets = (uint32_t*)q.ws;
    }
    if (valu_form) {
        const int R = gp.R;      // rows >= R are computed and discarded, so the buckets are kept fine-grained
        if (R <= 1) launch_gemv_valu<DT, 1>(q, gp);
        else if (R <= 2) launch_gemv_valu<DT, 2>(q, gp);
        else if (R <= 3) launch_gemv_valu<DT, 3>(q, gp);
        else if (R <= 4) launch_gemv_valu<DT, 4>(q, gp);
        else if (R <= 6) launch_gemv_valu<DT, 6>(q, gp);
        else if (R <= 8) launch_gemv_valu<DT, 8>(q, gp);
        else if (R <= 12) launch_gemv_valu<DT, 12>(q, gp);
        else launch_gemv_valu<DT, 16>(q, gp);
    } else {
        const int nm = q.sPb == 0 ? 1 : q.B;     // accumulator sets = distinct masks (extra sets of a bucket repeat the last mask)
        if (nm <= 1) launch_gemv_mfma<DT, 1>(q, gp);
        else if (nm <= 2) launch_gemv_mfma<DT, 2>(q, gp);
        else if (nm <= 3) launch_gemv_mfma<DT, 3>(q, gp);
        else if (nm <= 4) launch_gemv_mfma<DT, 4>(q, gp);
        else if (nm <= 6) launch_gemv_mfma<DT, 6>(q, gp);
        else if (nm <= 8) launch_gemv_mfma<DT, 8>(q, gp);
        else if (nm <= 12) launch_gemv_mfma<DT, 12>(q, gp);
        else launch_gemv_mfma<DT, 16>(q, gp);
    }
    if (gp.KS > 1 && !gp.tickets) {
        dim3 g2((unsigned)((q.N + 255) / 256), (unsigned)gp.R);
        hipLaunchKernelGGL((gemv_reduce_kernel<DT>), g2, dim3(256), 0, q.st, gp);
    }
    return launch_status();
}


inline GemmParams make_params(const Problem& q, int BM, int BN) {
    GemmParams p;
    p.A = (const char*)q.A; p.P = q.P; p.C = (char*)q.C; p.W = (const char*)q.W; p.alpha = q.alpha;
    p.M = q.M; p.N = q.N; p.K = q.K;
    p.tiles_m = (q.M + BM - 1) / BM; p.tiles_n = (q.N + BN - 1) / BN;
    p.tile_m0 = 0; p.tile_n0 = 0; p.ksplit = 1;
    p.sAb = q.sAb; p.sPb = q.sPb; p.sCb = q.sCb;
    p.sAm = (int)q.sAm; p.sCm = (int)q.sCm; p.ldw = (int)q.ldw;
    p.sAlb = (int)q.sAlb; p.gsz = q.N / q.G;
    p.round_mode = q.round_mode; p.accumulate = q.accumulate;
    p.nbatch = q.B; p.nent = q.B;
    // Tile walk order (profiles/r01_tile_order.txt): delta-only = n fastest (the XCD's run shares X row panels; the mask is tiny).
    // Fused = groups of 4 tile rows: each XCD's run covers a ~4 x 8 block of tiles, which minimises X + W bytes per XCD
    // (+3..5 % on the MLP shapes over n-fastest, equal to m-fastest on the single-round ones).
    p.group_m = q.W != nullptr ? (p.tiles_m < 4 ? p.tiles_m : 4) : 1;
    if (g_forced_group_m > 0) p.group_m = g_forced_group_m < p.tiles_m ? g_forced_group_m : p.tiles_m;
    return p;
}

template <class Cfg, int SCHED> struct TileKernel { static auto get() { return delta_gemm_kernel<Cfg>; } };
#ifdef BD_AB_VARIANTS
template <class Cfg> struct TileKernel<Cfg, 1> { static auto get() { return delta_gemm_pp_kernel<Cfg>; } };
#endif
template <class Cfg> struct TileKernel<Cfg, 2> { static auto get() { return delta_gemm_pf_kernel<Cfg>; } };
template <class Cfg> struct TileKernel<Cfg, 3> { static auto get() { return delta_gemm_fx_kernel<Cfg>; } };

// SCHED 0 = single barrier per k-tile (bd_gemm_mfma.h; the small-M tiles), 1 = half-tile ping-pong (bd_gemm_pp.h),
// 2 = full-tile ping-pong (bd_gemm_pf.h; the shipped delta-only schedule), 3 = one-pass fused, two accumulator sets (bd_gemm_fx.h).
template <class Cfg, int SCHED = 0>
int launch_tile(const Problem& q, int col0 = 0) {          // col0 > 0: only tile columns [col0, tiles_n) (launch_fused_tail_split)
    GemmParams p = make_params(q, Cfg::BM, Cfg::BN);
    if (col0 > 0) { p.tile_n0 = col0; p.tiles_n -= col0; if (p.tiles_n <= 0) return BD_OK; }
    auto kern = TileKernel<Cfg, SCHED>::get();
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    // Experiment kept behind bd_set_launch_chunking(1): issue a multi-round problem as several launches of at most one tile per CU
    // (an mc x nc block of tiles each, via GemmParams::tile_m0 / tile_n0).  Hypothesis: rounds inside one launch drift apart and that
    // is why 8192x4096x4096 runs 10 % below back-to-back 2048-row launches.  Measured (profiles/r01_launch_chunking.txt): no gain
    // (-0..3 %, -15 % with 6 tenants) -- the back-to-back figure was warm-cache re-reads of the same operands, not synchronisation.
    const long long cus = num_cus();
    const long long total = (long long)p.tiles_m * p.tiles_n * q.B;
    if (SCHED >= 2 && g_launch_chunking && total > cus && q.B <= cus) {
        const long long per = cus / q.B;                                   // tiles per launch and batch entry
        // orientation A: all tile rows x as many tile columns as fit; B: all tile columns x as many rows as fit
        long long mcA = p.tiles_m < per ? p.tiles_m : per, ncA = per / mcA; if (ncA > p.tiles_n) ncA = p.tiles_n;
        long long ncB = p.tiles_n < per ? p.tiles_n : per, mcB = per / ncB; if (mcB > p.tiles_m) mcB = p.tiles_m;
        const long long nA = ((p.tiles_m + mcA - 1) / mcA) * ((p.tiles_n + ncA - 1) / ncA);
        const long long nB = ((p.tiles_m + mcB - 1) / mcB) * ((p.tiles_n + ncB - 1) / ncB);
        const int mc = (int)(nA <= nB ? mcA : mcB), nc = (int)(nA <= nB ? ncA : ncB);
        for (int m0 = 0; m0 < p.tiles_m; m0 += mc)
            for (int n0 = 0; n0 < p.tiles_n; n0 += nc) {
                GemmParams c = p;
                c.tile_m0 = m0; c.tile_n0 = n0;
                c.tiles_m = p.tiles_m - m0 < mc ? p.tiles_m - m0 : mc;
                c.tiles_n = p.tiles_n - n0 < nc ? p.tiles_n - n0 : nc;
                if (c.group_m > c.tiles_m) c.group_m = c.tiles_m;
                dim3 grid((unsigned)(c.tiles_m * c.tiles_n), (unsigned)q.B);
                hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, c);
            }
        return launch_status();
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)q.B);
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    return launch_status();
}

// Four-wave persistent kernels (bd_gemm_w4.h): grid = min(batch x tiles, CUs) workgroups of 256 threads, each walking its share of the
// (batch entry, tile) stream.
// cols >= 0: only tile columns [0, cols) of the problem (the rest goes to another launch: launch_fused_tail_split)
// col0 > 0: only tile columns [col0, tiles_n) (the tail launch of launch_fused_tail_split)
template <class Cfg>
int launch_w4(const Problem& q, int cols = -1, int col0 = 0) {
    GemmParams p = make_params(q, Cfg::BM, Cfg::BN);
    if (cols >= 0) p.tiles_n = cols;
    if (col0 > 0) { p.tile_n0 = col0; p.tiles_n -= col0; if (p.tiles_n <= 0) return BD_OK; }
    auto kern = delta_gemm_w4_kernel<Cfg>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    const long long total = (long long)p.tiles_m * p.tiles_n * q.B, cus = num_cus();
    if (total <= 0) return BD_OK;
    dim3 grid((unsigned)(total < cus ? total : cus));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    return launch_status();
}
// share of the CU-rounds a tile count keeps busy (1.0 = every round full)
inline double round_fill(long long tiles) {
    const long long cus = num_cus();
    return (double)tiles / (double)(((tiles + cus - 1) / cus) * cus);
}

template <int DT, bool FUSED, bool OUT_F32>
int launch_generic(const Problem& q) {
    const GemmParams p = make_params(q, 64, 64);
    dim3 grid((unsigned)((q.N + 63) / 64), (unsigned)((q.M + 63) / 64), (unsigned)q.B);
    hipLaunchKernelGGL((delta_gemm_generic_kernel<DT, FUSED, OUT_F32>), grid, dim3(256), 0, q.st, p);
    return launch_status();
}

// 256x256 vs 256x128 ping-pong tile: rounds of CU-wide tile waves x measured relative cost of one tile
// (256x128 does half the work of a 256x256 tile at ~0.85x its MFMA rate; DESIGN.md "tile choice").
inline int choose_big_tile(const Problem& q) {
    const long long cus = num_cus();
    const long long tm = (q.M + 255) / 256;
    const long long t0 = tm * ((q.N + 255) / 256) * q.B, t5 = tm * ((q.N + 127) / 128) * q.B;
    const double c0 = (double)((t0 + cus - 1) / cus) * 1.0, c5 = (double)((t5 + cus - 1) / cus) * (0.5 / 0.85);
    return c5 < c0 ? 5 : 0;
}

// one-pass fused kernel: 256x128 tile vs 128x128 tile.  Rounds of CU-wide tile waves x measured relative tile cost: a 128x128 tile
// takes ~0.70 of a 256x128 tile's k loop (profiles/r01_small_m.txt), so it wins exactly when it does not add rounds (M <~ 1024 at
// N = 4096; M <= 256 at N = 11008).
// (round 5: the 128x128 tile on the four-wave schedule, variant 20, takes ~0.64 of a four-wave 256x128 tile's k loop -- tools/ab_w4_128.py:
//  M = 512 gate|up 3 rounds in 145.9 us vs 2 rounds in 149.9; M = 768 q|k|v 133.5 vs 138.7 -- and is 5..17 % faster than the 8-wave 128x128
//  tile on every shape measured: profiles/r05_mt_prefill_tiles.txt)
inline int choose_fused_tile(const Problem& q, double rel128 = 0.70) {
    const long long cus = num_cus();
    const long long tn = (q.N + 127) / 128;
    const long long t8 = (q.M + 255) / 256 * tn * q.B, t9 = (q.M + 127) / 128 * tn * q.B;
    const double c8 = (double)((t8 + cus - 1) / cus), c9 = (double)((t9 + cus - 1) / cus) * rel128;
    return c9 < c8 ? 9 : 8;
}

// Tail split of a multi-round fused launch (one mask, 16-bit output).  The persistent 256x128 kernel needs ceil(tiles / CUs) rounds; when
// the last round is mostly empty (Llama-2-7B gate|up at 2048 rows: 1376 tiles = 5.4 rounds -> 6), the tile columns of the FULL rounds go
// to it and the remaining columns to the 8-wave kernel on 128x128 tiles (twice as many, ~0.70 of a round each), launched behind it on
// the same stream: 5 + 0.53 instead of 6 rounds.  Returns the number of 128-wide tile columns for the main launch, or -1 = no split.
inline int fused_tail_split_cols(const Problem& q) {
    if (q.B != 1) return -1;
    const long long cus = num_cus();
    const long long tm = (q.M + 255) / 256, tn = (q.N + 127) / 128, T = tm * tn;
    if (T <= cus || T % cus == 0) return -1;
    const long long full = T / cus;                       // full rounds
    const long long n_main = full * cus / tm;             // whole tile columns inside them
    const long long rem = tn - n_main;
    if (n_main <= 0 || rem <= 0) return -1;
    const double now = (double)((T + cus - 1) / cus);
    const long long t_rem = ((q.M + 127) / 128) * rem;
    const double split = (double)((n_main * tm + cus - 1) / cus) + (double)((t_rem + cus - 1) / cus) * 0.70;
    return split < now - 0.2 ? (int)n_main : -1;
}

// Split-k for the one-pass fused kernel at mid-size M (16 < M <= 512: decode batches, speculative decoding, short prefills), where
// even 128x128 tiles leave most CUs idle and the launch is bound by how fast FEW CUs can stream W.  KS slices of k per tile (each
// block writes an fp32 partial [M][N] slab to the workspace) + one reduce launch.  Returns KS (1 = do not split).
constexpr int64_t SPLITK_WS_CAP = 64ll << 20;
inline int splitk_factor(int B, int M, int N, int K) {
    if (M <= 16 || M > 512 || K % 64 || N % 8) return 1;
    const long long cus = num_cus();
    const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128) * B;
    if (tiles * 2 > cus) return 1;                       // already at least half a round of tiles
    long long ks = cus / tiles;
    if (ks > 8) ks = 8;
    const long long by_k = (K / 64) / 8;                 // at least 8 k-tiles (512 k) per slice
    if (ks > by_k) ks = by_k;
    while (ks > 1 && (int64_t)B * ks * M * N * 4 > SPLITK_WS_CAP) --ks;
    return ks < 2 ? 1 : (int)ks;
}

// the reduce launch of the split-k tile kernels: partial slabs [B][KS][M][N] -> C.  A pending post-norm (Problem::pn_w) rides on it when the rows
// are whole 16-byte chunks of at most 8192 columns: one launch instead of reduce + RMSNorm.
template <int DT>
inline void launch_splitk_reduce(const Problem& q, const float* part, int KS, int accumulate) {
    if (q.pn_w && q.out_dtype != BD_F32 && q.N % 8 == 0 && q.N <= 8192 && aligned16(q.C) && q.sCm % 8 == 0 && q.sCb % 8 == 0) {
        hipLaunchKernelGGL((splitk_reduce_norm_kernel<DT>), dim3((unsigned)(q.B * q.M)), dim3(256), 0, q.st, part, (unsigned short*)q.C,
                           (const unsigned short*)q.pn_w, (unsigned short*)q.pn_h, KS, q.M, q.N, (long long)q.sCb, (long long)q.sCm,
                           (long long)q.sHb, (long long)q.sHm, (long long)q.s_pnw, accumulate, q.pn_eps);
        t_post_norm_done = 1;
        return;
    }
    const long long per = (long long)q.M * q.N / 4;
    dim3 g2((unsigned)((per + 255) / 256), (unsigned)q.B);
    hipLaunchKernelGGL((splitk_reduce_kernel<DT>), g2, dim3(256), 0, q.st, part, q.C, q.B, KS, q.M, q.N,
                       (long long)q.sCb, (int)q.sCm, q.out_dtype == BD_F32 ? 1 : 0, accumulate);
}

template <int DT, int BM>
int launch_fused_splitk_bm(const Problem& q, int KS) {
    const int64_t need = GEMV_TICKET_BYTES + (int64_t)q.B * KS * q.M * q.N * 4;
    if (!q.ws || q.ws_bytes < need) return BD_E_WORKSPACE;
    float* part = (float*)((char*)q.ws + GEMV_TICKET_BYTES);
    Problem c = q;
    c.C = part; c.out_dtype = BD_F32; c.sCm = q.N; c.sCb = (int64_t)q.M * q.N;     // slab y = b * KS + ks
    using Cfg = FxCfg<DT, BM, 128, 4, true, 1>;
    GemmParams p = make_params(c, Cfg::BM, Cfg::BN);
    p.ksplit = KS;
    auto kern = delta_gemm_fx_kernel<Cfg>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(q.B * KS));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    launch_splitk_reduce<DT>(q, part, KS, 0);
    return launch_status();
}
// up to 64 rows per mask: 64-row tiles (half the MFMA / LDS work of a padded 128-row tile)
template <int DT>
int launch_fused_splitk(const Problem& q, int KS) {
    return q.M <= 64 ? launch_fused_splitk_bm<DT, 64>(q, KS) : launch_fused_splitk_bm<DT, 128>(q, KS);
}

// ---- pair tiles (bd_gemm_fx.h, PAIR): two batch entries of <= 64 rows share one 128 x 128 tile (and its W stream); optional split-k
inline bool pair_ok(const Problem& q) {
    return q.W && q.B >= 2 && q.M >= 1 && q.M <= 64 && fast_ok(q) && q.sAb >= 0 && q.sAb < (1ll << 30) && q.sPb >= 0 && q.sPb < (1ll << 29);
}
static const int g_pair_ks_env = env_int("BD_PAIR_SPLITK", 0);       // A/B: force the k-slice count of the split pair tiles (0 = the rule)
inline int pair_splitk_dims(int B, int N, int K) {    // k slices so that pairs x column tiles x slices ~ fills the CUs (>= 512 k per slice)
    const long long cus = num_cus();
    const long long tiles = (long long)((B + 1) / 2) * ((N + 127) / 128);
    if (K % 64 || N % 8 || tiles * 2 > cus) return 1;
    if (g_pair_ks_env > 1) { int f = g_pair_ks_env; while (f > 1 && (K / f < 512 || (K / 64) % f)) --f; return f; }
    long long ks = cus / tiles;
    if (ks > 8) ks = 8;
    while (ks > 1 && K / ks < 512) --ks;
    return (int)ks;
}
inline int pair_splitk(const Problem& q) { return pair_splitk_dims(q.B, q.N, q.K); }
template <int DT, bool OUT_F32>
int launch_pair(const Problem& q) {
    using Cfg = FxCfg<DT, 128, 128, 4, OUT_F32, 1, 1>;
    GemmParams p = make_params(q, Cfg::BM, Cfg::BN);          // tiles_m = 1: the tile's two halves are two batch entries
    auto kern = delta_gemm_fx_kernel<Cfg>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)((q.B + 1) / 2));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    return launch_status();
}
template <int DT>
int launch_pair_splitk(const Problem& q, int KS) {
    const int64_t need = GEMV_TICKET_BYTES + (int64_t)q.B * KS * q.M * q.N * 4;
    if (!q.ws || q.ws_bytes < need) return BD_E_WORKSPACE;
    float* part = (float*)((char*)q.ws + GEMV_TICKET_BYTES);
    Problem c = q;
    c.C = part; c.out_dtype = BD_F32; c.sCm = q.N; c.sCb = (int64_t)q.M * q.N;     // slab y = entry * KS + ks
    c.accumulate = 0;                                                                // (the residual is added by the reduce launch)
    using Cfg = FxCfg<DT, 128, 128, 4, true, 1, 1>;
    GemmParams p = make_params(c, Cfg::BM, Cfg::BN);
    p.ksplit = KS;
    auto kern = delta_gemm_fx_kernel<Cfg>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(((q.B + 1) / 2) * KS));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    launch_splitk_reduce<DT>(q, part, KS, q.accumulate);
    return launch_status();
}

// four-wave PAIR tiles (W4Cfg PAIR = 1): two batch entries of <= 64 rows per 128 x 128 tile, persistent over (pair, column tile)
template <int DT, bool OUT_F32, int EPI = 0>
int launch_w4_pair(const Problem& q) {
    using Cfg = W4Cfg<DT, 128, 128, true, OUT_F32, 1 | 8192, EPI, 1>;
    GemmParams p = make_params(q, Cfg::BM, Cfg::BN);          // tiles_m = 1: the tile's two wave rows are two batch entries
    p.nent = q.B; p.nbatch = (q.B + 1) / 2;
    auto kern = delta_gemm_w4_kernel<Cfg>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    const long long total = (long long)p.tiles_n * p.nbatch, cus = num_cus();
    if (total <= 0) return BD_OK;
    dim3 grid((unsigned)(total < cus ? total : cus));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    return launch_status();
}
// ... + split-k: the persistent stream runs over (pair, k slice, column tile); fp32 partial slabs [entry * KS + slice][M][N], then the reduce launch
// of launch_pair_splitk (residual added there).  KS must divide the k-tile count.
template <int DT>
int launch_w4_pair_splitk(const Problem& q, int KS) {
    const int64_t need = GEMV_TICKET_BYTES + (int64_t)q.B * KS * q.M * q.N * 4;
    if (!q.ws || q.ws_bytes < need) return BD_E_WORKSPACE;
    if ((q.K / 64) % KS) return BD_E_BAD_SHAPE;
    float* part = (float*)((char*)q.ws + GEMV_TICKET_BYTES);
    Problem c = q;
    c.C = part; c.out_dtype = BD_F32; c.sCm = q.N; c.sCb = (int64_t)q.M * q.N;     // slab = entry * KS + slice
    c.accumulate = 0;
    using Cfg = W4Cfg<DT, 128, 128, true, true, 1 | 8192, 0, 1>;
    GemmParams p = make_params(c, Cfg::BM, Cfg::BN);
    p.ksplit = KS; p.nent = q.B; p.nbatch = ((q.B + 1) / 2) * KS;
    auto kern = delta_gemm_w4_kernel<Cfg>;
    static std::atomic<uint64_t> lds_done{0};
    if (!ensure_dyn_lds((const void*)kern, Cfg::LDS_BYTES, lds_done)) return BD_E_LAUNCH;
    const long long total = (long long)p.tiles_n * p.nbatch, cus = num_cus();
    dim3 grid((unsigned)(total < cus ? total : cus));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, q.st, p);
    launch_splitk_reduce<DT>(q, part, KS, q.accumulate);
    return launch_status();
}

template <int DT, bool FUSED, bool OUT_F32>
int dispatch3(const Problem& q) {
    int v = g_forced_variant;
    if (v > 200 && v <= 264) v = 200;        // 200 + KS: decode path with a forced k-split (test hook)
    if (v > 300 && v <= 364) v = 300;        // 300 (+ KS): force the VALU sign-flip decode kernel
    if (v > 400 && v <= 464) v = 400;        // 400 (+ KS): force the MFMA + LUT decode kernel
    if (v > 500 && v <= 564) v = 500;        // 500 (+ KS): force the no-split-k 16-column decode kernel
    if (v > 600 && v <= 664) v = 600;        // 600 (+ columns per block / 4): force the streaming decode kernel
    if (v > 800 && v <= 804) v = 800;        // 800 (+ masks per block): force the one-mask-per-row delta kernel (bd_gemv_rows.h)
    if (v == 700) {                           // 700: the loader / consumer decode kernel (packed sign layout, fused Linear only); the
        if (q.mask_tiled != 2 || !q.W) return BD_E_BAD_SHAPE;     // stream path picks it up (launch_gemv_stream_chunk, ring_wanted)
        v = 600;
    }
    if (q.epilogue == 1 && q.M > GEMV_MAX_M) {        // prefill-size SwiGLU launch (bd_binary_linear_swiglu): the four-wave kernel's 256 x 128 tile,
        if (v >= 0 && v != 15 && v != 21) return BD_E_BAD_SHAPE;                  // or its pair tile for several entries of <= 64 rows (round 6)
        if (v < 0) v = (FUSED && q.M <= 64 && q.B >= 2 && pair_ok(q)) ? 21 : 15;
    }
    if (v < 0) {
        if (gemv_ok(q)) v = 200;
        else if (!fast_ok(q)) v = 100;
        else if (FUSED && q.M > 16 && q.M <= 64 && q.B >= 2 && pair_ok(q)) {
            // several batch entries of <= 64 rows (multi-tenant prefill of short prompts, demo_backend.py:297-299): two entries per 128-row
            // tile share one W stream; narrow outputs add split-k (tools/bench_mt_prefill.py, 6 tenants x 64 rows, Mistral-7B shapes:
            // q|k|v 59 -> 50 us, o 39 -> 33, gate|up 201 -> 177, down 130 -> 91; profiles/r04_mt_prefill_tiles.txt)
            // Round 5: the same tiles on the FOUR-WAVE schedule (bd_gemm_w4.h PAIR; variants 18 / 19) -- q|k|v 51.6 -> 51.4 us, o 35.3 -> 32.2,
            // gate|up 192 -> 162 (1.12 PF useful), down 94.6 -> 83.4: -12 % per layer (profiles/r05_mt_prefill_tiles.txt).  Its split-k
            // slices are whole runs of k-tiles: when the slice count does not divide K / 64 the 8-wave kernel keeps the launch.
            const int kp = pair_splitk(q);
            const bool split = kp > 1 && q.N % 8 == 0 && q.sCm % 4 == 0 && q.sCb % 4 == 0 && q.ws &&
                               q.ws_bytes >= GEMV_TICKET_BYTES + (int64_t)q.B * kp * q.M * q.N * 4;
            v = split ? (((q.K / 64) % kp == 0) ? 19 : 17) : 18;
        }
        else if (FUSED && q.M > 16 && !q.accumulate && q.sCm % 4 == 0 && q.sCb % 4 == 0 && splitk_factor(q.B, q.M, q.N, q.K) > 1 && q.ws &&
                 q.ws_bytes >= GEMV_TICKET_BYTES + (int64_t)q.B * splitk_factor(q.B, q.M, q.N, q.K) * q.M * q.N * 4) v = 10;
        else if (FUSED && q.M > 16 && q.M <= 64)                  // (not split: there are enough 64-row tiles)
            // 64x256 tiles (twice the MFMAs per X fragment read) once they fill at least half the CUs, else 64x128 for the parallelism:
            // 6 tenants x 64 rows: q+k+v 56 vs 76 us, gate+up 207 vs 253 us, but o 54 vs 38 us (tools/bench_mt_prefill.py)
            v = ((long long)((q.N + 255) / 256) * q.B * 2 >= num_cus()) ? 12 : 11;
        else if (FUSED && q.M > 16) {
            v = choose_fused_tile(q, OUT_F32 ? 0.70 : 0.64);     // fused: one-pass kernel (profiles/r01_fx_vs_two_loop.txt, r01_small_m.txt,
                                          // r01_mid_m.txt: also for 16 < M <= 64, rows padded to the 128-row tile)
            // same tiles on the four-wave persistent schedule: 256x128 +4..7 % on every shape measured (profiles/r03_w4_*.txt), 128x128
            // +5..17 % (round 5)
            if (v == 8 && !OUT_F32) v = 14;
            if (v == 9 && !OUT_F32) v = 20;
        } else if (q.M > 128) {
            v = choose_big_tile(q);
            // four-wave persistent 256x256 kernel once its tiles keep >= 80 % of the CU-rounds busy (it has no 256x128 form)
            if (round_fill((long long)((q.M + 255) / 256) * ((q.N + 255) / 256) * q.B) >= 0.8) v = 13;
        }
        else if (q.M > 64) v = 1;
        else if (q.M > 32) v = 2;
        else v = 3;
    } else {
        if ((v == 200 || v == 300 || v == 400 || v == 500 || v == 600) && !gemv_ok(q)) return BD_E_BAD_SHAPE;
        if (v >= 0 && v <= 20 && !fast_ok(q)) return BD_E_BAD_SHAPE;
        if ((v == 16 || v == 17 || v == 18 || v == 19 || v == 20) && !FUSED) return BD_E_BAD_SHAPE;
        if (v == 15 && (!FUSED || OUT_F32 || q.epilogue != 1)) return BD_E_BAD_SHAPE;
        if (v == 13 && FUSED) return BD_E_BAD_SHAPE;
        if (v == 14 && !FUSED) return BD_E_BAD_SHAPE;
        if ((v == 8 || v == 9 || v == 10 || v == 11 || v == 12) && !FUSED) return BD_E_BAD_SHAPE;
        if (v == 10 && (q.N % 8 || q.sCm % 4 || q.sCb % 4 || q.accumulate)) return BD_E_BAD_SHAPE;
        if (FUSED && q.accumulate && !(v == 8 || v == 9 || v == 11 || v == 12 || v == 14 || v == 16 || v == 17 || v == 18 || v == 19 || v == 20 || v == 200 || v == 300 || v == 400 || v == 500 || v == 600))
            return BD_E_BAD_SHAPE;            // residual epilogue: one-pass fused tiles and the decode kernels only
    }
    t_last_variant = v;
    switch (v) {
        // Shipped MFMA tile kernels: 0 / 5 = delta-only full-tile ping-pong at 256x256 / 256x128 (bd_gemm_pf.h), 8 / 9 / 10 = one-pass
        // fused (bd_gemm_fx.h), 1 / 2 / 3 = small-M single-barrier tiles (bd_gemm_mfma.h).  The rejected schedules -- 4 (256x256
        // single barrier), 6 / 7 (half-tile ping-pong), fused 0 / 5 (two k loops over one accumulator set) -- are compiled only with
        // -DBD_AB_VARIANTS (the native harness); the shipped library answers BD_E_BAD_SHAPE for them.
        case 0:
            if constexpr (!FUSED) return launch_tile<GemmCfg<DT, 256, 256, 2, 4, 4, FUSED, OUT_F32, 1>, 2>(q);    // full-tile ping-pong, LUT sign expansion
#ifdef BD_AB_VARIANTS
            else return launch_tile<GemmCfg<DT, 256, 256, 2, 4, 4, FUSED, OUT_F32, 2>, 1>(q);   // 2-slot base ring
#else
            else return BD_E_BAD_SHAPE;
#endif
        case 5:
            if constexpr (!FUSED) return launch_tile<GemmCfg<DT, 256, 128, 2, 4, 4, FUSED, OUT_F32, 0>, 2>(q);
#ifdef BD_AB_VARIANTS
            else return launch_tile<GemmCfg<DT, 256, 128, 2, 4, 4, FUSED, OUT_F32, 0>, 2>(q);
#else
            else return BD_E_BAD_SHAPE;
#endif
#ifdef BD_AB_VARIANTS
        case 6: return launch_tile<GemmCfg<DT, 256, 256, 2, 4, 4, FUSED, OUT_F32, 2>, 1>(q);      // half-tile ping-pong (A/B reference)
        case 7: return launch_tile<GemmCfg<DT, 256, 128, 2, 4, 4, FUSED, OUT_F32, 2>, 1>(q);      // half-tile ping-pong 256x128 (A/B)
        case 4: return launch_tile<GemmCfg<DT, 256, 256, 2, 4, 4, FUSED, OUT_F32>>(q);   // single-barrier 256x256 (A/B reference)
#endif
        case 8:
            if constexpr (FUSED) return launch_tile<FxCfg<DT, 256, 128, 3, OUT_F32, 1>, 3>(q);
            else return BD_E_BAD_SHAPE;
        case 10: {   // one-pass fused, 128x128 tile, split-k over blockIdx.y + reduce launch (forced: KS from the rule, at least 2)
            if constexpr (FUSED) {
                int ks = splitk_factor(q.B, q.M, q.N, q.K);
                if (ks < 2) ks = (q.K / 64 >= 2) ? 2 : 1;
                if (ks < 2) return BD_E_BAD_SHAPE;
                return launch_fused_splitk<DT>(q, ks);
            } else return BD_E_BAD_SHAPE;
        }
        case 11:     // one-pass fused, 64x128 tile: up to 64 rows per mask (multi-tenant prefill of short prompts, demo_backend.py:297-299:
                     // M = 64 x 6 tenants) -- the 128-row tile does twice the MFMA and LDS work for rows that do not exist
            if constexpr (FUSED) return launch_tile<FxCfg<DT, 64, 128, 4, OUT_F32, 1>, 3>(q);
            else return BD_E_BAD_SHAPE;
        case 12:     // one-pass fused, 64x256 tile: wide outputs / many tenants (see the rule above)
            if constexpr (FUSED) return launch_tile<FxCfg<DT, 64, 256, 3, OUT_F32, 1>, 3>(q);
            else return BD_E_BAD_SHAPE;
        case 16:     // pair tiles: two batch entries of <= 64 rows per 128x128 tile (multi-tenant prefill of short prompts)
            if constexpr (FUSED) { if (!pair_ok(q)) return BD_E_BAD_SHAPE; return launch_pair<DT, OUT_F32>(q); }
            else return BD_E_BAD_SHAPE;
        case 17: {   // pair tiles + split-k (narrow outputs: o / down of a 6-tenant request are 96 tiles on 256 CUs)
            if constexpr (FUSED) {
                if (!pair_ok(q) || q.N % 8 || q.sCm % 4 || q.sCb % 4) return BD_E_BAD_SHAPE;
                int ks = pair_splitk(q);
                if (ks < 2) ks = (q.K / 64 >= 2) ? 2 : 1;
                if (ks < 2 || ks > q.K / 64) return BD_E_BAD_SHAPE;       // every k slice holds at least one 64-k tile (no empty slices)
                return launch_pair_splitk<DT>(q, ks);
            } else return BD_E_BAD_SHAPE;
        }
        case 18:     // FOUR-WAVE pair tiles (bd_gemm_w4.h PAIR): one wave per SIMD, wave tile 64 x 64, persistent over (pair, column tile)
            if constexpr (FUSED) { if (!pair_ok(q)) return BD_E_BAD_SHAPE; return launch_w4_pair<DT, OUT_F32>(q); }
            else return BD_E_BAD_SHAPE;
        case 21:     // 18 with the SwiGLU epilogue: the gate|up projection of a multi-tenant request of short prompts (6 x 64 rows), no [M, 2 I] round trip
            if constexpr (FUSED && !OUT_F32) { if (!pair_ok(q) || q.epilogue != 1) return BD_E_BAD_SHAPE; return launch_w4_pair<DT, false, 1>(q); }
            else return BD_E_BAD_SHAPE;
        case 20:     // four-wave fused 128x128 tile, ONE batch entry per tile (prompts of 65 .. 128 rows per tenant; W4Cfg TM = 2, PAIR = 0)
            if constexpr (FUSED) return launch_w4<W4Cfg<DT, 128, 128, true, OUT_F32, 1 | 8192, 0, 0>>(q);
            else return BD_E_BAD_SHAPE;
        case 19: {   // four-wave pair tiles + split-k (the narrow outputs of a multi-tenant request)
            if constexpr (FUSED) {
                if (!pair_ok(q) || q.N % 8 || q.sCm % 4 || q.sCb % 4) return BD_E_BAD_SHAPE;
                int ks = pair_splitk(q);
                if (ks < 2) ks = 2;
                while (ks > 1 && (q.K / 64) % ks) --ks;                   // the slices are whole, equal runs of k-tiles
                if (ks < 2) return BD_E_BAD_SHAPE;
                return launch_w4_pair_splitk<DT>(q, ks);
            } else return BD_E_BAD_SHAPE;
        }
        case 9:      // one-pass fused, 128x128 tile, 4-slot ring: twice the tiles when 256x128 cannot fill the CUs (128 < M <~ 768)
            if constexpr (FUSED) return launch_tile<FxCfg<DT, 128, 128, 4, OUT_F32, 1>, 3>(q);
            else return BD_E_BAD_SHAPE;
        // (W4Cfg OPT = 1 | 8192: LUT sign expansion + the k loop unrolled by four -- bit-identical outputs, -3.6 / -5.5 % cycles per k-tile,
        //  +0.6 % on the timed prefill step in a same-box A/B of two library builds; profiles/r05_w4_cycles.txt)
        case 13:     // four-wave persistent delta-only kernel, 256x256 tile, LUT sign expansion (bd_gemm_w4.h)
            if constexpr (!FUSED) return launch_w4<W4Cfg<DT, 256, 256, false, OUT_F32, 1 | 8192>>(q);
            else return BD_E_BAD_SHAPE;
        case 14:     // four-wave persistent one-pass fused kernel, 256x128 tile, LUT sign expansion (+1.5 % over VALU expansion, same box)
            if constexpr (FUSED) {
                if constexpr (!OUT_F32) {
                    const int cols = (g_forced_variant < 0 && g_tail_split) ? fused_tail_split_cols(q) : -1;
                    if (cols > 0) {
                        const int rc = launch_w4<W4Cfg<DT, 256, 128, true, false, 1 | 8192>>(q, cols);
                        if (rc != BD_OK) return rc;
                        // the remaining tile columns on 128x128 tiles -- four-wave since round 5 (variant 20's kernel: 5..17 % faster than the
                        // 8-wave 128x128 tile on every shape measured); same 128-wide tile columns
                        return launch_w4<W4Cfg<DT, 128, 128, true, false, 1 | 8192, 0, 0>>(q, -1, cols);
                    }
                }
                return launch_w4<W4Cfg<DT, 256, 128, true, OUT_F32, 1 | 8192>>(q);      // (fp32 output: general-form epilogue only)
            } else return BD_E_BAD_SHAPE;
        case 15:     // 14 with the SwiGLU epilogue (bd_binary_linear_swiglu: 8-interleaved gate|up pair, C has N/2 columns)
            if constexpr (FUSED && !OUT_F32) return launch_w4<W4Cfg<DT, 256, 128, true, false, 1 | 8192, 1>>(q);
            else return BD_E_BAD_SHAPE;
        case 1: return launch_tile<GemmCfg<DT, 128, 256, 1, 4, 4, FUSED, OUT_F32>>(q);
        case 2: return launch_tile<GemmCfg<DT, 64, 256, 1, 4, 4, FUSED, OUT_F32>>(q);
        case 3: return launch_tile<GemmCfg<DT, 32, 256, 1, 4, 4, FUSED, OUT_F32>>(q);
        case 100: return launch_generic<DT, FUSED, OUT_F32>(q);
        // decode: the VALU sign-flip kernel is as fast or faster on the fused Linear shapes (profiles/r01_decode_kernels.txt);
        // the MFMA + LUT kernel wins when the delta dominates (delta-only with >= 8 masks) and on narrow outputs (k/v projections)
        // ... and whenever a mask is shared by >= 2 rows (M > 1 or a broadcast mask): it expands each word once for all rows
        case 200: {
            // one mask per row, delta only, reference layout (binary_bmm at M = 1): 64-column super-tiles, the whole batch in one launch
            if constexpr (!FUSED) { if (g_forced_variant < 0 && rows_auto(q)) return launch_rows<DT>(q); }
            const int cb = GEMV_MAX_R / q.M, bc = q.B < cb ? q.B : cb, nmask = q.sPb == 0 ? 1 : bc;
            // streaming kernel: one launch, one block per CU, everything in flight from the first cycle (profiles/r02_decode_*.txt)
            if (stream_ok(q, bc * q.M, nmask)) { t_last_variant = 600; return launch_gemv<DT>(q, false, false, true); }
            // no-split-k kernel (one launch) when the chunk's activations fit in LDS and there are enough 16-column blocks to fill
            // the chip but not several rounds of them: 4096 < ... <= 8192 columns (+3..10 % over the two-launch kernels there)
            // (all of its N/16 blocks must be resident at once: a second round of 8-wave blocks pays the start-up latency again --
            // 8192x8192, 4 tenants: 50.8 vs 37.5 us)
            // up to 8 rows, fused: the wave-specialised VALU kernel is the fastest on every Llama / Mistral shape
            // (profiles/r01_decode_kernels.txt: 4096^2 T=6 14.4 us vs 15.8 one-launch kernel vs 16.7 single-role)
            if (q.W && bc * q.M <= 8 && bc * q.M < 2 * nmask && g_gemv_wave_spec) return launch_gemv<DT>(q, true);
            int ks16, ksl16;
            col16_split(bc * q.M, q.K, 0, ks16, ksl16);
            const long long lds16 = 4096 + (long long)bc * q.M * (ksl16 * 2 + 16) + 16 * 1024 + 512;
            const long long per_cu = lds16 > 0 ? (160 * 1024) / lds16 : 0;
            if (ks16 == 1 && q.N > 2048 && (q.N + 15) / 16 <= (long long)num_cus() * (per_cu < 2 ? per_cu : 2))
                return launch_gemv<DT>(q, false, true);
            return launch_gemv<DT>(q, !((!q.W && nmask >= 8) || q.N <= 2048 || bc * q.M >= 2 * nmask));
        }
        case 300: return launch_gemv<DT>(q, true);
        case 400: return launch_gemv<DT>(q, false);
        case 500: return launch_gemv<DT>(q, false, true);
        case 600: {
            const int cb = GEMV_MAX_R / q.M, bc = q.B < cb ? q.B : cb, nmask = q.sPb == 0 ? 1 : bc;
            if (!stream_ok(q, bc * q.M, nmask)) return BD_E_BAD_SHAPE;
            return launch_gemv<DT>(q, false, false, true);
        }
        case 800:    // delta_rows_kernel (801 / 802 / 804: masks per block forced)
            if constexpr (!FUSED) return launch_rows<DT>(q);
            else return BD_E_BAD_SHAPE;
        default: return BD_E_BAD_SHAPE;
    }
}

int dispatch(const Problem& q) {
    if (q.B < 0 || q.M < 0 || q.N < 0 || q.K < 0) return BD_E_BAD_SHAPE;
    if (q.K % 32) return BD_E_K_NOT_MULTIPLE;
    if (q.dtype != BD_F16 && q.dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (q.out_dtype != q.dtype && q.out_dtype != BD_F32) return BD_E_BAD_DTYPE;
    if (q.G < 1 || (q.N > 0 && q.N % q.G)) return BD_E_BAD_GROUPS;
    if (q.B == 0 || q.M == 0 || q.N == 0) return BD_OK;
    if (q.B > 65535) return BD_E_BAD_SHAPE;
    if (!q.A || !q.P || !q.C) return BD_E_NULL;
    if ((q.W || q.accumulate) && !q.alpha) return BD_E_NULL;
    const bool fused = q.W != nullptr, f32 = q.out_dtype == BD_F32;
    if (q.K == 0) return BD_E_BAD_SHAPE;
#define BD_D(DT) (fused ? (f32 ? dispatch3<DT, true, true>(q) : dispatch3<DT, true, false>(q)) \
                        : (f32 ? dispatch3<DT, false, true>(q) : dispatch3<DT, false, false>(q)))
    return q.dtype == BD_BF16 ? BD_D(DT_BF16) : BD_D(DT_F16);
#undef BD_D
}

}  // namespace

extern "C" int64_t bd_gemm_workspace_bytes(int B, int M, int N, int K) {
    if (B <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
    if (M > GEMV_MAX_M || (int64_t)B * M > 4 * GEMV_MAX_R) {
        // mid-size M: split-k partial slabs of the fused tile kernel -- only where the automatic rule really splits (or a test
        // forces the split-k variant); every other tile-kernel launch needs no scratch at all
        if (M > 16 && M <= 512 && K % 64 == 0 && N % 8 == 0) {
            int ks = splitk_factor(B, M, N, K);
            if (ks < 2 && g_forced_variant == 10) ks = 2;
            if (B >= 2 && M <= 64) {                      // pair tiles (variants 16 / 17) split by their own rule
                int kp = pair_splitk_dims(B, N, K);
                if (kp < 2 && (g_forced_variant == 17 || g_forced_variant == 19)) kp = 2;
                if (kp > ks) ks = kp;
            }
            if (ks < 2) return 0;
            const int64_t need = (int64_t)B * ks * M * N * 4;
            return need <= SPLITK_WS_CAP * 2 ? GEMV_TICKET_BYTES + need : 0;
        }
        return 0;
    }
    Problem q{};
    { const int cb = GEMV_MAX_R / M; B = B < cb ? B : cb; }     // the decode path works on chunks of <= 16 rows
    q.B = B; q.M = M; q.N = N; q.K = K;
    int KS, kslice;
    gemv_split(q, KS, kslice);
    int KSmax = K / 512 < 1 ? 1 : K / 512;               // upper bound over the auto rule and the forced-KS test hook
    if (KS > KSmax) KSmax = KS;
    return GEMV_TICKET_BYTES + (int64_t)KSmax * B * M * N * 4;
}

extern "C" int bd_delta_bmm(const void* A, const int32_t* P, void* C, int B, int M, int N, int K, int64_t sAb, int64_t sAm,
                            int64_t sPb, int64_t sCb, int64_t sCm, int dtype, int out_dtype, int round_mode,
                            const float* alpha, int64_t sAlb, int G, int accumulate, void* ws, int64_t ws_bytes,
                            void* stream) {
    Problem q{};
    q.A = A; q.P = P; q.C = C; q.W = nullptr; q.alpha = alpha;
    q.B = B; q.M = M; q.N = N; q.K = K;
    q.sAb = sAb; q.sAm = sAm; q.sPb = sPb; q.sCb = sCb; q.sCm = sCm; q.ldw = 0; q.sAlb = sAlb;
    q.G = alpha ? G : 1; q.dtype = dtype; q.out_dtype = out_dtype; q.round_mode = round_mode;
    q.accumulate = accumulate ? 1 : 0;
    q.ws = ws; q.ws_bytes = ws_bytes; q.st = (hipStream_t)stream;
    if (alpha && !accumulate) {
        // C = alpha * acc without a C_in: express as the fused epilogue of an all-zero base is not worth a kernel;
        // zero C then accumulate (two tiny ops on the output only).
        if (B > 0 && M > 0 && N > 0 && C) {
            const size_t esz = out_dtype == BD_F32 ? 4 : 2;
            if (sCm == N && (sCb == (int64_t)M * N || B == 1)) {
                if (hipMemsetAsync(C, 0, (size_t)B * M * N * esz, q.st) != hipSuccess) return BD_E_LAUNCH;
            } else {
                for (int b = 0; b < B; ++b)
                    if (hipMemset2DAsync((char*)C + (size_t)b * sCb * esz, (size_t)sCm * esz, 0, (size_t)N * esz, M, q.st) !=
                        hipSuccess)
                        return BD_E_LAUNCH;
            }
        }
        q.accumulate = 1;
    }
    return dispatch(q);
}

static int binary_linear_impl(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y, int B, int M,
                              int N, int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                              int64_t sYb, int64_t sYm, int dtype, int out_dtype, int accumulate, int mask_tiled, int t_pad,
                              void* ws, int64_t ws_bytes, void* stream, const void* norm_w = nullptr, int64_t sNw = 0,
                              float eps = 0.f, int epilogue = 0, const float* ssq_in = nullptr, float* ssq_out = nullptr,
                              void* xw_out = nullptr, const void* pn_w = nullptr, int64_t s_pnw = 0, float pn_eps = 0.f, void* pn_h = nullptr,
                              int64_t sHb = 0, int64_t sHm = 0) {
    if (B > 0 && M > 0 && N > 0 && !W) return BD_E_NULL;
    Problem q{};
    q.A = X; q.P = P; q.C = Y; q.W = W; q.alpha = alpha;
    q.B = B; q.M = M; q.N = N; q.K = K;
    q.sAb = sXb; q.sAm = sXm; q.sPb = sPb; q.sCb = sYb; q.sCm = sYm; q.ldw = ldw; q.sAlb = sAlb;
    q.G = G; q.dtype = dtype; q.out_dtype = out_dtype; q.round_mode = 0; q.accumulate = accumulate ? 1 : 0;
    q.mask_tiled = mask_tiled; q.t_pad = t_pad;
    q.w_tiled = 0;
    if (ldw == 0 && W) {      // tile-major base weight (the decode copy made by the serving side): packed sign layout only, whole blocks
        if (mask_tiled != 2 || N % 16 || K % 128 || M != 1) return BD_E_BAD_SHAPE;
        q.w_tiled = 1;
        q.ldw = K;            // (extent checks below; the kernel does not use it)
    }
    q.norm_w = norm_w; q.sNw = sNw; q.eps = eps; q.epilogue = epilogue;
    q.pn_w = pn_w; q.s_pnw = s_pnw; q.pn_eps = pn_eps; q.pn_h = pn_h; q.sHb = sHb; q.sHm = sHm;
    q.ssq_in = ssq_in; q.ssq_out = ssq_out;
    if (ssq_in || ssq_out || xw_out) {
        // RMSNorm hand-off (gemv_stream_kernel, StreamParams::ssq_in / ssq_out / xw_out): packed layout, one row per tenant, <= 8 tenants
        if (mask_tiled != 2 || M != 1 || B > 8 || (ssq_in && ssq_out)) return BD_E_BAD_SHAPE;
        if (ssq_in && (norm_w || !aligned16(ssq_in) || K % 16 || K > 16 * 256 * 2)) return BD_E_BAD_SHAPE;   // (X = the producer's pre-multiplied copy)
        if (ssq_out && (N % 16 || out_dtype == BD_F32 || epilogue || !aligned16(ssq_out))) return BD_E_BAD_SHAPE;
        if (xw_out && (!ssq_out || !norm_w || sNw < 0)) return BD_E_BAD_SHAPE;
        if (ssq_out && norm_w && !xw_out) return BD_E_BAD_SHAPE;
        if (ssq_out) {            // the producer's norm_w is the NEXT norm's weight (epilogue multiply), not a prologue on its own input
            q.nw_next = norm_w; q.sNwNext = sNw; q.xw_out = xw_out;
            q.norm_w = nullptr; norm_w = nullptr;
        }
    }
    if (norm_w || epilogue) {
        // fused prologue / epilogue of the packed streaming kernel: see gemv_stream_kernel (XL, EPI)
        if (mask_tiled != 2 || (epilogue != 0 && epilogue != 1)) return BD_E_BAD_SHAPE;
        if (norm_w) {
            if (!aligned16(norm_w) || sNw % 8 || sNw < 0) return BD_E_BAD_SHAPE;
            // one row per tenant (M == 1: the decode step), K = 2048 * 2^s, all rows in 16 x 16-byte chunks per thread
            // (the hand-off form needs no power of two: its copy is flat)
            if (M != 1 || K < 2048 || (!ssq_in && (K & (K - 1))) || (int64_t)B * K > 16 * 2048) return BD_E_BAD_SHAPE;
            if ((int64_t)STREAM_XS_OFF + (int64_t)B * M * (2 * (int64_t)K + 16) > STREAM_LDS_MAX) return BD_E_BAD_SHAPE;
            if ((int64_t)(B - 1) * sNw + K >= (1ll << 30)) return BD_E_BAD_SHAPE;
        }
        if (epilogue && (N % 16 || G != 2 || out_dtype != dtype || accumulate)) return BD_E_BAD_SHAPE;
    }
    q.ws = ws; q.ws_bytes = ws_bytes; q.st = (hipStream_t)stream;
    // tile-major masks exist for the streaming decode kernel only (serving-side repack; the reference layout works everywhere)
    if (q.mask_tiled) {
        const bool forced_other = g_forced_variant >= 0 && g_forced_variant != 200 && g_forced_variant != 700 &&
                                  !(g_forced_variant >= 600 && g_forced_variant <= 664);
        if (M < 1 || M > GEMV_MAX_M || forced_other || !gemv_ok(q)) return BD_E_BAD_SHAPE;
        const int cb = GEMV_MAX_R / M, bc = B < cb ? B : cb;
        if (!stream_ok(q, bc * M, sPb == 0 ? 1 : bc)) return BD_E_BAD_SHAPE;
        if (q.mask_tiled == 2) {      // interleaved tenants: the whole batch is one chunk, one dword slot per tenant
            const bool tp_ok = t_pad == 1 || t_pad == 2 || t_pad == 4 || t_pad == 6 || t_pad == 8 || t_pad == 12 || t_pad == 16;
            if (!tp_ok || B > t_pad || B > cb || (sPb == 0 && B > 1 && t_pad != 1)) return BD_E_BAD_SHAPE;
            q.sPb = t_pad == 1 ? 0 : 1;             // only "broadcast or not" matters to the kernel in this layout
            if (t_pad == 1 && B > 1) q.sPb = 0;
        } else if (q.mask_tiled != 1) return BD_E_BAD_SHAPE;
    }
    // Y += ... is an epilogue of the decode kernels only (a residual add costs a launch per Linear there; at prefill sizes it is
    // noise next to the GEMM and stays with the caller)
    // Y += ... (residual epilogue): the decode kernels and the one-pass fused tile kernels (M > 16 on their fast path)
    if (q.accumulate && !gemv_ok(q) && !(M > 16 && fast_ok(q))) return BD_E_BAD_SHAPE;
    return dispatch(q);
}

extern "C" int bd_binary_linear(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y, int B, int M,
                                int N, int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                                int64_t sYb, int64_t sYm, int dtype, int out_dtype, void* ws, int64_t ws_bytes,
                                void* stream) {
    return binary_linear_impl(X, W, P, alpha, Y, B, M, N, K, sXb, sXm, ldw, sPb, sAlb, G, sYb, sYm, dtype, out_dtype, 0, 0, 0, ws,
                              ws_bytes, stream);
}

extern "C" int bd_binary_linear_swiglu(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y, int B, int M, int N,
                                       int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int64_t sYb, int64_t sYm,
                                       int dtype, void* stream) {
    if (B > 0 && M > 0 && N > 0 && !W) return BD_E_NULL;
    if (M <= GEMV_MAX_M) return BD_E_BAD_SHAPE;                    // decode shapes: bd_binary_linear_decode_fused (packed sign layout)
    if (N % 16 || K % 64 || sYm % 8 || sYb % 8 || !aligned16(Y)) return BD_E_BAD_SHAPE;
    Problem q{};
    q.A = X; q.P = P; q.C = Y; q.W = W; q.alpha = alpha;
    q.B = B; q.M = M; q.N = N; q.K = K;
    q.sAb = sXb; q.sAm = sXm; q.sPb = sPb; q.sCb = sYb; q.sCm = sYm; q.ldw = ldw; q.sAlb = sAlb;
    q.G = 2; q.dtype = dtype; q.out_dtype = dtype; q.round_mode = 0; q.accumulate = 0;
    q.epilogue = 1;
    q.st = (hipStream_t)stream;
    if (!fast_ok(q)) return BD_E_BAD_SHAPE;
    return dispatch(q);
}

extern "C" int bd_binary_linear_decode(const void* X, const void* W, const int32_t* P, int mask_layout, int t_pad, const float* alpha,
                                       void* Y, int B, int M, int N, int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb,
                                       int64_t sAlb, int G, int64_t sYb, int64_t sYm, int dtype, int out_dtype, int accumulate,
                                       void* stream) {
    if (mask_layout != 1 && mask_layout != 2) return BD_E_BAD_SHAPE;
    return binary_linear_impl(X, W, P, alpha, Y, B, M, N, K, sXb, sXm, ldw, sPb, sAlb, G, sYb, sYm, dtype, out_dtype,
                              accumulate, mask_layout, t_pad, nullptr, 0, stream);
}

extern "C" int bd_binary_linear_decode_fused(const void* X, const void* W, const int32_t* P, int t_pad, const float* alpha, void* Y,
                                             int B, int M, int N, int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb,
                                             int64_t sAlb, int G, int64_t sYb, int64_t sYm, int dtype, int out_dtype,
                                             int accumulate, const void* norm_w, int64_t s_norm, float eps, int epilogue,
                                             void* stream) {
    if (!norm_w && epilogue != 1) return BD_E_NULL;
    if (B < 1 || M < 1 || N < 1 || K < 1) return BD_E_BAD_SHAPE;
    return binary_linear_impl(X, W, P, alpha, Y, B, M, N, K, sXb, sXm, ldw, sPb, sAlb, G, sYb, sYm, dtype, out_dtype, accumulate, 2,
                              t_pad, nullptr, 0, stream, norm_w, s_norm, eps, epilogue);
}

extern "C" int bd_binary_linear_decode_handoff(const void* X, const void* W, const int32_t* P, int t_pad, const float* alpha, void* Y,
                                               int B, int M, int N, int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb,
                                               int64_t sAlb, int G, int64_t sYb, int64_t sYm, int dtype, int out_dtype,
                                               int accumulate, const void* norm_w, int64_t s_norm, float eps, int epilogue,
                                               const float* ssq_in, float* ssq_out, void* xw_out, void* stream) {
    if (!ssq_in && !ssq_out) return BD_E_NULL;
    if (B < 1 || M < 1 || N < 1 || K < 1) return BD_E_BAD_SHAPE;
    return binary_linear_impl(X, W, P, alpha, Y, B, M, N, K, sXb, sXm, ldw, sPb, sAlb, G, sYb, sYm, dtype, out_dtype, accumulate, 2,
                              t_pad, nullptr, 0, stream, norm_w, s_norm, eps, epilogue, ssq_in, ssq_out, xw_out);
}

extern "C" int bd_binary_linear_residual(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y, int B,
                                         int M, int N, int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb,
                                         int64_t sAlb, int G, int64_t sYb, int64_t sYm, int dtype, int out_dtype, void* ws,
                                         int64_t ws_bytes, void* stream) {
    return binary_linear_impl(X, W, P, alpha, Y, B, M, N, K, sXb, sXm, ldw, sPb, sAlb, G, sYb, sYm, dtype, out_dtype, 1, 0, 0, ws,
                              ws_bytes, stream);
}

// residual Linear + the RMSNorm that follows it (prefill sizes: the o / down projections of a decoder layer and the norm in front of the next
// Linear).  The norm rides on the split-k reduce launch when the dispatcher splits (the multi-tenant request of short prompts); otherwise it is
// the ordinary per-tenant norm launch behind the Linear -- the same arithmetic either way.
extern "C" int bd_binary_linear_residual_norm(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y, int B, int M, int N,
                                              int K, int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G, int64_t sYb,
                                              int64_t sYm, int dtype, const void* norm_w, int64_t s_nw, float eps, void* H, int64_t sHb,
                                              int64_t sHm, void* ws, int64_t ws_bytes, void* stream) {
    if (B < 0 || M < 0 || N < 0) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (B == 0 || M == 0 || N == 0) return BD_OK;
    if (!norm_w || !H || !Y) return BD_E_NULL;
    // the norm kernels' envelope: whole 16-byte chunks, rows of at most 8192 columns (both norm forms), dense [B, M] row grid for the fallback launch
    if (M <= GEMV_MAX_M || N % 8 || N > 8192 || sYm % 8 || sYb % 8 || sHm % 8 || sHb % 8 || s_nw % 8 || s_nw < 0 || !aligned16(Y) || !aligned16(H) ||
        !aligned16(norm_w) || sYb != (int64_t)M * sYm || sHb != (int64_t)M * sHm)
        return BD_E_BAD_SHAPE;
    t_post_norm_done = 0;
    const int rc = binary_linear_impl(X, W, P, alpha, Y, B, M, N, K, sXb, sXm, ldw, sPb, sAlb, G, sYb, sYm, dtype, dtype, 1, 0, 0, ws, ws_bytes,
                                      stream, nullptr, 0, 0.f, 0, nullptr, nullptr, nullptr, norm_w, s_nw, eps, H, sHb, sHm);
    if (rc != BD_OK || t_post_norm_done) return rc;
    return bd_srv_rmsnorm(Y, norm_w, H, B * M, N, sYm, sHm, s_nw, M, eps, dtype, stream);
}

// ------------------------------------------------------------------ per-tenant dense Linear (serving: lm_head per tenant)
extern "C" int bd_tenant_linear(const void* X, const void* W, void* Y, int T, int M, int N, int K, int64_t sXt, int64_t sXm,
                                int64_t sWt, int64_t ldw, int64_t sYt, int64_t sYm, int dtype, int out_dtype, void* stream) {
    if (T < 0 || M < 0 || N < 0 || K < 0) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (out_dtype != dtype && out_dtype != BD_F32) return BD_E_BAD_DTYPE;
    if (T == 0 || M == 0 || N == 0) return BD_OK;
    if (!X || !W || !Y) return BD_E_NULL;
    if (K % 32) return BD_E_K_NOT_MULTIPLE;
    // decode shapes only (the weight-streaming kernel): larger M is a plain batched GEMM and belongs to the BLAS library
    if (M > GEMV_MAX_M || T > 65535 || K == 0) return BD_E_BAD_SHAPE;
    if (!aligned16(X) || !aligned16(W) || sXm % 8 || sXt % 8 || ldw % 8 || sWt % 8) return BD_E_BAD_SHAPE;
    const int64_t lim = (1ll << 31) - 64;
    const int64_t xb = ((int64_t)(M - 1) * sXm + K) * 2, wb = ((int64_t)(N - 1) * ldw + K) * 2;
    if (xb >= lim || wb >= lim || sXm < 0) return BD_E_BAD_SHAPE;
    StreamParams sp{};
    GemvParams& gp = sp.g;
    gp.X = (const unsigned short*)X; gp.W = (const unsigned short*)W; gp.P = nullptr; gp.alpha = nullptr; gp.C = Y;
    gp.ws = nullptr; gp.tickets = nullptr;
    gp.B = 1; gp.M = M; gp.N = N; gp.K = K; gp.R = M;
    gp.sXb = 0; gp.sPb = 0; gp.sCb = 0;
    gp.sXm = (int)sXm; gp.sCm = (int)sYm; gp.ldw = (int)ldw; gp.sAlb = 0; gp.gsz = N;
    gp.KS = 1; gp.kslice = K; gp.round_mode = 0; gp.accumulate = 0; gp.out_f32 = (out_dtype == BD_F32);
    sp.sXt = sXt; sp.sWt = sWt; sp.sCt = sYt;
    sp.x_bytes = (uint32_t)xb; sp.w_bytes = (uint32_t)wb; sp.p_bytes = 0; sp.pts = 16; sp.prs = (uint32_t)N;
    int bpt = num_cus() / T;                       // blocks per tenant: all tenants stream concurrently, ~one block per CU in total
    if (bpt < 1) bpt = 1;
    int cpb = (N + bpt - 1) / bpt;
    cpb = (cpb + 3) & ~3;
    if (cpb < 4) cpb = 4;
    sp.cpb = cpb;
    dim3 grid((unsigned)((N + cpb - 1) / cpb), (unsigned)T);
    hipStream_t st = (hipStream_t)stream;
    // A/B (round 6; bd_set_stream_tuning bits 11 / 12): natural-order weight loads (a load instruction = 16 rows x 64 contiguous bytes) with / without the
    // non-temporal policy, against the shipped word-row order.  profiles/r06_decode_step.txt has the outcome.
    int rc;
    if (g_stream_tune & 2048)
        rc = dtype == BD_BF16 ? launch_stream_inst<DT_BF16, 0, true, 8, 4, 1, 2>(sp, grid, st) : launch_stream_inst<DT_F16, 0, true, 8, 4, 1, 2>(sp, grid, st);
    else if (g_stream_tune & 4096)
        rc = dtype == BD_BF16 ? launch_stream_inst<DT_BF16, 0, true, 8, 4, 1, 0>(sp, grid, st) : launch_stream_inst<DT_F16, 0, true, 8, 4, 1, 0>(sp, grid, st);
    else
        rc = dtype == BD_BF16 ? launch_stream_tuned<DT_BF16, 0, true, 8>(sp, grid, st)
                              : launch_stream_tuned<DT_F16, 0, true, 8>(sp, grid, st);      // (NS 6 / 4 measured equal or slower here)
    if (rc != BD_OK) return rc;
    return launch_status();
}

// ------------------------------------------------------------------ serving-loop glue (decode step; callers of the path)
extern "C" int bd_srv_rmsnorm(const void* X, const void* Wt, void* Y, int rows, int H, int64_t sx, int64_t sy, int64_t sw,
                              int rows_per_tenant, float eps, int dtype, void* stream) {
    if (rows < 0 || H < 0 || rows_per_tenant < 1) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (rows == 0 || H == 0) return BD_OK;
    if (!X || !Wt || !Y) return BD_E_NULL;
    if (H % 8 || sx % 8 || sy % 8 || sw % 8 || !aligned16(X) || !aligned16(Wt) || !aligned16(Y)) return BD_E_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // many rows of 2048 / 4096 / 6144 / 8192 elements: one wave per row, the row in registers (rmsnorm_rows_kernel; bit-identical results)
    if (rows >= g_norm_rows_min && H % 2048 == 0 && H <= 8192) {
        const dim3 grid((unsigned)((rows + 3) / 4));
#define BD_NR(DT, NCH) hipLaunchKernelGGL((rmsnorm_rows_kernel<DT, NCH>), grid, dim3(256), 0, st, (const unsigned short*)X, (const unsigned short*)Wt, \
                                          (unsigned short*)Y, rows, (long long)sx, (long long)sy, (long long)sw, rows_per_tenant, eps)
        const int nch = H / 2048;
        if (dtype == BD_BF16) { if (nch == 1) BD_NR(DT_BF16, 1); else if (nch == 2) BD_NR(DT_BF16, 2); else if (nch == 3) BD_NR(DT_BF16, 3); else BD_NR(DT_BF16, 4); }
        else { if (nch == 1) BD_NR(DT_F16, 1); else if (nch == 2) BD_NR(DT_F16, 2); else if (nch == 3) BD_NR(DT_F16, 3); else BD_NR(DT_F16, 4); }
#undef BD_NR
        return launch_status();
    }
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((rmsnorm_tenant_kernel<DT_BF16>), dim3(rows), dim3(256), 0, st, (const unsigned short*)X, (const unsigned short*)Wt,
                           (unsigned short*)Y, H, (long long)sx, (long long)sy, (long long)sw, rows_per_tenant, eps);
    else
        hipLaunchKernelGGL((rmsnorm_tenant_kernel<DT_F16>), dim3(rows), dim3(256), 0, st, (const unsigned short*)X, (const unsigned short*)Wt,
                           (unsigned short*)Y, H, (long long)sx, (long long)sy, (long long)sw, rows_per_tenant, eps);
    return launch_status();
}

extern "C" int bd_srv_add_rmsnorm(const void* resid, const float* y32, const void* Wt, void* x_out, void* h_out, int rows, int H, int64_t s_r,
                                  int64_t s_y, int64_t s_x, int64_t s_h, int64_t sw, int rows_per_tenant, float eps, int dtype, void* stream) {
    if (rows < 0 || H < 0 || rows_per_tenant < 1) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (rows == 0 || H == 0) return BD_OK;
    if (!resid || !y32 || !Wt || !x_out || !h_out) return BD_E_NULL;
    if (H % 8 || H > 8192 || s_r % 8 || s_y % 4 || s_x % 8 || s_h % 8 || sw % 8 || !aligned16(resid) || !aligned16(y32) || !aligned16(Wt) ||
        !aligned16(x_out) || !aligned16(h_out))
        return BD_E_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((add_rmsnorm_kernel<DT_BF16>), dim3(rows), dim3(256), 0, st, (const unsigned short*)resid, y32, (const unsigned short*)Wt,
                           (unsigned short*)x_out, (unsigned short*)h_out, H, (long long)s_r, (long long)s_y, (long long)s_x, (long long)s_h,
                           (long long)sw, rows_per_tenant, eps);
    else
        hipLaunchKernelGGL((add_rmsnorm_kernel<DT_F16>), dim3(rows), dim3(256), 0, st, (const unsigned short*)resid, y32, (const unsigned short*)Wt,
                           (unsigned short*)x_out, (unsigned short*)h_out, H, (long long)s_r, (long long)s_y, (long long)s_x, (long long)s_h,
                           (long long)sw, rows_per_tenant, eps);
    return launch_status();
}

extern "C" int bd_srv_swiglu(const void* G, const void* U, void* Y, int rows, int I, int64_t sg, int64_t su, int64_t sy,
                             int interleaved8, int dtype, void* stream) {
    if (rows < 0 || I < 0 || rows > 65535) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (rows == 0 || I == 0) return BD_OK;
    if (!G || !U || !Y) return BD_E_NULL;
    if (I % 8 || sg % 8 || su % 8 || sy % 8 || !aligned16(G) || !aligned16(U) || !aligned16(Y)) return BD_E_BAD_SHAPE;
    if (interleaved8 && G == Y) return BD_E_BAD_SHAPE;       // (in place is fine for the split form only)
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((I / 8 + 255) / 256), (unsigned)rows);
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((swiglu_kernel<DT_BF16>), grid, dim3(256), 0, st, (const unsigned short*)G, (const unsigned short*)U,
                           (unsigned short*)Y, I, (long long)sg, (long long)su, (long long)sy, interleaved8 ? 1 : 0);
    else
        hipLaunchKernelGGL((swiglu_kernel<DT_F16>), grid, dim3(256), 0, st, (const unsigned short*)G, (const unsigned short*)U,
                           (unsigned short*)Y, I, (long long)sg, (long long)su, (long long)sy, interleaved8 ? 1 : 0);
    return launch_status();
}

extern "C" int bd_srv_rope(void* X, const void* cos_t, const void* sin_t, int rows, int heads, int head_dim, int64_t sx, int seq,
                           int pos0, int dtype, void* stream) {
    if (rows < 0 || heads < 1 || seq < 1 || pos0 < 0) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (rows == 0) return BD_OK;
    if (!X || !cos_t || !sin_t) return BD_E_NULL;
    if (head_dim != 128 || sx % 8 || !aligned16(X) || !aligned16(cos_t) || !aligned16(sin_t)) return BD_E_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((rope_kernel<DT_BF16>), dim3(rows), dim3(256), 0, st, (unsigned short*)X, (const unsigned short*)cos_t,
                           (const unsigned short*)sin_t, heads, (long long)sx, seq, pos0);
    else
        hipLaunchKernelGGL((rope_kernel<DT_F16>), dim3(rows), dim3(256), 0, st, (unsigned short*)X, (const unsigned short*)cos_t,
                           (const unsigned short*)sin_t, heads, (long long)sx, seq, pos0);
    return launch_status();
}

extern "C" int bd_srv_rope_kv_append(void* QKV, const void* cos_t, const void* sin_t, void* kcache, void* vcache, int T, int S, int H, int KVH,
                                     int head_dim, int64_t sx, int Lc, int pos0, int dtype, void* stream) {
    if (T < 0 || S < 0 || H < 1 || KVH < 1 || pos0 < 0 || Lc < 1) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (T == 0 || S == 0) return BD_OK;
    if (!QKV || !cos_t || !sin_t || !kcache || !vcache) return BD_E_NULL;
    if (head_dim != 128 || sx % 8 || sx < (int64_t)(H + 2 * KVH) * 128 || (int64_t)pos0 + S > Lc || (int64_t)T * S >= (1ll << 31) ||
        !aligned16(QKV) || !aligned16(cos_t) || !aligned16(sin_t) || !aligned16(kcache) || !aligned16(vcache))
        return BD_E_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((rope_kv_append_kernel<DT_BF16>), dim3((unsigned)(T * S)), dim3(256), 0, st, (unsigned short*)QKV,
                           (const unsigned short*)cos_t, (const unsigned short*)sin_t, (unsigned short*)kcache, (unsigned short*)vcache, H, KVH,
                           (long long)sx, S, pos0, Lc);
    else
        hipLaunchKernelGGL((rope_kv_append_kernel<DT_F16>), dim3((unsigned)(T * S)), dim3(256), 0, st, (unsigned short*)QKV,
                           (const unsigned short*)cos_t, (const unsigned short*)sin_t, (unsigned short*)kcache, (unsigned short*)vcache, H, KVH,
                           (long long)sx, S, pos0, Lc);
    return launch_status();
}

extern "C" int bd_srv_step_begin(const void* embed, int64_t sEt, int64_t sEv, const int64_t* tok, void* X, int64_t sx, void* valid, int Lc,
                                 const int64_t* pos, int T, int V, int H, void* stream) {
    if (T < 0 || V < 1 || H < 1 || Lc < 1) return BD_E_BAD_SHAPE;
    if (T == 0) return BD_OK;
    if (!embed || !tok || !X || !valid || !pos) return BD_E_NULL;
    if (H % 8 || sEv % 8 || sEt % 8 || sx % 8 || sEv < H || sx < H || sEt < 0 || !aligned16(embed) || !aligned16(X)) return BD_E_BAD_SHAPE;
    hipLaunchKernelGGL(step_begin_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)embed, (long long)sEt,
                       (long long)sEv, (const long long*)tok, (unsigned short*)X, (long long)sx, (unsigned char*)valid, Lc, (const long long*)pos, V, H);
    return launch_status();
}

extern "C" int bd_srv_step_end(const void* logits, int64_t sl, int V, int64_t* tok, int64_t* out, int64_t s_out, int out_cap,
                               const int64_t* stop_ids, int ns, void* stopped, int64_t* pos, int64_t* step, void* ticket, int T, int dtype,
                               void* stream) {
    if (T < 0 || V < 1 || ns < 0 || out_cap < 0) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (T == 0) return BD_OK;
    if (!logits || !tok || !out || !stopped || !pos || !step || !ticket || (ns > 0 && !stop_ids)) return BD_E_NULL;
    if (V % 8 || sl % 8 || sl < V || s_out < out_cap || !aligned16(logits) || ((uintptr_t)ticket & 3)) return BD_E_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((step_end_kernel<DT_BF16>), dim3((unsigned)T), dim3(256), 0, st, (const unsigned short*)logits, (long long)sl, V,
                           (long long*)tok, (long long*)out, (long long)s_out, out_cap, (const long long*)stop_ids, ns, (unsigned char*)stopped,
                           (long long*)pos, (long long*)step, (unsigned int*)ticket, T);
    else
        hipLaunchKernelGGL((step_end_kernel<DT_F16>), dim3((unsigned)T), dim3(256), 0, st, (const unsigned short*)logits, (long long)sl, V,
                           (long long*)tok, (long long*)out, (long long)s_out, out_cap, (const long long*)stop_ids, ns, (unsigned char*)stopped,
                           (long long*)pos, (long long*)step, (unsigned int*)ticket, T);
    return launch_status();
}

extern "C" int bd_srv_cache_warm(const void* p0, int64_t bytes0, const void* p1, int64_t bytes1, int blocks, void* stream) {
    if (bytes0 < 0 || bytes1 < 0 || blocks < 0) return BD_E_BAD_SHAPE;
    if ((bytes0 && !p0) || (bytes1 && !p1)) return BD_E_NULL;
    if ((bytes0 && !aligned16(p0)) || (bytes1 && !aligned16(p1))) return BD_E_BAD_SHAPE;
    if (bytes0 / 16 + bytes1 / 16 == 0) return BD_OK;
    if (blocks == 0) blocks = num_cus();
    hipLaunchKernelGGL(cache_warm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)p0, (long long)(bytes0 / 16),
                       (const u32x4_t*)p1, (long long)(bytes1 / 16));
    return launch_status();
}

#ifndef BD_ATTN_SPLITS
#define BD_ATTN_SPLITS 4
#endif
constexpr int ATTN_SPLITS = BD_ATTN_SPLITS;   // (a build-time knob for A/B runs; 4 measured best at 6 tenants x 8 kv heads, profiles/r03_decode_attn_splits.txt, r05_decode_step.txt)
constexpr int ATTN_SPLITS_MAX = 16;
static_assert(ATTN_SPLITS <= 4, "decode_attn_kernel<.., MAXS = 4> merges at most 4 partials");
constexpr int64_t ATTN_TICKET_BYTES = 16384;       // one arrival counter per (tenant, kv head): T * KVH <= 4096
static thread_local int g_attn_splits_max = env_int("BD_ATTN_SPLITS_MAX", ATTN_SPLITS_MAX);       // A/B hook: 4 = the fixed split count of rounds 3 - 5
// key-range splits of a launch: ATTN_SPLITS, doubled while the launch would still cover at most a QUARTER of the CUs (one or two sequences of a
// grouped-query model: a single sequence on 8 kv heads is 32 blocks at 4 splits -- and the launch is a dependent chain of memory round trips, so
// fewer rows per block = fewer load rounds), never more than one split per 32 cache rows.  Same-box A/B, Mistral-7B, 4 -> this rule (tools/gpu_r5y.sh):
// 1 tenant 3.319 -> 3.275 ms/step (-1.3 %, 16 splits), 2 tenants 3.548 -> 3.544 (8 splits); doubling up to HALF of the CUs was +0.6 % at 4 tenants
// (8 splits, 256 blocks), so 32 (tenant, kv head) pairs and more keep 4.  A function of (T, KVH, Lc) only.
inline int attn_splits(int T, int KVH, int Lc) {
    if (Lc < 256) return 1;                                       // short caches run unsplit
    int ns = ATTN_SPLITS;
    const int cap = g_attn_splits_max < ATTN_SPLITS ? ATTN_SPLITS : (g_attn_splits_max > ATTN_SPLITS_MAX ? ATTN_SPLITS_MAX : g_attn_splits_max);
    while (ns * 2 <= cap && (int64_t)T * KVH * ns * 4 <= num_cus() && ns * 2 * 32 <= Lc) ns *= 2;
    return ns;
}
extern "C" int64_t bd_srv_decode_attention_workspace_bytes(int T, int H, int KVH, int head_dim, int Lc) {
    if (T <= 0 || H <= 0 || KVH <= 0 || Lc < 256) return 0;       // short caches run unsplit
    return ATTN_TICKET_BYTES + (int64_t)T * H * ATTN_SPLITS_MAX * (head_dim + 2) * 4;      // (sized for the largest split count: independent of the A/B hook)
}

extern "C" int bd_srv_decode_attention(const void* QKV, const void* cos_t, const void* sin_t, void* kcache, void* vcache,
                                       void* valid, const int64_t* pos, void* out, int T, int H, int KVH, int head_dim, int Lc,
                                       int64_t s_qkv, int64_t s_out, int dtype, void* ws, int64_t ws_bytes, void* stream) {
    if (T < 0 || H < 1 || KVH < 1 || H % KVH || Lc < 1) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (T == 0) return BD_OK;
    if (!QKV || !cos_t || !sin_t || !kcache || !vcache || !valid || !pos || !out) return BD_E_NULL;
    const int G = H / KVH;
    if (head_dim != 128 || (G != 1 && G != 4 && G != 8) || s_qkv % 8 || !aligned16(QKV) || !aligned16(kcache) || !aligned16(vcache))
        return BD_E_BAD_SHAPE;                         // other head geometries: the caller keeps its torch attention
    AttnParams p;
    p.qkv = (const unsigned short*)QKV; p.cos = (const unsigned short*)cos_t; p.sin = (const unsigned short*)sin_t;
    p.kc = (unsigned short*)kcache; p.vc = (unsigned short*)vcache; p.valid = (unsigned char*)valid; p.pos = (const long long*)pos;
    p.out = (unsigned short*)out; p.T = T; p.H = H; p.KVH = KVH; p.Lc = Lc; p.s_qkv = s_qkv; p.s_out = s_out;
    p.scale = 1.0f / sqrtf((float)head_dim);
    // split the key range over 4 blocks per (tenant, kv head) when the cache is long enough and a workspace is given: T * KVH blocks
    // alone leave most CUs (and most of HBM) idle -- one CU streams only ~12-25 GB/s
    // A workspace smaller than bd_srv_decode_attention_workspace_bytes() -- e.g. sized by a caller with the round-4 formula, which covered 4
    // splits -- gets the LARGEST split count it holds (16 -> 8 -> 4 -> 2), not a silent fall to the unsplit launch: partials are
    // indexed [tenant][head][split of nsplit], so any count whose T * H * nsplit records fit behind the ticket area is valid.
    const int64_t need = bd_srv_decode_attention_workspace_bytes(T, H, KVH, head_dim, Lc);
    p.nsplit = 1;
    if (need > 0 && ws && ws_bytes > ATTN_TICKET_BYTES && (int64_t)T * KVH * 4 <= ATTN_TICKET_BYTES) {
        const int64_t fit = (ws_bytes - ATTN_TICKET_BYTES) / ((int64_t)T * H * (head_dim + 2) * 4);
        int ns = attn_splits(T, KVH, Lc);
        while (ns > 1 && ns > fit) ns >>= 1;
        p.nsplit = ns;
    }
    p.tickets = (unsigned*)ws;
    p.ws = (float*)((char*)ws + ATTN_TICKET_BYTES);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(T * KVH), (unsigned)p.nsplit);
#define BD_ATT(DT, GG)                                                                                      \
    do {                                                                                                    \
        const bool d2_ = g_attn_depth ? g_attn_depth == 2 : Lc <= 2048;                                                    \
        if (p.nsplit <= 4) {                                                                                               \
            if (d2_) hipLaunchKernelGGL((decode_attn_kernel<DT, GG, 2, 4>), grid, dim3(512), 0, st, p);                    \
            else hipLaunchKernelGGL((decode_attn_kernel<DT, GG, 4, 4>), grid, dim3(512), 0, st, p);                        \
        } else {                                                                                                           \
            if (d2_) hipLaunchKernelGGL((decode_attn_kernel<DT, GG, 2, 16>), grid, dim3(512), 0, st, p);                   \
            else hipLaunchKernelGGL((decode_attn_kernel<DT, GG, 4, 16>), grid, dim3(512), 0, st, p);                       \
        }                                                                                                                  \
    } while (0)
    // G = query heads per kv head: 1 (Llama-2-7B), 4 (Mistral-7B), 8 (Llama-2-70B -- also per rank under tensor parallelism).
    // Depth of the K / V row ring: 2 iterations for caches of up to 2048 positions (a split is then <= 16 iterations; 4.698 -> 4.661 ms on the
    // 6-tenant step at 512 keys, one iteration: 4.649 but exposed to the full load latency on long splits), 4 beyond
    if (dtype == BD_BF16) { if (G == 1) BD_ATT(DT_BF16, 1); else if (G == 4) BD_ATT(DT_BF16, 4); else BD_ATT(DT_BF16, 8); }
    else { if (G == 1) BD_ATT(DT_F16, 1); else if (G == 4) BD_ATT(DT_F16, 4); else BD_ATT(DT_F16, 8); }
#undef BD_ATT
    return launch_status();
}

extern "C" int bd_srv_prefill_attention(const void* Q, const void* K, const void* V, void* O, int B, int S, int H, int KVH, int head_dim,
                                        int64_t sqb, int64_t sqs, int64_t skb, int64_t sks, int64_t svb, int64_t svs,
                                        int64_t sob, int64_t sos, const int32_t* kv_start, float scale, int causal, int dtype,
                                        void* stream) {
    if (B < 0 || S < 0 || H < 1 || KVH < 1 || H % KVH) return BD_E_BAD_SHAPE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (B == 0 || S == 0) return BD_OK;
    if (!Q || !K || !V || !O) return BD_E_NULL;
    // other geometries: the caller keeps its torch attention
    if (head_dim != 128 || S % 64 || (sqs | sks | svs | sos | sqb | skb | svb | sob) % 8 || !aligned16(Q) || !aligned16(K) || !aligned16(V) ||
        !aligned16(O) || (int64_t)B * H * ((S + 127) / 128) > 0x7fffffffLL)
        return BD_E_BAD_SHAPE;
    PrefillAttnParams p;
    p.q = (const unsigned short*)Q; p.k = (const unsigned short*)K; p.v = (const unsigned short*)V; p.o = (unsigned short*)O;
    p.sqb = sqb; p.sqs = sqs; p.skb = skb; p.sks = sks; p.svb = svb; p.svs = svs; p.sob = sob; p.sos = sos;
    p.kv_start = kv_start; p.B = B; p.S = S; p.H = H; p.KVH = KVH; p.nqb = (S + 127) / 128;
    p.c = scale * 1.4426950408889634f; p.causal = causal ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(p.nqb * H * B));
    static std::atomic<uint64_t> lds_done[2];               // the kernel needs more than the default dynamic LDS limit
    const int di = dtype == BD_BF16 ? 1 : 0;
    const void* fn = di ? (const void*)prefill_attn_kernel<DT_BF16> : (const void*)prefill_attn_kernel<DT_F16>;
    if (!ensure_dyn_lds(fn, PREFILL_ATTN_LDS, lds_done[di])) return BD_E_LAUNCH;
    if (di) hipLaunchKernelGGL((prefill_attn_kernel<DT_BF16>), grid, dim3(256), PREFILL_ATTN_LDS, st, p);
    else hipLaunchKernelGGL((prefill_attn_kernel<DT_F16>), grid, dim3(256), PREFILL_ATTN_LDS, st, p);
    return launch_status();
}

// ------------------------------------------------------------------ binarize / merge
extern "C" int64_t bd_binarize_workspace_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    return ((N + 63) / 64) * ((K + 255) / 256) * 4;
}

extern "C" int bd_binarize(const void* base, const void* fine, int64_t N, int64_t K, int64_t ld, int dtype, int32_t* mask,
                           float* coeff, void* ws, int64_t ws_bytes, void* stream) {
    if (N < 0 || K < 0 || N > (1LL << 30) || K > (1LL << 30)) return BD_E_BAD_SHAPE;
    if (K % 32) return BD_E_K_NOT_MULTIPLE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (N == 0 || K == 0) return BD_OK;
    if (!base || !fine || !mask || !coeff) return BD_E_NULL;
    const int64_t need = bd_binarize_workspace_bytes(N, K);
    if (!ws || ws_bytes < need) return BD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((K + 255) / 256));
    if (grid.y > 65535) return BD_E_BAD_SHAPE;
    const int nparts = (int)(grid.x * grid.y);
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((binarize_kernel<DT_BF16>), grid, dim3(256), 0, st, (const unsigned short*)base,
                           (const unsigned short*)fine, (uint32_t*)mask, (float*)ws, (int)N, (int)K, (long long)ld);
    else
        hipLaunchKernelGGL((binarize_kernel<DT_F16>), grid, dim3(256), 0, st, (const unsigned short*)base,
                           (const unsigned short*)fine, (uint32_t*)mask, (float*)ws, (int)N, (int)K, (long long)ld);
    hipLaunchKernelGGL(binarize_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nparts,
                       1.0 / ((double)N * (double)K), coeff);
    return launch_status();
}

extern "C" int bd_merge_delta(void* W, int64_t ldw, const int32_t* P, const float* coeff, int64_t N, int64_t K, int dtype,
                              void* stream) {
    if (N < 0 || K < 0 || N > (1LL << 30) || K > (1LL << 30)) return BD_E_BAD_SHAPE;
    if (K % 32) return BD_E_K_NOT_MULTIPLE;
    if (dtype != BD_F16 && dtype != BD_BF16) return BD_E_BAD_DTYPE;
    if (N == 0 || K == 0) return BD_OK;
    if (!W || !P || !coeff) return BD_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((K + 255) / 256));
    if (grid.y > 65535) return BD_E_BAD_SHAPE;
    if (dtype == BD_BF16)
        hipLaunchKernelGGL((merge_kernel<DT_BF16>), grid, dim3(256), 0, st, (unsigned short*)W, (const uint32_t*)P, coeff,
                           (int)N, (int)K, (long long)ldw);
    else
        hipLaunchKernelGGL((merge_kernel<DT_F16>), grid, dim3(256), 0, st, (unsigned short*)W, (const uint32_t*)P, coeff,
                           (int)N, (int)K, (long long)ldw);
    return launch_status();
}
