"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Every check goes Python surface -> C ABI -> HIP kernel and is
compared with the CPU oracle (oracle/bd_oracle.c, pinned to the reference by tests/test_oracle_golden.py) and/or with
the golden vectors the reference itself produced (tests/golden/).

Tolerances (BASELINE.json north star: 1e-3 relative on 16-bit, bit-exact on the sign pack/unpack):
  * pack / unpack / binarize mask / merge:  bit-exact.
  * fp32-output mode of every GEMM family:  rel-Frobenius and mean-rel <= 1e-5 (fp32 accumulation-order noise only).
  * fp16 outputs: <= 1e-3 rel-Frobenius vs the oracle, and within 1 fp16 ulp per element.
  * bf16 outputs: every element within 1 bf16 ulp of the rounded oracle (a flat 1e-3 is below bf16's own rounding
    floor), >= 99 % bit-equal.
"""
import os

import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from canary import CanaryOut, canary_workspace  # noqa: E402

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def bd():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import bitdelta_amd
    from bitdelta_amd import _lib
    _lib.lib()          # fails loudly if the HIP library is missing: there is no fallback
    return bitdelta_amd


def dev(t):
    return t.cuda()


def ulp_diff(a, b):
    """element-wise distance in units of the 16-bit format's ulp (via the monotonic integer mapping)."""
    def key(t):
        i = t.view(torch.int16).int()
        return torch.where(i < 0, -(i & 0x7fff), i)
    return (key(a) - key(b)).abs()


def within_one_ulp(got, ref, K):
    """<= 1 ulp of the 16-bit format, or -- for results that cancel to ~0, where an fp32 accumulation-order difference is many ulps
    of a tiny number -- within a floor that SCALES with the problem: 2^-22 * sqrt(K) * max|ref| over the element's ROW (fp32 rounding of partial sums as
    large as the largest output, random-walked over K terms).  At K = 4096 and fused outputs of magnitude ~5 that is 8e-5."""
    d = ulp_diff(got, ref)
    floor = 2.0 ** -22 * (K ** 0.5) * ref.float().abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)      # per output ROW (one x row's dot products)
    ok = (d <= 1) | ((got.float() - ref.float()).abs() <= floor)
    return bool(ok.all()), (d == 0).float().mean().item()


def relerr(a, ref):
    a, ref = a.double(), ref.double()
    return ((a - ref).norm() / ref.norm()).item(), ((a - ref).abs().mean() / ref.abs().mean()).item()


def rand_problem(B, M, K, N, dtype, tenants, seed=0, wscale=0.02):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(B, M, K, generator=g).to(dtype)
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (tenants, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
    w = (torch.randn(N, K, generator=g) * wscale).to(dtype)
    alpha = (torch.rand(tenants, 1, generator=g) * 2e-4 + 3e-4).float()
    return a, p, w, alpha


# ------------------------------------------------------------------ pack / unpack
def test_pack_unpack_golden_bit_exact(bd, golden):
    for case in golden["g1_pack32"]:
        p = bd.pack(dev(case["bits"]))
        assert p.dtype == torch.int32 and torch.equal(p.cpu(), case["packed"])
        assert torch.equal(bd.unpack(dev(case["packed"])).cpu(), case["bits"])
    for nb, case in golden["g1_pack_nbits"].items():
        p = bd.pack(dev(case["bits"]), n_bits=nb)
        assert p.dtype == case["packed"].dtype and torch.equal(p.cpu(), case["packed"])
        assert torch.equal(bd.unpack(dev(case["packed"]), n_bits=nb).cpu(), case["bits"])
    t = golden["g1_pack_transposed"]
    assert torch.equal(bd.pack(dev(t["bits_NK"]).T).cpu(), t["packed"])


def test_pack_errors(bd):
    with pytest.raises(AssertionError):
        bd.pack(torch.zeros(33, 4, dtype=torch.bool, device="cuda"))
    with pytest.raises(UnboundLocalError):
        bd.pack(torch.zeros(24, 4, dtype=torch.bool, device="cuda"), n_bits=12)
    e = bd.pack(torch.zeros(0, 5, dtype=torch.bool, device="cuda"))
    assert e.shape == (0, 5)
    e = bd.unpack(torch.zeros(2, 0, dtype=torch.int32, device="cuda"))
    assert e.shape == (64, 0)


def test_pack_unpack_full_size_roundtrip(bd, oracle):
    # Llama-2-7B gate_proj sized mask [4096 -> 11008]: round trip + oracle spot rows + both layouts
    torch.manual_seed(1)
    bits = torch.rand(4096, 11008, device="cuda") > 0.5
    p = bd.pack(bits)
    assert p.shape == (128, 11008)
    assert torch.equal(bd.unpack(p), bits)
    assert torch.equal(p[:2].cpu(), oracle.pack(bits[:64].cpu()))
    pt = bd.pack(bits.T.contiguous().T)          # k-major view, as BinaryDiff.__init__ packs
    assert torch.equal(pt, p)
    allones = bd.pack(torch.ones(64, 7, dtype=torch.bool, device="cuda"))
    assert (allones == -1).all()


# ------------------------------------------------------------------ BinaryDiff.__init__
def test_binarize_golden(bd, golden, oracle):
    from bitdelta_amd.diff import binarize
    for key in ("g2_binarydiff", "g2_binarydiff_fp16"):
        g = golden[key]
        mask, coeff = binarize(dev(g["base"]), dev(g["fine"]))
        assert torch.equal(mask.cpu(), g["mask"])
        assert abs(coeff.item() - g["coeff"].item()) <= 2e-7 * g["coeff"].item()
    # ragged edges + full size vs oracle / sampled
    torch.manual_seed(2)
    base = (torch.randn(200, 96) * 0.02).bfloat16()
    fine = (base.float() + torch.randn(200, 96) * 5e-4).bfloat16()
    fine[7, 5] = base[7, 5]
    m, c = binarize(dev(base), dev(fine))
    mo, co = oracle.binarize(base, fine)
    assert torch.equal(m.cpu(), mo) and abs(c.item() - co.item()) <= 2e-7 * co.item()


def test_binarydiff_module_surface(bd, golden):
    g = golden["g2_binarydiff"]
    m = bd.BinaryDiff(dev(g["base"]), dev(g["fine"]))
    assert list(m.state_dict().keys()) == g["state_keys"] == ["coeff", "mask", "base"]
    assert tuple(m.base.shape) == g["base_buf_shape"] and tuple(m.base.stride()) == g["base_buf_stride"]
    assert isinstance(m.coeff, torch.nn.Parameter) and m.coeff.requires_grad and m.coeff.dtype == torch.float32
    assert m.coeff.dim() == 0 and m.mask.dtype == torch.int32
    assert torch.equal(m.mask.cpu(), g["mask"])


# ------------------------------------------------------------------ delta GEMM
def test_binary_bmm_golden_fp16(bd, golden):
    for case in golden["g3_bmm_fp16"]:
        c = bd.binary_bmm(dev(case["a"]), dev(case["packed"]))
        assert c.dtype == torch.float16 and not c.requires_grad
        d = ulp_diff(c.cpu(), case["c"])
        assert d.max().item() <= 1
        assert (d == 0).float().mean().item() >= 0.999
    g = golden["g3_mm_fp16"]
    c = bd.binary_matmul(dev(g["a"]), dev(g["packed"]))
    assert ulp_diff(c.cpu(), g["c"]).max().item() <= 1
    g = golden["g3_bmm_fp16_big"]     # fp16 epilogue overflow -> inf, like the reference
    c = bd.binary_bmm(dev(g["a"]), dev(g["packed"]))
    assert torch.equal(torch.isinf(c.cpu()), torch.isinf(g["c"]))
    fin = ~torch.isinf(g["c"])
    assert ulp_diff(c.cpu()[fin], g["c"][fin]).max().item() <= 1


def test_binary_bmm_asserts(bd):
    a = torch.zeros(2, 4, 64, dtype=torch.float16, device="cuda")
    b = torch.zeros(2, 2, 8, dtype=torch.int32, device="cuda")
    with pytest.raises(AssertionError):
        bd.binary_bmm(a[0], b)                       # A must be 3D
    with pytest.raises(AssertionError):
        bd.binary_bmm(a, b[:, :1])                   # incompatible dimensions
    with pytest.raises(AssertionError):
        bd.binary_bmm(a, b[:1])                      # batch mismatch
    with pytest.raises(AssertionError):
        bd.binary_bmm(a.transpose(1, 2).contiguous().transpose(1, 2), b)   # A not contiguous
    out = bd.binary_bmm(a, b, activation="leaky_relu")   # accepted and ignored, like the reference
    assert out.shape == (2, 4, 8)


SHAPES = [
    # B, M, K, N, tenants, forced variant (None = auto)
    (1, 1, 64, 32, 1, None), (2, 16, 64, 32, 2, None), (3, 17, 96, 40, 3, None),      # golden-sized / generic (K=96)
    (16, 128, 512, 1024, 16, None),                                                    # notebook check shape (ipynb:519-531)
    (1, 512, 512, 512, 1, None),                                                       # notebook 2-D check (ipynb:281-292)
    (2, 200, 256, 520, 2, 0), (2, 200, 256, 520, 2, 1), (2, 200, 256, 520, 2, 2), (2, 200, 256, 520, 2, 3),
    (2, 200, 256, 520, 2, 5), (2, 200, 256, 520, 2, 100),        # (4 / 6 / 7: rejected schedules, harness-only build)
    (3, 130, 128, 300, 1, 0),                                                          # broadcast mask
    (1, 70, 64, 77, 1, None),                                                          # odd N
    (6, 1, 1024, 1000, 6, None), (3, 2, 512, 512, 3, None), (16, 1, 2048, 256, 1, None), (1, 1, 4096, 4096, 1, None),  # decode
    (6, 64, 1024, 1024, 6, None),                                                      # demo prefill
    (4, 4, 160, 200, 4, None), (16, 1, 1184, 520, 16, None), (5, 1, 1536, 300, 5, 203),  # decode: M > 1, K % 128 != 0, 16 masks, forced k-split
    (1, 16, 512, 300, 1, None), (40, 1, 256, 200, 40, None), (7, 3, 192, 136, 7, None), (20, 1, 512, 264, 1, None),  # decode: 16 rows on one mask; > 16 rows -> chunks
    (6, 1, 1024, 1000, 6, 300), (3, 2, 512, 512, 1, 300), (16, 1, 1184, 520, 16, 300),   # forced: VALU sign-flip decode kernel
    (6, 1, 1024, 1000, 6, 400), (3, 2, 512, 512, 1, 400), (4, 4, 160, 200, 4, 400), (5, 1, 1536, 300, 5, 403),   # forced: MFMA + LUT decode kernel
    (6, 1, 1024, 1000, 6, 500), (3, 2, 512, 512, 1, 500), (4, 4, 160, 200, 4, 500), (5, 1, 1536, 300, 5, 503),   # forced: no-split-k 16-column kernel
    (16, 1, 8192, 72, 16, 500), (1, 16, 512, 300, 1, 500), (40, 1, 256, 200, 40, 500),
    # forced: streaming decode kernel (600 + columns-per-block / 4).  Ragged N, K % 128 != 0, K < 1024 (waves with an empty k range),
    # M > 1, a broadcast mask shared by 16 rows, 8 masks, several 16-column tiles per block (640: 160 columns), 4-column blocks (601)
    (6, 1, 1024, 1000, 6, 600), (3, 2, 512, 512, 1, 600), (4, 4, 160, 520, 4, 600), (5, 1, 1536, 700, 5, 601),
    (8, 1, 2048, 1024, 8, 600), (1, 16, 512, 600, 1, 600), (6, 1, 1184, 1000, 6, 640), (2, 1, 4096, 1040, 2, 610),
    (1, 1, 4096, 4096, 1, 600), (3, 1, 8192, 520, 3, 603), (6, 2, 2176, 777, 6, 605),
    # forced: four-wave persistent kernels (bd_gemm_w4.h; 13 = delta-only 256x256, 14 = fused 256x128 -- 13 is refused for fused launches
    # and 14 for delta-only ones, so each list below keeps only its own).  Ragged M / N, a single k-tile (k shorter than the ring and than
    # the DMA look-ahead), several batch entries in one persistent stream, a broadcast mask, odd N (general-form epilogue)
    (2, 200, 256, 520, 2, 13), (3, 130, 64, 300, 1, 13), (1, 700, 512, 1032, 1, 13), (1, 257, 128, 77, 1, 13),
    (2, 200, 256, 520, 2, 14), (3, 130, 64, 300, 1, 14), (1, 700, 512, 1032, 1, 14), (1, 257, 128, 77, 1, 14), (6, 64, 512, 640, 6, 14),
]
# 800 (+ masks per block) = delta_rows_kernel (bd_gemv_rows.h): delta only, <= 16 activation rows per block, 64- or 32-column super-tiles.
# One mask per row (M = 1): a ragged last chunk (5 masks in chunks of 4), waves with an empty k range (K = 640: 5 iterations over 4 waves;
# K = 128: one), fewer iterations than prefetch stages, long k (4 rounds of stages), more than 16 rows in ONE launch, the automatic choice.
# M > 1 rows per mask (two masks x 4 rows per block, 5 rows per mask), ONE mask shared by every row (binary_matmul with M = 16; 2 x 8 and
# 3 x 2 rows of a batch; 16 batch entries on one mask), N % 64 != 0 (32-column super-tiles), and the automatic choice for 16 rows on one mask
ROWS_SHAPES = [(8, 1, 1024, 256, 8, 804), (16, 1, 4096, 128, 16, 804), (5, 1, 512, 192, 5, 804), (3, 1, 640, 64, 3, 802),
               (7, 1, 128, 320, 7, 801), (16, 1, 8192, 64, 16, 802), (40, 1, 256, 192, 40, 800), (4, 1, 2048, 4096, 4, None),
               (1, 16, 512, 320, 1, 800), (1, 1, 1024, 96, 1, 800), (4, 4, 640, 160, 4, 800), (4, 4, 640, 128, 4, 802), (3, 5, 256, 64, 3, 800),
               (2, 8, 384, 192, 1, 800), (3, 2, 512, 512, 1, 800), (16, 1, 2048, 256, 1, 800), (1, 16, 2048, 4096, 1, None),
               (9, 3, 384, 8192, 9, None)]
DELTA_SHAPES = [sh for sh in SHAPES if sh[5] != 14] + ROWS_SHAPES


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", DELTA_SHAPES)
def test_delta_bmm_vs_oracle(bd, oracle, dtype, shape):
    from bitdelta_amd import _lib
    B, M, K, N, T, variant = shape
    a, p, _, _ = rand_problem(B, M, K, N, dtype, T, seed=(B * 131 + M * 17 + K * 7 + N * 3 + T) & 0xffff)
    L = _lib.lib()
    L.bd_set_gemm_variant(-1 if variant is None else variant)
    try:
        c32 = bd.delta_bmm(dev(a), dev(p), out_dtype=torch.float32, round_mode=0)
        c16 = bd.delta_bmm(dev(a), dev(p), round_mode=0)
        c16r = bd.delta_bmm(dev(a), dev(p), round_mode=1)
    finally:
        L.bd_set_gemm_variant(-1)
    ref32 = oracle.delta_bmm(a, p, out_dtype=torch.float32, round_mode=0)
    fro, mrel = relerr(c32.cpu(), ref32)
    assert fro <= 1e-5 and mrel <= 1e-5, (fro, mrel)
    for got, mode in ((c16, 0), (c16r, 1)):
        ref = oracle.delta_bmm(a, p, round_mode=mode)
        ok, same = within_one_ulp(got.cpu(), ref, K)
        assert ok and same >= 0.99, (ok, same)
        if dtype == torch.float16:
            assert relerr(got.cpu(), ref32)[0] <= 1e-3
    # canaries: the same three launches into an output surrounded by >= one tile of poisoned margin on every side, with exactly the
    # scratch bytes bd_gemm_workspace_bytes() names: nothing outside the output / the request may change, and the values are the same
    import bitdelta_amd.binary_gemm_kernel as bgk
    L.bd_set_gemm_variant(-1 if variant is None else variant)
    try:
        with canary_workspace(bgk) as (ws_ok, _):
            for plain, od, rm in ((c32, torch.float32, 0), (c16, dtype, 0), (c16r, dtype, 1)):
                cn = CanaryOut(B, M, N, od)
                bd.delta_bmm(dev(a), dev(p), out=cn.view, out_dtype=od, round_mode=rm)
                assert torch.equal(cn.result(), plain), (od, rm)
                assert cn.untouched_outside(), ("output margin overwritten", od, rm)
            assert ws_ok(), "wrote past bd_gemm_workspace_bytes()"
    finally:
        L.bd_set_gemm_variant(-1)


# the one-pass fused kernel (variant 8, bd_binary_linear only): multi-tenant, ragged M/N, k shorter than its 3-slot ring
# fused launches: forced 0 / 5 are the two-loop A/B references (harness-only build) -> the shipped library refuses them (tested below)
LINEAR_SHAPES = [sh for sh in SHAPES if not (sh[5] in (0, 5, 13))] + [(3, 130, 128, 300, 1, None),
                          (2, 200, 256, 520, 2, 8), (1, 257, 64, 136, 1, 8), (3, 300, 128, 264, 1, 8), (1, 512, 1024, 384, 1, 8),
                          (2, 200, 256, 520, 2, 9), (1, 257, 64, 136, 1, 9), (3, 300, 128, 264, 1, 9), (1, 512, 1024, 384, 1, 9),   # 9 = 128x128 tile
                          (2, 200, 256, 520, 2, 10), (1, 128, 512, 384, 1, 10), (1, 40, 1024, 264, 1, None), (3, 33, 2048, 1024, 3, None),  # split-k (mid M)
                          (1, 96, 4096, 1024, 1, None),
                          # 11 = 64x128 tile (up to 64 rows per mask: multi-tenant prefill of short prompts; also under split-k)
                          (6, 64, 512, 640, 6, None), (6, 64, 512, 640, 6, 11), (3, 33, 256, 264, 3, 11), (2, 200, 256, 520, 2, 11),
                          (1, 64, 1024, 384, 1, 10), (4, 17, 192, 136, 4, 11), (6, 64, 4096, 512, 6, None),
                          (6, 64, 512, 640, 6, 12), (3, 33, 256, 264, 3, 12), (2, 200, 256, 520, 2, 12), (4, 64, 256, 8200, 4, None),
                          # 16 / 17 = pair tiles (two batch entries of <= 64 rows per 128x128 tile; 17 adds split-k): even and odd batches, ragged
                          # rows per entry, a broadcast mask shared by the pair, a single k-tile per slice, wide N, and the automatic choice
                          (6, 64, 512, 640, 6, 16), (3, 33, 256, 264, 3, 16), (5, 64, 1024, 384, 5, 16), (2, 17, 192, 136, 2, 16),
                          (4, 64, 512, 640, 1, 16), (4, 64, 256, 8200, 4, 16), (6, 64, 4096, 512, 6, 17), (3, 33, 2048, 264, 3, 17),
                          (5, 40, 1024, 1032, 5, 17), (2, 64, 128, 136, 2, 17), (6, 48, 2048, 1024, 6, None), (2, 64, 4096, 256, 2, None),
                          # 18 = FOUR-WAVE pair tiles (bd_gemm_w4.h PAIR: one wave per SIMD, wave tile 64 x 64, persistent over (pair, column tile)):
                          # even / odd batches, ragged rows per entry, a broadcast mask, k shorter than the ring, ragged and wide N, several rounds
                          (6, 64, 512, 640, 6, 18), (3, 33, 256, 264, 3, 18), (5, 64, 1024, 384, 5, 18), (2, 17, 192, 136, 2, 18),
                          (4, 64, 512, 640, 1, 18), (4, 64, 256, 8200, 4, 18), (2, 64, 64, 136, 2, 18), (7, 48, 128, 33000, 7, 18),
                          # 19 = the same + split-k (slices of whole k-tiles; fp32 slabs + the reduce launch)
                          (6, 64, 4096, 512, 6, 19), (3, 33, 2048, 264, 3, 19), (5, 40, 1024, 1032, 5, 19), (2, 64, 128, 136, 2, 19),
                          (4, 64, 1536, 640, 1, 19),
                          # 20 = four-wave 128x128 tile, one entry per tile (65 .. 128 rows per tenant; also ragged M / several row tiles)
                          (6, 128, 512, 640, 6, 20), (3, 100, 256, 264, 3, 20), (2, 200, 256, 520, 2, 20), (1, 257, 64, 136, 1, 20),
                          (4, 128, 256, 8200, 1, 20)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", LINEAR_SHAPES)
def test_binary_linear_vs_oracle(bd, oracle, dtype, shape):
    from bitdelta_amd import _lib
    B, M, K, N, T, variant = shape
    a, p, w, alpha = rand_problem(B, M, K, N, dtype, T, seed=(B * 131 + M * 17 + K * 7 + N * 3 + T + 7) & 0xffff)
    L = _lib.lib()
    L.bd_set_gemm_variant(-1 if variant is None else variant)
    try:
        y32 = bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), out_dtype=torch.float32)
        y16 = bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha))
    finally:
        L.bd_set_gemm_variant(-1)
    ref32 = oracle.binary_linear(a, w, p, alpha, out_dtype=torch.float32, round_mode=0)
    fro, mrel = relerr(y32.cpu(), ref32)
    assert fro <= 1e-5 and mrel <= 2e-5, (fro, mrel)
    ref16 = ref32.to(dtype)
    ok, same = within_one_ulp(y16.cpu(), ref16, K)
    assert ok and same >= 0.99, (ok, same)
    fro16 = relerr(y16.cpu(), ref32)[0]
    assert fro16 <= (1e-3 if dtype == torch.float16 else 3e-3)
    # not worse than the reference's own multi-rounding chain (SURVEY.md 7b iv)
    ref_chain = oracle.binary_linear(a, w, p, alpha, round_mode=1)
    assert fro16 <= relerr(ref_chain, ref32)[0] * 1.05 + 1e-7
    # canaries (tests/canary.py): output inside poisoned margins, scratch of exactly bd_gemm_workspace_bytes(); same values, nothing else touched.
    # Covers the ragged-M / ragged-N guarded epilogues of every tile family in LINEAR_SHAPES, the pair tiles' absent partner, the
    # split-k slabs and the fused-residual epilogue (in place on the canary view).
    import bitdelta_amd.binary_gemm_kernel as bgk
    L.bd_set_gemm_variant(-1 if variant is None else variant)
    try:
        with canary_workspace(bgk) as (ws_ok, _):
            for plain, od in ((y32, torch.float32), (y16, dtype)):
                cn = CanaryOut(B, M, N, od)
                bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), out_dtype=od, out=cn.view)
                assert torch.equal(cn.result(), plain), od
                assert cn.untouched_outside(), ("output margin overwritten", od)
            if variant is None or variant in (8, 9, 11, 12, 14, 16, 17, 18, 19, 20) or variant >= 200:      # families with a residual epilogue (bd_api.hip)
                cn = CanaryOut(B, M, N, dtype)
                r = torch.randn(B, M, N).to(dtype)
                cn.view.copy_(dev(r))
                bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), residual=cn.view)
                got = cn.result().cpu().float()
                # (up to one ulp of the Linear output + one ulp of the sum: two roundings at M > 16, and the kernel may sit 1 ulp from the oracle)
                assert ((got - (r.float() + ref32)).abs() <= (r.float().abs() + ref32.abs()) * (2 ** -6 if dtype == torch.bfloat16 else 2 ** -9) + 1e-4).all()
                assert cn.untouched_outside(), "residual epilogue wrote outside the output"
            assert ws_ok(), "wrote past bd_gemm_workspace_bytes()"
    finally:
        L.bd_set_gemm_variant(-1)


def test_pruned_ab_variants_are_refused_not_silently_replaced(bd):
    """The shipped library holds only dispatched kernels; a forced A/B-only variant answers BD_E_BAD_SHAPE, never a fallback."""
    from bitdelta_amd import _lib
    L = _lib.lib()
    a, p, w, alpha = rand_problem(1, 200, 256, 520, torch.bfloat16, 1, seed=3)
    for variant, fused in ((4, False), (6, False), (7, True), (0, True), (5, True)):
        L.bd_set_gemm_variant(variant)
        try:
            with pytest.raises(_lib.BitDeltaHipError):
                if fused:
                    bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha))
                else:
                    bd.delta_bmm(dev(a), dev(p))
        finally:
            L.bd_set_gemm_variant(-1)
    # ... and the round-4 loader / consumer decode kernel (variant 700, tests/native/ab/bd_gemv_ring.h: lost its A/B by 26-45 %) is a
    # harness-only kernel too: forcing it on a packed decode launch is refused, and bd_set_decode_engine(1) changes nothing
    from bitdelta_amd.binary_gemm_kernel import binary_linear_decode, pack_decode_masks
    a, p, w, alpha = rand_problem(6, 1, 1024, 1024, torch.float16, 6, seed=4)
    pk = pack_decode_masks(dev(p))
    ref = binary_linear_decode(dev(a), dev(w), pk, dev(alpha), layout="packed")
    L.bd_set_gemm_variant(700)
    try:
        with pytest.raises(_lib.BitDeltaHipError):
            binary_linear_decode(dev(a), dev(w), pk, dev(alpha), layout="packed")
    finally:
        L.bd_set_gemm_variant(-1)
    L.bd_set_decode_engine(1)
    try:
        got = binary_linear_decode(dev(a), dev(w), pk, dev(alpha), layout="packed")
        assert L.bd_last_gemm_variant() == 600 and torch.equal(got, ref)
    finally:
        L.bd_set_decode_engine(-1)


def test_decode_in_launch_reduction_matches_two_launch_form(bd):
    """Split-k partials are summed in k-slice order by whichever block arrives last (ticket) -- bit-identical to the two-launch
    reduce kernel, on every repeat (the ticket area of the persistent workspace is handed back zeroed)."""
    from bitdelta_amd import _lib
    L = _lib.lib()
    a, p, w, alpha = rand_problem(6, 1, 4096, 1024, torch.float16, 6, seed=21)
    a, p, w, alpha = dev(a), dev(p), dev(w), dev(alpha)
    try:
        L.bd_set_gemm_variant(300)             # the split-k VALU decode kernel (the automatic choice here is the streaming kernel)
        L.bd_set_decode_two_launch(1)
        ref = bd.binary_linear(a, w, p, alpha).clone()
        L.bd_set_decode_two_launch(0)
        for _ in range(5):
            got = bd.binary_linear(a, w, p, alpha)
            assert L.bd_last_gemm_variant() == 300
            assert torch.equal(got, ref)
    finally:
        L.bd_set_decode_two_launch(1)          # library default
        L.bd_set_gemm_variant(-1)


def test_four_wave_persistent_kernels(bd, oracle):
    """bd_gemm_w4.h beyond the per-shape oracle checks above: (i) a persistent workgroup walking SEVERAL tiles and batch entries (more
    tiles than CUs), (ii) bit-identity with the 8-wave kernels it replaces (same per-element MFMA sequence), (iii) every epilogue
    form: reference fp16 rounding, C += alpha * acc with scale groups, fused + residual, grouped alpha, strided C rows."""
    from bitdelta_amd import _lib
    L = _lib.lib()
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def forced(v, fn):
        L.bd_set_gemm_variant(v)
        try:
            out = fn()
            assert L.bd_last_gemm_variant() == v
            return out
        finally:
            L.bd_set_gemm_variant(-1)

    # (i) + (ii): 3 x (5 x 64) = 960 fused tiles / 3 x (5 x 32) = 480 delta tiles on <= 256 CUs, ragged edges
    B, M, K, N = 3, 1100, 320, 8100 + 8
    a, p, w, alpha = rand_problem(B, M, K, N, torch.bfloat16, B, seed=21)
    assert B * ((M + 255) // 256) * ((N + 255) // 256) > cus
    d13 = forced(13, lambda: bd.delta_bmm(dev(a), dev(p), round_mode=1))
    d0 = forced(0, lambda: bd.delta_bmm(dev(a), dev(p), round_mode=1))
    assert torch.equal(d13, d0)
    y14 = forced(14, lambda: bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha)))
    y8 = forced(8, lambda: bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha)))
    assert torch.equal(y14, y8)
    rows = torch.tensor([0, 255, 256, 511, 777, 1099])
    ref = oracle.binary_linear(a[:, rows].contiguous(), w, p, alpha, out_dtype=torch.float32)
    ok, same = within_one_ulp(y14[:, rows].cpu(), ref.bfloat16(), K)
    assert ok and same >= 0.99
    # the automatic choice at these sizes IS the four-wave kernel
    bd.delta_bmm(dev(a), dev(p))
    assert L.bd_last_gemm_variant() == 13
    bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha))
    assert L.bd_last_gemm_variant() == 14

    # (iii) epilogue forms
    a, p, w, _ = rand_problem(2, 300, 256, 512, torch.bfloat16, 2, seed=22)
    alpha = torch.tensor([[3e-4, 4e-4, 5e-4, 6e-4], [1e-3, 2e-3, 3e-3, 4e-3]])
    scale = alpha.repeat_interleave(128, dim=1)[:, None, :]
    ref = oracle.delta_bmm(a, p, out_dtype=torch.float32, round_mode=0)
    base = (a.float() @ w.float().T).bfloat16()
    out = dev(base.clone())
    forced(13, lambda: bd.delta_bmm(dev(a), dev(p), out=out, alpha=dev(alpha), accumulate=True, groups=4))
    assert ulp_diff(out.cpu(), (base.float() + scale * ref).bfloat16()).max().item() <= 1
    z = forced(13, lambda: bd.delta_bmm(dev(a), dev(p), alpha=dev(alpha), groups=4, out_dtype=torch.float32))
    assert relerr(z.cpu(), scale * ref)[0] <= 1e-5
    y = forced(14, lambda: bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), groups=4))
    yo = oracle.binary_linear(a, w, p, alpha, G=4, out_dtype=torch.float32)
    ok, same = within_one_ulp(y.cpu(), yo.bfloat16(), 256)
    assert ok and same >= 0.99
    y32 = forced(14, lambda: bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), groups=4, out_dtype=torch.float32))
    assert relerr(y32.cpu(), yo)[0] <= 1e-5
    for dt in (torch.bfloat16, torch.float16):          # fused + residual: two roundings, like the separate add
        a, p, w, al1 = rand_problem(2, 300, 256, 512, dt, 2, seed=23)
        r = torch.randn(2, 300, 512).to(dt)
        got = forced(14, lambda: bd.binary_linear(dev(a), dev(w), dev(p), dev(al1), residual=dev(r).clone()))
        y32 = oracle.binary_linear(a, w, p, al1, out_dtype=torch.float32)
        want = (r.float() + y32.to(dt).float()).to(dt)
        # one ulp at the magnitude of the larger addend (the sum may cancel to something much smaller than its terms) ...
        d = (got.cpu().float() - want.float()).abs()
        # (ulp(y) + ulp(sum): the kernel's y may sit 1 ulp from the oracle's and the sum can then round the other way -- seen at 2 of 80
        #  residual draws with the former half-width band while the kernel equalled the separate ops bit for bit; tools/dbg_fourwave.py)
        assert (d <= (r.float().abs() + y32.abs()) * (2 ** -9 if dt == torch.float16 else 2 ** -6) + 1e-4).all()
        assert (got.cpu() == want).float().mean().item() >= 0.99
        # ... and exactly the separate ops wherever the Linear output itself is bit-equal
        y16 = forced(14, lambda: bd.binary_linear(dev(a), dev(w), dev(p), dev(al1)))
        assert torch.equal(got, dev(r) + y16)
    # strided output rows (a column slice of a wider buffer): the aligned fast form with sCm != N, and an unaligned slice (general form)
    a, p, w, al1 = rand_problem(1, 300, 128, 256, torch.bfloat16, 1, seed=24)
    ref = oracle.delta_bmm(a, p, round_mode=1)
    for off in (8, 3):
        wide = torch.zeros(1, 300, 256 + 16, dtype=torch.bfloat16, device="cuda")
        forced(13, lambda: bd.delta_bmm(dev(a), dev(p), out=wide[:, :, off:off + 256]))
        ok, same = within_one_ulp(wide[:, :, off:off + 256].cpu().contiguous(), ref, 128)
        assert ok and same >= 0.99 and (wide[:, :, :off] == 0).all() and (wide[:, :, off + 256:] == 0).all()


def test_prefill_swiglu_epilogue_is_the_two_launches(bd, oracle):
    """bd_binary_linear_swiglu (fused gate|up GEMM + SwiGLU epilogue, bd_gemm_w4.h EPI = 1) == bd_binary_linear followed by
    bd_srv_swiglu, bit for bit, and both agree with the oracle's Linear pushed through the same activation arithmetic."""
    from bitdelta_amd.binary_gemm_kernel import binary_linear_swiglu
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd import _lib
    from bitdelta_amd.serving_loop import FusedDeltaLinear
    L = _lib.lib()
    for dt in (torch.bfloat16, torch.float16):
        for B, M, K, inter, T in ((1, 300, 256, 264, 1), (2, 130, 64, 1032, 2), (3, 17, 128, 136, 1), (1, 700, 512, 2752, 1)):
            g = torch.Generator().manual_seed(B * 7 + M)
            x = torch.randn(B, M, K, generator=g).to(dt)
            ws, ms, cs = [], [], []
            for _ in range(2):
                w = (torch.randn(inter, K, generator=g) * 0.02).to(dt)
                ws.append(w.cuda())
                ms.append(torch.randint(-2 ** 31, 2 ** 31 - 1, (T, K // 32, inter), generator=g, dtype=torch.int64).to(torch.int32).cuda())
                cs.append((torch.rand(T, generator=g) * 2e-4 + 3e-4).cuda())
            f = FusedDeltaLinear(ws, ms, cs, interleave8=True, decode_copies=False)
            assert f.swiglu_ok(x.cuda())
            got = f.forward_swiglu(x.cuda())
            # (several entries of <= 64 rows: the pair tile with the same epilogue, round 6)
            assert L.bd_last_gemm_variant() == (21 if (B >= 2 and M <= 64) else 15) and got.shape == (B, M, inter)
            two = ops.swiglu_interleaved8(f(x.cuda()))
            assert torch.equal(got, two)
            # oracle: the two projections separately, then round(silu(round(g))) * round(u)
            go = oracle.binary_linear(x, ws[0].cpu(), ms[0].cpu(), cs[0].cpu().reshape(T, 1), out_dtype=torch.float32).to(dt).float()
            uo = oracle.binary_linear(x, ws[1].cpu(), ms[1].cpu(), cs[1].cpu().reshape(T, 1), out_dtype=torch.float32).to(dt).float()
            want = ((go / (1 + torch.exp(-go))).to(dt).float() * uo).to(dt)
            d = (got.cpu().float() - want.float()).abs()
            tol = (go.abs() * uo.abs() + uo.abs() + go.abs()) * (2 ** -9 if dt == torch.float16 else 2 ** -6) + 1e-4
            assert (d <= tol).all() and (got.cpu() == want).float().mean().item() >= 0.97
    with pytest.raises(Exception):          # decode-size launches are refused (their fused form is bd_binary_linear_decode_fused)
        binary_linear_swiglu(torch.zeros(1, 8, 64, device="cuda", dtype=torch.bfloat16), torch.zeros(32, 64, device="cuda", dtype=torch.bfloat16),
                             torch.zeros(1, 2, 32, device="cuda", dtype=torch.int32), torch.ones(1, 2, device="cuda"))


def test_delta_bmm_alpha_accumulate_and_groups(bd, oracle):
    a, p, w, _ = rand_problem(2, 150, 256, 512, torch.bfloat16, 2, seed=11)
    alpha = torch.tensor([[3e-4, 4e-4, 5e-4, 6e-4], [1e-3, 2e-3, 3e-3, 4e-3]])
    base = (a.float() @ w.float().T).bfloat16()
    out = dev(base.clone())
    bd.delta_bmm(dev(a), dev(p), out=out, alpha=dev(alpha), accumulate=True, groups=4)
    ref = oracle.delta_bmm(a, p, out_dtype=torch.float32, round_mode=0)
    scale = alpha.repeat_interleave(128, dim=1)[:, None, :]
    want = (base.float() + scale * ref).bfloat16()
    assert ulp_diff(out.cpu(), want).max().item() <= 1
    y = bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), groups=4, out_dtype=torch.float32)
    yo = oracle.binary_linear(a, w, p, alpha, G=4, out_dtype=torch.float32)
    assert relerr(y.cpu(), yo)[0] <= 1e-5
    z = bd.delta_bmm(dev(a), dev(p), alpha=dev(alpha), groups=4, out_dtype=torch.float32)
    assert relerr(z.cpu(), scale * ref)[0] <= 1e-5


def test_decode_stream_kernel_groups_accumulate_and_determinism(bd, oracle):
    """Streaming decode kernel: several scale groups inside one block's column range (scales come from its LDS table), the
    `C += alpha * acc` epilogue, and run-to-run determinism (fixed wave-order reduction, no atomics)."""
    from bitdelta_amd import _lib
    L = _lib.lib()
    a, p, w, _ = rand_problem(3, 2, 640, 1024, torch.float16, 3, seed=31)
    alpha = (torch.rand(3, 8) * 4e-4 + 2e-4).float()                    # 8 groups of 128 columns
    for variant in (640, 601):                                          # 160-column blocks span two groups; 4-column blocks one
        L.bd_set_gemm_variant(variant)
        try:
            y = bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), groups=8, out_dtype=torch.float32)
            assert L.bd_last_gemm_variant() == 600
            y2 = bd.binary_linear(dev(a), dev(w), dev(p), dev(alpha), groups=8, out_dtype=torch.float32)
            base = (a.float() @ w.float().T).half()
            out = dev(base.clone())
            bd.delta_bmm(dev(a), dev(p), out=out, alpha=dev(alpha), accumulate=True, groups=8)
        finally:
            L.bd_set_gemm_variant(-1)
        assert torch.equal(y, y2)
        yo = oracle.binary_linear(a, w, p, alpha, G=8, out_dtype=torch.float32)
        assert relerr(y.cpu(), yo)[0] <= 1e-5
        ref = oracle.delta_bmm(a, p, out_dtype=torch.float32, round_mode=0)
        want = (base.float() + alpha.repeat_interleave(128, dim=1)[:, None, :] * ref).half()
        assert ulp_diff(out.cpu(), want).max().item() <= 1


# ------------------------------------------------------------------ module forward / multi-tenant
def test_binarydiff_forward_golden(bd, golden):
    for tag in ("bf16", "fp16"):
        g = golden[f"g5_forward_{tag}"]
        mod = bd.BinaryDiff(dev(g["base"]), dev(g["fine"]))
        assert torch.equal(mod.mask.cpu(), g["mask"])
        with torch.no_grad():
            y = mod(dev(g["x"]))
        assert y.dtype == g["y"].dtype and y.shape == g["y"].shape
        # reference output carries 4 roundings (GEMM, fp16+store-cast, coeff*, +), ours 1: <= 1 ulp of the format each,
        # with an absolute floor for outputs where base and delta terms cancel (ulp distance is meaningless near 0)
        rt, at = (2 ** -7, 2e-4) if tag == "bf16" else (2 ** -10, 3e-5)
        assert torch.allclose(y.float().cpu(), g["y"].float(), rtol=rt, atol=at)
        assert relerr(y.cpu(), g["y"].float())[0] <= (6e-3 if tag == "bf16" else 1e-3)
        # training form (grad enabled): same composition as the reference, grads as the reference defines them
        mod.coeff.grad = None
        xg = dev(g["x"]).clone().requires_grad_(True)
        yt = mod(xg)
        assert torch.allclose(yt.detach().float().cpu(), g["y"].float(), rtol=rt, atol=at)
        yt.float().sum().backward()
        assert mod.coeff.grad is not None and xg.grad is not None
        wsum = g["base"].float().sum(0)                          # d/dx flows through `x @ base` only
        assert torch.allclose(xg.grad[0, 0].float().cpu(), wsum, rtol=2e-2, atol=1e-3)


def test_diffcompress_module_golden(bd, golden):
    g = golden["g6_multitenant_fp16"]
    lin = torch.nn.Linear(g["w"].shape[1], g["w"].shape[0], bias=False, dtype=torch.float16, device="cuda")
    with torch.no_grad():
        lin.weight.copy_(dev(g["w"]))
    mod = bd.DiffCompressModule(lin, dev(g["masks"]), dev(g["coeffs"]))
    with torch.no_grad():
        y = mod(dev(g["h"]))
    assert y.shape == g["y"].shape and y.dtype == torch.float16
    assert torch.allclose(y.float().cpu(), g["y"].float(), rtol=2 ** -10, atol=3e-5)
    assert relerr(y.cpu(), g["y"].float())[0] <= 1e-3


# ------------------------------------------------------------------ merge / diff.pt format
def test_merge_delta_golden_bit_exact(bd, golden):
    from bitdelta_amd.diff import merge_delta_
    for key in ("g7_merge_fp16", "g7_merge_bf16"):
        g = golden[key]
        w = merge_delta_(dev(g["w"]).clone(), dev(g["mask"]), dev(g["coeff"]))
        assert torch.equal(w.cpu().view(torch.int16), g["w_merged"].view(torch.int16))


def test_merge_full_size_vs_oracle_rows(bd, oracle):
    from bitdelta_amd.diff import merge_delta_
    torch.manual_seed(3)
    w = (torch.randn(1000, 4096) * 0.02).half()
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (128, 1000), dtype=torch.int64).to(torch.int32)
    got = merge_delta_(dev(w).clone(), dev(mask), torch.tensor(4.2e-4, device="cuda"))
    want = oracle.merge_delta(w.clone(), mask, 4.2e-4)
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


def test_diff_pt_format_and_load_diff(bd, tmp_path):
    transformers = pytest.importorskip("transformers")
    import copy
    gm = torch.load(os.path.join(GOLDEN, "tiny_llama_merged.pt"), weights_only=False)
    ref_diff = torch.load(os.path.join(GOLDEN, "tiny_llama_diff.pt"), weights_only=False)
    cfg = transformers.LlamaConfig(**{k: v for k, v in gm["config"].items()
                                      if k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers",
                                               "num_attention_heads", "num_key_value_heads", "max_position_embeddings")})
    base_m = transformers.LlamaForCausalLM(cfg).bfloat16()
    base_m.load_state_dict(gm["base_state"])
    fine_m = transformers.LlamaForCausalLM(cfg).bfloat16()
    fine_m.load_state_dict(gm["fine_state"])
    base_m, fine_m = base_m.cuda(), fine_m.cuda()
    comp = copy.deepcopy(fine_m)
    bd.compress_diff(base_m, fine_m, comp)
    path = str(tmp_path / "diff.pt")
    bd.save_diff(comp, path)
    ours = torch.load(path, weights_only=False)
    # same keys in the same order, same python types / dtypes / shapes, identical masks and (fp32) coeffs
    assert list(ours.keys()) == gm["diff_keys"]
    for k, v in ours.items():
        # python type: the reference stores `param.cpu()` -- the Parameter itself for a CPU model (how the golden file was
        # written), a plain Tensor copy for a GPU model (this run); both unpickle and load identically
        assert type(v).__name__ in ("Tensor", "Parameter")
        assert (str(v.dtype), tuple(v.shape)) == gm["diff_types"][k][1:], k
        if k.endswith(".mask"):
            assert torch.equal(v, ref_diff[k]), k
        elif k.endswith(".coeff"):
            assert abs(v.item() - ref_diff[k].item()) <= 2e-7 * ref_diff[k].item(), k
        else:
            assert torch.equal(v, ref_diff[k]), k
    # our load_diff on the REFERENCE-written file reproduces the reference's merged fp16 weights bit for bit
    eval_m = copy.deepcopy(base_m).half()
    bd.load_diff(eval_m, os.path.join(GOLDEN, "tiny_llama_diff.pt"))
    after = eval_m.state_dict()
    for k, v in gm["after"].items():
        assert torch.equal(after[k].cpu().view(torch.int16), v.view(torch.int16)), k


def _tiny_llama(transformers, gm):
    cfg = transformers.LlamaConfig(**{k: v for k, v in gm["config"].items()
                                      if k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers",
                                               "num_attention_heads", "num_key_value_heads", "max_position_embeddings")})
    m = transformers.LlamaForCausalLM(cfg).bfloat16()
    m.load_state_dict(gm["base_state"])
    return cfg, m


def test_load_diff_takes_all_three_branches_like_the_reference(bd):
    """G8 (tests/golden/make_golden_lowrank.py): a diff.pt with 1-bit `.mask` / `.coeff` entries, dense `.weight` replacements AND low-rank
    `.A` / `.B` pairs -- bitdelta/diff.py:88-104.  Our load_diff (HIP merge for the 1-bit entries) on that file leaves bit for bit the weights
    the reference's load_diff left (VERDICT r05 missing #3: the `.A/.B` branch was implemented but never executed)."""
    transformers = pytest.importorskip("transformers")
    import copy
    gm = torch.load(os.path.join(GOLDEN, "tiny_llama_merged.pt"), weights_only=False)
    g8 = torch.load(os.path.join(GOLDEN, "tiny_llama_lowrank.pt"), weights_only=False)
    assert any(k.endswith(".A") for k in g8["diff"]) and any(k.endswith(".mask") for k in g8["diff"]) and "lm_head.weight" in g8["diff"]
    _, base_m = _tiny_llama(transformers, gm)
    eval_m = copy.deepcopy(base_m).half().cuda()
    untouched = {k: v.detach().clone() for k, v in eval_m.state_dict().items() if k not in g8["after"]}
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "diff.pt")
        torch.save(g8["diff"], path)
        bd.load_diff(eval_m, path)
    after = eval_m.state_dict()
    for k, v in g8["after"].items():
        assert after[k].dtype == v.dtype and torch.equal(after[k].cpu().view(torch.int16), v.view(torch.int16)), k
    for k, v in untouched.items():                                   # nothing else moved
        assert torch.equal(after[k], v), k
    assert eval_m.config.vocab_size == eval_m.lm_head.weight.size(0)


def test_save_full_model_writes_the_merged_checkpoint(bd, oracle, tmp_path):
    """save_full_model (bitdelta/diff.py:108-116): base checkpoint + diff.pt -> a plain HF checkpoint + the fine-tune's tokenizer.  A tiny Llama
    saved to disk stands in for the hub models; the reloaded checkpoint's projections equal the oracle's merge line on the bf16 base weights bit for
    bit (G7 pins that line in bf16), every dense entry of the diff.pt replaced its tensor, and the tokenizer files are there."""
    transformers = pytest.importorskip("transformers")
    import json
    gm = torch.load(os.path.join(GOLDEN, "tiny_llama_merged.pt"), weights_only=False)
    diff = torch.load(os.path.join(GOLDEN, "tiny_llama_diff.pt"), weights_only=False)
    _, base_m = _tiny_llama(transformers, gm)
    base_dir, tok_dir, out_dir = tmp_path / "base", tmp_path / "tok", tmp_path / "out"
    base_m.save_pretrained(str(base_dir))
    tok_dir.mkdir()
    vocab = {chr(97 + i): i for i in range(26)}
    vocab["<|endoftext|>"] = 26
    json.dump(vocab, open(tok_dir / "vocab.json", "w"))
    open(tok_dir / "merges.txt", "w").write("#version: 0.2\n")
    json.dump({"tokenizer_class": "GPT2Tokenizer", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "bos_token": "<|endoftext|>"},
              open(tok_dir / "tokenizer_config.json", "w"))
    bd.save_full_model(str(base_dir), str(tok_dir), os.path.join(GOLDEN, "tiny_llama_diff.pt"), str(out_dir), "cuda")
    assert (out_dir / "config.json").exists() and any(f.name.startswith("tokenizer") for f in out_dir.iterdir())
    full = transformers.LlamaForCausalLM.from_pretrained(str(out_dir), torch_dtype=torch.bfloat16)
    sd, n_proj = full.state_dict(), 0
    for k, v in gm["base_state"].items():
        mod = k.rsplit(".", 1)[0]
        if mod + ".mask" in diff:
            want = oracle.merge_delta(v.detach().clone().contiguous(), diff[mod + ".mask"].contiguous(), float(diff[mod + ".coeff"]))
            assert torch.equal(sd[k].view(torch.int16), want.view(torch.int16)), k
            n_proj += 1
        elif k in diff:
            assert torch.equal(sd[k], diff[k].detach().to(sd[k].dtype)), k
        else:
            assert torch.equal(sd[k], v), k
    assert n_proj == 14                                              # 7 projections x 2 layers


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
def test_full_size_linearity_and_sign_flip(bd):
    # 4096x4096 layer, M = 4096 rows: size-independent properties instead of an O(MNK) CPU reference
    torch.manual_seed(5)
    K = N = 4096
    M = 4096
    x1 = torch.randn(1, M, K, device="cuda").bfloat16()
    x2 = torch.randn(1, M, K, device="cuda").bfloat16()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), device="cuda", dtype=torch.int64).to(torch.int32)
    f = lambda x, pp: bd.delta_bmm(x, pp, out_dtype=torch.float32, round_mode=0)
    y1, y2 = f(x1, p), f(x2, p)
    ys = f((x1.float() * 2).bfloat16(), p)                           # exact scaling by 2
    assert torch.equal(ys, y1 * 2)
    yn = f(x1, ~p)                                                   # flipping every sign bit negates the result
    # (exactly, up to the MFMA accumulator's rounding not being perfectly sign-symmetric: ~1e-5 of the elements differ
    #  by one fp32 ulp on MI355X)
    assert torch.allclose(yn, -y1, rtol=2e-6, atol=1e-5)
    assert (yn != -y1).float().mean().item() < 1e-3
    xs = (x1.float() + x2.float())
    exact = xs.bfloat16().float() == xs                              # rows where the bf16 sum is exact
    rows = exact.all(dim=-1)[0]
    if rows.any():
        y12 = f(xs.bfloat16(), p)
        assert torch.allclose(y12[0, rows], (y1 + y2)[0, rows], rtol=1e-5, atol=1e-2)
    # a column whose signs are all +1 reproduces the row sums
    p1 = p.clone()
    p1[0, :, 123] = -1
    ysum = f(x1, p1)[0, :, 123]
    assert torch.allclose(ysum, x1[0].float().sum(-1), rtol=1e-5, atol=1e-2)
    # spot rows against torch fp32 math on the device (independent of the oracle)
    s = (bd.unpack(p[0]).float() * 2 - 1)
    ref = x1[0, :64].float() @ s
    assert relerr(y1[0, :64], ref)[0] <= 1e-5


def test_full_size_fused_spot_rows(bd):
    torch.manual_seed(6)
    K, N, M = 4096, 11008, 2048                                      # Llama-2-7B gate_proj at prefill 2048
    x = torch.randn(1, M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    fine = (w.float() + torch.randn(N, K, device="cuda") * 5e-4).bfloat16()
    mod = bd.BinaryDiff(w, fine)
    with torch.no_grad():
        y = mod(x)
    rows = torch.tensor([0, 1, 255, 256, 1023, 2047], device="cuda")
    s = (bd.unpack(mod.mask).float() * 2 - 1)
    ref = x[0, rows].float() @ w.float().T + mod.coeff.detach() * (x[0, rows].float() @ s)
    # 1 bf16 ulp, with the absolute floor of fp32 accumulation noise for outputs that cancel to ~0 (|y| ~ 1e-5 among O(1) values)
    ok, same = within_one_ulp(y[0, rows].cpu(), ref.bfloat16().cpu(), K)
    assert ok and same >= 0.98, (ok, same)


def test_binary_gemm_accepts_every_reference_word_width(bd, oracle):
    """The reference kernels take n_bits (binary_gemm_kernel.py:109-111, :128, :153): 8 / 16 / 64-bit packed operands give the same
    result as the int32 pack of the same bits."""
    torch.manual_seed(17)
    bits = torch.rand(2, 256, 72, device="cuda") > 0.5
    a = torch.randn(2, 5, 256, device="cuda").half()
    want = bd.binary_bmm(a, bd.pack(bits, 32))
    ref = oracle.delta_bmm(a.cpu(), oracle.pack(bits.cpu(), 32), round_mode=1)
    assert ulp_diff(want.cpu(), ref).max().item() <= 1
    for nb in (8, 16, 64):
        got = bd.binary_bmm(a, bd.pack(bits, nb), n_bits=nb)
        assert torch.equal(got, want), nb
        got2 = bd.binary_matmul(a[0], bd.pack(bits[0], nb), n_bits=nb)
        assert torch.equal(got2, want[0]), nb
    with pytest.raises(AssertionError):
        bd.binary_bmm(a, bd.pack(bits, 8), n_bits=16)                  # word dtype must match n_bits


DECODE_LAYOUT_SHAPES = [
    # B (tenants), M, K, N, per-tenant masks?
    (6, 1, 1024, 1000, True), (1, 1, 4096, 4096, True), (3, 2, 512, 520, True), (5, 1, 1184, 777, True), (8, 2, 288, 640, True),
    (2, 4, 160, 520, True), (1, 16, 512, 600, True), (4, 1, 2048, 1040, True), (6, 1, 4096, 6144, True), (7, 1, 384, 512, True),
    # 9 .. 16 tenants in ONE launch (packed layout only, t_pad 12 / 16): the reference's batched benchmark runs B = 16
    # (notebooks/binary_gemm_kernel_triton.ipynb:759)
    (12, 1, 1024, 1000, True), (16, 1, 4096, 4096, True), (16, 1, 1184, 520, True), (9, 1, 2048, 1040, True), (11, 1, 512, 640, True),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", DECODE_LAYOUT_SHAPES)
def test_binary_linear_decode_layouts_vs_oracle(bd, oracle, dtype, shape):
    """bd_binary_linear_decode with tile-major and with packed (interleaved tenants, natural k order) sign words: same values as the
    reference-layout call and the oracle; the repacks themselves are exact byte shuffles (checked against their definition)."""
    B, M, K, N, _ = shape
    a, p, w, alpha = rand_problem(B, M, K, N, dtype, B, seed=(B * 77 + M * 13 + K + N) & 0xffff)
    ref32 = oracle.binary_linear(a, w, p, alpha, out_dtype=torch.float32, round_mode=0)
    pd = dev(p)
    tiled = bd.tile_masks(pd)
    packed = bd.pack_decode_masks(pd)
    assert packed.shape[4] == next(v for v in (1, 2, 4, 6, 8, 12, 16) if v >= B)
    # definition of the tile-major repack
    n = torch.arange(N)
    assert torch.equal(tiled[:, n // 16, :, n % 16].permute(1, 2, 0).cpu(), p)
    # definition of the packed repack: byte s of [tile][it][g][c][t] = byte g of word row 4 it + s
    pk = packed.cpu().view(torch.uint8).view(*packed.shape, 4)
    pb = torch.nn.functional.pad(p, (0, 0, 0, (-p.shape[1]) % 4)).contiguous().view(torch.uint8).view(B, -1, N, 4)
    for (t, i, nn_) in ((0, 0, 0), (B - 1, p.shape[1] - 1, N - 1), (B // 2, p.shape[1] // 2, N // 3)):
        for g in range(4):
            assert pk[nn_ // 16, i // 4, g, nn_ % 16, t, i % 4] == pb[t, i, nn_, g]
    base = bd.binary_linear(dev(a), dev(w), pd, dev(alpha), out_dtype=torch.float32)
    from bitdelta_amd import _lib
    for layout, m in ((("tile", tiled), ("packed", packed)) if B <= 8 else (("packed", packed),)):
        y32 = bd.binary_linear_decode(dev(a), dev(w), m, dev(alpha), layout=layout, out_dtype=torch.float32)
        assert _lib.lib().bd_last_gemm_variant() == 600                # one launch of the streaming kernel, also for 9 .. 16 tenants
        fro, mrel = relerr(y32.cpu(), ref32)
        assert fro <= 1e-5 and mrel <= 2e-5, (layout, fro, mrel)
        assert relerr(y32.cpu(), base.cpu())[0] <= 2e-6
        y16 = bd.binary_linear_decode(dev(a), dev(w), m, dev(alpha), layout=layout)
        ok, same = within_one_ulp(y16.cpu(), ref32.to(dtype), K)
        assert ok and same >= 0.99, (layout, ok, same)
        r = torch.randn(B, M, N).to(dtype)
        out = bd.binary_linear_decode(dev(a), dev(w), m, dev(alpha), layout=layout, residual=dev(r).clone())
        want = (r.float() + ref32).to(dtype)
        d = (out.cpu().float() - want.float()).abs()
        assert (d <= (r.float().abs() + ref32.abs()) * (2 ** -6 if dtype == torch.bfloat16 else 2 ** -9) + 1e-4).all()
        # canaries: ragged N / several tenants / 16-row chunks of the streaming kernel into poisoned margins -- same bits, nothing else touched
        for plain, od in ((y32, torch.float32), (y16, dtype)):
            cn = CanaryOut(B, M, N, od)
            bd.binary_linear_decode(dev(a), dev(w), m, dev(alpha), layout=layout, out_dtype=od, out=cn.view)
            assert torch.equal(cn.result(), plain), (layout, od)
            assert cn.untouched_outside(), ("output margin overwritten", layout, od)
        cn = CanaryOut(B, M, N, dtype)
        cn.view.copy_(dev(r))
        bd.binary_linear_decode(dev(a), dev(w), m, dev(alpha), layout=layout, residual=cn.view)
        assert torch.equal(cn.result(), out) and cn.untouched_outside(), ("residual form", layout)


@pytest.mark.gpu
def test_per_device_kernel_attributes_on_second_gpu(bd):
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are per-device properties: the one-pass fused GEMM (151 KB
    of LDS) and the streaming decode kernel (98 KB) must launch on cuda:1 after they have run on cuda:0 (round 1 raised the limit
    once per process, i.e. only on the first device).  Needs two visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    outs = []
    g = torch.Generator().manual_seed(7)                                      # same host data on every device
    x_h = torch.randn(1, 256, 512, generator=g).to(torch.bfloat16)
    w_h = (torch.randn(640, 512, generator=g) * 0.02).to(torch.bfloat16)
    m_h = torch.randint(-2**31, 2**31 - 1, (1, 16, 640), generator=g, dtype=torch.int64).to(torch.int32)
    for dev in ("cuda:0", "cuda:1", "cuda:0"):
        x, w, mask = x_h.to(dev), w_h.to(dev), m_h.to(dev)
        alpha = torch.full((1, 1), 3e-4, device=dev)
        y_big = bd.binary_linear(x, w, mask, alpha)                              # fused MFMA GEMM
        y_dec = bd.binary_linear(x[:, :1].contiguous(), w, mask, alpha)          # streaming decode kernel
        torch.cuda.synchronize(dev)
        outs.append((y_big.cpu(), y_dec.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[2][0])
    assert torch.equal(outs[0][1], outs[0][0][:, :1]) or (outs[0][1].float() - outs[0][0][:, :1].float()).abs().max() < 1e-2


def test_environment_override_of_the_variant_table(bd):
    """BD_GEMM_VARIANT in the environment = every thread's initial forced variant (SURVEY.md section 5 build note): a child process with
    BD_GEMM_VARIANT=20 runs a fused Linear the automatic rule would give to another tile, and reports the variant that ran."""
    import subprocess
    code = ("import torch, bitdelta_amd as bd; from bitdelta_amd import _lib; "
            "x = torch.randn(1, 256, 512, device='cuda').bfloat16(); w = (torch.randn(1024, 512, device='cuda') * 0.02).bfloat16(); "
            "p = torch.randint(-2**31, 2**31 - 1, (1, 16, 1024), device='cuda', dtype=torch.int64).to(torch.int32); "
            "bd.binary_linear(x, w, p, torch.full((1, 1), 4e-4, device='cuda')); print('variant', _lib.lib().bd_last_gemm_variant())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for v in (None, "20", "9"):
        env = dict(os.environ)
        env.pop("BD_GEMM_VARIANT", None)
        if v is not None:
            env["BD_GEMM_VARIANT"] = v
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        outs[v] = int(r.stdout.strip().split()[-1])
    assert outs["20"] == 20 and outs["9"] == 9 and outs[None] not in (-1,), outs
