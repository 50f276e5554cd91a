"""CPU tests of the serving-side sign-word repacks (host logic, exact integer work): the tile-major and the packed decode layouts of
include/bitdelta_hip.h are checked bit by bit against the reference bit order (bit j of word [i, n] <-> k = 32 i + j,
bitdelta/binary_gemm_kernel.py:109-111), through the oracle's unpack."""
import pytest
import torch

from bitdelta_amd.binary_gemm_kernel import pack_decode_masks, tile_masks, tile_weight
from bitdelta_amd.serving_loop import FusedDeltaLinear, padded_length


def rand_masks(T, K, N, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-2 ** 31, 2 ** 31 - 1, (T, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)


def bits_of(mask):
    """[T, K/32, N] int32 -> bool [T, K, N] in the reference bit order (little-endian within a word)"""
    T, KW, N = mask.shape
    j = torch.arange(32, dtype=torch.int64)
    b = (mask.to(torch.int64)[:, :, None, :] >> j[None, None, :, None]) & 1
    return b.reshape(T, KW * 32, N).bool()


@pytest.mark.parametrize("T,K,N", [(1, 128, 16), (3, 256, 40), (6, 384, 100), (8, 96, 33), (2, 4096, 64)])
def test_tile_major_layout(T, K, N):
    m = rand_masks(T, K, N, seed=K + N)
    t = tile_masks(m)
    assert t.shape == (T, (N + 15) // 16, K // 32, 16) and t.dtype == torch.int32 and t.is_contiguous()
    for n in range(N):
        assert torch.equal(t[:, n // 16, :, n % 16], m[:, :, n])
    if N % 16:
        assert int(t[:, -1, :, N % 16:].abs().sum()) == 0            # padding columns are zero words


@pytest.mark.parametrize("T,K,N", [(1, 128, 16), (3, 256, 40), (6, 384, 100), (5, 96, 33), (8, 160, 16), (2, 1024, 64), (9, 160, 33), (12, 256, 40),
                                   (16, 128, 16)])
def test_packed_decode_layout(T, K, N):
    """element [tile][it][g][c][t], byte s, bit e  ==  sign bit of k = 128 it + 32 s + 8 g + e of column 16 tile + c, tenant t;
    k past K, columns past N and tenants past T are zero"""
    m = rand_masks(T, K, N, seed=K * 3 + N)
    bits = bits_of(m)                                                # [T, K, N]
    p = pack_decode_masks(m)
    tp = next(v for v in (1, 2, 4, 6, 8, 12, 16) if v >= T)
    ntile, nit = (N + 15) // 16, (K + 127) // 128
    assert p.shape == (ntile, nit, 4, 16, tp) and p.dtype == torch.int32 and p.is_contiguous()
    by = p.view(torch.uint8).view(ntile, nit, 4, 16, tp, 4)          # little-endian: byte s of the dword
    e = torch.arange(8)
    for tile in range(ntile):
        for it in range(nit):
            for g in range(4):
                for s in range(4):
                    k0 = 128 * it + 32 * s + 8 * g
                    got = ((by[tile, it, g, :, :, s].to(torch.int64)[..., None] >> e) & 1).bool()          # [c, t, e]
                    want = torch.zeros(16, tp, 8, dtype=torch.bool)
                    cols = min(16, N - 16 * tile)
                    if k0 < K and cols > 0:
                        want[:cols, :T, :] = bits[:, k0:k0 + 8, 16 * tile:16 * tile + cols].permute(2, 0, 1)
                    assert torch.equal(got, want), (tile, it, g, s)


def test_fused_delta_linear_interleave_is_a_row_permutation():
    """gate|up interleaved in blocks of 8: stored row 16 j + c is gate row 8 j + c (c < 8) or up row 8 j + c - 8; split() undoes it"""
    T, K, inter = 2, 64, 24
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(inter, K, generator=g).half() for _ in range(2)]
    ms = [rand_masks(T, K, inter, seed=i) for i in range(2)]
    cs = [torch.rand(T, generator=g) for _ in range(2)]
    il = FusedDeltaLinear(ws, ms, cs, interleave8=True)
    plain = FusedDeltaLinear(ws, ms, cs)
    for j in range(inter // 8):
        assert torch.equal(il.weight[16 * j:16 * j + 8], ws[0][8 * j:8 * j + 8])
        assert torch.equal(il.weight[16 * j + 8:16 * j + 16], ws[1][8 * j:8 * j + 8])
        assert torch.equal(il.mask[:, :, 16 * j:16 * j + 8], ms[0][:, :, 8 * j:8 * j + 8])
        assert torch.equal(il.mask[:, :, 16 * j + 8:16 * j + 16], ms[1][:, :, 8 * j:8 * j + 8])
    y = torch.arange(2 * inter, dtype=torch.float32).expand(3, 1, 2 * inter)
    a, b = il.split(y)
    assert a.shape == (3, 1, inter) and torch.equal(a[0, 0, :8], torch.arange(8.)) and torch.equal(b[0, 0, :8], torch.arange(8., 16.))
    for t in range(T):
        ca = il.column_alpha(t)
        assert torch.allclose(ca[0:8], cs[0][t].expand(8)) and torch.allclose(ca[8:16], cs[1][t].expand(8))
        assert torch.allclose(plain.column_alpha(t)[:inter], cs[0][t].expand(inter))
    assert il.alpha_pair.shape == (T, 2) and plain.alpha_pair is None


def test_padded_length_rule():
    """demo/demo_backend.py:297-299: next power of two, at least 64"""
    assert [padded_length(n) for n in (1, 63, 64, 65, 128, 129, 1000, 1024)] == [64, 64, 64, 128, 128, 256, 1024, 1024]


@pytest.mark.parametrize("N,K", [(16, 128), (48, 384), (32, 1024)])
def test_tile_major_weight_layout(N, K):
    """W'[tile][it][s][c][g][e] = W[16 tile + c][128 it + 32 s + 8 g + e]; a pure permutation, returned flat as [N, K]"""
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K).half()            # distinct small integers are exact in fp16 up to 2048 ...
    w = (torch.arange(N * K) % 2039).float().reshape(N, K).half()                 # ... so use residues: still a strong permutation check
    t = tile_weight(w)
    assert t.shape == (N, K) and t.is_contiguous()
    v = t.view(N // 16, K // 128, 4, 16, 4, 8)
    for tile in range(N // 16):
        for it in range(K // 128):
            for s_ in range(4):
                for g in range(4):
                    k0 = 128 * it + 32 * s_ + 8 * g
                    assert torch.equal(v[tile, it, s_, :, g, :], w[16 * tile:16 * tile + 16, k0:k0 + 8])
    assert torch.equal(torch.sort(t.flatten())[0], torch.sort(w.flatten())[0])


def test_fused_norm_envelope_matches_the_kernel_side():
    """the Python envelope of the fused RMSNorm prologue must not admit what the kernel rejects -- packed layouts of 12 / 16
    tenants have no norm_w form (BD_PKL in csrc/bd_api.hip returns BD_E_BAD_SHAPE), so 9..16 tenants are outside it even at K = 2048."""
    from bitdelta_amd.binary_gemm_kernel import fused_norm_ok
    assert fused_norm_ok(8, 1, 2048) and fused_norm_ok(6, 1, 4096) and fused_norm_ok(1, 1, 8192)
    for B in range(9, 17):
        assert not fused_norm_ok(B, 1, 2048)
    assert not fused_norm_ok(6, 2, 4096) and not fused_norm_ok(6, 1, 6144) and not fused_norm_ok(8, 1, 8192)


def test_canary_helper_detects_writes_outside_the_output():
    """tests/canary.py on CPU tensors: a write one row past M, one column past N, into the next batch entry's margin or into the poisoned
    tail of a workspace must be reported; writes inside the output must not."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from canary import CanaryOut, canary_workspace
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        c = CanaryOut(2, 5, 7, dt, device="cpu", row_margin=3, col_margin=8)
        c.view.copy_(torch.randn(2, 5, 7).to(dt))
        assert c.untouched_outside() and c.result().shape == (2, 5, 7)
        for where in ((1, 3 + 5, 8), (1, 3, 8 + 7), (1, 2, 8), (0, 4, 9), (3, 3, 8)):       # below, right, above, previous / next batch entry
            c2 = CanaryOut(2, 5, 7, dt, device="cpu", row_margin=3, col_margin=8)
            c2.buf[where] = 0.5
            assert not c2.untouched_outside(), (dt, where)

    class Target:
        @staticmethod
        def workspace(nbytes, device, zeroed=False):
            raise AssertionError("the original allocator must not be called inside the block")
    with canary_workspace(Target) as (ok, made):
        buf, n = Target.workspace(1000, "cpu", zeroed=True)
        assert n == 1000 and bool((buf[:1000] == 0).all())
        buf[999] = 3
        assert ok()
        buf[1000] = 3
        assert not ok()
    assert Target.workspace.__qualname__.startswith("test_canary_helper")         # restored
