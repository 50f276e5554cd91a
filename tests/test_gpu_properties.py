"""Property tests on the GPU (hypothesis): pack/unpack are inverse bijections on every shape the reference accepts, binarize agrees
with pack(sign) + mean|diff|, the delta GEMM is linear in its activations and odd in its signs.  Small shapes, many cases."""
import pytest
import torch

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bd():
    import bitdelta_amd
    from bitdelta_amd import _lib
    _lib.lib()
    return bitdelta_amd


@settings(max_examples=40, deadline=None)
@given(lead=st.lists(st.integers(1, 3), max_size=2), kw=st.integers(1, 5), n=st.integers(1, 70),
       n_bits=st.sampled_from([8, 16, 32, 64]), seed=st.integers(0, 2 ** 16))
def test_pack_unpack_roundtrip(bd, lead, kw, n, n_bits, seed):
    g = torch.Generator().manual_seed(seed)
    bits = (torch.rand(*lead, kw * n_bits, n, generator=g) > 0.5).cuda()
    p = bd.pack(bits, n_bits=n_bits)
    assert p.shape == (*lead, kw, n)
    assert torch.equal(bd.unpack(p, n_bits=n_bits), bits)
    # word value = sum(bit << j) in two's complement (reference binary_gemm_kernel.py:16-19, :23-32)
    b64 = bits.reshape(-1, kw, n_bits, n).to(torch.int64).cpu()
    want = (b64 << torch.arange(n_bits)[None, None, :, None]).sum(-2).to(p.dtype).reshape(*lead, kw, n)
    assert torch.equal(p.cpu(), want)


@settings(max_examples=25, deadline=None)
@given(n=st.integers(1, 130), kw=st.integers(1, 9), dtype=st.sampled_from([torch.bfloat16, torch.float16]), seed=st.integers(0, 2 ** 16))
def test_binarize_matches_definition(bd, n, kw, dtype, seed):
    from bitdelta_amd.diff import binarize
    g = torch.Generator().manual_seed(seed)
    base = (torch.randn(n, kw * 32, generator=g) * 0.02).to(dtype).cuda()
    fine = (base.float() + torch.randn(n, kw * 32, generator=g).cuda() * 5e-4).to(dtype)
    mask, coeff = binarize(base, fine)
    diff = fine - base                                           # in the weights' dtype, like diff.py:11
    assert torch.equal(mask, bd.pack((diff >= 0).T))             # zero -> bit 1 (diff.py:14-15)
    assert abs(coeff.item() - diff.float().abs().mean().item()) <= 1e-6 * max(diff.float().abs().mean().item(), 1e-12) + 1e-12


@settings(max_examples=20, deadline=None)
@given(b=st.integers(1, 3), m=st.integers(1, 70), kw=st.integers(1, 6), n=st.integers(1, 90), seed=st.integers(0, 2 ** 16),
       dtype=st.sampled_from([torch.bfloat16, torch.float16]))
def test_delta_bmm_linear_and_odd(bd, b, m, kw, n, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    K = kw * 32
    x = torch.randn(b, m, K, generator=g).to(dtype).cuda()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (b, kw, n), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    f = lambda xx, pp: bd.delta_bmm(xx, pp, out_dtype=torch.float32, round_mode=0)
    y = f(x, p)
    s = bd.unpack(p).float() * 2 - 1
    assert torch.allclose(y, torch.bmm(x.float(), s), rtol=1e-5, atol=1e-4 * K ** 0.5)
    assert torch.allclose(f(x, ~p), -y, rtol=1e-5, atol=1e-5)
    x2 = (x.float() * 2).to(dtype)
    assert torch.equal(f(x2, p), 2 * y)


@settings(max_examples=40, deadline=None)
@given(data=st.data(), m=st.integers(1, 16), shared=st.booleans(), n32=st.integers(1, 10), nit=st.integers(1, 11), seed=st.integers(0, 2 ** 16),
       dtype=st.sampled_from([torch.bfloat16, torch.float16]))
def test_rows_kernel_random_geometries(bd, data, m, shared, n32, nit, seed, dtype):
    """delta_rows_kernel (variant 800 + masks per block; bd_gemv_rows.h) over random (batch, rows per mask, shared / per-entry masks, 32-column
    units, 128-k iterations): equals unpack + fp32 bmm, is odd in its signs and exactly linear under a power-of-two scaling of the
    activations -- and every batch entry uses ITS mask"""
    from bitdelta_amd import _lib
    per = {800: 1, 801: 1, 802: 2, 804: 4}
    if shared:                                   # one mask for every row: at most 16 rows in the launch
        b, mc = data.draw(st.integers(1, 16 // m)), 800
    else:                                        # per-entry masks: masks per block x rows per mask <= 16
        mc = data.draw(st.sampled_from([v for v in per if per[v] * m <= 16]))
        b = data.draw(st.integers(2, max(2, 48 // m)))
    g = torch.Generator().manual_seed(seed)
    N, K = 32 * n32, 128 * nit
    if mc == 804:
        N = 64 * ((n32 + 1) // 2)                                       # (four masks per block: 64-column super-tiles only)
    x = torch.randn(b, m, K, generator=g).to(dtype).cuda()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 if shared else b, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    L = _lib.lib()
    L.bd_set_gemm_variant(mc)
    try:
        f = lambda xx, pp: bd.delta_bmm(xx, pp, out_dtype=torch.float32, round_mode=0)
        y = f(x, p)
        assert L.bd_last_gemm_variant() == 800
        s = bd.unpack(p).float() * 2 - 1
        assert torch.allclose(y, torch.matmul(x.float(), s), rtol=1e-5, atol=1e-4 * K ** 0.5)
        assert torch.allclose(f(x, ~p), -y, rtol=1e-5, atol=1e-5)
        assert torch.equal(f((x.float() * 2).to(dtype), p), 2 * y)
        perm = torch.randperm(b, generator=g).cuda()                      # entries and masks permuted together: the same rows come back
        assert torch.equal(f(x[perm].contiguous(), p if shared else p[perm].contiguous()), y[perm])
    finally:
        L.bd_set_gemm_variant(-1)


def test_prefill_attention_random_geometries():
    """bd_srv_prefill_attention over random (batch, length, heads, kv heads, padding, causal) against fp32 softmax attention"""
    import random
    from bitdelta_amd import serving_ops as ops
    rnd = random.Random(11)
    dev = "cuda"
    for case in range(24):
        B = rnd.choice([1, 2, 3, 5])
        S = 64 * rnd.randint(1, 9)
        KVH = rnd.choice([1, 2, 4])
        H = KVH * rnd.choice([1, 2, 4, 8])
        causal = rnd.random() < 0.75
        dtype = rnd.choice([torch.bfloat16, torch.float16])
        g = torch.Generator(device=dev).manual_seed(case)
        # separately allocated q / k / v with their own (padded) strides
        q = torch.randn(B, S, H * 128 + 64, device=dev, generator=g).to(dtype)[..., :H * 128].view(B, S, H, 128)
        k = (torch.randn(B, S, KVH * 128, device=dev, generator=g) * rnd.choice([0.3, 1.0, 3.0])).to(dtype).view(B, S, KVH, 128)
        v = torch.randn(B, S + 3, KVH * 128, device=dev, generator=g).to(dtype)[:, :S].view(B, S, KVH, 128)
        kv_start = None
        if rnd.random() < 0.6:
            kv_start = torch.tensor([rnd.randint(0, S - 1) for _ in range(B)], dtype=torch.int32, device=dev)
        assert ops.prefill_attention_supported(q, k, v)
        out = ops.prefill_attention(q, k, v, kv_start=kv_start, causal=causal).float().view(B, S, H, 128)
        G = H // KVH
        sc = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float().repeat_interleave(G, dim=2)) * 128 ** -0.5
        keys = torch.arange(S, device=dev)
        ok = torch.ones(B, 1, S, S, dtype=torch.bool, device=dev)
        if causal:
            ok &= (keys[None, :] <= keys[:, None])[None, None]
        if kv_start is not None:
            ok &= keys[None, None, None, :] >= kv_start.view(B, 1, 1, 1)
        p = torch.softmax(sc.masked_fill(~ok, float("-inf")), dim=-1).nan_to_num(0.0)
        ref = torch.einsum("bhqk,bkhd->bqhd", p, v.float().repeat_interleave(G, dim=2))
        assert torch.isfinite(out).all()
        err = (out - ref).abs().max().item()
        tol = 4e-2 if dtype == torch.bfloat16 else 5e-3
        assert err <= tol, (case, B, S, H, KVH, causal, dtype, err)


# ---- round 6 glue kernels: exact integer / copy semantics on random geometry
@settings(max_examples=30, deadline=None)
@given(T=st.integers(1, 7), S=st.integers(1, 40), KVH=st.sampled_from([1, 2, 4]), G=st.sampled_from([1, 2, 4]), pad=st.integers(0, 24),
       dtype=st.sampled_from([torch.bfloat16, torch.float16]), seed=st.integers(0, 2 ** 16), data=st.data())
def test_rope_kv_append_equals_rope_plus_cache_copies_on_random_geometry(bd, T, S, KVH, G, pad, dtype, seed, data):
    from bitdelta_amd import serving_ops as ops
    H = KVH * G
    Lc = S + data.draw(st.integers(0, 30))
    pos0 = data.draw(st.integers(0, Lc - S))
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = (H + 2 * KVH) * 128
    full = torch.randn(T, S, W + 8 * pad, device="cuda", generator=g).to(dtype)
    qkv = full[..., :W]
    f = torch.outer(torch.arange(Lc, device="cuda", dtype=torch.float32), 1.0 / (10000.0 ** (torch.arange(0, 128, 2, device="cuda") / 128.0)))
    emb = torch.cat([f, f], -1)
    cos = emb.cos().to(dtype).contiguous()
    sin = (emb.sin() * torch.cat([-torch.ones(64, device="cuda"), torch.ones(64, device="cuda")])).to(dtype).contiguous()
    kc = torch.randn(T, KVH, Lc, 128, device="cuda", generator=g).to(dtype)
    vc = torch.randn(T, KVH, Lc, 128, device="cuda", generator=g).to(dtype)
    ref_full, kc_ref, vc_ref = full.clone(), kc.clone(), vc.clone()
    ref = ref_full[..., :W]
    ops.rope_(ref[..., :(H + KVH) * 128], cos, sin, H + KVH, S, pos0)
    kc_ref[:, :, pos0:pos0 + S] = ref[..., H * 128:(H + KVH) * 128].reshape(T, S, KVH, 128).transpose(1, 2)
    vc_ref[:, :, pos0:pos0 + S] = ref[..., (H + KVH) * 128:].reshape(T, S, KVH, 128).transpose(1, 2)
    ops.rope_kv_append_(qkv, cos, sin, kc, vc, H, KVH, pos0)
    assert torch.equal(full, ref_full) and torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref)


@settings(max_examples=40, deadline=None)
@given(T=st.integers(1, 9), v8=st.integers(1, 700), dtype=st.sampled_from([torch.bfloat16, torch.float16]), seed=st.integers(0, 2 ** 16),
       plateau=st.booleans(), nans=st.integers(0, 3), pad=st.integers(0, 3))
def test_step_end_argmax_matches_torch_on_random_logits(bd, T, v8, dtype, seed, plateau, nans, pad):
    """the argmax of bd_srv_step_end == torch.argmax on coarse (many exact ties), plateau and NaN-bearing logits of any width V % 8 == 0, rows padded"""
    from bitdelta_amd import serving_ops as ops
    V = 8 * v8
    g = torch.Generator(device="cuda").manual_seed(seed)
    full = (torch.randn(T, V + 8 * pad, device="cuda", generator=g) * 3).round().to(dtype)      # integers in a small range: ties everywhere
    logits = full[:, :V]
    if plateau:
        logits[:, V // 3:] = logits.max()
    for i in range(nans):
        logits[i % T, int(torch.randint(0, V, (1,), generator=torch.Generator().manual_seed(seed + i)))] = float("nan")
    tok = torch.zeros(T, 1, dtype=torch.long, device="cuda")
    out = torch.zeros(T, 4, dtype=torch.long, device="cuda")
    step, pos = torch.tensor([2], device="cuda"), torch.tensor([17], device="cuda")
    stop_ids = torch.full((T, 1), -1, dtype=torch.long, device="cuda")
    stopped = torch.zeros(T, dtype=torch.bool, device="cuda")
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.step_end(logits, tok, out, step, pos, stop_ids, stopped, ticket)
    ref = torch.argmax(logits, dim=-1)
    assert torch.equal(tok[:, 0], ref) and torch.equal(out[:, 2], ref) and int(step) == 3 and int(pos) == 18 and int(ticket) == 0 and not stopped.any()
