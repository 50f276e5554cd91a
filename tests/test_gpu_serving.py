"""GPU tests of the serving-side pieces: per-tenant dense weights, residual epilogue,
the opt-in differentiable delta term.  Checked against the REFERENCE EXPRESSIONS restated in plain torch fp32 on the device
(demo/demo_backend.py:62-79 for the per-tenant loop, bitdelta/diff.py:39 for the training composition) and the CPU oracle."""
import os
import sys
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bd():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import bitdelta_amd
    from bitdelta_amd import _lib
    _lib.lib()
    return bitdelta_amd


def relerr(a, ref):
    a, ref = a.double(), ref.double()
    return ((a - ref).norm() / ref.norm()).item()


def reference_loop(module, weight_list, hidden_states):
    """DataParallelModule.forward exactly as the reference writes it (demo/demo_backend.py:69-79)."""
    outputs = []
    for i in range(len(weight_list)):
        module.weight.data = weight_list[i]
        outputs.append(module(hidden_states[i, None]))
    nt = torch.nested.as_nested_tensor([outputs[i][0] for i in range(len(outputs))])
    return torch.nested.to_padded_tensor(nt, torch.finfo(nt.dtype).min)


class RMSNorm(nn.Module):                      # the HF Llama / Mistral norm (fp32 internals, weight * normalised)
    def __init__(self, dim, eps=1e-5, dtype=torch.float16):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype))
        self.variance_epsilon = eps

    def forward(self, x):
        dt = x.dtype
        v = x.to(torch.float32)
        v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * v.to(dt)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tenant_linear_vs_oracle_and_bmm(bd, oracle, dtype):
    g = torch.Generator().manual_seed(3)
    for T, M, N, K in ((6, 1, 32000, 4096), (4, 1, 1000, 256), (3, 5, 520, 1184), (2, 16, 640, 64)):
        x = torch.randn(T, M, K, generator=g).to(dtype)
        w = (torch.randn(T, N, K, generator=g) * 0.02).to(dtype)
        y = bd.tenant_linear(x.cuda(), w.cuda())
        y32 = bd.tenant_linear(x.cuda(), w.cuda(), out_dtype=torch.float32)
        assert y.shape == (T, M, N) and y.dtype == dtype
        ref = torch.bmm(x.double(), w.double().transpose(1, 2))
        assert relerr(y32.cpu(), ref) <= 1e-5
        d = (y.float().cpu() - ref.to(dtype).float()).abs()
        assert (d <= ref.abs().float() * 2 ** -7 + 1e-4).all()
    # prefill shapes go to torch.bmm
    x = torch.randn(3, 40, 256, generator=g).half().cuda()
    w = torch.randn(3, 100, 256, generator=g).half().cuda()
    assert torch.equal(bd.tenant_linear(x, w), torch.bmm(x, w.transpose(1, 2)))


def test_data_parallel_module_matches_reference_loop(bd):
    """embedding / norm / lm_head with per-tenant weights: batched implementation == the reference's weight-swapping loop"""
    from bitdelta_amd.serving import DataParallelModule
    torch.manual_seed(5)
    T, S, hid, vocab = 4, 7, 256, 1000
    dev = "cuda"
    # embedding: exact
    emb = nn.Embedding(vocab, hid).half().to(dev)
    ws = [torch.randn(vocab, hid, device=dev).half() for _ in range(T)]
    ids = torch.randint(0, vocab, (T, S), device=dev)
    got = DataParallelModule(emb, ws)(ids)
    assert torch.equal(got, reference_loop(emb, ws, ids))
    # RMSNorm: bit-identical (unit-weight evaluation times the stacked weights)
    norm = RMSNorm(hid).to(dev)
    nws = [(torch.rand(hid, device=dev) + 0.5).half() for _ in range(T)]
    h = torch.randn(T, S, hid, device=dev).half()
    m = DataParallelModule(norm, nws)
    assert m.kind == "scale"
    got = m(h)
    assert norm.weight.data.data_ptr() == m.original_weight.data_ptr()          # the base weight is back in place
    assert torch.equal(got, reference_loop(norm, nws, h))
    # lm_head at decode (HIP kernel) and prefill (bmm): same values up to fp32 accumulation order
    head = nn.Linear(hid, vocab, bias=False).half().to(dev)
    hws = [(torch.randn(vocab, hid, device=dev) * 0.05).half() for _ in range(T)]
    for s in (1, S):
        x = torch.randn(T, s, hid, device=dev).half()
        m = DataParallelModule(head, hws)
        assert m.kind == "linear"
        got, want = m(x), reference_loop(head, hws, x)
        assert got.shape == want.shape
        assert torch.allclose(got.float(), want.float(), rtol=2 ** -9, atol=2e-3)
    # ragged vocabularies: padded outputs are finfo.min, exactly like the reference's nested-tensor padding
    hws2 = [hws[0], hws[1][:900].contiguous(), hws[2], hws[3][:950].contiguous()]
    x = torch.randn(T, 1, hid, device=dev).half()
    got, want = DataParallelModule(head, hws2)(x), reference_loop(head, hws2, x)
    assert got.shape == want.shape == (T, 1, vocab)
    assert torch.equal(got == torch.finfo(torch.float16).min, want == torch.finfo(torch.float16).min)
    assert torch.allclose(got.float(), want.float(), rtol=2 ** -9, atol=2e-3)
    # a leaf this class does not know runs the reference loop
    ln = nn.LayerNorm(hid).half().to(dev)
    lws = [(torch.rand(hid, device=dev) + 0.5).half() for _ in range(T)]
    m = DataParallelModule(ln, lws)
    assert m.kind == "loop" and torch.equal(m(h), reference_loop(ln, lws, h))

    # 1-D-weight leaves whose last step is NOT `weight * normed` must not take the scale fast path: a Gemma-style norm
    # (normed * (1 + weight)) is kept off it by name, an RMSNorm-named module with the same arithmetic is caught by the first-call probe
    class GemmaRMSNorm(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.weight = nn.Parameter(torch.zeros(d))

        def forward(self, x):
            v = x.float()
            v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6)
            return (v * (1.0 + self.weight.float())).to(x.dtype)

    class SneakyRMSNorm(GemmaRMSNorm):
        pass

    for cls, kind0 in ((GemmaRMSNorm, "loop"), (SneakyRMSNorm, "scale")):
        gn = cls(hid).half().to(dev)
        m = DataParallelModule(gn, nws)
        assert m.kind == kind0
        assert torch.equal(m(h), reference_loop(gn, nws, h))
        assert m.kind == "loop"
        assert torch.equal(m(h), reference_loop(gn, nws, h))
    pr = nn.PReLU(hid).half().to(dev)
    m = DataParallelModule(pr, nws)
    assert m.kind == "loop"


def test_binary_linear_residual_epilogue(bd, oracle):
    g = torch.Generator().manual_seed(7)
    # M <= 16: decode kernels; M > 16 with K % 64 == 0: the one-pass fused GEMM's epilogue (64-row, 128-row and 256-row tiles,
    # ragged M / N, bf16-free fp16 here; bf16 below); K % 64 != 0: caller-side add
    # (6, 64, 4096, 512) and (5, 33, 2048, 264): pair tiles + split-k, the residual is added by the reduce launch
    for B, M, K, N, T in ((6, 1, 1024, 1000, 6), (2, 3, 512, 520, 1), (2, 40, 256, 520, 2), (6, 64, 512, 640, 6), (1, 300, 256, 264, 1),
                          (1, 130, 128, 136, 1), (1, 600, 1024, 384, 1), (1, 40, 96, 136, 1), (6, 64, 4096, 512, 6), (5, 33, 2048, 264, 5)):
        a = torch.randn(B, M, K, generator=g).half()
        p = torch.randint(-2 ** 31, 2 ** 31 - 1, (T, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
        w = (torch.randn(N, K, generator=g) * 0.02).half()
        alpha = (torch.rand(T, 1, generator=g) * 2e-4 + 3e-4).float()
        res = torch.randn(B, M, N, generator=g).half()
        r = res.cuda().clone()
        out = bd.binary_linear(a.cuda(), w.cuda(), p.cuda(), alpha.cuda(), residual=r)
        assert out.data_ptr() == r.data_ptr()
        y32 = oracle.binary_linear(a, w, p, alpha, out_dtype=torch.float32)
        if M <= 16:        # fused epilogue: one rounding of residual + y
            want = (res.float() + y32).half()
        else:              # caller-side add: y rounded, then the sum rounded
            want = (res + y32.half())
        d = (out.cpu().float() - want.float()).abs()
        # one fp16 ulp at the magnitude of the larger addend (the sum may cancel to something much smaller than its terms)
        assert (d <= (res.float().abs() + y32.abs()) * 2 ** -10 + 1e-4).all()
        if M > 16 and K % 64 == 0:     # the GEMM epilogue reproduces the separate ops exactly where the Linear output is bit-equal
            y16 = bd.binary_linear(a.cuda(), w.cuda(), p.cuda(), alpha.cuda())
            assert torch.equal(out, res.cuda() + y16)


def test_differentiable_delta_term(bd):
    """Opt-in full gradient: dx = g.W + coeff * g.S^T, dcoeff = sum(g * x.S); compared with autograd through the
    dense fp32 expression x @ (W^T + coeff * S); and the default path still reproduces the reference composition's gradients."""
    torch.manual_seed(9)
    N, K, M = 96, 128, 10
    base = (torch.randn(N, K) * 0.05).bfloat16().cuda()
    fine = (base.float() + torch.randn(N, K, device="cuda") * 2e-3).bfloat16()
    mod = bd.BinaryDiff(base, fine)
    S = bd.unpack(mod.mask).float() * 2 - 1                                   # [K, N]
    x = torch.randn(2, M // 2, K, device="cuda").bfloat16()
    gout = torch.randn(2, M // 2, N, device="cuda").bfloat16()

    # dense fp32 reference of the full gradient
    xr = x.float().clone().requires_grad_(True)
    cr = mod.coeff.detach().clone().requires_grad_(True)
    yr = xr @ (base.float().T + cr * S)
    yr.backward(gout.float())

    mod.delta_input_grad = True
    xg = x.clone().requires_grad_(True)
    mod.coeff.grad = None
    y = mod(xg)
    assert y.requires_grad and torch.allclose(y.float(), yr.detach(), rtol=2 ** -7, atol=2e-3)
    y.backward(gout)
    assert relerr(xg.grad.float(), xr.grad) <= 8e-3                           # bf16 storage of dx
    assert abs(mod.coeff.grad.item() - cr.grad.item()) <= 2e-2 * abs(cr.grad.item()) + 1e-3
    # the delta part of dx is really there: without it the error is the size of coeff * g.S^T
    no_delta = (gout.float() @ base.float())
    assert relerr(xg.grad.float(), xr.grad) < 0.2 * relerr(no_delta, xr.grad)

    # default: reference composition -- same dcoeff, dx WITHOUT the delta term (the reference's quirk)
    mod.delta_input_grad = False
    xq = x.clone().requires_grad_(True)
    mod.coeff.grad = None
    mod(xq).backward(gout)
    assert relerr(xq.grad.float(), no_delta) <= 8e-3
    assert abs(mod.coeff.grad.item() - cr.grad.item()) <= 3e-2 * abs(cr.grad.item()) + 1e-3

    # ... and that default path is ONE HIP launch in forward (the fused kernel through an autograd.Function) whose gradients are
    # IDENTICAL, bit for bit, to those of the reference's four-launch composition (x @ base + coeff * binary_bmm(x, mask))
    g_fused = (xq.grad.clone(), mod.coeff.grad.clone())
    mod.reference_composition = True
    try:
        xc = x.clone().requires_grad_(True)
        mod.coeff.grad = None
        yc = mod(xc)
        yc.backward(gout)
        assert torch.equal(xc.grad, g_fused[0]) and torch.equal(mod.coeff.grad, g_fused[1])
    finally:
        mod.reference_composition = False
    from torch.profiler import profile, ProfilerActivity
    xq2 = x.clone().requires_grad_(True)
    kernels = None
    try:
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            yq = mod(xq2)
            torch.cuda.synchronize()
        kernels = [e.key for e in prof.key_averages() if getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) > 0]
    except Exception:                      # no kernel tracer in this build: the grad_fn check below still pins the path
        yq = mod(xq2)
    if kernels:
        assert len(kernels) == 1 and "bd::" in kernels[0], kernels   # one launch, and it is this library's kernel
    fn = yq.grad_fn                                  # (the module reshapes the kernel output back to x's leading dims)
    while fn is not None and "View" in type(fn).__name__:
        fn = fn.next_functions[0][0]
    assert type(fn).__name__ == "_RefLinearFnBackward", type(fn).__name__
    # forward values: single rounding of the fused kernel vs the composition's four -- both within bf16 rounding of the fp32 truth
    assert torch.allclose(yq.float(), yr.detach(), rtol=2 ** -7, atol=2e-3) and torch.allclose(yc.float(), yr.detach(), rtol=2 ** -6, atol=4e-3)

    # the lazily built W^T / S^T copies of the opt-in backward follow their sources: an in-place update of the buffers
    # must rebuild them
    mod.delta_input_grad = True
    mt0, wk0 = mod._transposed_mask(), mod._transposed_weight()
    assert mod._transposed_mask() is mt0 and mod._transposed_weight() is wk0          # cached while nothing changed
    with torch.no_grad():
        mod.mask.bitwise_not_()
        mod.base.mul_(2)
    assert mod._transposed_mask() is not mt0 and torch.equal(mod._transposed_mask(), ~mt0)
    assert mod._transposed_weight() is not wk0 and torch.equal(mod._transposed_weight(), wk0 * 2)
    with torch.no_grad():
        mod.mask.bitwise_not_()
        mod.base.div_(2)
    mod.delta_input_grad = False

    # finite difference on coeff through the opt-in path (fp32 output of the kernel)
    with torch.no_grad():
        c0 = mod.coeff.item()
        eps = 1e-3
        f = lambda c: (bd.binary_linear(x.reshape(1, -1, K), mod._weight_nk(), mod.mask[None], torch.tensor([[c]], device="cuda"),
                                        out_dtype=torch.float32).reshape(2, M // 2, N) * gout.float()).sum().item()
        fd = (f(c0 + eps) - f(c0 - eps)) / (2 * eps)
    assert abs(fd - cr.grad.item()) <= 1e-2 * abs(cr.grad.item()) + 1e-3


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_delta_linear_backward_at_baseline_shapes(bd, N, K):
    """dx and dcoeff of the opt-in differentiable op at the Llama-2-7B projection shapes, M = 512 (4 x 128 tokens, the reference's
    distillation batch): forward through the fused kernel, dx through the SAME kernel on (W^T, S^T) in one launch, dcoeff through the
    delta GEMM -- against stock autograd on the dense fp32 expression x @ (W^T + coeff * S)."""
    from bitdelta_amd import _lib
    torch.manual_seed(N + K)
    M = 512
    base = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    fine = (base.float() + torch.randn(N, K, device="cuda") * 5e-4).bfloat16()
    mod = bd.BinaryDiff(base, fine)
    mod.delta_input_grad = True
    S = bd.unpack(mod.mask).float() * 2 - 1                                   # [K, N]
    x = torch.randn(4, M // 4, K, device="cuda").bfloat16()
    gout = (torch.randn(4, M // 4, N, device="cuda") * 0.1).bfloat16()
    xr = x.float().clone().requires_grad_(True)
    cr = mod.coeff.detach().clone().requires_grad_(True)
    yr = xr @ (base.float().T + cr * S)
    yr.backward(gout.float())
    xg = x.clone().requires_grad_(True)
    y = mod(xg)
    assert relerr(y.float(), yr.detach()) <= 3e-3                              # bf16 output rounding
    y.backward(gout)
    assert _lib.lib().bd_last_gemm_variant() in (0, 1, 5, 8, 9, 10, 13, 14)   # an MFMA tile kernel ran the last backward GEMM
    assert relerr(xg.grad.float(), xr.grad) <= 4e-3                            # bf16 storage of dx
    assert abs(mod.coeff.grad.item() - cr.grad.item()) <= 5e-3 * abs(cr.grad.item()) + 1e-2 * gout.float().abs().mean().item()
    # the delta part of dx is there: dropping it leaves an error of the size of coeff * g.S^T
    no_delta = gout.float() @ base.float()
    assert relerr(xg.grad.float(), xr.grad) < 0.5 * relerr(no_delta, xr.grad)
    del S, yr, xr


def test_scale_distillation_step_matches_dense_fp32(bd):
    """tools/distill_step.py's loop (reference train.py:60-88: AdamW over the compressed model's parameters, MSE on logits, batch 4 x
    128) on a small synthetic Llama pair: the student whose 14 BinaryDiff modules run forward and backward through the HIP ops follows
    the loss trace of its dense fp32 twin under stock autograd, with and without the delta term in dx."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import copy
    import distill_step as ds
    from bitdelta_amd.diff import BinaryDiff
    steps = 8
    keep = BinaryDiff.delta_input_grad
    try:
        for full in (True, False):
            BinaryDiff.delta_input_grad = full
            base, fine, student = ds.build(256, 704, 2, 4, 512, "cuda", torch.bfloat16, seed=3, sigma=2e-3)
            assert sum(isinstance(m, BinaryDiff) for m in student.modules()) == 14
            g = torch.Generator(device="cuda").manual_seed(1)
            batches = [torch.randint(0, 512, (4, 128), device="cuda", generator=g) for _ in range(steps)]
            twin = ds.dense_twin(student)
            c0 = [m.coeff.item() for m in student.modules() if isinstance(m, BinaryDiff)]
            losses, _ = ds.run(student, fine, batches, 1e-4, steps)
            l32, _ = ds.run(twin, copy.deepcopy(fine).float(), batches, 1e-4, steps)
            c1 = [m.coeff.item() for m in student.modules() if isinstance(m, BinaryDiff)]
            assert all(torch.isfinite(torch.tensor(losses))) and any(abs(a - b) > 0 for a, b in zip(c0, c1))       # the coeffs trained
            rel = [abs(a - b) / b for a, b in zip(losses, l32)]
            # bf16 activations / logits vs the fp32 twin: the traces agree to a few per cent at every step (the default path drops
            # d/dx of the delta term, as the reference does: its parameters drift from the twin's, slowly, so it gets the wider gate)
            assert max(rel) <= (0.08 if full else 0.15), (full, losses, l32)
    finally:
        BinaryDiff.delta_input_grad = keep


# ------------------------------------------------------------------------------------------------ serving loop (section 8f row 3)
def _dense_reference_logits(bd, dec, t, ids, am):
    """Tenant t's model as ONE dense fp32 network (delta merged into the weights, W + alpha * S^T), no KV cache, HF-style
    semantics: raw positions, causal mask, left pads masked as keys.  ids / am: 1-D tensors (full padded sequence so far)."""
    import torch.nn.functional as F
    from bitdelta_amd.serving_loop import _rope
    hid, inter, nl, heads, kvh, vocab = dec.cfg
    hd = dec.hd
    L = ids.shape[0]

    def merged(fl):
        S = (bd.unpack(fl.mask[t]).float() * 2 - 1).T                              # [N, K]
        a = fl.column_alpha(t)                                                      # [N]
        return fl.weight.float() + a[:, None] * S

    def rms(x, w):
        v = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + dec.eps)
        return v * w.float()

    cos, sin = dec.cos[:L].float(), dec.sin[:L].float()
    x = dec.embed[t][ids].float()                                                  # [L, hid]
    keymask = am[None, :] & torch.ones(L, L, dtype=torch.bool, device=ids.device).tril()
    keymask = keymask | torch.eye(L, dtype=torch.bool, device=ids.device)
    for layer in dec.layers:
        h = rms(x, layer.norm1[t])
        qkv = h @ merged(layer.qkv).T
        q, k, v = layer.qkv.split(qkv)
        q = _rope(q.view(L, heads, hd).transpose(0, 1)[None], cos, sin)[0]         # [H, L, hd]
        k = _rope(k.view(L, kvh, hd).transpose(0, 1)[None], cos, sin)[0]
        v = v.view(L, kvh, hd).transpose(0, 1)
        rep = heads // kvh
        k, v = k.repeat_interleave(rep, 0), v.repeat_interleave(rep, 0)
        s = (q @ k.transpose(1, 2)) / hd ** 0.5
        s = s.masked_fill(~keymask[None], float("-inf"))
        a = (s.softmax(-1) @ v).transpose(0, 1).reshape(L, heads * hd)
        x = x + a @ merged(layer.o).T
        h = rms(x, layer.norm2[t])
        g, u = layer.gate_up.split(h @ merged(layer.gate_up).T)
        x = x + (F.silu(g) * u) @ merged(layer.down).T
    return rms(x[-1], dec.final_norm[t]) @ dec.lm_head[t].float().T


def test_serving_loop_full_width_layer_matches_dense_models(bd):
    """The same comparison at FULL WIDTH (loop-level parity was tiny-sized only): one Mistral-7B layer (hidden 4096, 32 / 8
    heads, intermediate 14336), 6 tenants -- the BASELINE configs[2] launches: packed decode kernels with q|k|v (G = 6) and gate|up ->
    SwiGLU fused, decode attention with 4 query heads per kv head, the 64-row multi-tenant prefill tiles -- prefill logits and three
    teacher-forced decode steps against 6 independent dense fp32 models (delta merged into the weights)."""
    from bitdelta_amd.serving_loop import TenantDecoder
    T = 6
    dec = TenantDecoder.synthetic("mistral-1layer", T, "cuda", dtype=torch.float16, seed=21, max_len=128)
    g = torch.Generator().manual_seed(2)
    prompts = [torch.randint(1, 512, (n,), generator=g).tolist() for n in (9, 64, 33, 50, 17, 60)]
    ids, am = dec.prepare(prompts)
    assert ids.shape == (T, 64)
    cache = dec.new_cache()
    lg = dec.prefill(ids, am, cache)
    for t in range(T):
        ref = _dense_reference_logits(bd, dec, t, ids[t], am[t])
        # one fp16 layer: every stored activation carries <= 2^-11 relative rounding, the kernels themselves are at 1e-5 / 1 ulp; the measured
        # aggregate is ~1e-3.  3e-3 (round 6; 2e-2 before) would expose a wrong scale group or an off-by-one position of small magnitude.
        assert relerr(lg[t].float(), ref) <= 3e-3, (t, relerr(lg[t].float(), ref))
    steps = 3
    toks, n = dec.generate(prompts, max_new_tokens=steps, use_graph=True)
    toks_e, _ = dec.generate(prompts, max_new_tokens=steps, use_graph=False)
    assert n == steps and torch.equal(toks, toks_e)
    for t in range(T):
        seq, msk = ids[t].clone(), am[t].clone()
        for s_ in range(steps):
            ref = _dense_reference_logits(bd, dec, t, seq, msk)
            top2 = ref.topk(2).values
            tok = int(toks[t, s_])
            if (top2[0] - top2[1]).item() > 0.05:
                assert tok == int(ref.argmax()), (t, s_)
            else:
                assert ref[tok] >= top2[0] - 0.1
            seq = torch.cat([seq, torch.tensor([tok], device=seq.device)])
            msk = torch.cat([msk, torch.tensor([True], device=seq.device)])


def test_serving_loop_matches_dense_per_tenant_models(bd):
    """Left-pad to a power of two >= 64, prefill, greedy argmax feedback with the KV cache, per-tenant embedding / norms / lm_head,
    fused q+k+v / gate+up launches, hipGraph replay -- against T independent dense fp32 models fed the same tokens."""
    from bitdelta_amd.serving_loop import TenantDecoder, padded_length
    assert padded_length(5) == 64 and padded_length(64) == 64 and padded_length(65) == 128 and padded_length(1000) == 1024
    T = 3
    dec = TenantDecoder.synthetic("tiny", T, "cuda", dtype=torch.float16, seed=11, max_len=128)
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(1, 512, (n,), generator=g).tolist() for n in (5, 37, 64)]
    with pytest.raises(ValueError):
        dec.prepare([list(range(1, 1100))] * T)
    ids, am = dec.prepare(prompts)
    assert ids.shape == (T, 64) and int(am[0].sum()) == 5 and bool(am[0, -5:].all()) and int(ids[0, :59].abs().sum()) == 0
    steps = 6
    toks_graph, n1 = dec.generate(prompts, max_new_tokens=steps, use_graph=True)
    toks_eager, n2 = dec.generate(prompts, max_new_tokens=steps, use_graph=False)
    assert n1 == n2 == steps and torch.equal(toks_graph, toks_eager)          # graph replay == eager, token for token
    # teacher-forced comparison with the dense per-tenant models
    for t in range(T):
        seq, msk = ids[t].clone(), am[t].clone()
        for s in range(steps):
            ref = _dense_reference_logits(bd, dec, t, seq, msk)
            top2 = ref.topk(2).values
            margin = (top2[0] - top2[1]).item()
            tok = int(toks_graph[t, s])
            if margin > 0.05:                                                  # beyond fp16 noise the greedy choice must agree
                assert tok == int(ref.argmax()), (t, s, margin)
            else:
                assert ref[tok] >= top2[0] - 0.1
            seq = torch.cat([seq, torch.tensor([tok], device=seq.device)])
            msk = torch.cat([msk, torch.tensor([True], device=seq.device)])
    # logits of the prefill themselves
    cache = dec.new_cache()
    lg = dec.prefill(ids, am, cache)
    for t in range(T):
        ref = _dense_reference_logits(bd, dec, t, ids[t], am[t])
        assert relerr(lg[t].float(), ref) <= 3e-3, (t, relerr(lg[t].float(), ref))          # two fp16 layers (gate was 2e-2 until round 6)
    # stop tokens: generation ends once every tenant has produced one
    stop = [[int(toks_graph[t, 1])] for t in range(T)]
    toks_s, n = dec.generate(prompts, max_new_tokens=steps, stop_token_ids=stop, use_graph=False)
    assert n == 2 and torch.equal(toks_s, toks_graph[:, :2])


def test_decode_glue_kernels_vs_torch_ops(bd):
    """per-tenant RMSNorm, SwiGLU and single-token attention (RoPE + cache append + GQA softmax) against the stock torch ops they
    replace in the decode step"""
    import torch.nn.functional as F
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.serving_loop import _rope, _rope_tables
    torch.manual_seed(13)
    dev = "cuda"
    for dtype in (torch.float16, torch.bfloat16):
        T, M, H = 6, 1, 4096
        x = torch.randn(T, M, H, device=dev).to(dtype)
        w = (1 + 0.1 * torch.randn(T, H, device=dev)).to(dtype)
        want = F.rms_norm(x, (H,), None, 1e-5) * w[:, None, :]
        got = ops.rmsnorm_tenant(x, w, 1e-5)
        assert torch.allclose(got.float(), want.float(), rtol=2 ** -7 if dtype == torch.bfloat16 else 2 ** -10, atol=1e-3)
        I = 1024
        gu = torch.randn(T, M, 2 * I, device=dev).to(dtype)
        want = F.silu(gu[..., :I]) * gu[..., I:]
        got = ops.swiglu(gu, I)
        assert torch.allclose(got.float(), want.float(), rtol=2 ** -7 if dtype == torch.bfloat16 else 2 ** -9, atol=1e-3)
        # fused in-place RoPE (prefill): bit-identical to the stock composition torch.addcmul(x * cos, rotate_half(x), sin)
        Bq, Sq, Hq = 2, 37, 5
        cos_t, sin_t = _rope_tables(64, 128, dev, dtype)
        xq = torch.randn(Bq, Sq, Hq * 128, device=dev).to(dtype)
        want = _rope(xq.view(Bq, Sq, Hq, 128).transpose(1, 2), cos_t[3:3 + Sq], sin_t[3:3 + Sq]).transpose(1, 2).reshape(Bq, Sq, Hq * 128)
        got = ops.rope_(xq.clone(), cos_t, sin_t, Hq, Sq, 3)
        assert torch.equal(got, want)
        g2, u2 = torch.randn(2, 9, 264, device=dev).to(dtype), torch.randn(2, 9, 264, device=dev).to(dtype)
        assert torch.allclose(ops.swiglu2(g2, u2).float(), (F.silu(g2) * u2).float(), rtol=2 ** -7 if dtype == torch.bfloat16 else 2 ** -9, atol=1e-3)
        # G = 4 (Mistral-style GQA) and G = 1 (Llama-2-7B-style MHA); short cache (one block per kv head) and long cache (key range
        # split over 4 blocks + combine launch), position in the first / a middle / the last split
        for heads, kvh, Lc, pos in ((8, 2, 96, 70), (4, 4, 96, 70), (8, 2, 320, 300), (8, 2, 1088, 40), (4, 4, 512, 511), (8, 2, 640, 129)):
            hd = 128
            cos, sin = _rope_tables(Lc, hd, dev, dtype)
            kc = torch.randn(T, kvh, Lc, hd, device=dev).to(dtype)
            vc = torch.randn(T, kvh, Lc, hd, device=dev).to(dtype)
            valid = torch.zeros(T, Lc, dtype=torch.bool, device=dev)
            for t in range(T):
                valid[t, 5 * t:pos] = True                               # left padding of different lengths
            qkv = torch.randn(T, 1, (heads + 2 * kvh) * hd, device=dev).to(dtype)
            # torch reference on copies
            kr, vr, vm = kc.clone(), vc.clone(), valid.clone()
            q, k, v = qkv.split([heads * hd, kvh * hd, kvh * hd], dim=-1)
            pidx = torch.tensor([pos], device=dev)
            q = _rope(q.view(T, 1, heads, hd).transpose(1, 2), cos[pidx], sin[pidx])
            k = _rope(k.view(T, 1, kvh, hd).transpose(1, 2), cos[pidx], sin[pidx])
            kr.index_copy_(2, pidx, k)
            vr.index_copy_(2, pidx, v.view(T, 1, kvh, hd).transpose(1, 2))
            vm[:, pos] = True
            rep = heads // kvh
            s = (q.float() @ kr.float().repeat_interleave(rep, 1).transpose(2, 3)) / hd ** 0.5
            s = s.masked_fill(~vm[:, None, None, :], float("-inf"))
            want = (s.softmax(-1) @ vr.float().repeat_interleave(rep, 1)).transpose(1, 2).reshape(T, 1, heads * hd)
            got = ops.decode_attention(qkv, cos, sin, kc, vc, valid, pidx, heads, kvh)
            assert relerr(got.float(), want) <= (1e-2 if dtype == torch.bfloat16 else 2e-3)
            assert torch.equal(kc, kr) and torch.equal(vc, vr) and torch.equal(valid, vm)     # cache append is bit-exact


def test_decode_attention_split_merge_is_deterministic(bd):
    """The in-launch merge of the key-range splits hands partial (acc, max, sum) triples from four blocks to the block that arrives last,
    through agent-scope stores / a ticket / agent-scope loads (csrc/bd_serving.h).  A visibility race in that hand-over would show up as
    run-to-run differences: 3000 launches on the decode step's geometry (6 tenants, 32 / 8 heads, 512 keys) must all be bit-identical."""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.serving_loop import _rope_tables
    torch.manual_seed(5)
    dev, dtype, T, heads, kvh, hd, Lc, pos = "cuda", torch.float16, 6, 32, 8, 128, 576, 512
    cos, sin = _rope_tables(Lc, hd, dev, dtype)
    kc = torch.randn(T, kvh, Lc, hd, device=dev).to(dtype)
    vc = torch.randn(T, kvh, Lc, hd, device=dev).to(dtype)
    valid = torch.zeros(T, Lc, dtype=torch.bool, device=dev)
    valid[:, :pos] = True
    qkv = torch.randn(T, 1, (heads + 2 * kvh) * hd, device=dev).to(dtype)
    pidx = torch.tensor([pos], device=dev)
    first = ops.decode_attention(qkv, cos, sin, kc, vc, valid, pidx, heads, kvh).clone()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(3000):
        got = ops.decode_attention(qkv, cos, sin, kc, vc, valid, pidx, heads, kvh)
        bad += (got != first).sum()
    assert int(bad) == 0


def test_decode_attention_uses_the_largest_split_count_a_small_workspace_holds(bd):
    """bd_srv_decode_attention_workspace_bytes grew 4x in round 5 (sized for 16 splits); a caller that still hands over a
    round-4-sized scratch (4 splits) -- or anything between -- must get the key-range split that FITS, not the unsplit launch.  One Mistral
    sequence (8 kv heads, 2048 keys: the rule picks 16 splits): full / half / round-4-sized / ticket-only workspaces all agree with the full one
    up to the merge order."""
    from bitdelta_amd import _lib
    from bitdelta_amd._lib import DTYPE_CODE, check, ptr, stream_ptr
    from bitdelta_amd.serving_loop import _rope_tables
    torch.manual_seed(6)
    dev, dtype, T, heads, kvh, hd, Lc, pos = "cuda", torch.float16, 1, 32, 8, 128, 2112, 2048
    cos, sin = _rope_tables(Lc, hd, dev, dtype)
    kc0 = torch.randn(T, kvh, Lc, hd, device=dev).to(dtype)
    vc0 = torch.randn(T, kvh, Lc, hd, device=dev).to(dtype)
    valid = torch.zeros(T, Lc, dtype=torch.bool, device=dev)
    valid[:, :pos] = True
    qkv = torch.randn(T, 1, (heads + 2 * kvh) * hd, device=dev).to(dtype)
    pidx = torch.tensor([pos], device=dev)
    L = _lib.lib()
    need = L.bd_srv_decode_attention_workspace_bytes(T, heads, kvh, hd, Lc)
    tick = 16384
    per = T * heads * (hd + 2) * 4
    assert need == tick + 16 * per
    outs = {}
    for name, nbytes in (("full", need), ("half", tick + 8 * per), ("round4", tick + 4 * per), ("three", tick + 3 * per), ("tickets_only", tick)):
        ws = torch.zeros(need + 4096, dtype=torch.uint8, device=dev)
        ws[nbytes:] = 0x7F                                                     # poison past what the caller says it owns
        out = torch.empty(T, 1, heads * hd, device=dev, dtype=dtype)
        kc, vc, vl = kc0.clone(), vc0.clone(), valid.clone()
        check(L.bd_srv_decode_attention(ptr(qkv), ptr(cos), ptr(sin), ptr(kc), ptr(vc), ptr(vl), ptr(pidx), ptr(out), T, heads, kvh, hd, Lc,
                                        qkv.stride(0), out.stride(0), DTYPE_CODE[dtype], ptr(ws), nbytes, stream_ptr()), "srv_decode_attention")
        torch.cuda.synchronize()
        assert bool((ws[nbytes:] == 0x7F).all()), name                        # never writes past the size it was given
        assert bool((ws[:tick] == 0).all()), name                             # arrival counters restored
        outs[name] = out.float()
    for name, o in outs.items():
        assert relerr(o, outs["full"]) <= 2e-3, name                          # same attention, another merge order (fp16 output rounding)


def test_serving_loop_fast_glue_matches_torch_glue(bd):
    """head_dim-128 decoder: greedy decode with the HIP glue kernels (and graph replay) == the same loop on stock torch ops"""
    from bitdelta_amd.serving_loop import TenantDecoder
    T = 4
    dec = TenantDecoder.synthetic("tiny128", T, "cuda", dtype=torch.float16, seed=21, max_len=160)
    g = torch.Generator().manual_seed(2)
    prompts = [torch.randint(1, 512, (n,), generator=g).tolist() for n in (9, 64, 33, 70)]
    dec.fast_glue = True
    fast, _ = dec.generate(prompts, max_new_tokens=8, use_graph=True)
    dec.fast_glue = False
    slow, _ = dec.generate(prompts, max_new_tokens=8, use_graph=False)
    agree = (fast == slow).float().mean().item()
    assert agree >= 0.9, (agree, fast, slow)                              # greedy paths may fork at a near-tie; they must not diverge wholesale
    assert torch.equal(fast[:, 0], slow[:, 0])                            # same prefill token through either attention


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,M,K,N", [(6, 1, 4096, 6144), (8, 1, 4096, 1024), (3, 1, 2048, 2560), (1, 1, 4096, 4096), (4, 1, 8192, 1024),
                                     (2, 1, 2048, 512)])
def test_fused_rmsnorm_prologue_is_bit_identical_to_separate_launches(bd, dtype, T, M, K, N):
    """bd_binary_linear_decode_fused(norm_w) == bd_srv_rmsnorm then bd_binary_linear_decode, bit for bit (same arithmetic, same order;
    the activations come from LDS instead of L2).  Also with the residual epilogue and a broadcast norm weight."""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.binary_gemm_kernel import binary_linear_decode, fused_norm_ok, pack_decode_masks
    assert fused_norm_ok(T, M, K) and not fused_norm_ok(T, 2, K) and not fused_norm_ok(T, 1, 6144) and not fused_norm_ok(9, 1, 4096)
    g = torch.Generator(device="cuda").manual_seed(K + N + T)
    x = (torch.randn(T, M, K, device="cuda", generator=g) * 1.5).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dtype)
    mask = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    alpha = torch.rand(T, 1, device="cuda", generator=g) * 1e-3
    nw = (1 + 0.1 * torch.randn(T, K, device="cuda", generator=g)).to(dtype)
    pk = pack_decode_masks(mask)
    for norm in (nw, nw[:1]):
        normed = ops.rmsnorm_tenant(x, norm.expand(T, K).contiguous(), 1e-5)
        ref = binary_linear_decode(normed, w, pk, alpha, layout="packed")
        got = binary_linear_decode(x, w, pk, alpha, layout="packed", norm_weight=norm, eps=1e-5)
        assert torch.equal(got, ref)
    res = torch.randn(T, M, N, device="cuda", generator=g).to(dtype)
    ref = binary_linear_decode(ops.rmsnorm_tenant(x, nw, 1e-5), w, pk, alpha, layout="packed", residual=res.clone())
    got = binary_linear_decode(x, w, pk, alpha, layout="packed", residual=res.clone(), norm_weight=nw, eps=1e-5)
    assert torch.equal(got, ref)
    # and against stock torch RMSNorm (tolerance: torch reduces in a different order)
    t_norm = torch.nn.functional.rms_norm(x.float(), (K,), None, 1e-5).to(dtype) * nw[:, None, :]
    t_ref = binary_linear_decode(t_norm.contiguous(), w, pk, alpha, layout="packed")
    assert (got - res - t_ref).abs().max().item() <= 0.02 * t_ref.abs().max().item() + 0.02 * res.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,K,inter", [(6, 4096, 14336), (3, 2048, 1024), (8, 4096, 512), (1, 4096, 11008)])
def test_fused_swiglu_epilogue_is_bit_identical_to_separate_launches(bd, dtype, T, K, inter):
    """RMSNorm -> gate|up (rows interleaved in blocks of 8) -> SwiGLU in one launch == the three separate launches, and == the
    un-interleaved fused gate|up Linear followed by act_fn(gate) * up."""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.serving_loop import FusedDeltaLinear
    g = torch.Generator(device="cuda").manual_seed(K + inter + T)
    ws = [(torch.randn(inter, K, device="cuda", generator=g) * 0.02).to(dtype) for _ in range(2)]
    masks = [torch.randint(-2**31, 2**31 - 1, (T, K // 32, inter), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
             for _ in range(2)]
    coeffs = [torch.rand(T, device="cuda", generator=g) * 1e-3 for _ in range(2)]
    il = FusedDeltaLinear(ws, masks, coeffs, interleave8=True)
    plain = FusedDeltaLinear(ws, masks, coeffs)
    x = (torch.randn(T, 1, K, device="cuda", generator=g) * 1.5).to(dtype)
    nw = (1 + 0.1 * torch.randn(T, K, device="cuda", generator=g)).to(dtype)
    assert il.fusable(x, swiglu=True) and not plain.fusable(x, swiglu=True)
    h = ops.rmsnorm_tenant(x, nw, 1e-5)
    gu_il, gu_plain = il(h), plain(h)
    g_il, u_il = il.split(gu_il)
    assert torch.equal(torch.cat([g_il, u_il], -1), gu_plain)                       # the interleave is a pure row permutation
    sep = ops.swiglu_interleaved8(gu_il)
    assert torch.equal(sep, ops.swiglu(gu_plain, inter))
    fused = il.forward_fused(x, nw, 1e-5, swiglu=True)
    assert fused.shape == (T, 1, inter) and torch.equal(fused, sep)
    epi_only = il.forward_fused(h, None, 1e-5, swiglu=True)                          # SwiGLU epilogue without the norm prologue
    assert torch.equal(epi_only, sep)
    t_ref = torch.nn.functional.silu(gu_plain[..., :inter]) * gu_plain[..., inter:]
    assert (fused.float() - t_ref.float()).abs().max().item() <= 0.01 * t_ref.float().abs().max().item() + 1e-3


def test_serving_loop_fused_glue_matches_separate_launches(bd):
    """hidden = 2048 decoder: the decode step with RMSNorm / SwiGLU folded into the Linear launches produces the same tokens AND the
    same logits (bit for bit) as the step with separate glue launches; eager and hipGraph."""
    from bitdelta_amd.serving_loop import TenantDecoder
    T = 3
    dec = TenantDecoder.synthetic("tiny2048", T, "cuda", dtype=torch.float16, seed=5, max_len=160)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(1, 512, (n,), generator=g).tolist() for n in (12, 64, 40)]
    outs = {}
    dec.fuse_qkv_norm = dec.fuse_gateup_norm = True                       # every fusion on (the shipped default leaves q|k|v's norm separate)
    dec.norm_handoff = False        # (the RMSNorm hand-off moves one rounding: it has its own tolerance tests; this one is about bit-identity)
    for fuse in (True, False):
        dec.fuse_glue = fuse
        for graph in (True, False):
            outs[(fuse, graph)], _ = dec.generate(prompts, max_new_tokens=6, use_graph=graph)
    dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm = True, False, False      # SwiGLU epilogue only
    outs[("epi", True)], _ = dec.generate(prompts, max_new_tokens=6, use_graph=True)
    dec.fuse_qkv_norm = dec.fuse_gateup_norm = True
    ref = outs[(False, False)]
    for k, v in outs.items():
        assert torch.equal(v, ref), k
    # one decode step, logits compared exactly
    ids, am = dec.prepare(prompts)
    logits = {}
    for fuse in (True, False):
        dec.fuse_glue = fuse
        cache = dec.new_cache()
        dec.prefill(ids, am, cache)
        pos = torch.tensor([ids.shape[1]], device="cuda")
        cache["valid"].index_fill_(1, pos, True)
        tok = torch.full((T, 1), 7, dtype=torch.long, device="cuda")
        logits[fuse] = dec.forward(tok, pos, cache, cache["valid"][:, None, None, :])
    assert torch.equal(logits[True], logits[False])


def test_serving_loop_single_tenant(bd):
    """T = 1 (config 2's decode steps through the same loop): size-1 batch dims carry arbitrary strides -- prefill, graph decode and
    eager decode agree token for token."""
    from bitdelta_amd.serving_loop import TenantDecoder
    dec = TenantDecoder.synthetic("tiny128", 1, "cuda", dtype=torch.float16, seed=3, max_len=160)
    prompts = [list(range(1, 41))]
    a, n1 = dec.generate(prompts, max_new_tokens=6, use_graph=True)
    b, n2 = dec.generate(prompts, max_new_tokens=6, use_graph=False)
    assert n1 == n2 == 6 and torch.equal(a, b)


def test_serving_loop_twelve_tenants_decode_in_one_launch(bd):
    """More than 8 tenants per GPU (the reference's batched benchmark runs B = 16): the packed decode layout takes t_pad 12 / 16, so a
    decode Linear over 12 tenants is ONE launch of the streaming kernel (round 3: chunks through the generic path).  Graph decode, eager
    decode and the stock-torch glue agree token for token; the Linear itself agrees with per-tenant launches."""
    from bitdelta_amd import _lib
    from bitdelta_amd.serving_loop import TenantDecoder
    T = 12
    dec = TenantDecoder.synthetic("tiny128", T, "cuda", dtype=torch.float16, seed=11, max_len=160)
    lay = dec.layers[0]
    assert lay.qkv.mask_packed is not None and lay.qkv.mask_packed.shape[4] == 12
    x = torch.randn(T, 1, 512, device="cuda", dtype=torch.float16)
    y = lay.qkv(x)
    assert _lib.lib().bd_last_gemm_variant() == 600
    for t in (0, 7, 11):                                   # tenant t alone: its own mask and scales through the reference-layout path
        yt = bd.binary_linear(x[t:t + 1], lay.qkv.weight, lay.qkv.mask[t:t + 1], lay.qkv.alpha[t:t + 1], groups=lay.qkv.groups)
        assert relerr(y[t:t + 1].float(), yt.float()) <= 2e-3
    prompts = [list(range(1 + t, 30 + 2 * t)) for t in range(T)]
    a, n1 = dec.generate(prompts, max_new_tokens=5, use_graph=True)
    b, n2 = dec.generate(prompts, max_new_tokens=5, use_graph=False)
    dec.fast_glue = False
    c, _ = dec.generate(prompts, max_new_tokens=5, use_graph=False)
    assert n1 == n2 == 5 and torch.equal(a, b) and (a == c).float().mean().item() >= 0.9      # (stock glue rounds differently: near-ties may flip)


def test_serving_loop_static_state_is_bounded(bd):
    """generate() with caller-chosen max_new_tokens and stop-list widths must not grow device memory without bound: one KV
    cache per decoder whatever the request, static buffers keyed by the stop-table width only, at most MAX_STATIC_SLOTS captured graphs
    -- and a slot that was evicted and comes back still decodes the same tokens."""
    from bitdelta_amd.serving_loop import TenantDecoder
    dec = TenantDecoder.synthetic("tiny128", 2, "cuda", dtype=torch.float16, seed=5, max_len=200)
    prompts = [list(range(1, 30)), list(range(5, 50))]
    ref, _ = dec.generate(prompts, max_new_tokens=12, use_graph=False)
    kv = dec._kv_cache
    assert kv is not None
    for n_new in (3, 5, 7, 9, 12, 4, 6):                       # seven different max_new_tokens: ONE slot (round 3: seven caches + graphs)
        out, n = dec.generate(prompts, max_new_tokens=n_new, use_graph=True)
        assert n == n_new and torch.equal(out, ref[:, :n_new])
    assert len(dec._static) == 1 and dec._kv_cache is kv
    for width in (1, 8, 9, 17, 33, 65, 129):                   # stop tables of 8 .. 256 ids: six slots requested, MAX_STATIC_SLOTS kept
        stop = [[9999 + i for i in range(width)], []]
        out, n = dec.generate(prompts, max_new_tokens=12, stop_token_ids=stop, use_graph=True)
        assert n == 12 and torch.equal(out, ref)
        assert dec._kv_cache is kv and all(sl["st"]["cache"] is kv for sl in dec._static.values())
    assert len(dec._static) == dec.MAX_STATIC_SLOTS
    out, n = dec.generate(prompts, max_new_tokens=12, use_graph=True)          # the first (evicted) width again: re-captured, same tokens
    assert torch.equal(out, ref) and len(dec._static) == dec.MAX_STATIC_SLOTS


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,K,N", [(6, 4096, 6144), (1, 4096, 4096), (2, 1024, 1024), (4, 14336, 4096), (8, 2048, 512), (3, 128, 528),
                                   (1, 11008, 4096), (2, 14336, 4096), (3, 5120, 1024), (1, 1152, 528), (2, 1280, 512),
                                   (1, 2304, 528)])     # resident rows at any K (1152 = 9 iterations: a wave would be empty -> plain form)
def test_tile_major_weight_is_bit_identical(bd, dtype, T, K, N):
    """the tile-major decode copy of the base weight (ldw = 0 at the C ABI) gives the same bits as the row-major operand: plain,
    with the residual epilogue, with the SwiGLU epilogue, with RMSNorm + SwiGLU and with the RMSNorm prologue alone"""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.binary_gemm_kernel import binary_linear_decode, fused_norm_ok, pack_decode_masks, tile_weight
    g = torch.Generator(device="cuda").manual_seed(K + N + T)
    x = (torch.randn(T, 1, K, device="cuda", generator=g) * 1.5).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dtype)
    mask = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    alpha = torch.rand(T, 2, device="cuda", generator=g) * 1e-3
    pk, wt = pack_decode_masks(mask), tile_weight(w)
    assert not torch.equal(wt, w)
    a1 = alpha[:, :1].contiguous()
    assert torch.equal(binary_linear_decode(x, wt, pk, a1, layout="packed", weight_tiled=True),
                       binary_linear_decode(x, w, pk, a1, layout="packed"))
    res = torch.randn(T, 1, N, device="cuda", generator=g).to(dtype)
    assert torch.equal(binary_linear_decode(x, wt, pk, a1, layout="packed", weight_tiled=True, residual=res.clone()),
                       binary_linear_decode(x, w, pk, a1, layout="packed", residual=res.clone()))
    assert torch.equal(binary_linear_decode(x, wt, pk, alpha, layout="packed", groups=2, swiglu=True, weight_tiled=True),
                       binary_linear_decode(x, w, pk, alpha, layout="packed", groups=2, swiglu=True))
    if fused_norm_ok(T, 1, K):
        nw = (1 + 0.1 * torch.randn(T, K, device="cuda", generator=g)).to(dtype)
        assert torch.equal(binary_linear_decode(x, wt, pk, alpha, layout="packed", groups=2, swiglu=True, norm_weight=nw, weight_tiled=True),
                           binary_linear_decode(x, w, pk, alpha, layout="packed", groups=2, swiglu=True, norm_weight=nw))
        # RMSNorm prologue alone (the q|k|v launch of a decoder layer): tile-major == row-major == separate norm launch + Linear
        y_t = binary_linear_decode(x, wt, pk, a1, layout="packed", norm_weight=nw, weight_tiled=True)
        assert torch.equal(y_t, binary_linear_decode(x, w, pk, a1, layout="packed", norm_weight=nw))
        assert torch.equal(y_t, binary_linear_decode(ops.rmsnorm_tenant(x, nw, 1e-5), wt, pk, a1, layout="packed", weight_tiled=True))


# ---------------------------------------------------------------------------------------------------- prefill attention (caller glue)
def _attention_fp32(q, k, v, kv_start, causal):
    """plain fp32 softmax attention on [B, S, heads, 128] views; rows without a valid key -> 0"""
    B, S, H, _ = q.shape
    G = H // k.shape[2]
    qf = q.float().transpose(1, 2)
    kf = k.float().transpose(1, 2).repeat_interleave(G, dim=1)
    vf = v.float().transpose(1, 2).repeat_interleave(G, dim=1)
    sc = qf @ kf.transpose(-1, -2) * (128 ** -0.5)
    keys = torch.arange(S, device=q.device)
    ok = torch.ones(B, 1, S, S, dtype=torch.bool, device=q.device)
    if causal:
        ok &= (keys[None, :] <= keys[:, None])[None, None]
    if kv_start is not None:
        ok &= (keys[None, None, None, :] >= kv_start.view(B, 1, 1, 1))
    sc = sc.masked_fill(~ok, float("-inf"))
    p = torch.softmax(sc, dim=-1).nan_to_num(0.0)
    return (p @ vf).transpose(1, 2).reshape(B, S, H * 128)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,S,H,KVH,causal,pad", [(1, 2048, 32, 32, True, False),      # configs[1]: Llama-2-7B prefill
                                                  (2, 1024, 32, 8, True, True),        # Mistral heads, left-padded tenant batch
                                                  (3, 64, 4, 4, True, True), (1, 192, 8, 2, True, False),
                                                  (2, 256, 8, 1, False, False), (1, 4096, 8, 8, True, False)])
def test_prefill_attention_vs_fp32_softmax(bd, dtype, B, S, H, KVH, causal, pad):
    """bd_srv_prefill_attention on the three slices of a fused q|k|v buffer (as the prefill step passes them) against fp32 softmax attention
    on the same 16-bit inputs and against torch's own SDPA: the gate is 'not worse than stock SDPA + a 16-bit rounding'."""
    import torch.nn.functional as F
    from bitdelta_amd import serving_ops as ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(B * 1000 + S + H)
    qkv = torch.randn(B, S, (H + 2 * KVH) * 128, device=dev, generator=g).to(dtype)
    q4 = qkv[..., :H * 128].view(B, S, H, 128)
    k4 = qkv[..., H * 128:(H + KVH) * 128].view(B, S, KVH, 128)
    v4 = qkv[..., (H + KVH) * 128:].view(B, S, KVH, 128)
    kv_start = None
    if pad:
        kv_start = ((torch.arange(B, device=dev) * 37 + 5) % (S // 2)).to(torch.int32)
    assert ops.prefill_attention_supported(q4, k4, v4)
    out = ops.prefill_attention(q4, k4, v4, kv_start=kv_start, causal=causal)
    assert out.shape == (B, S, H * 128) and out.dtype == dtype
    ref = _attention_fp32(q4, k4, v4, kv_start, causal)
    assert torch.isfinite(out).all()
    err = (out.float() - ref).abs().max().item()
    rel = relerr(out, ref)
    # stock SDPA on the same inputs (no padding): the yardstick for the tolerance
    if not pad:
        sd = F.scaled_dot_product_attention(q4.transpose(1, 2), k4.transpose(1, 2), v4.transpose(1, 2), is_causal=causal,
                                            enable_gqa=(KVH != H)).transpose(1, 2).reshape(B, S, H * 128)
        rel_sd = relerr(sd, ref)
        assert rel <= max(1.5 * rel_sd, 4e-3 if dtype == torch.bfloat16 else 6e-4), (rel, rel_sd)
    assert rel <= (4e-3 if dtype == torch.bfloat16 else 6e-4), rel
    assert err <= (3e-2 if dtype == torch.bfloat16 else 4e-3), err
    if pad:      # query rows in the padding have no valid key: zeros, not NaN
        for b in range(B):
            assert (out[b, :int(kv_start[b])] == 0).all()


def test_prefill_attention_rejects_other_geometries(bd):
    from bitdelta_amd import serving_ops as ops
    x = torch.randn(1, 96, 4, 128, device="cuda", dtype=torch.bfloat16)          # S % 64 != 0
    assert not ops.prefill_attention_supported(x, x, x)
    y = torch.randn(1, 128, 4, 64, device="cuda", dtype=torch.bfloat16)          # head_dim 64
    assert not ops.prefill_attention_supported(y, y, y)
    with pytest.raises(AssertionError):
        ops.prefill_attention(x, x, x)


def test_prefill_layer_with_hip_attention_matches_sdpa_layer(bd):
    """the prefill step's decoder layer (bench_model.DecoderLayer) with the HIP attention against the same layer through torch SDPA"""
    import bench_model as bm
    dev, dtype = "cuda", torch.bfloat16
    gen = torch.Generator(device=dev).manual_seed(3)
    layer = bm.DecoderLayer((1024, 2816, 1, 8, 8, 1000), dev, dtype, gen)
    x = torch.randn(1, 256, 1024, device=dev, generator=gen).to(dtype)
    cos, sin = bm.rope_tables(256, 128, dev)
    cs, sn = cos.to(dtype), sin.to(dtype)
    half = torch.cat([-sn[:, :64], sn[:, 64:]], dim=1).contiguous()
    rope = (cs.contiguous(), half, 0)
    with torch.no_grad():
        layer.hip_attention = True
        a = layer(x.clone(), cs, sn, None, rope)
        layer.hip_attention = False
        b = layer(x.clone(), cs, sn, None, rope)
    assert relerr(a, b) < 3e-3, relerr(a, b)


@pytest.mark.parametrize("name,lens", [("tiny128", (9, 64, 33, 70)), ("mistral-1layer", (5, 200, 128, 77, 256, 31))])
def test_serving_loop_prefill_through_hip_attention_matches_torch_attention(bd, name, lens):
    """the prefill call of the serving loop (left-padded tenant batch): RoPE + bd_srv_prefill_attention against rope + SDPA over the
    [T, 1, L, Lc] mask -- logits of the last position and the cached K / V rows of the valid positions"""
    from bitdelta_amd.serving_loop import TenantDecoder
    T = len(lens)
    dec = TenantDecoder.synthetic(name, T, "cuda", dtype=torch.bfloat16, seed=5, max_len=320, shared_heads=True)
    g = torch.Generator().manual_seed(4)
    prompts = [torch.randint(1, 500, (n,), generator=g).tolist() for n in lens]
    ids, am = dec.prepare(prompts)
    out = {}
    for flag in (True, False):
        dec.hip_prefill_attention = flag
        cache = dec.new_cache(ids.shape[1] + 8)
        logits = dec.prefill(ids, am, cache)
        assert cache["kv_start"] is None
        out[flag] = (logits.float(), [k.float().clone() for k in cache["k"]], [v.float().clone() for v in cache["v"]])
    a, b = out[True], out[False]
    assert relerr(a[0][:, -1], b[0][:, -1]) < 2e-2
    L = ids.shape[1]
    for t, n in enumerate(lens):                     # valid rows only: the padding rows differ by construction (zeros vs self-attention)
        for ka, kb_ in zip(a[1], b[1]):
            assert relerr(ka[t, :, L - n:L], kb_[t, :, L - n:L]) < 2e-2
        for va, vb in zip(a[2], b[2]):
            assert relerr(va[t, :, L - n:L], vb[t, :, L - n:L]) < 2e-2


# ------------------------------------------------------------------------------------------------ RMSNorm by hand-off (round 5)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,hid,N2,swiglu", [(6, 4096, 6144, False), (6, 4096, 28672, True), (1, 4096, 12288, False), (8, 4096, 1024, False),
                                             (3, 2048, 2048, True), (4, 8192, 1024, False), (2, 5120, 2560, False)])
def test_rmsnorm_handoff_between_two_decode_launches(bd, dtype, T, hid, N2, swiglu):
    """bd_binary_linear_decode_handoff.  PRODUCER (a residual Linear, K1 -> hid): same output bits as the plain launch, the [hid/16, 16]
    buffer holds the per-row sums of squares of the STORED values, 16 columns at a time, and xw_out their product with the next norm's weight.
    CONSUMER (hid -> N2, optional SwiGLU): xw_out as resident rows + 1/rms on the accumulators == RMSNorm launch followed by the Linear, up to
    the position of one rounding: compared
    with the separate launches (a few 16-bit ulps) and, like them, with a dense fp32 evaluation of the same layer."""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.binary_gemm_kernel import handoff_ok
    from bitdelta_amd.serving_loop import FusedDeltaLinear
    assert handoff_ok(T, 1, hid) and not handoff_ok(9, 1, 4096) and not handoff_ok(T, 2, hid) and not handoff_ok(1, 1, 1024)
    g = torch.Generator(device="cuda").manual_seed(hid + N2 + T)
    K1 = 2048

    def lin(n_out, n_in, il8=False, parts=1):
        ws = [(torch.randn(n_out // parts, n_in, device="cuda", generator=g) * 0.02).to(dtype) for _ in range(parts)]
        ms = [torch.randint(-2**31, 2**31 - 1, (T, n_in // 32, n_out // parts), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
              for _ in range(parts)]
        cs = [torch.rand(T, device="cuda", generator=g) * 1e-3 for _ in range(parts)]
        return FusedDeltaLinear(ws, ms, cs, interleave8=il8)
    prod = lin(hid, K1)
    cons = lin(N2, hid, il8=swiglu, parts=2 if swiglu else 1)
    a = torch.randn(T, 1, K1, device="cuda", generator=g).to(dtype)
    resid = (torch.randn(T, 1, hid, device="cuda", generator=g) * 2.0).to(dtype)
    nw = (1 + 0.1 * torch.randn(T, hid, device="cuda", generator=g)).to(dtype)
    assert prod.handoff_producer_ok(a) and cons.handoff_consumer_ok(resid, swiglu=swiglu)
    # ---- producer
    plain = prod(a, residual=resid.clone())
    ssq = torch.full((hid // 16, 16), float("nan"), device="cuda")
    x = prod(a, residual=resid.clone(), ssq_out=ssq)
    assert torch.equal(x, plain)
    want = (x.float()[:, 0, :].reshape(T, hid // 16, 16) ** 2).sum(-1).T                 # [hid/16, T]
    assert torch.allclose(ssq[:, :T], want, rtol=1e-5, atol=1e-6)
    assert torch.isnan(ssq[:, T:]).all()                                                  # rows past the tenants are not touched
    # ---- ... and the pre-multiplied copy round(x * nw) for the consumer (a broadcast weight, then the per-tenant one used below)
    ssq2 = torch.zeros_like(ssq)
    xw = torch.full_like(x, float("nan"))
    for nrm in (nw[:1], nw):
        x2 = prod(a, residual=resid.clone(), ssq_out=ssq2, next_norm=nrm, xw_out=xw)
        assert torch.equal(x2, plain) and torch.equal(ssq2[:, :T], ssq[:, :T])
        assert torch.equal(xw, (x.float() * nrm.float()[:, None, :]).to(dtype))
    # ---- consumer: resident rows = xw, 1/rms on the accumulators
    h = ops.rmsnorm_tenant(x, nw, 1e-5)
    sep = cons.forward_fused(h, None, 1e-5, swiglu=True) if swiglu else cons(h)
    got = cons.forward_fused(xw, None, 1e-5, swiglu=swiglu, ssq_in=ssq2)
    assert got.shape == sep.shape and got.dtype == dtype
    # dense fp32 evaluation of the same layer (HF RMSNorm in fp32, merged weights)
    S = bd.unpack(cons.mask).float() * 2 - 1                                             # [T, hid, N2]
    wm = cons.weight.float().T[None] + torch.stack([cons.column_alpha(t) for t in range(T)])[:, None, :] * S
    xn = torch.nn.functional.rms_norm(x.float(), (hid,), None, 1e-5) * nw.float()[:, None, :]
    dense = torch.bmm(xn, wm)
    if swiglu:
        v = dense.reshape(T, 1, N2 // 16, 2, 8)
        dense = (torch.nn.functional.silu(v[..., 0, :]) * v[..., 1, :]).reshape(T, 1, N2 // 2)
    rel = lambda u, r: ((u.float() - r).norm() / r.norm()).item()
    tol = 1.5e-3 if dtype == torch.float16 else 8e-3
    assert rel(got, dense) <= tol and rel(sep, dense) <= tol, (rel(got, dense), rel(sep, dense))
    assert rel(got, dense) <= 1.3 * rel(sep, dense) + 1e-4                               # as accurate as the separate launches
    assert rel(got, sep.float()) <= tol
    # outside the envelope the entry point refuses (no silent fallback): a consumer handed a norm weight, a producer with fp32 output
    from bitdelta_amd._lib import BitDeltaHipError
    with pytest.raises((BitDeltaHipError, AssertionError)):
        cons.forward_fused(xw, nw, 1e-5, swiglu=swiglu, ssq_in=ssq2)
    with pytest.raises((BitDeltaHipError, AssertionError)):
        prod(a, residual=resid.float(), out_dtype=torch.float32, ssq_out=ssq)


def test_decoder_with_norm_handoff_matches_the_separate_launch_decoder(bd):
    """TenantDecoder.norm_handoff: two Llama-width layers x 6 tenants, decode steps after a prefill -- logits within the 16-bit rounding band of
    the separate-launch decoder's, same greedy tokens, and the hand-off really runs (no bd_srv_rmsnorm launch inside layers >= 1)."""
    from bitdelta_amd.serving_loop import TenantDecoder
    dec = TenantDecoder.synthetic("tiny4096", 6, "cuda", dtype=torch.float16, seed=11)
    prompts = [[(7 * t + 3 * i) % 500 + 1 for i in range(5 + t)] for t in range(6)]
    outs = {}
    for flag in (False, True):
        dec.norm_handoff = flag
        toks, _ = dec.generate(prompts, max_new_tokens=6, use_graph=False)
        ids, am = dec.prepare(prompts)
        cache = dec.new_cache()
        dec.prefill(ids, am, cache)
        pos = torch.tensor([ids.shape[1]], device="cuda")
        cache["valid"].index_fill_(1, pos, True)
        logits = dec.forward(torch.full((6, 1), 17, device="cuda"), pos, cache, cache["valid"][:, None, None, :])
        outs[flag] = (toks, logits.float())
    assert torch.equal(outs[True][0], outs[False][0])
    d = (outs[True][1] - outs[False][1]).abs().max().item()
    assert d <= 2e-2 * outs[False][1].abs().max().item(), d
    # graph replay == eager with the hand-off on
    dec.norm_handoff = True
    tg, _ = dec.generate(prompts, max_new_tokens=6, use_graph=True)
    assert torch.equal(tg, outs[True][0])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,M,H", [(1, 2048, 4096), (6, 64, 4096), (3, 100, 2048), (1, 257, 8192), (2, 96, 6144), (1, 70, 4096)])
def test_rmsnorm_many_rows_kernel_is_bit_identical_to_the_row_per_block_kernel(bd, dtype, T, M, H):
    """rmsnorm_rows_kernel (one wave per row, the row in registers; bd_srv_rmsnorm from 64 rows on) forms the four per-wave sums of
    rmsnorm_tenant_kernel with the same lanes in the same order: the two launches must agree BIT FOR BIT (a child process with
    BD_NORM_ROWS_MIN raised runs the block-per-row kernel), on strided rows too, and both match the HF definition"""
    import subprocess
    import torch.nn.functional as F
    from bitdelta_amd import serving_ops as ops
    g = torch.Generator().manual_seed(T * 1000 + M + H)
    xfull = torch.randn(T, M, H + 64, generator=g).to(dtype).cuda()
    x = xfull[..., :H]                                        # row stride H + 64: rows are not back to back
    w = (1 + 0.1 * torch.randn(T, H, generator=g)).to(dtype).cuda()
    got = ops.rmsnorm_tenant(x, w, 1e-5)
    v = x.float()
    want = (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype) * w[:, None, :]       # HF LlamaRMSNorm
    assert torch.allclose(got.float(), want.float(), rtol=2 ** -7 if dtype == torch.bfloat16 else 2 ** -10, atol=1e-3)
    # the block-per-row kernel in a child process (the dispatch threshold is read once per thread from the environment)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", f"_norm_ab_{os.getpid()}.pt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"x": xfull.cpu(), "w": w.cpu(), "H": H}, path)
    code = ("import torch, sys; from bitdelta_amd import serving_ops as ops; d = torch.load(sys.argv[1]); "
            "x = d['x'].cuda()[..., :d['H']]; y = ops.rmsnorm_tenant(x, d['w'].cuda(), 1e-5); torch.save(y.cpu(), sys.argv[1] + '.out')")
    try:
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, cwd=root, timeout=300,
                           env=dict(os.environ, BD_NORM_ROWS_MIN="1000000000"))
        assert r.returncode == 0, r.stderr[-1500:]
        old = torch.load(path + ".out")
    finally:
        for f in (path, path + ".out"):
            if os.path.exists(f):
                os.remove(f)
    assert torch.equal(got.cpu(), old)


# ------------------------------------------------------------------------------------------------ fine-grid decode form (round 6)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,K,N", [(6, 4096, 6144), (4, 4096, 6144), (1, 4096, 6144), (2, 4096, 8192), (6, 4096, 4096), (5, 2048, 4112),
                                   (3, 8192, 5120), (6, 4096, 4096 + 16 * 7)])
def test_fine_grid_decode_form_is_bit_identical(bd, oracle, dtype, T, K, N):
    """gemv_stream_kernel<..., FG = 1> (single-tile blocks, two per CU, nibble sign table; bd_set_stream_tuning 512 = wherever eligible, 256 = never):
    the same bits as the one-block-per-CU form for the plain, residual, SwiGLU and hand-off launches, checked against the C oracle too; Mistral's
    fused q|k|v shape (384 tiles) takes it by default."""
    from bitdelta_amd import _lib
    from bitdelta_amd.binary_gemm_kernel import binary_linear_decode, pack_decode_masks, tile_weight
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(K + N + T)
    x = (torch.randn(T, 1, K, device="cuda", generator=g) * 1.5).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dtype)
    mask = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    alpha = torch.rand(T, 2, device="cuda", generator=g) * 1e-3
    a1 = alpha[:, :1].contiguous()
    res = torch.randn(T, 1, N, device="cuda", generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(T, N, device="cuda", generator=g)).to(dtype)
    pk, wt = pack_decode_masks(mask), tile_weight(w)
    sin = torch.rand(K // 16, 16, device="cuda", generator=g) * 8 + 1          # "partial sums of squares" of the hand-off consumer launches
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from canary import CanaryOut                         # poisoned margins around an output: a store outside [T, 1, N] is caught
    outs = {}
    try:
        for flag in (256, 512, 0):
            L.bd_set_stream_tuning(flag)
            o = {}
            o["plain"] = binary_linear_decode(x, wt, pk, a1, layout="packed", weight_tiled=True)
            form = L.bd_last_decode_form()
            assert form == (1 if flag == 512 or (flag == 0 and 256 < N // 16 <= 512) else 0), (flag, form)
            can = CanaryOut(T, 1, N, dtype, row_margin=4, col_margin=64)
            can.view.copy_(res)
            binary_linear_decode(x, wt, pk, a1, layout="packed", weight_tiled=True, residual=can.view)
            o["resid"] = can.result()
            assert can.untouched_outside()
            o["swiglu"] = binary_linear_decode(x, wt, pk, alpha, layout="packed", groups=2, swiglu=True, weight_tiled=True)
            ssq = torch.zeros(N // 16, 16, device="cuda")
            xw = torch.zeros(T, 1, N, device="cuda", dtype=dtype)
            o["prod"] = binary_linear_decode(x, wt, pk, a1, layout="packed", weight_tiled=True, residual=res.clone(), ssq_out=ssq, norm_weight=nw, xw_out=xw)
            o["ssq"], o["xw"] = ssq, xw
            if K % 16 == 0 and T <= 8 and K >= 2048:
                o["cons"] = binary_linear_decode(x, wt, pk, a1, layout="packed", weight_tiled=True, ssq_in=sin, eps=1e-5)
                o["cons_sw"] = binary_linear_decode(x, wt, pk, alpha, layout="packed", groups=2, swiglu=True, weight_tiled=True, ssq_in=sin, eps=1e-5)
            outs[flag] = o
    finally:
        L.bd_set_stream_tuning(0)
    for k in outs[256]:
        assert torch.equal(outs[256][k], outs[512][k]), k
        assert torch.equal(outs[256][k], outs[0][k]), k
    ref = oracle.binary_linear(x.cpu(), w.cpu(), mask.cpu(), a1.cpu(), out_dtype=torch.float32)
    assert relerr(outs[512]["plain"].float().cpu(), ref) < (1e-3 if dtype == torch.float16 else 4e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("row_scale,nw_scale", [(1e-3, 0.02), (1e-3, 5.0), (1.0, 0.02), (1e3, 0.02), (1e3, 5.0), (30.0, 1.0)])
def test_rmsnorm_handoff_activation_range(bd, dtype, row_scale, nw_scale):
    """the hand-off stores round16(x_raw * norm_w) of the UN-normalised residual stream, where HF normalises first -- in fp16
    that product can overflow on a massive-activation row with a large norm weight (row scale 1e3, weights ~5: 2e3 * 5 * 4 sigma > 65504) or go
    subnormal on small rows.  serving_loop.handoff_norm removes the overflow by construction (norm weight / s with s = 2^k >= max |nw|, so
    |xw| <= |x|; 1 / s^2 on the producer's sums of squares and eps / s^2 on the consumer give the same product).  Residual rows scaled by
    1e-3 ... 1e+3, one massive element per row set (as real Llama / Mistral checkpoints have), norm weights of about 0.02 ... 5: the launch pair
    stays finite and as accurate against the dense fp32 evaluation as the separate RMSNorm + Linear launches; the UNSCALED protocol does overflow
    in the (1e3, 5) fp16 case, which is what the scaled one is for."""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.serving_loop import FusedDeltaLinear, handoff_norm
    T, hid, N2, K1 = 6, 4096, 6144, 2048
    g = torch.Generator(device="cuda").manual_seed(int(row_scale * 1000) + int(nw_scale * 100) + 7)

    def lin(n_out, n_in):
        w = (torch.randn(n_out, n_in, device="cuda", generator=g) * 0.02).to(dtype)
        m = torch.randint(-2**31, 2**31 - 1, (T, n_in // 32, n_out), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
        return FusedDeltaLinear([w], [m], [torch.rand(T, device="cuda", generator=g) * 1e-3])
    prod, cons = lin(hid, K1), lin(N2, hid)
    a = (torch.randn(T, 1, K1, device="cuda", generator=g) * min(row_scale, 1.0)).to(dtype)
    resid = torch.randn(T, 1, hid, device="cuda", generator=g) * 2.0 * row_scale
    resid[0, 0, 1415] = min(60.0 * 2.0 * row_scale, 3.0e4)           # one massive element (Llama-2-7B: dims 1415 / 2533 sit 50-100 sigma out)
    resid = resid.to(dtype)
    assert torch.isfinite(resid.float()).all()
    nw = (nw_scale * (1 + 0.3 * torch.randn(T, hid, device="cuda", generator=g))).to(dtype)
    nwh, s = handoff_norm(nw)
    assert float(nwh.float().abs().max()) <= 1.0 and s >= float(nw.float().abs().max()) and torch.equal((nwh.float() * s).to(dtype), nw)
    ssq = torch.zeros(hid // 16, 16, device="cuda")
    xw = torch.zeros(T, 1, hid, device="cuda", dtype=dtype)
    x = prod(a, residual=resid.clone(), ssq_out=ssq, next_norm=nwh, xw_out=xw, ssq_scale=1.0 / (s * s))
    assert bool((xw.float().abs() <= x.float().abs()).all())         # |nw / s| <= 1: the pre-multiplied copy never exceeds the stream itself
    got = cons.forward_fused(xw, None, 1e-5 / (s * s), ssq_in=ssq)
    sep = cons(ops.rmsnorm_tenant(x, nw, 1e-5))
    assert torch.isfinite(xw.float()).all() and torch.isfinite(got.float()).all() and torch.isfinite(sep.float()).all()
    S = bd.unpack(cons.mask).float() * 2 - 1
    wm = cons.weight.float().T[None] + torch.stack([cons.column_alpha(t) for t in range(T)])[:, None, :] * S
    xn = torch.nn.functional.rms_norm(x.float(), (hid,), None, 1e-5) * nw.float()[:, None, :]
    dense = torch.bmm(xn, wm)
    rel = lambda u: ((u.float() - dense).norm(dim=-1) / dense.norm(dim=-1)).max().item()          # worst ROW, not the aggregate
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    assert rel(got) <= tol and rel(sep) <= tol, (rel(got), rel(sep))
    assert rel(got) <= 1.5 * rel(sep) + 2e-4, (rel(got), rel(sep))
    # the scaled protocol is an exact reparametrisation of the unscaled one wherever that one stays in range (powers of two throughout)
    ssq0 = torch.zeros_like(ssq)
    xw0 = torch.zeros_like(xw)
    prod(a, residual=resid.clone(), ssq_out=ssq0, next_norm=nw, xw_out=xw0)
    if torch.isfinite(xw0.float()).all() and float(xw0.float().abs().min()) >= 0.0:
        got0 = cons.forward_fused(xw0, None, 1e-5, ssq_in=ssq0)
        normal = xw0.float().abs() >= (6.2e-5 if dtype == torch.float16 else 0.0)                  # fp16 subnormals round differently after the division
        if bool(normal.all()) and bool((xw.float().abs() >= (6.2e-5 if dtype == torch.float16 else 0.0)).all()):
            assert torch.equal(got0, got)
    elif dtype == torch.float16:
        assert row_scale >= 1e3 and nw_scale >= 5.0                  # the case the scaling exists for: the unscaled copy overflowed


def test_norm_handoff_token_level_agreement_over_32_greedy_steps(bd):
    """VERDICT r05 weak #1c: the default decode path is the RMSNorm HAND-OFF, which moves one rounding relative to HF's order and was only
    compared with dense twins.  Three full-width Mistral-7B layers (hidden 4096, 32 / 8 heads, intermediate 14336: both hand-offs of a layer
    run, o -> gate|up and down -> next q|k|v), 6 tenants, 33 greedy steps: the hand-off decoder is teacher-forced on the tokens the
    separate-launch decoder chose; at every step and tenant the two argmax tokens agree unless the separate-launch logits are in a near-tie,
    and the logits stay within the fp16 rounding band of each other."""
    from bitdelta_amd.serving_loop import TenantDecoder
    T, steps = 6, 33
    dec = TenantDecoder.synthetic((4096, 14336, 3, 32, 8, 512), T, "cuda", dtype=torch.float16, seed=31, max_len=128)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(1, 512, (n,), generator=g).tolist() for n in (9, 64, 33, 50, 17, 60)]
    ids, am = dec.prepare(prompts)
    L = ids.shape[1]

    def run(flag, forced=None):
        dec.norm_handoff = flag
        cache = dec.new_cache()
        lg = dec.prefill(ids, am, cache)
        logits, toks = [lg.float()], []
        for s_ in range(steps):
            tok = forced[s_] if forced is not None else torch.argmax(logits[-1], dim=-1)
            toks.append(tok)
            pos = torch.tensor([L + s_], device="cuda")
            cache["valid"].index_fill_(1, pos, True)
            logits.append(dec.forward(tok[:, None], pos, cache, cache["valid"][:, None, None, :]).float())
        return logits, toks
    try:
        l_off, t_off = run(False)
        l_on, _ = run(True, forced=t_off)
    finally:
        dec.norm_handoff = True
    n_tie = 0
    for s_ in range(1, steps + 1):                                   # (logits[0] is the prefill: identical code in both runs)
        a, b = l_off[s_], l_on[s_]
        assert relerr(b, a) <= 4e-3, (s_, relerr(b, a))
        top2 = a.topk(2, dim=-1).values
        margin = (top2[:, 0] - top2[:, 1])
        same = a.argmax(-1) == b.argmax(-1)
        near_tie = margin <= 4e-3 * top2[:, 0].abs().clamp_min(1.0)
        assert bool((same | near_tie).all()), (s_, margin.tolist(), same.tolist())
        n_tie += int((~same).sum())
    assert n_tie <= 2                                                # near-ties are rare; a systematic disagreement is a bug
    assert torch.equal(l_off[0], l_on[0])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,M,H", [(1, 1, 8192), (1, 1, 4096), (6, 1, 4096), (2, 3, 2056), (1, 16, 8192), (3, 1, 264)])
def test_add_rmsnorm_is_bit_identical_to_cast_add_norm(bd, dtype, T, M, H):
    """bd_srv_add_rmsnorm (round 6: what follows every row-parallel Linear's all-reduce in the tensor-parallel decoder, tp.py) == the three launches it
    replaces -- `y32.to(dtype)`, the 16-bit residual add, rmsnorm_tenant -- bit for bit, on strided inputs too, and HF's RMSNorm within rounding."""
    from bitdelta_amd import serving_ops as ops
    g = torch.Generator(device="cuda").manual_seed(T * 100 + M * 10 + H)
    resid_full = (torch.randn(T, M, H + 64, device="cuda", generator=g) * 2).to(dtype)
    resid = resid_full[..., :H]                               # row stride H + 64
    y32 = torch.randn(T, M, H, device="cuda", generator=g) * 0.7
    w = (1 + 0.1 * torch.randn(T, H, device="cuda", generator=g)).to(dtype)
    x, h = ops.add_rmsnorm(resid, y32, w, 1e-5)
    x_ref = resid + y32.to(dtype)
    h_ref = ops.rmsnorm_tenant(x_ref.contiguous(), w, 1e-5)
    assert x.dtype == dtype and torch.equal(x, x_ref) and torch.equal(h, h_ref)
    v = x_ref.float()
    hf = (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype) * w[:, None, :]
    assert torch.allclose(h.float(), hf.float(), rtol=2 ** -7 if dtype == torch.bfloat16 else 2 ** -10, atol=1e-3)


# ------------------------------------------------------------------------------------------------ short-prompt prefill fusions (round 6)
def _mt_linear(T, N, K, dtype, g, interleave8=False):
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dtype)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (T, K // 32, N), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    return w, mask


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,M,N,K", [(6, 64, 4096, 4096), (6, 64, 4096, 14336), (5, 40, 4096, 4096), (2, 64, 2048, 2048), (6, 17, 1024, 512),
                                     (3, 128, 4096, 4096), (6, 64, 8192, 1024), (1, 64, 4096, 4096)])
def test_residual_linear_with_following_norm_is_bit_identical_to_the_two_launches(bd, oracle, dtype, T, M, N, K):
    """bd_binary_linear_residual_norm: (residual + Linear(x), per-tenant RMSNorm of it) == bd_binary_linear_residual then bd_srv_rmsnorm, bit for
    bit -- on the shapes whose Linear is split over k (the o / down projections of a 6-tenant request of <= 64-row prompts: the norm rides on the
    reduce launch) and on shapes that are not split (the norm is its own launch behind the Linear); and the oracle's Linear on sampled columns."""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.binary_gemm_kernel import binary_linear_residual_norm
    g = torch.Generator(device="cuda").manual_seed(T * 1000 + M + N + K)
    x = torch.randn(T, M, K, device="cuda", generator=g).to(dtype)
    w, mask = _mt_linear(T, N, K, dtype, g)
    alpha = (torch.rand(T, 1, device="cuda", generator=g) * 1e-3 + 2e-4)
    resid = torch.randn(T, M, N, device="cuda", generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(T, N, device="cuda", generator=g)).to(dtype)
    r_ref = resid.clone()
    bd.binary_linear(x, w, mask, alpha, residual=r_ref)
    from bitdelta_amd import _lib
    used_plain = _lib.lib().bd_last_gemm_variant()
    h_ref = ops.rmsnorm_tenant(r_ref, nw, 1e-5)
    r = resid.clone()
    r2, h = binary_linear_residual_norm(x, w, mask, alpha, r, nw, 1e-5)
    assert r2.data_ptr() == r.data_ptr()
    assert torch.equal(r, r_ref), (used_plain, (r.float() - r_ref.float()).abs().max().item())
    assert torch.equal(h, h_ref)
    # the Linear itself against the oracle on a few columns of tenant 0 and the last tenant
    cols = torch.tensor([0, 1, N // 2, N - 1])
    ref = oracle.binary_linear(x.cpu(), w.cpu()[cols].contiguous(), mask.cpu()[:, :, cols].contiguous(), alpha.cpu(), out_dtype=torch.float32)
    want = (resid.cpu()[:, :, cols].float() + ref.to(dtype).float()).to(dtype)           # the two roundings of `hidden = residual + proj(x)`
    got = r.cpu()[:, :, cols]
    assert relerr(got, want) < (1e-2 if dtype == torch.bfloat16 else 2e-3), relerr(got, want)


def test_residual_norm_entry_refuses_what_it_cannot_run(bd):
    from bitdelta_amd.binary_gemm_kernel import binary_linear_residual_norm
    from bitdelta_amd._lib import BitDeltaHipError
    g = torch.Generator(device="cuda").manual_seed(3)
    dtype = torch.bfloat16
    for (T, M, N, K) in [(6, 1, 4096, 4096), (2, 64, 8200, 512), (2, 64, 16384, 512)]:        # decode rows; N % 8 != 0 is impossible with int32 masks -> N > 8192
        x = torch.randn(T, M, K, device="cuda", generator=g).to(dtype)
        w, mask = _mt_linear(T, N, K, dtype, g)
        resid = torch.zeros(T, M, N, device="cuda", dtype=dtype)
        nw = torch.ones(T, N, device="cuda", dtype=dtype)
        with pytest.raises(BitDeltaHipError):
            binary_linear_residual_norm(x, w, mask, torch.full((T, 1), 1e-3, device="cuda"), resid, nw, 1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,S,H,KVH,Lc,pos0", [(6, 64, 32, 8, 320, 0), (2, 128, 32, 32, 128, 0), (3, 17, 8, 2, 64, 5), (1, 64, 4, 4, 96, 32)])
def test_rope_kv_append_is_bit_identical_to_rope_and_two_copies(bd, dtype, T, S, H, KVH, Lc, pos0):
    """bd_srv_rope_kv_append == bd_srv_rope on the q and k heads + `kcache[:, :, pos0:pos0 + S] = k.transpose(1, 2)` + the v twin; cache rows outside
    the written positions are untouched"""
    from bitdelta_amd import serving_ops as ops
    g = torch.Generator(device="cuda").manual_seed(T + S + H)
    W = (H + 2 * KVH) * 128
    full = torch.randn(T, S, W + 64, device="cuda", generator=g).to(dtype)
    qkv = full[..., :W]                                                          # padded rows (row stride W + 64)
    pos = torch.arange(Lc, device="cuda", dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, device="cuda", dtype=torch.float32) / 128))
    f = torch.outer(pos, inv)
    emb = torch.cat([f, f], -1)
    cos = emb.cos().to(dtype).contiguous()
    sin = (emb.sin() * torch.cat([-torch.ones(64, device="cuda"), torch.ones(64, device="cuda")])).to(dtype).contiguous()
    kc = torch.randn(T, KVH, Lc, 128, device="cuda", generator=g).to(dtype)
    vc = torch.randn(T, KVH, Lc, 128, device="cuda", generator=g).to(dtype)
    ref_full = full.clone()
    ref = ref_full[..., :W]
    ops.rope_(ref[..., :(H + KVH) * 128], cos, sin, H + KVH, S, pos0)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    kc_ref[:, :, pos0:pos0 + S] = ref[..., H * 128:(H + KVH) * 128].reshape(T, S, KVH, 128).transpose(1, 2)
    vc_ref[:, :, pos0:pos0 + S] = ref[..., (H + KVH) * 128:].reshape(T, S, KVH, 128).transpose(1, 2)
    ops.rope_kv_append_(qkv, cos, sin, kc, vc, H, KVH, pos0)
    assert torch.equal(full, ref_full)                                           # (the padding columns too: untouched)
    assert torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,M,I,K", [(6, 64, 14336, 4096), (5, 33, 1024, 512), (2, 64, 11008, 4096), (3, 17, 512, 1024)])
def test_pair_tile_swiglu_epilogue_is_bit_identical_to_linear_plus_swiglu(bd, dtype, T, M, I, K):
    """bd_binary_linear_swiglu on several tenants of <= 64 rows (round 6: the four-wave PAIR tile with the SwiGLU epilogue, variant 21) == the
    interleaved gate|up Linear followed by the SwiGLU pass"""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.binary_gemm_kernel import binary_linear_swiglu
    from bitdelta_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(T + M + I + K)
    x = torch.randn(T, M, K, device="cuda", generator=g).to(dtype)
    w, mask = _mt_linear(T, 2 * I, K, dtype, g)
    alpha = torch.rand(T, 2, device="cuda", generator=g) * 1e-3 + 2e-4
    y = binary_linear_swiglu(x, w, mask, alpha)
    assert L.bd_last_gemm_variant() == 21
    L.bd_set_gemm_variant(18)                        # the unsplit pair tile: a split-k launch sums k in another order (1-ulp differences on narrow N)
    try:
        gu = bd.binary_linear(x, w, mask, alpha.repeat(1, I // 8).contiguous(), groups=2 * I // 8)  # columns in blocks of 8: gate, up, gate, up, ...
    finally:
        L.bd_set_gemm_variant(-1)
    ref = ops.swiglu_interleaved8(gu)
    assert y.shape == (T, M, I) and torch.equal(y, ref)


@pytest.mark.parametrize("epi", [False, True])
def test_short_prompt_prefill_fusions_change_no_bit(bd, epi):
    """the serving loop's prefill of a 6-tenant request of short prompts with the round-6 fusions (RoPE + cache append in one launch, the norms on
    the split-k reduce launches; epi: SwiGLU in the pair-tile epilogue too) == the same request without them: logits, every cached K / V row"""
    from bitdelta_amd.serving_loop import TenantDecoder
    lens = (9, 64, 33, 50, 17, 64)
    dec = TenantDecoder.synthetic("mistral-1layer", len(lens), "cuda", dtype=torch.bfloat16, seed=11, max_len=128, shared_heads=True)
    g = torch.Generator().manual_seed(2)
    prompts = [torch.randint(1, 500, (n,), generator=g).tolist() for n in lens]
    ids, am = dec.prepare(prompts)
    assert ids.shape[1] == 64
    out = {}
    for flag in (True, False):
        dec.short_prompt_fusions = flag
        dec.swiglu_epilogue = flag and epi
        cache = dec.new_cache(128)
        logits = dec.prefill(ids, am, cache)
        out[flag] = (logits.clone(), [k.clone() for k in cache["k"]], [v.clone() for v in cache["v"]])
    a, b = out[True], out[False]
    assert torch.equal(a[0], b[0])
    for ka, kb_ in zip(a[1], b[1]):
        assert torch.equal(ka, kb_)
    for va, vb in zip(a[2], b[2]):
        assert torch.equal(va, vb)


# ------------------------------------------------------------------------------------------------ the two ends of a decode step (round 6)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,V,H,shared", [(6, 32000, 4096, False), (1, 512, 256, False), (4, 32008, 2048, True), (12, 1000, 64, False)])
def test_step_begin_and_step_end_do_what_the_stock_ops_do(bd, dtype, T, V, H, shared):
    """bd_srv_step_begin == `valid.index_fill_(1, pos, True)` + the per-tenant embedding gather; bd_srv_step_end == argmax (torch's order: first
    maximum, NaN wins) + `tok.copy_` + `out.index_copy_` + the stop-flag update + `pos += 1; step += 1` -- exact, over several steps in a row,
    with ties, NaNs and stop tokens in the logits"""
    from bitdelta_amd import serving_ops as ops
    g = torch.Generator(device="cuda").manual_seed(T + V + H)
    embed = (torch.randn(V, H, device="cuda", generator=g) if shared else torch.randn(T, V, H, device="cuda", generator=g)).to(dtype)
    Lc, ns, cap = 40, 3, 9
    tok = torch.randint(0, V, (T, 1), device="cuda", generator=g)
    valid = torch.zeros(T, Lc, dtype=torch.bool, device="cuda")
    valid[:, :5] = True
    pos, step = torch.tensor([5], device="cuda"), torch.tensor([1], device="cuda")
    out = torch.zeros(T, cap, dtype=torch.long, device="cuda")
    stopped = torch.zeros(T, dtype=torch.bool, device="cuda")
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    r = {k: v.clone() for k, v in dict(tok=tok, valid=valid, pos=pos, step=step, out=out, stopped=stopped).items()}
    stop_ids = torch.full((T, ns), -1, dtype=torch.long, device="cuda")
    for it in range(cap + 2):                                    # runs past the end of `out`: the surplus steps must not write
        x = ops.step_begin(embed, tok, valid, pos)
        r["valid"].index_fill_(1, r["pos"], True)
        x_ref = (embed[r["tok"][:, 0]] if shared else embed[torch.arange(T, device="cuda"), r["tok"][:, 0]])[:, None, :]
        assert torch.equal(x, x_ref) and torch.equal(valid, r["valid"])
        logits = torch.randn(T, V, device="cuda", generator=g).to(dtype)
        if it == 1:                                              # ties: the maximum appears several times (first index wins)
            logits[:, [7, 3, V - 1]] = 100.0
        if it == 2:                                              # NaN beats everything, the first NaN wins
            logits[:, [V - 5, 11]] = float("nan")
            logits[0, 2] = float("inf")
        if it == 3:                                              # negative zero ties positive zero; all-equal rows
            logits.fill_(0.0)
            logits[:, 1] = -0.0
        if it == 4:                                              # this step's winner is a stop token of tenants 0 and T - 1
            logits[:, 123 % V] = 50.0
            stop_ids[0, 1] = 123 % V
            stop_ids[T - 1, 2] = 123 % V
        ops.step_end(logits, tok, out, step, pos, stop_ids, stopped, ticket)
        nxt = torch.argmax(logits, dim=-1)
        r["tok"].copy_(nxt[:, None])
        if int(r["step"]) < cap:
            r["out"].index_copy_(1, r["step"], nxt[:, None])
        r["stopped"] |= (nxt[:, None] == stop_ids).any(dim=1)
        r["pos"] += 1
        r["step"] += 1
        for k, v in dict(tok=tok, pos=pos, step=step, out=out, stopped=stopped).items():
            assert torch.equal(v, r[k]), (it, k, v, r[k])
        assert int(ticket) == 0
    assert bool(stopped[0]) and bool(stopped[T - 1]) and (T <= 2 or not bool(stopped[1]))


@pytest.mark.parametrize("name,T,dtype", [("tiny128", 4, torch.float16), ("mistral-1layer", 6, torch.bfloat16)])
def test_decode_loop_with_step_kernels_generates_the_same_tokens(bd, name, T, dtype):
    """TenantDecoder.generate with the step kernels (graph replay and eager) == the same loop with the stock ops at both ends of the step: every
    token, the step count, the stop behaviour"""
    from bitdelta_amd.serving_loop import TenantDecoder
    dec = TenantDecoder.synthetic(name, T, "cuda", dtype=dtype, seed=31, max_len=192, shared_heads=True)
    g = torch.Generator().manual_seed(9)
    prompts = [torch.randint(1, 500, (n,), generator=g).tolist() for n in (9, 64, 33, 70, 5, 12)[:T]]
    dec.step_kernels = False
    ref, n_ref = dec.generate(prompts, max_new_tokens=12, use_graph=True)
    stops = [[int(ref[t, 5])] if t % 2 == 0 else [] for t in range(T)]        # a stop token that tenants 0, 2, ... really produce at step 5
    ref_s, n_ref_s = dec.generate(prompts, max_new_tokens=12, stop_token_ids=[s if s else [int(ref[t, 7])] for t, s in enumerate(stops)], use_graph=True)
    for use_graph in (True, False):
        dec.step_kernels = True
        got, n = dec.generate(prompts, max_new_tokens=12, use_graph=use_graph)
        assert n == n_ref and torch.equal(got, ref)
        got_s, n_s = dec.generate(prompts, max_new_tokens=12, stop_token_ids=[s if s else [int(ref[t, 7])] for t, s in enumerate(stops)], use_graph=use_graph)
        assert n_s == n_ref_s and torch.equal(got_s, ref_s) and n_s < 12


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_residual_norm_with_scale_groups_shared_mask_and_shared_norm_weight(bd, dtype):
    """bd_binary_linear_residual_norm off the serving loop's beaten path: a fused projection with several scale groups (alpha [T, G]), ONE mask shared
    by all entries (mask batch 1), an odd number of entries -- still bit-identical to residual Linear + RMSNorm"""
    from bitdelta_amd import serving_ops as ops
    from bitdelta_amd.binary_gemm_kernel import binary_linear_residual_norm
    g = torch.Generator(device="cuda").manual_seed(5)
    for (T, M, N, K, G, shared) in [(5, 48, 4096, 4096, 4, False), (3, 64, 2048, 1024, 2, True), (2, 33, 1024, 512, 1, True)]:
        x = torch.randn(T, M, K, device="cuda", generator=g).to(dtype)
        w, mask = _mt_linear(1 if shared else T, N, K, dtype, g)
        alpha = torch.rand(1 if shared else T, G, device="cuda", generator=g) * 1e-3 + 2e-4
        resid = torch.randn(T, M, N, device="cuda", generator=g).to(dtype)
        nw = (1 + 0.1 * torch.randn(T, N, device="cuda", generator=g)).to(dtype)
        r_ref = resid.clone()
        bd.binary_linear(x, w, mask, alpha, groups=G, residual=r_ref)
        h_ref = ops.rmsnorm_tenant(r_ref, nw, 1e-5)
        r = resid.clone()
        _, h = binary_linear_residual_norm(x, w, mask, alpha, r, nw, 1e-5, groups=G)
        assert torch.equal(r, r_ref) and torch.equal(h, h_ref), (T, M, N, K, G, shared)
