"""GPU parity at the BASELINE.json shapes, every config by name (run on the MI355X box: `pytest -m gpu`).

Each case goes Python surface -> C ABI -> HIP kernel at the FULL problem size and is compared with the CPU oracle
(oracle/bd_oracle.c, pinned to the reference by tests/test_oracle_golden.py).  Where the full O(M*N*K) oracle would take minutes
(prefill M = 2048) the oracle is run on a SAMPLE OF OUTPUT COLUMNS -- it takes the sliced W rows / packed-word columns as its
operands, so every sampled column is checked exactly over all M rows and all of k -- and the columns are spread over the whole
width (tile edges included).  Decode shapes (M = 1) are checked in full.

Tolerances: fp32-output mode <= 1e-5 rel-Frobenius vs the exact-sum oracle; 16-bit outputs within 1 ulp of the rounded oracle
(>= 99 % bit-equal), the same gates as tests/test_gpu_parity.py.

Shapes:  Llama-2-7B q/k/v/o 4096x4096, gate/up 4096->11008, down 11008->4096;  Mistral-7B q/o 4096x4096,
k/v 4096->1024, gate/up 4096->14336, down 14336->4096;  Llama-2-70B TP=8 shards: q 8192->1024, k/v 8192->128, o 1024->8192,
gate/up 8192->3584, down 3584->8192  (reference model pair: scripts/multigpu_train_example.bash:1-13).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bd():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import bitdelta_amd
    from bitdelta_amd import _lib
    _lib.lib()
    return bitdelta_amd


def ulp_diff(a, b):
    def key(t):
        i = t.view(torch.int16).int()
        return torch.where(i < 0, -(i & 0x7fff), i)
    return (key(a) - key(b)).abs()


def relerr(a, ref):
    a, ref = a.double(), ref.double()
    return ((a - ref).norm() / ref.norm()).item()


def sample_columns(N, n=64, seed=0):
    """n output columns spread over [0, N): both ends, 16/128/256-column tile edges, and random ones."""
    g = torch.Generator().manual_seed(seed)
    fixed = [0, 1, 15, 16, 127, 128, 255, 256, N // 2 - 1, N // 2, N - 257, N - 129, N - 17, N - 16, N - 2, N - 1]
    fixed = [c for c in fixed if 0 <= c < N]
    rnd = torch.randint(0, N, (n,), generator=g).tolist()
    cols = sorted(set(fixed + rnd))[:max(n, len(fixed))]
    return torch.tensor(cols, dtype=torch.long)


def make_layer(N, K, dtype, tenants, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dtype)
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (tenants, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
    alpha = (torch.rand(tenants, 1, generator=g) * 2e-4 + 3e-4).float()
    return w, p, alpha


def check_linear(bd, oracle, x, w, p, alpha, cols=None, groups=1):
    """binary_linear on the GPU at full size vs the oracle on all / sampled columns."""
    B, M, K = x.shape
    N = w.shape[0]
    xd, wd, pd, ad = x.cuda(), w.cuda(), p.cuda(), alpha.cuda()
    y32 = bd.binary_linear(xd, wd, pd, ad, out_dtype=torch.float32, groups=groups).cpu()
    y16 = bd.binary_linear(xd, wd, pd, ad, groups=groups).cpu()
    if cols is None:
        ref32 = oracle.binary_linear(x, w, p, alpha, G=groups, out_dtype=torch.float32, round_mode=0)
        got32, got16 = y32, y16
    else:
        assert groups == 1
        ref32 = oracle.binary_linear(x, w[cols].contiguous(), p[:, :, cols].contiguous(), alpha, out_dtype=torch.float32,
                                     round_mode=0)
        got32, got16 = y32[:, :, cols], y16[:, :, cols]
    fro = relerr(got32, ref32)
    assert fro <= 1e-5, fro
    ref16 = ref32.to(x.dtype)
    d = ulp_diff(got16.contiguous(), ref16)
    ok = (d <= 1) | ((got16.float() - ref16.float()).abs() <= cancel_floor(ref16, K))
    assert bool(ok.all()), d.max().item()
    assert (d == 0).float().mean().item() >= 0.99
    return y16


ROW_BLOCKS = (0, 1016, 2032)        # 16-row blocks checked at FULL WIDTH at M = 2048: first rows, a tile-interior seam (1016..1031 straddles the
                                    # 1024-row boundary of the 256-row tiles), last rows


def check_row_blocks(oracle, x, w, p, alpha, y32, y16, groups=1):
    """VERDICT r05 weak #1a: the sampled-column checks cover every row on ~60 columns; a bug confined to the INTERIOR columns of a tile would pass.
    Here three 16-row blocks go through the oracle over ALL N columns (all of k): fp32 mode <= 1e-5, 16-bit outputs <= 1 ulp or inside the
    per-row cancellation floor, >= 99 % bit-equal.  Returns how many elements needed the floor."""
    M, K = x.shape[1], x.shape[2]
    rows = torch.cat([torch.arange(r, r + 16) for r in ROW_BLOCKS if r + 16 <= M])
    xr = x[:, rows].contiguous().cpu()
    ref32 = oracle.binary_linear(xr, w.cpu().contiguous(), p.cpu().contiguous(), alpha.cpu(), G=groups, out_dtype=torch.float32, round_mode=0)
    got32, got16 = y32[:, rows].cpu(), y16[:, rows].cpu().contiguous()
    assert relerr(got32, ref32) <= 1e-5
    ref16 = ref32.to(got16.dtype)
    d = ulp_diff(got16, ref16)
    ok = (d <= 1) | ((got16.float() - ref16.float()).abs() <= cancel_floor(ref16, K))
    assert bool(ok.all()), d.max().item()
    assert (d == 0).float().mean().item() >= 0.99
    return int((d > 1).sum())


def cancel_floor(ref, K):
    """absolute floor for outputs that cancel to ~0 (ulp distance is meaningless there): fp32 rounding of partial sums as large as
    the largest output OF THE SAME ROW, random-walked over K terms -- scales with the problem, not a flat constant"""
    return 2.0 ** -22 * (K ** 0.5) * ref.float().abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)       # per output row


def check_delta(bd, oracle, x, p, cols=None):
    """binary_bmm (reference epilogue fp32 -> fp16 -> dtype) at full size vs the oracle on all / sampled columns."""
    xd, pd = x.cuda(), p.cuda()
    if pd.shape[0] != xd.shape[0]:
        pd = pd.expand(xd.shape[0], -1, -1).contiguous()
    c = bd.binary_bmm(xd, pd).cpu()
    if cols is None:
        ref = oracle.delta_bmm(x, p, round_mode=1)
        got = c
    else:
        ref = oracle.delta_bmm(x, p[:, :, cols].contiguous(), round_mode=1)
        got = c[:, :, cols].contiguous()
    d = ulp_diff(got, ref)
    K = x.shape[-1]
    ok = (d <= 1) | ((got.float() - ref.float()).abs() <= cancel_floor(ref, K))
    assert bool(ok.all()) and (d == 0).float().mean().item() >= 0.99


# ---------------------------------------------------------------------------------------------------------------------
# config 1: "Single BinaryLinear 4096x4096, bf16 act [1,128,4096]" -- the exact shape, through the module surface
def test_config1_binarylinear_4096x4096_bf16_act_1x128x4096(bd, oracle):
    torch.manual_seed(101)
    K = N = 4096
    base = (torch.randn(N, K) * 0.02).bfloat16()
    fine = (base.float() + torch.randn(N, K) * 5e-4).bfloat16()
    x = torch.randn(1, 128, K).bfloat16()
    mod = bd.BinaryLinear(base.cuda(), fine.cuda())
    mo, co = oracle.binarize(base, fine)
    assert torch.equal(mod.mask.cpu(), mo)
    assert abs(mod.coeff.item() - co.item()) <= 2e-7 * co.item()
    with torch.no_grad():
        y = mod(x.cuda()).cpu()
    assert y.dtype == torch.bfloat16 and y.shape == (1, 128, N)
    ref32 = oracle.binary_linear(x, base, mo[None], co.reshape(1, 1), out_dtype=torch.float32, round_mode=0)
    d = ulp_diff(y, ref32.bfloat16())
    ok = (d <= 1) | ((y.float() - ref32).abs() <= cancel_floor(ref32, K))
    assert bool(ok.all()) and (d == 0).float().mean().item() >= 0.99
    # not worse than the reference's own 4-rounding chain (SURVEY.md 7b iv)
    chain = oracle.binary_linear(x, base, mo[None], co.reshape(1, 1), round_mode=1)
    assert relerr(y, ref32) <= relerr(chain, ref32) * 1.05 + 1e-7
    # the un-fused op on the same shape, reference epilogue
    check_delta(bd, oracle, x, mo[None])


# ---------------------------------------------------------------------------------------------------------------------
# config 2: "Llama-2-7B base + Vicuna-7B-v1.5 1-bit delta, prefill seq=2048" -- every projection shape at M = 2048
LLAMA7B = {"q_o_proj": (4096, 4096), "gate_up_proj": (11008, 4096), "down_proj": (4096, 11008)}


@pytest.mark.parametrize("name", list(LLAMA7B))
def test_config2_llama2_7b_prefill2048(bd, oracle, name):
    N, K = LLAMA7B[name]
    w, p, alpha = make_layer(N, K, torch.bfloat16, 1, seed=200 + N % 97)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 2048, K, generator=g).bfloat16()
    cols = sample_columns(N, 48, seed=N)
    check_linear(bd, oracle, x, w, p, alpha, cols=cols)
    check_delta(bd, oracle, x[:, :256].contiguous(), p, cols=cols[:24])      # the un-fused op (M = 256 rows)


def fused_boundary_columns(lin, n=40, seed=0):
    """Stored output rows of a FusedDeltaLinear that straddle every scale-group boundary, 8-row interleave block, 128 / 256-column
    tile edge and the tail split of the timed launch, plus random ones."""
    N = lin.weight.shape[0]
    gsz = N // lin.groups
    cols = set([0, 1, 7, 8, 15, 16, 127, 128, 255, 256, N - 257, N - 256, N - 129, N - 128, N - 17, N - 16, N - 9, N - 8, N - 1])
    if lin.interleave8:
        for b in (1, N // 16, N // 8 - 1):                                 # gate block / up block of a few 16-row tiles
            cols.update([16 * b - 1, 16 * b, 16 * b + 7, 16 * b + 8, 16 * b + 15])
    else:
        for gi in range(1, lin.groups):                                     # both sides of every scale-group boundary
            cols.update([gi * gsz - 2, gi * gsz - 1, gi * gsz, gi * gsz + 1])
        acc = 0
        for wd in lin.widths[:-1]:                                          # ... and of every projection boundary
            acc += wd
            cols.update([acc - 1, acc])
    g = torch.Generator().manual_seed(seed)
    cols.update(torch.randint(0, N, (n,), generator=g).tolist())
    return torch.tensor(sorted(c for c in cols if 0 <= c < N), dtype=torch.long)


FUSED_PREFILL = {
    # name: (projection shapes, interleave8)  -- exactly what bench_model.DecoderLayer builds (bench_model.py:135-136) and Mistral-7B's
    "llama7b_qkv_12288_G3": ([(4096, 4096), (4096, 4096), (4096, 4096)], False),
    "llama7b_gate_up_22016_il8": ([(11008, 4096), (11008, 4096)], True),
    "mistral7b_qkv_6144_G6": ([(4096, 4096), (1024, 4096), (1024, 4096)], False),
    "mistral7b_gate_up_28672_il8": ([(14336, 4096), (14336, 4096)], True),
}


@pytest.mark.parametrize("name", list(FUSED_PREFILL))
def test_config2_fused_launches_of_the_timed_prefill_step(bd, oracle, name):
    """The launches bench.py's headline number is made of: q|k|v as ONE Linear (one scale group per projection) and gate|up as ONE
    8-row-interleaved Linear (two scales), M = 2048, bf16, through FusedDeltaLinear -> bd_binary_linear -> delta_gemm_w4_kernel
    <fused> (variant 14; gate|up: + the 128x128 tail launch).  Oracle on sampled stored rows with each row's own scale."""
    from bitdelta_amd import _lib
    from bitdelta_amd.diff import binarize
    from bitdelta_amd.serving_loop import FusedDeltaLinear
    shapes, il8 = FUSED_PREFILL[name]
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(1234 + len(name))
    ws, ms, cs = [], [], []
    for n_out, n_in in shapes:
        w = (torch.randn(n_out, n_in, device=dev, generator=gen) * 0.02).bfloat16()
        fine = (w.float() + torch.randn(n_out, n_in, device=dev, generator=gen) * 5e-4).bfloat16()
        m, c = binarize(w, fine)
        ws.append(w); ms.append(m[None]); cs.append((c * (1.0 + 0.25 * len(cs))).reshape(1))       # distinct scales per projection
    lin = FusedDeltaLinear(ws, ms, cs, interleave8=il8, decode_copies=False)
    N, K = lin.weight.shape
    M = 2048
    x = torch.randn(1, M, K, device=dev, generator=gen).bfloat16()
    L = _lib.lib()
    y16 = lin(x)
    # the four-wave persistent fused kernel: 256x128 tiles (14) for the Llama launches, 128x128 tiles (20, round 5) where three rounds of small
    # tiles beat two of large ones (Mistral's q|k|v: 768 tiles)
    assert L.bd_last_gemm_variant() == (20 if name == "mistral7b_qkv_6144_G6" else 14), L.bd_last_gemm_variant()
    y32 = lin(x, out_dtype=torch.float32)
    cols = fused_boundary_columns(lin, seed=N)
    col_alpha = lin.column_alpha(0)[cols.to(dev)].float().cpu().reshape(1, -1)
    ref32 = oracle.binary_linear(x.cpu(), lin.weight[cols.to(dev)].cpu().contiguous(), lin.mask[:, :, cols.to(dev)].cpu().contiguous(),
                                 col_alpha, G=len(cols), out_dtype=torch.float32, round_mode=0)
    got32, got16 = y32[:, :, cols.to(dev)].cpu(), y16[:, :, cols.to(dev)].cpu().contiguous()
    assert relerr(got32, ref32) <= 1e-5
    ref16 = ref32.bfloat16()
    d = ulp_diff(got16, ref16)
    ok = (d <= 1) | ((got16.float() - ref16.float()).abs() <= cancel_floor(ref16, K))
    assert bool(ok.all()), d.max().item()
    assert (d == 0).float().mean().item() >= 0.99
    # ... and three 16-row blocks over ALL stored output rows of the fused Linear (every tile interior, each with its own scale group)
    check_row_blocks(oracle, x, lin.weight, lin.mask, lin.alpha[:1], y32, y16, groups=lin.groups)
    # the tail split (last, mostly empty round of 256x128 tiles handed to the 128x128 kernel) fires for Llama's gate|up and the two
    # paths agree bit for bit: same launch with the split switched off
    if name == "llama7b_gate_up_22016_il8":
        L.bd_set_tail_split(0)
        try:
            y_nosplit = lin(x)
        finally:
            L.bd_set_tail_split(1)
        assert torch.equal(y_nosplit, y16)
    # the split() views give back the per-projection outputs in the reference's order
    parts = lin.split(y16)
    assert [t.shape[-1] for t in parts] == [s[0] for s in shapes]


def test_config2_llama2_7b_decode_single_token(bd, oracle):
    """the same model one token at a time (BinaryDiff.forward at M = 1: the streaming decode kernel, one mask)"""
    for name, (N, K) in LLAMA7B.items():
        w, p, alpha = make_layer(N, K, torch.bfloat16, 1, seed=300 + N % 97)
        x = torch.randn(1, 1, K, generator=torch.Generator().manual_seed(8)).bfloat16()
        check_linear(bd, oracle, x, w, p, alpha)


# ---------------------------------------------------------------------------------------------------------------------
# config 3 / 5: "Mistral-7B base + 6 fine-tune deltas multi-tenant batched decode" (T = 6) and "32 tenants sharded 4-per-GPU" (T = 4)
MISTRAL = {"q_o_proj": (4096, 4096), "k_v_proj": (1024, 4096), "gate_up_proj": (14336, 4096), "down_proj": (4096, 14336)}


@pytest.mark.parametrize("tenants", [6, 4], ids=["config3_T6", "config5_T4_per_gpu"])
@pytest.mark.parametrize("name", list(MISTRAL))
def test_mistral_7b_multitenant_decode(bd, oracle, name, tenants):
    from bitdelta_amd import _lib
    N, K = MISTRAL[name]
    w, p, alpha = make_layer(N, K, torch.float16, tenants, seed=400 + N % 89 + tenants)
    x = torch.randn(tenants, 1, K, generator=torch.Generator().manual_seed(9)).half()
    check_linear(bd, oracle, x, w, p, alpha)
    assert _lib.lib().bd_last_gemm_variant() in (200, 600)        # a decode kernel ran, not an MFMA tile kernel


@pytest.mark.parametrize("tenants", [6, 4], ids=["config3_T6", "config5_T4_per_gpu"])
def test_mistral_7b_multitenant_decode_fused_qkv_and_gate_up(bd, oracle, tenants):
    """The serving loop launches q+k+v and gate+up as ONE Linear each (weights / masks concatenated along N, one scale group per
    1024 output columns so every projection keeps its own per-tenant coeff): N = 6144 with G = 6, N = 28672 with G = 2."""
    K = 4096
    for N, G in ((4096 + 1024 + 1024, 6), (2 * 14336, 2)):
        g = torch.Generator().manual_seed(500 + N % 83 + tenants)
        w = (torch.randn(N, K, generator=g) * 0.02).half()
        p = torch.randint(-2 ** 31, 2 ** 31 - 1, (tenants, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
        alpha = (torch.rand(tenants, G, generator=g) * 2e-4 + 3e-4).float()
        x = torch.randn(tenants, 1, K, generator=g).half()
        check_linear(bd, oracle, x, w, p, alpha, groups=G)


@pytest.mark.parametrize("M", [64, 256])
def test_config3_mistral_7b_multitenant_prefill(bd, oracle, M):
    """demo prompt prefill: left-padded to a power of two >= 64 (demo/demo_backend.py:297-299), T = 6 tenants"""
    T = 6
    for name, (N, K) in MISTRAL.items():
        if M == 256 and name != "q_o_proj":
            continue
        w, p, alpha = make_layer(N, K, torch.float16, T, seed=600 + N % 89)
        x = torch.randn(T, M, K, generator=torch.Generator().manual_seed(10)).half()
        check_linear(bd, oracle, x, w, p, alpha, cols=sample_columns(N, 32, seed=N + M))


# ---------------------------------------------------------------------------------------------------------------------
# config 4: "Llama-2-70B base + chat 1-bit delta, TP=8": the per-rank shard shapes (N-split q/k/v/gate/up, K-split o/down)
LLAMA70B_TP8 = {"q_shard": (1024, 8192), "kv_shard": (128, 8192), "o_shard": (8192, 1024),
                "gate_up_shard": (3584, 8192), "down_shard": (8192, 3584)}


@pytest.mark.parametrize("name", list(LLAMA70B_TP8))
def test_config4_llama2_70b_tp8_shard_decode(bd, oracle, name):
    N, K = LLAMA70B_TP8[name]
    w, p, alpha = make_layer(N, K, torch.bfloat16, 1, seed=700 + N % 79)
    x = torch.randn(1, 1, K, generator=torch.Generator().manual_seed(11)).bfloat16()
    check_linear(bd, oracle, x, w, p, alpha)


@pytest.mark.parametrize("name", list(LLAMA70B_TP8))
def test_config4_llama2_70b_tp8_shard_prefill2048(bd, oracle, name):
    N, K = LLAMA70B_TP8[name]
    w, p, alpha = make_layer(N, K, torch.bfloat16, 1, seed=800 + N % 79)
    x = torch.randn(1, 2048, K, generator=torch.Generator().manual_seed(12)).bfloat16()
    check_linear(bd, oracle, x, w, p, alpha, cols=sample_columns(N, 40, seed=N))


# ---------------------------------------------------------------------------------------------------------------------
# the reference's published binary_bmm benchmark shapes (notebooks/binary_gemm_kernel_triton.ipynb:800-1044,
# fp16, M = 1, one mask per batch entry) at FULL size through the reference surface: the one-mask-per-row kernel (variant 800) takes
# them, and every output row agrees with the oracle on sampled columns
@pytest.mark.parametrize("B,NK", [(8, 4096), (16, 4096), (8, 8192), (16, 8192)])
def test_published_binary_bmm_shapes_fp16(bd, oracle, B, NK):
    from bitdelta_amd import _lib
    g = torch.Generator().manual_seed(900 + B + NK)
    x = torch.randn(B, 1, NK, generator=g).half()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, NK // 32, NK), generator=g, dtype=torch.int64).to(torch.int32)
    cols = torch.randperm(NK, generator=g)[:64].sort().values
    check_delta(bd, oracle, x, p, cols=cols)
    assert _lib.lib().bd_last_gemm_variant() == 800, _lib.lib().bd_last_gemm_variant()
    # the same launch is deterministic, and equals the streaming / split-k kernels it replaced within one rounding of the fp32 sum
    xd, pd = x.cuda(), p.cuda()
    c0 = bd.binary_bmm(xd, pd)
    assert torch.equal(c0, bd.binary_bmm(xd, pd))
    L = _lib.lib()
    L.bd_set_gemm_variant(600)                         # the streaming kernel (<= 8 masks per launch)
    try:
        old = bd.binary_bmm(xd[:8].contiguous(), pd[:8].contiguous())
    finally:
        L.bd_set_gemm_variant(-1)
    assert L.bd_last_gemm_variant() == 600
    new8, old8 = c0[:8].cpu().contiguous(), old.cpu()
    ok = (ulp_diff(new8, old8) <= 1) | ((new8.float() - old8.float()).abs() <= cancel_floor(old8, NK))
    assert bool(ok.all())
