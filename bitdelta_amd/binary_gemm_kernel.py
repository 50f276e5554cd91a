"""MI355X drop-in for the reference's ``bitdelta/binary_gemm_kernel.py``.

Same public names, signatures, asserts and result semantics:

    pack(x, n_bits=32)                          reference :6-32
    unpack(x, n_bits=32)                        reference :34-46
    binary_matmul(a, b, n_bits=32, activation="")   reference :153-184 (kernel :48-151)
    binary_bmm(a, b, n_bits=32, activation="")      reference :297-335 (kernel :186-295)

Every function launches hand-written HIP (gfx950) through the C ABI in include/bitdelta_hip.h; there is no
Triton, no torch-op composition and no CPU path.  Extra keyword-only arguments (``round_mode``, ``out_dtype``) expose
what the reference hard-codes; their defaults reproduce the reference.
"""
import torch

from . import _lib
from ._lib import DTYPE_CODE, WORD_DTYPE, check, lib, ptr, require_gpu, stream_ptr, workspace

__all__ = ["pack", "unpack", "binary_matmul", "binary_bmm", "delta_bmm", "binary_linear", "binary_linear_residual_norm", "tenant_linear", "tile_masks",
           "pack_decode_masks", "binary_linear_decode", "decode_shape_ok"]


def pack(x, n_bits=32):
    """
    pack n_bits of x into a single integer

    x: bool tensor (*, K, N)
    return: int tensor (*, K // n_bits, N)
    """
    assert x.shape[-2] % n_bits == 0, "K must be divisible by n_bits"
    if n_bits not in WORD_DTYPE:
        # reference: `dtype` is never bound for other widths (binary_gemm_kernel.py:23-32)
        raise UnboundLocalError("local variable 'dtype' referenced before assignment")
    require_gpu(x)
    if x.dtype != torch.bool:
        x = x != 0
    lead = tuple(x.shape[:-2])
    K, N = x.shape[-2], x.shape[-1]
    if x.dim() == 2:
        x3 = x.unsqueeze(0)
    elif x.dim() == 3:
        x3 = x
    else:
        x3 = x.reshape(-1, K, N)
    if x.numel() == 0:
        return torch.empty((*lead, K // n_bits, N), dtype=WORD_DTYPE[n_bits], device=x.device)
    out = torch.empty((x3.shape[0], K // n_bits, N), dtype=WORD_DTYPE[n_bits], device=x.device)
    with torch.cuda.device(x.device):
        check(lib().bd_pack(ptr(x3), x3.shape[0], K, N, x3.stride(0), x3.stride(1), x3.stride(2), ptr(out), n_bits,
                            stream_ptr()), "pack")
    return out.view(*lead, K // n_bits, N)


def unpack(x, n_bits=32):
    """
    unpack n_bits of x into a single integer

    x: int tensor (*, K // n_bits, N)
    return: bool tensor (*, K, N)
    """
    if n_bits not in WORD_DTYPE:
        raise ValueError("n_bits must be 8, 16, 32 or 64")
    require_gpu(x)
    if x.dtype != WORD_DTYPE[n_bits]:
        # the reference shifts whatever integer dtype it is given; only the low n_bits matter
        x = x.to(WORD_DTYPE[n_bits])
    lead = tuple(x.shape[:-2])
    KW, N = x.shape[-2], x.shape[-1]
    if x.numel() == 0:
        return torch.empty((*lead, KW * n_bits, N), dtype=torch.bool, device=x.device)
    xc = x.contiguous().view(-1, KW, N)
    out = torch.empty((xc.shape[0], KW * n_bits, N), dtype=torch.bool, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().bd_unpack(ptr(xc), xc.shape[0], KW, N, ptr(out), n_bits, stream_ptr()), "unpack")
    return out.view(*lead, KW * n_bits, N)


def delta_bmm(a, b, *, out=None, out_dtype=None, round_mode=1, alpha=None, accumulate=False, groups=1):
    """C[i] = a[i] . (2*unpack(b[i]) - 1)  (+ optional fused ``alpha *`` / ``C +=`` epilogue).

    a: (B, M, K) fp16/bf16, rows k-contiguous.  b: (B, K/32, N) or (1, K/32, N) int32 contiguous -- a leading 1
    broadcasts ONE mask over the batch (what bitdelta/diff.py:38 materialises with ``mask.repeat``).
    alpha: fp32 tensor of shape (B or 1, groups) or None.
    """
    require_gpu(a, b, alpha, out)
    assert a.dim() == 3 and b.dim() == 3
    B, M, K = a.shape
    N = b.shape[2]
    assert b.dtype == torch.int32 and b.is_contiguous()
    assert b.shape[0] in (1, B) and b.shape[1] * 32 == K
    assert a.dtype in (torch.float16, torch.bfloat16) and a.stride(2) == 1
    out_dtype = out_dtype or a.dtype
    if out is None:
        assert not accumulate
        out = torch.empty((B, M, N), device=a.device, dtype=out_dtype)
    else:
        assert out.shape == (B, M, N) and out.dtype == out_dtype and out.stride(2) == 1
    sPb = 0 if (b.shape[0] == 1 and B > 1) else b.stride(0)
    al_ptr, sAlb = ptr(None), 0
    if alpha is not None:
        alpha = alpha.detach()
        if alpha.dtype != torch.float32 or not alpha.is_contiguous():
            alpha = alpha.float().contiguous()
        alpha = alpha.reshape(-1, groups)
        assert alpha.shape[0] in (1, B)
        al_ptr, sAlb = ptr(alpha), (0 if alpha.shape[0] == 1 else groups)
    L = lib()
    ws, ws_bytes = workspace(L.bd_gemm_workspace_bytes(B, M, N, K), a.device, zeroed=True)
    with torch.cuda.device(a.device):
        check(L.bd_delta_bmm(ptr(a), ptr(b), ptr(out), B, M, N, K, a.stride(0), a.stride(1), sPb, out.stride(0),
                             out.stride(1), DTYPE_CODE[a.dtype], DTYPE_CODE[out_dtype], int(round_mode), al_ptr, sAlb,
                             groups, 1 if accumulate else 0, ptr(ws), ws_bytes, stream_ptr()), "delta_bmm")
    return out


def binary_linear(x, weight, mask, alpha, *, out_dtype=None, groups=1, residual=None, out=None):
    """Fused 16-bit base + 1-bit delta Linear:  y[i] = x[i] . weight^T + alpha[i] * (x[i] . S[i])  in ONE launch.

    x: (B, M, K); weight: (N, K) rows k-contiguous; mask: (B or 1, K/32, N) int32; alpha: fp32 (B or 1, groups).
    Replaces the four launches of BinaryDiff.forward (bitdelta/diff.py:38-39) / DiffCompressModule.forward
    (demo/demo_backend.py:95-98).  fp32 accumulation, one rounding to ``out_dtype`` (default: x.dtype).
    residual: optional (B, M, N) tensor of out_dtype that is updated IN PLACE to residual + y and returned (the decoder layer's
    `hidden = residual + proj(...)`): folded into the kernel epilogue at decode shapes (fp32 sum, one rounding) and on the fast
    path of the fused GEMM at M > 16 (output rounded, then the sum: the two roundings of the separate ops); a separate add otherwise.
    out: optional (B, M, N) destination of out_dtype with unit column stride (e.g. a peer-mapped all-reduce buffer: the row-parallel
    Linear of tp.py writes its fp32 partial sums straight into it); not combined with `residual`.
    """
    require_gpu(x, weight, mask, alpha, residual, out)
    assert x.dim() == 3 and mask.dim() == 3 and weight.dim() == 2
    B, M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and weight.stride(1) == 1 and weight.dtype == x.dtype
    assert mask.dtype == torch.int32 and mask.is_contiguous() and mask.shape[1] * 32 == K and mask.shape[2] == N
    assert mask.shape[0] in (1, B) and x.stride(2) == 1
    out_dtype = out_dtype or x.dtype
    alpha = alpha.detach()
    if alpha.dtype != torch.float32 or not alpha.is_contiguous():
        alpha = alpha.float().contiguous()
    alpha = alpha.reshape(-1, groups)
    assert alpha.shape[0] in (1, B)
    sPb = 0 if (mask.shape[0] == 1 and B > 1) else mask.stride(0)
    sAlb = 0 if alpha.shape[0] == 1 else groups
    L = lib()
    aligned = (x.data_ptr() % 16 == 0 and x.stride(0) % 8 == 0 and x.stride(1) % 8 == 0 and weight.data_ptr() % 16 == 0 and
               weight.stride(0) % 8 == 0)
    fused_residual = residual is not None and aligned and ((M <= 16 and B * M <= 64 and K % 32 == 0) or (M > 16 and K % 64 == 0))
    if residual is not None:
        assert residual.shape == (B, M, N) and residual.dtype == out_dtype and residual.stride(2) == 1
    if out is not None:
        assert residual is None, "out= and residual= are exclusive"
        assert out.shape == (B, M, N) and out.dtype == out_dtype and out.stride(2) == 1 and out.device == x.device
        y = out
    else:
        y = residual if fused_residual else torch.empty((B, M, N), device=x.device, dtype=out_dtype)
    fn = L.bd_binary_linear_residual if fused_residual else L.bd_binary_linear
    ws, ws_bytes = workspace(L.bd_gemm_workspace_bytes(B, M, N, K), x.device, zeroed=True)
    with torch.cuda.device(x.device):
        check(fn(ptr(x), ptr(weight), ptr(mask), ptr(alpha), ptr(y), B, M, N, K, x.stride(0),
                 x.stride(1), weight.stride(0), sPb, sAlb, groups, y.stride(0), y.stride(1),
                 DTYPE_CODE[x.dtype], DTYPE_CODE[out_dtype], ptr(ws), ws_bytes, stream_ptr()),
              "binary_linear")
    if residual is not None and not fused_residual:
        residual += y
        return residual
    return y


def binary_linear_residual_norm(x, weight, mask, alpha, residual, norm_weight, eps, *, groups=1):
    """binary_linear(..., residual=residual) at prefill size (M > 16) TOGETHER WITH the per-tenant RMSNorm that follows it in the decoder
    layer: returns (residual, h) -- residual updated in place to residual + Linear(x), h = rmsnorm_tenant(residual, norm_weight, eps) --
    bit-identical to the two calls.  One launch fewer when the Linear is split over k (several tenants of <= 64 rows: the o / down projections
    of a short-prompt request), where the norm rides on the split-k reduce.  norm_weight [B, N].  Raises BitDeltaHipError(BD_E_BAD_SHAPE)
    outside the envelope (M > 16, N % 8 == 0, N <= 8192, contiguous residual)."""
    require_gpu(x, weight, mask, alpha, residual, norm_weight)
    assert x.dim() == 3 and mask.dim() == 3 and weight.dim() == 2
    B, M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and weight.stride(1) == 1 and weight.dtype == x.dtype
    assert mask.dtype == torch.int32 and mask.is_contiguous() and mask.shape[1] * 32 == K and mask.shape[2] == N
    assert mask.shape[0] in (1, B) and x.stride(2) == 1
    assert residual.shape == (B, M, N) and residual.dtype == x.dtype and residual.is_contiguous()
    assert norm_weight.shape == (B, N) and norm_weight.dtype == x.dtype and norm_weight.stride(1) == 1
    alpha = alpha.detach()
    if alpha.dtype != torch.float32 or not alpha.is_contiguous():
        alpha = alpha.float().contiguous()
    alpha = alpha.reshape(-1, groups)
    assert alpha.shape[0] in (1, B)
    sPb = 0 if (mask.shape[0] == 1 and B > 1) else mask.stride(0)
    sAlb = 0 if alpha.shape[0] == 1 else groups
    L = lib()
    h = torch.empty_like(residual)
    ws, ws_bytes = workspace(L.bd_gemm_workspace_bytes(B, M, N, K), x.device, zeroed=True)
    with torch.cuda.device(x.device):
        check(L.bd_binary_linear_residual_norm(ptr(x), ptr(weight), ptr(mask), ptr(alpha), ptr(residual), B, M, N, K, x.stride(0), x.stride(1),
                                               weight.stride(0), sPb, sAlb, groups, residual.stride(0), residual.stride(1), DTYPE_CODE[x.dtype],
                                               ptr(norm_weight), norm_weight.stride(0), float(eps), ptr(h), h.stride(0), h.stride(1),
                                               ptr(ws), ws_bytes, stream_ptr()), "binary_linear_residual_norm")
    return residual, h


def binary_linear_swiglu(x, weight, mask, alpha):
    """Prefill-size fused gate|up projection with SwiGLU in its epilogue (bd_binary_linear_swiglu): weight / mask describe the two
    BinaryDiff projections `gate_proj` and `up_proj` stored as ONE projection with the output rows interleaved in blocks of 8
    (serving_loop.FusedDeltaLinear(interleave8=True)), alpha (B or 1, 2) = (gate, up) scales.  Returns (B, M, N/2):
    round(silu(round(gate))) * round(up) -- bit-identical to binary_linear followed by serving_ops.swiglu, one launch and no
    [M, N] intermediate.  Raises BitDeltaHipError(BD_E_BAD_SHAPE) outside its envelope (M > 16, N % 16 == 0, K % 64 == 0, aligned rows)."""
    require_gpu(x, weight, mask, alpha)
    assert x.dim() == 3 and mask.dim() == 3 and weight.dim() == 2
    B, M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and weight.stride(1) == 1 and weight.dtype == x.dtype
    assert mask.dtype == torch.int32 and mask.is_contiguous() and mask.shape[1] * 32 == K and mask.shape[2] == N
    assert mask.shape[0] in (1, B) and x.stride(2) == 1 and N % 16 == 0
    alpha = alpha.detach()
    if alpha.dtype != torch.float32 or not alpha.is_contiguous():
        alpha = alpha.float().contiguous()
    alpha = alpha.reshape(-1, 2)
    assert alpha.shape[0] in (1, B)
    sPb = 0 if (mask.shape[0] == 1 and B > 1) else mask.stride(0)
    y = torch.empty((B, M, N // 2), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        check(lib().bd_binary_linear_swiglu(ptr(x), ptr(weight), ptr(mask), ptr(alpha), ptr(y), B, M, N, K, x.stride(0), x.stride(1),
                                            weight.stride(0), sPb, 0 if alpha.shape[0] == 1 else 2, y.stride(0), y.stride(1),
                                            DTYPE_CODE[x.dtype], stream_ptr()), "binary_linear_swiglu")
    return y


def tile_masks(mask):
    """Repack packed sign words [T, K/32, N] (the reference / diff.pt layout) into the tile-major order of the streaming decode
    kernel, [T, ceil(N/16), K/32, 16]: every 16-column tile's words become one contiguous run over k.  Done once per registered
    tenant set on the serving side; columns past N are zero padding."""
    assert mask.dim() == 3 and mask.dtype == torch.int32
    T, KW, N = mask.shape
    Np = (N + 15) // 16 * 16
    if Np != N:
        mask = torch.nn.functional.pad(mask, (0, Np - N))
    return mask.view(T, KW, Np // 16, 16).permute(0, 2, 1, 3).contiguous()


def tile_weight(weight):
    """Decode copy of a base weight [N, K] (N % 16 == 0, K % 128 == 0) in the streaming kernel's TILE-MAJOR order
    [N/16][K/128][4 steps s][16 rows c][4 groups g][8]  with  W'[tile][it][s][c][g][e] = W[16 tile + c][128 it + 32 s + 8 g + e]:
    one (16-column tile, 128-k iteration) stage is ONE contiguous 4-KiB block read as four 1-KiB runs, and a wave's consecutive
    stages are consecutive blocks (row-major: 16 rows 2K bytes apart per load instruction).  Same values, same arithmetic; returned
    with shape [N, K] (a flat reinterpretation) so it can stand in for `weight` in binary_linear_decode(..., weight_tiled=True)."""
    N, K = weight.shape
    assert N % 16 == 0 and K % 128 == 0 and weight.stride(1) == 1
    return weight.reshape(N // 16, 16, K // 128, 4, 4, 8).permute(0, 2, 3, 1, 4, 5).contiguous().view(N, K)


def pack_decode_masks(mask):
    """Repack packed sign words [T, K/32, N] (reference / diff.pt layout) into the PACKED layout of the streaming decode kernel:
    int32 [ceil(N/16), ceil(K/128), 4, 16, t_pad] with element [tile][it][g][c][t] = tenant t's dword whose byte s holds the 8 signs
    of k = 128 it + 32 s + 8 g .. + 7 of column 16 tile + c; tenants interleaved, zero-padded to t_pad in {1, 2, 4, 6, 8, 12, 16}
    (up to 16 tenants in one launch: the reference's own batched benchmark runs B = 16, notebooks/binary_gemm_kernel_triton.ipynb:759).
    Exact byte shuffling (torch ops on the device), done once per registered tenant set."""
    assert mask.dim() == 3 and mask.dtype == torch.int32
    T, KW, N = mask.shape
    assert T <= 16, "the packed decode layout holds at most 16 tenants per call"
    tp = next(v for v in (1, 2, 4, 6, 8, 12, 16) if v >= T)
    KW4, N16 = (KW + 3) // 4 * 4, (N + 15) // 16 * 16
    if KW4 != KW or N16 != N:
        mask = torch.nn.functional.pad(mask, (0, N16 - N, 0, KW4 - KW))
    # bytes: [t][it][s][tile][c][g]   (byte g of word row 4 it + s covers k = 128 it + 32 s + 8 g .. + 7; little-endian)
    b = mask.contiguous().view(torch.uint8).view(T, KW4 // 4, 4, N16 // 16, 16, 4)
    b = b.permute(3, 1, 5, 4, 0, 2).contiguous()                       # [tile][it][g][c][t][s]
    out = b.view(torch.int32).view(N16 // 16, KW4 // 4, 4, 16, T)      # the 4 bytes over s form one dword
    if tp != T:
        out = torch.nn.functional.pad(out, (0, tp - T))
    return out.contiguous()


def decode_shape_ok(B, M, N, K, n_masks, layout="packed"):
    """True when bd_binary_linear_decode accepts the problem (the streaming decode kernel's envelope): up to 16 activation rows, and up
    to 16 masks in the packed layout (8 in the tile-major one)."""
    if M < 1 or M > 16 or N < 512 or K % 32:
        return False
    chunk = min(B, 16 // M)
    return (1 if n_masks == 1 else chunk) <= (16 if layout == "packed" else 8)


def fused_norm_ok(B, M, K):
    """envelope of the fused RMSNorm prologue of bd_binary_linear_decode_fused (include/bitdelta_hip.h): at most 8 tenants -- the kernel
    side (BD_PKL in csrc/bd_api.hip) answers BD_E_BAD_SHAPE for a norm_w launch on the 12 / 16-tenant packed layouts"""
    return M == 1 and B <= 8 and K >= 2048 and K & (K - 1) == 0 and B * K <= 16 * 2048 and 83968 + B * (2 * K + 16) <= 160 * 1024


def handoff_ok(B, M, K):
    """envelope of the RMSNorm hand-off consumer of bd_binary_linear_decode_handoff (include/bitdelta_hip.h): the resident-row form"""
    nit = (K + 127) // 128
    return (M == 1 and B <= 8 and K >= 2048 and K % 128 == 0 and 3 * ((nit + 3) // 4) < nit and B * K <= 16 * 2048 and
            83968 + B * (2 * K + 16) <= 160 * 1024)


def binary_linear_decode(x, weight, mask, alpha, *, layout="tile", out_dtype=None, groups=1, residual=None,
                         norm_weight=None, eps=1e-5, swiglu=False, weight_tiled=False, out=None, ssq_in=None, ssq_out=None,
                         xw_out=None, ssq_scale=1.0):
    """binary_linear for decode shapes with repacked masks: one launch of the streaming kernel.
    x: (B, M, K), M <= 16; weight (N, K); alpha fp32 (B or 1, groups);
    layout "tile":   mask = tile_masks(...)        (B or 1, ceil(N/16), K/32, 16)
    layout "packed": mask = pack_decode_masks(...) (ceil(N/16), ceil(K/128), 4, 16, t_pad), B <= t_pad tenants, B*M <= 16.
    Packed layout only: norm_weight (B or 1, K) fuses the HF RMSNorm of x (x = the un-normalised residual stream) into the launch;
    weight_tiled=True: `weight` is tile_weight(W) (M == 1, N % 16 == 0, K % 128 == 0);
    swiglu=True (with or without norm_weight) treats weight/mask as a gate|up pair interleaved in blocks of 8 output rows, alpha (B or 1, 2) =
    (gate, up) scales, and returns act_fn(gate) * up, (B, M, N/2).  Both bit-identical to the separate launches.
    RMSNorm hand-off (bd_binary_linear_decode_handoff; packed layout, M == 1, B <= 8, tile-major weight):
      ssq_out: fp32 (N/16, 16) buffer -- the launch (a residual Linear: o_proj / down_proj) also writes its output's per-row sums of squares,
               16 columns at a time; with xw_out (B, M, N) and norm_weight = the weight of the RMSNorm that FOLLOWS, also the pre-multiplied
               copy round(y * norm_weight); ssq_scale multiplies the written sums (serving_loop.handoff_norm: norm_weight = nw / s, ssq_scale = 1 / s^2,
               consumer eps / s^2 -- overflow-safe in fp16, same product);
      ssq_in:  the buffer the PREVIOUS launch filled -- x is that launch's xw_out, 1/rms scales the accumulators: no stand-alone norm launch,
               no per-block reduction, no multiply (`handoff_ok`); norm_weight must be None."""
    require_gpu(x, weight, mask, alpha, residual, norm_weight)
    B, M, K = x.shape
    N = weight.shape[0]
    assert mask.dtype == torch.int32 and mask.is_contiguous()
    if layout == "tile":
        assert mask.dim() == 4 and mask.shape[1:] == ((N + 15) // 16, K // 32, 16) and mask.shape[0] in (1, B)
        code, t_pad = 1, 0
        sPb = 0 if (mask.shape[0] == 1 and B > 1) else mask.stride(0)
    else:
        assert layout == "packed" and mask.dim() == 5
        assert mask.shape[:4] == ((N + 15) // 16, (K + 127) // 128, 4, 16) and B <= mask.shape[4] and B * M <= 16
        code, t_pad = 2, mask.shape[4]
        sPb = 1
    assert weight.shape[1] == K and weight.stride(1) == 1 and weight.dtype == x.dtype and x.stride(2) == 1
    out_dtype = out_dtype or x.dtype
    ldw = weight.stride(0)
    if weight_tiled:
        assert layout == "packed" and M == 1 and N % 16 == 0 and K % 128 == 0 and weight.is_contiguous()
        ldw = 0                                     # the C ABI's marker for the tile-major decode copy
    alpha = alpha.detach()
    if alpha.dtype != torch.float32 or not alpha.is_contiguous():
        alpha = alpha.float().contiguous()
    alpha = alpha.reshape(-1, groups)
    assert alpha.shape[0] in (1, B)
    sAlb = 0 if alpha.shape[0] == 1 else groups
    s_norm = 0
    handoff = ssq_in is not None or ssq_out is not None
    if handoff:
        require_gpu(ssq_in, ssq_out, xw_out)
        assert layout == "packed" and M == 1 and B <= 8 and (ssq_in is None or ssq_out is None)
        for t_, cols in ((ssq_in, K), (ssq_out, N)):
            assert t_ is None or (t_.dtype == torch.float32 and t_.is_contiguous() and t_.shape == (cols // 16, 16) and cols % 16 == 0)
        if ssq_in is not None:        # x = the producer's pre-multiplied copy: no norm weight here
            assert weight_tiled and handoff_ok(B, M, K) and xw_out is None and norm_weight is None and K <= 8192
        if ssq_out is not None:
            assert not swiglu and out_dtype == x.dtype and (xw_out is None) == (norm_weight is None)
            if xw_out is not None:          # norm_weight = the NEXT norm's weight, over this launch's OUTPUT columns
                assert xw_out.shape == (B, M, N) and xw_out.dtype == x.dtype and xw_out.stride(2) == 1
                assert norm_weight.dim() == 2 and norm_weight.shape[1] == N and norm_weight.shape[0] in (1, B)
                assert norm_weight.dtype == x.dtype and norm_weight.stride(1) == 1
    else:
        assert xw_out is None
    if (norm_weight is not None and ssq_out is None) or swiglu:
        assert layout == "packed" and M == 1
        if norm_weight is not None:
            assert fused_norm_ok(B, M, K)
            assert norm_weight.dim() == 2 and norm_weight.shape[1] == K and norm_weight.shape[0] in (1, B)
            assert norm_weight.dtype == x.dtype and norm_weight.stride(1) == 1
            s_norm = 0 if (norm_weight.shape[0] == 1 and B > 1) else norm_weight.stride(0)
        assert not swiglu or (groups == 2 and N % 16 == 0 and residual is None and out_dtype == x.dtype)
    if residual is not None:
        assert out is None and residual.shape == (B, M, N) and residual.dtype == out_dtype and residual.stride(2) == 1
        y = residual
    elif out is not None:                # caller-provided destination (any row / batch stride, unit column stride)
        require_gpu(out)
        assert out.shape == (B, M, N // 2 if swiglu else N) and out.dtype == out_dtype and out.stride(2) == 1
        y = out
    else:
        y = torch.empty((B, M, N // 2 if swiglu else N), device=x.device, dtype=out_dtype)
    if handoff:
        if xw_out is not None:
            assert xw_out.stride(0) == y.stride(0) and xw_out.stride(1) == y.stride(1), "xw_out uses the output's strides"
            s_norm = 0 if (norm_weight.shape[0] == 1 and B > 1) else norm_weight.stride(0)
        with torch.cuda.device(x.device):
            check(lib().bd_binary_linear_decode_handoff(ptr(x), ptr(weight), ptr(mask), t_pad, ptr(alpha), ptr(y), B, M, N, K,
                                                        x.stride(0), x.stride(1), ldw, sPb, sAlb, groups, y.stride(0),
                                                        y.stride(1), DTYPE_CODE[x.dtype], DTYPE_CODE[out_dtype],
                                                        1 if residual is not None else 0, ptr(norm_weight), s_norm,
                                                        # (producer launches: the eps slot carries the factor on the written sums of squares)
                                                        float(ssq_scale) if ssq_out is not None else float(eps),
                                                        1 if swiglu else 0, ptr(ssq_in), ptr(ssq_out), ptr(xw_out), stream_ptr()),
                  "binary_linear_decode_handoff")
        return y
    if norm_weight is not None or swiglu:
        with torch.cuda.device(x.device):
            check(lib().bd_binary_linear_decode_fused(ptr(x), ptr(weight), ptr(mask), t_pad, ptr(alpha), ptr(y), B, M, N, K,
                                                      x.stride(0), x.stride(1), ldw, sPb, sAlb, groups, y.stride(0),
                                                      y.stride(1), DTYPE_CODE[x.dtype], DTYPE_CODE[out_dtype],
                                                      1 if residual is not None else 0, ptr(norm_weight), s_norm, float(eps),
                                                      1 if swiglu else 0, stream_ptr()), "binary_linear_decode_fused")
        return y
    with torch.cuda.device(x.device):
        check(lib().bd_binary_linear_decode(ptr(x), ptr(weight), ptr(mask), code, t_pad, ptr(alpha), ptr(y), B, M, N, K,
                                            x.stride(0), x.stride(1), ldw, sPb, sAlb, groups, y.stride(0),
                                            y.stride(1), DTYPE_CODE[x.dtype], DTYPE_CODE[out_dtype],
                                            1 if residual is not None else 0, stream_ptr()), "binary_linear_decode")
    return y


def tenant_linear(x, weights, *, out_dtype=None):
    """Per-tenant dense Linear:  y[t] = x[t] . weights[t]^T   (row block t uses tenant t's own matrix).

    x: (T, M, K); weights: (T, N, K), rows k-contiguous.  The batched counterpart of the weight-swapping loop in the reference's
    DataParallelModule.forward (demo/demo_backend.py:69-79) for nn.Linear leaves such as per-tenant lm_heads.  Decode shapes
    (M <= 16) stream every tenant's weights once in ONE HIP launch; larger M is an ordinary batched GEMM (torch.bmm / rocBLAS).
    """
    require_gpu(x, weights)
    assert x.dim() == 3 and weights.dim() == 3 and x.shape[0] == weights.shape[0] and x.shape[2] == weights.shape[2]
    assert x.dtype == weights.dtype and x.dtype in (torch.float16, torch.bfloat16)
    T, M, K = x.shape
    N = weights.shape[1]
    out_dtype = out_dtype or x.dtype
    if M > 16 or K % 32 or T == 0 or N == 0:
        return torch.bmm(x, weights.transpose(1, 2)).to(out_dtype)
    if x.stride(2) != 1 or x.stride(1) % 8 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    if weights.stride(2) != 1 or weights.stride(1) % 8 or weights.stride(0) % 8 or weights.data_ptr() % 16:
        weights = weights.contiguous()
    y = torch.empty((T, M, N), device=x.device, dtype=out_dtype)
    with torch.cuda.device(x.device):
        check(lib().bd_tenant_linear(ptr(x), ptr(weights), ptr(y), T, M, N, K, x.stride(0), x.stride(1), weights.stride(0),
                                     weights.stride(1), y.stride(0), y.stride(1), DTYPE_CODE[x.dtype], DTYPE_CODE[out_dtype],
                                     stream_ptr()), "tenant_linear")
    return y


def _words32(b, n_bits):
    """Packed sign words of any reference width as int32 words [.., K/32, N] with the SAME bit <-> k mapping: the reference kernels
    read bit (k % n_bits) of word row k // n_bits (bitdelta/binary_gemm_kernel.py:109-111, :128 / :251-253, :270); regrouping the
    rows into 32-bit words keeps k = 32 i + j <-> bit j of word i.  A re-layout of the operand (torch integer ops on the device,
    exact), not a compute path: the GEMM itself always runs on the int32 kernel."""
    if n_bits == 32:
        assert b.dtype == torch.int32, "n_bits = 32 expects int32 words"
        return b
    assert n_bits in (8, 16, 64), "n_bits must be 8, 16, 32 or 64"
    assert b.dtype == WORD_DTYPE[n_bits], f"n_bits = {n_bits} expects {WORD_DTYPE[n_bits]} words"
    KW, N = b.shape[-2], b.shape[-1]
    K = KW * n_bits
    assert K % 32 == 0, "K must be a multiple of 32"
    lead = b.shape[:-2]
    if n_bits == 64:
        lo = (b & 0xffffffff).to(torch.int64)
        hi = (b >> 32) & 0xffffffff
        w = torch.stack([lo, hi], dim=-2).reshape(*lead, KW * 2, N)                     # word i of width 64 -> rows 2i, 2i+1
    else:
        per = 32 // n_bits
        mask = (1 << n_bits) - 1
        u = (b.to(torch.int64) & mask).reshape(*lead, KW // per, per, N)
        sh = torch.arange(per, device=b.device, dtype=torch.int64).view(*([1] * len(lead)), 1, per, 1) * n_bits
        w = (u << sh).sum(dim=-2)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)                                         # wrap to int32 (bit 31 -> negative)
    return w.to(torch.int32).contiguous()


def binary_matmul(a, b, n_bits=32, activation="", *, round_mode=1, out_dtype=None):
    """
        a: float tensor (M, K)
        b: int tensor (K, N)
        n_bits: int, number of bits that each element in b represents
    """
    # Check constraints (same as the reference, binary_gemm_kernel.py:159-162).
    assert a.shape[1] == b.shape[0] * n_bits, "Incompatible dimensions"
    assert a.is_contiguous(), "Matrix A must be contiguous"
    assert b.is_contiguous(), "Matrix B must be contiguous"
    # `activation` is accepted and ignored, exactly like the reference (:73, :141-142)
    return delta_bmm(a.unsqueeze(0), _words32(b, n_bits).unsqueeze(0), round_mode=round_mode, out_dtype=out_dtype)[0]


def binary_bmm(a, b, n_bits=32, activation="", *, round_mode=1, out_dtype=None):
    """
        a: float tensor (B, M, K)
        b: int tensor (B, K, N)
        n_bits: int, number of bits that each element in b represents
    """
    assert a.dim() == 3, "Matrix A must be 3D"
    assert b.dim() == 3, "Matrix B must be 3D"
    assert a.shape[2] == b.shape[1] * n_bits, "Incompatible dimensions"
    assert a.shape[0] == b.shape[0], "Incompatible batch dimensions"
    assert a.is_contiguous(), "Matrix A must be contiguous"
    assert b.is_contiguous(), "Matrix B must be contiguous"
    assert a.device == b.device, "A and B must be on the same device"
    # output: non-differentiable tensor of a.dtype, fp32 accumulate -> fp16 -> a.dtype (reference :287, :314)
    return delta_bmm(a, _words32(b, n_bits), round_mode=round_mode, out_dtype=out_dtype)
