"""ctypes front-end of the CPU oracle (oracle/bd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (bitdelta_amd/) must never import this module.

All functions take/return torch CPU tensors (torch is only the buffer holder here).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BD_ORACLE_SO: an alternative build of the same source (tests/test_oracle_sanitized.py loads the ASan + UBSan build this way)
_SO = os.environ.get("BD_ORACLE_SO") or os.path.join(_HERE, "libbd_oracle.so")
_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
_WORD_DT = {8: torch.uint8, 16: torch.int16, 32: torch.int32, 64: torch.int64}


def build(force=False):
    src = os.path.join(_HERE, "bd_oracle.c")
    if os.environ.get("BD_ORACLE_SO"):
        return _SO                      # built by whoever set the override
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        L.bdo_pack.argtypes = [vp, i64, i64, i64, i64, i64, i64, vp, ci]
        L.bdo_unpack.argtypes = [vp, i64, i64, i64, ci, vp]
        L.bdo_delta_bmm.argtypes = [vp, vp, vp] + [i64] * 9 + [ci] * 4
        L.bdo_binary_linear.argtypes = [vp] * 5 + [i64] * 9 + [ci] + [i64] * 2 + [ci] * 3
        L.bdo_binarize.argtypes = [vp, vp, i64, i64, i64, ci, vp, vp]
        L.bdo_merge_delta.argtypes = [vp, i64, vp, ctypes.c_float, i64, i64, ci]
        L.bdo_f32_to_f16.argtypes = [ctypes.c_float]
        L.bdo_f32_to_f16.restype = ctypes.c_uint16
        L.bdo_f32_to_bf16.argtypes = [ctypes.c_float]
        L.bdo_f32_to_bf16.restype = ctypes.c_uint16
        L.bdo_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.bdo_f16_to_f32.restype = ctypes.c_float
        _lib = L
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise AssertionError(f"oracle {what} failed rc={rc}")


def pack(x, n_bits=32):
    """bitdelta/binary_gemm_kernel.py:6-32.  x: bool (*, K, N), any strides."""
    assert x.dtype == torch.bool and x.device.type == "cpu"
    assert x.shape[-2] % n_bits == 0, "K must be divisible by n_bits"
    lead = x.shape[:-2]
    K, N = x.shape[-2:]
    if x.dim() == 2:
        x3 = x.unsqueeze(0)          # keeps the (possibly transposed) strides, as diff.py:16 passes them
    elif x.dim() == 3:
        x3 = x
    else:
        x3 = x.reshape(-1, K, N)
    xb = x3.view(torch.uint8)
    out = torch.empty((xb.shape[0], K // n_bits, N), dtype=_WORD_DT[n_bits])
    _chk(lib().bdo_pack(xb.data_ptr(), xb.shape[0], K, N, xb.stride(0), xb.stride(1), xb.stride(2),
                        out.data_ptr(), n_bits), "pack")
    return out.view(*lead, K // n_bits, N)


def unpack(x, n_bits=32):
    """bitdelta/binary_gemm_kernel.py:34-46.  x: int (*, K/n_bits, N) -> bool (*, K, N)."""
    assert x.device.type == "cpu" and x.dtype == _WORD_DT[n_bits]
    lead = x.shape[:-2]
    KW, N = x.shape[-2:]
    xc = x.contiguous().view(-1, KW, N)
    out = torch.empty((xc.shape[0], KW * n_bits, N), dtype=torch.uint8)
    _chk(lib().bdo_unpack(xc.data_ptr(), xc.shape[0], KW, N, n_bits, out.data_ptr()), "unpack")
    return out.view(torch.bool).view(*lead, KW * n_bits, N)


def delta_bmm(a, p, out_dtype=None, round_mode=1, acc_mode=1):
    """C[b] = A[b] . (2*unpack(P[b])-1).  a [B,M,K]; p [B or 1,K/32,N] int32 (1 = broadcast)."""
    a = a.contiguous()
    p = p.contiguous()
    B, M, K = a.shape
    N = p.shape[-1]
    assert p.shape[-2] * 32 == K and p.dtype == torch.int32 and p.shape[0] in (1, B)
    out_dtype = out_dtype or a.dtype
    c = torch.empty((B, M, N), dtype=out_dtype)
    sPb = 0 if p.shape[0] == 1 and B > 1 else p.stride(0)
    _chk(lib().bdo_delta_bmm(a.data_ptr(), p.data_ptr(), c.data_ptr(), B, M, N, K,
                             a.stride(0), a.stride(1), sPb, c.stride(0), c.stride(1),
                             _DT[a.dtype], _DT[out_dtype], round_mode, acc_mode), "delta_bmm")
    return c


def binary_linear(x, w, p, alpha, G=1, out_dtype=None, round_mode=0):
    """y = x W^T + alpha * (x S).  x [B,M,K]; w [N,K]; p [B or 1,K/32,N]; alpha fp32 [B or 1, G]."""
    x = x.contiguous()
    w = w.contiguous()
    p = p.contiguous()
    B, M, K = x.shape
    N = w.shape[0]
    alpha = alpha.detach().float().reshape(-1, G).contiguous()
    assert alpha.shape[0] in (1, B) and p.shape[0] in (1, B)
    out_dtype = out_dtype or x.dtype
    y = torch.empty((B, M, N), dtype=out_dtype)
    sPb = 0 if p.shape[0] == 1 else p.stride(0)
    sAlb = 0 if alpha.shape[0] == 1 else G
    _chk(lib().bdo_binary_linear(x.data_ptr(), w.data_ptr(), p.data_ptr(), alpha.data_ptr(), y.data_ptr(),
                                 B, M, N, K, x.stride(0), x.stride(1), w.stride(0), sPb, sAlb, G,
                                 y.stride(0), y.stride(1), _DT[x.dtype], _DT[out_dtype], round_mode),
         "binary_linear")
    return y


def binarize(base, fine):
    """BinaryDiff.__init__ buffers (bitdelta/diff.py:9-31) -> (mask int32 [K/32,N], coeff fp32 0-dim)."""
    base = base.contiguous()
    fine = fine.contiguous()
    N, K = base.shape
    mask = torch.empty((K // 32, N), dtype=torch.int32)
    coeff = torch.empty((), dtype=torch.float32)
    _chk(lib().bdo_binarize(base.data_ptr(), fine.data_ptr(), N, K, base.stride(0), _DT[base.dtype],
                            mask.data_ptr(), coeff.data_ptr()), "binarize")
    return mask, coeff


def merge_delta(w, p, coeff):
    """In place W += ((unpack(P)*2-1)*coeff).T.to(W.dtype)   (bitdelta/diff.py:93-95)."""
    assert w.is_contiguous() and p.is_contiguous()
    N, K = w.shape
    _chk(lib().bdo_merge_delta(w.data_ptr(), w.stride(0), p.data_ptr(), float(coeff), N, K, _DT[w.dtype]),
         "merge_delta")
    return w
