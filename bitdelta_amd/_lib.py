"""ctypes binding of libbitdelta_hip.so (include/bitdelta_hip.h).

The product path has NO fallback: if the shared library is missing or a tensor is not on a ROCm device the
call raises.  Nothing here imports oracle/.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BD_HIP_LIB: another BUILD of the same library (same-box A/Bs of compile-time kernel switches, tools/ab_lib.sh); never a fallback
LIB_PATH = os.environ.get("BD_HIP_LIB") or os.path.join(_HERE, "lib", "libbitdelta_hip.so")

BD_F16, BD_BF16, BD_F32 = 0, 1, 2
DTYPE_CODE = {torch.float16: BD_F16, torch.bfloat16: BD_BF16, torch.float32: BD_F32}
WORD_DTYPE = {8: torch.uint8, 16: torch.int16, 32: torch.int32, 64: torch.int64}

_i64, _vp, _ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int

# name -> (restype, argtypes); must list every symbol declared in include/bitdelta_hip.h (tests check this)
SIGNATURES = {
    "bd_version": (_ci, []),
    "bd_error_string": (ctypes.c_char_p, [_ci]),
    "bd_pack": (_ci, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _ci, _vp]),
    "bd_unpack": (_ci, [_vp, _i64, _i64, _i64, _vp, _ci, _vp]),
    "bd_delta_bmm": (_ci, [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci, _ci, _ci,
                           _vp, _i64, _ci, _ci, _vp, _i64, _vp]),
    "bd_binary_linear": (_ci, [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci,
                               _i64, _i64, _ci, _ci, _vp, _i64, _vp]),
    "bd_binary_linear_swiglu": (_ci, [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ci, _vp]),
    "bd_binary_linear_decode": (_ci, [_vp, _vp, _vp, _ci, _ci, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci,
                                      _i64, _i64, _ci, _ci, _ci, _vp]),
    "bd_binary_linear_residual": (_ci, [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci,
                                        _i64, _i64, _ci, _ci, _vp, _i64, _vp]),
    "bd_binary_linear_residual_norm": (_ci, [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci,
                                             _i64, _i64, _ci, _vp, _i64, ctypes.c_float, _vp, _i64, _i64, _vp, _i64, _vp]),
    "bd_tenant_linear": (_ci, [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _i64, _ci, _ci, _vp]),
    "bd_srv_rmsnorm": (_ci, [_vp, _vp, _vp, _ci, _ci, _i64, _i64, _i64, _ci, ctypes.c_float, _ci, _vp]),
    "bd_srv_add_rmsnorm": (_ci, [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci, ctypes.c_float, _ci, _vp]),
    "bd_srv_swiglu": (_ci, [_vp, _vp, _vp, _ci, _ci, _i64, _i64, _i64, _ci, _ci, _vp]),
    "bd_binary_linear_decode_fused": (_ci, [_vp, _vp, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci,
                                            _i64, _i64, _ci, _ci, _ci, _vp, _i64, ctypes.c_float, _ci, _vp]),
    "bd_binary_linear_decode_handoff": (_ci, [_vp, _vp, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _ci,
                                              _i64, _i64, _ci, _ci, _ci, _vp, _i64, ctypes.c_float, _ci, _vp, _vp, _vp, _vp]),
    "bd_srv_cache_warm": (_ci, [_vp, _i64, _vp, _i64, _ci, _vp]),
    "bd_srv_step_begin": (_ci, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _ci, _vp, _ci, _ci, _ci, _vp]),
    "bd_srv_step_end": (_ci, [_vp, _i64, _ci, _vp, _vp, _i64, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _vp]),
    "bd_srv_rope_kv_append": (_ci, [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _i64, _ci, _ci, _ci, _vp]),
    "bd_srv_rope": (_ci, [_vp, _vp, _vp, _ci, _ci, _ci, _i64, _ci, _ci, _ci, _vp]),
    "bd_srv_decode_attention": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _i64, _i64, _ci, _vp, _i64, _vp]),
    "bd_srv_decode_attention_workspace_bytes": (_i64, [_ci, _ci, _ci, _ci, _ci]),
    "bd_srv_prefill_attention": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                       _vp, ctypes.c_float, _ci, _ci, _vp]),
    "bd_gemm_workspace_bytes": (_i64, [_ci, _ci, _ci, _ci]),
    "bd_binarize": (_ci, [_vp, _vp, _i64, _i64, _i64, _ci, _vp, _vp, _vp, _i64, _vp]),
    "bd_binarize_workspace_bytes": (_i64, [_i64, _i64]),
    "bd_merge_delta": (_ci, [_vp, _i64, _vp, _vp, _i64, _i64, _ci, _vp]),
    "bd_set_gemm_variant": (_ci, [_ci]),
    "bd_last_gemm_variant": (_ci, []),
    "bd_set_tile_group_m": (_ci, [_ci]),
    "bd_set_launch_chunking": (_ci, [_ci]),
    "bd_set_tail_split": (_ci, [_ci]),
    "bd_set_decode_two_launch": (_ci, [_ci]),
    "bd_set_decode_wave_spec": (_ci, [_ci]),
    "bd_set_decode_small_lut": (_ci, [_ci]),
    "bd_set_stream_tuning": (_ci, [_ci]),
    "bd_last_decode_form": (_ci, []),
    "bd_set_decode_generic_loop": (_ci, [_ci]),
    "bd_set_decode_engine": (_ci, [_ci]),
    "bd_set_ring_tuning": (_ci, [_ci]),
}

_lib = None


class BitDeltaHipError(RuntimeError):
    pass


def lib():
    """Load libbitdelta_hip.so (built by __graft_entry__.build() / bitdelta_amd/build.py).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BitDeltaHipError(
                f"{LIB_PATH} not found: build it with `python -m bitdelta_amd.build` (hipcc, gfx950). "
                "bitdelta_amd has no CPU / PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().bd_error_string(rc).decode()
        if rc in (-1,):
            raise AssertionError(f"{what}: {msg}")          # the reference raises AssertionError here
        raise BitDeltaHipError(f"{what}: {msg} (code {rc})")


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise BitDeltaHipError(
                "bitdelta_amd ops run on a ROCm device only (got a %s tensor); there is no CPU fallback" % t.device.type)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_WORKSPACES = {}          # (device index, stream handle) -> [scratch tensors, newest (largest) last]


def workspace(nbytes, device, zeroed=False):
    """Scratch for one launch.  `zeroed=True` (the GEMM paths): a persistent buffer per (device, stream), zero-filled once when it
    is allocated -- include/bitdelta_hip.h's contract for the decode path's optional ticket area; the library restores the zeros,
    and launches on one stream are ordered, so the buffer is reused by every call on that stream.

    Growth is GEOMETRIC (request rounded up to a power of two, at least 1 MiB): a caller whose problem size creeps up call by call
    (generation without a KV cache, length-sorted batches) allocates O(log size) buffers and the memory retained per stream stays
    below 2 x the largest request.  Superseded buffers are kept alive rather than freed -- a captured hipGraph may still hold their
    addresses, and nothing here can tell whether one does -- which is what bounds the retained total at ~2 x peak instead of 1 x."""
    if nbytes <= 0:
        return None, 0
    nbytes = int(nbytes)
    if not zeroed:
        return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    bufs = _WORKSPACES.setdefault(key, [])
    if not bufs or bufs[-1].numel() < nbytes:
        size = 1 << 20
        while size < nbytes:
            size <<= 1
        bufs.append(torch.zeros(size, dtype=torch.uint8, device=device))
    return bufs[-1], bufs[-1].numel()


def workspace_bytes_retained():
    """Total bytes held by the persistent GEMM workspaces (diagnostic; tests use it to pin the growth policy)."""
    return sum(b.numel() for bufs in _WORKSPACES.values() for b in bufs)
