"""bitdelta_amd: MI355X-native (gfx950) implementation of BitDelta's 1-bit-delta Linear hot path.

Drop-in module map (reference -> here):
    bitdelta.binary_gemm_kernel  ->  bitdelta_amd.binary_gemm_kernel   (pack, unpack, binary_matmul, binary_bmm)
    bitdelta.diff                ->  bitdelta_amd.diff                 (BinaryDiff, compress_diff, save_diff, load_diff, save_full_model)
    demo.demo_backend (modules)  ->  bitdelta_amd.serving              (DiffCompressModule, DataParallelModule, register_diff_compress, ...)
"""
from .binary_gemm_kernel import (binary_bmm, binary_linear, binary_linear_decode, binary_matmul, delta_bmm, pack,  # noqa: F401
                                 pack_decode_masks, tenant_linear, tile_masks, unpack)
from .diff import BinaryDiff, BinaryLinear, compress_diff, load_diff, save_diff, save_full_model  # noqa: F401
from .serving import DiffCompressModule  # noqa: F401

__version__ = "0.1.0"
