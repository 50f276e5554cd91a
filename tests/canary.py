"""Guard / canary buffers for the GPU parity tests ("race detection / sanitizers"): an output allocated INSIDE a larger
buffer filled with a byte pattern, with at least one whole kernel tile of margin on every side (rows above and below, columns left and
right, one whole batch entry before and after), handed to the kernel through its `out=` / stride arguments.  After the launch every byte
outside the output must still carry the pattern: a tile kernel whose guarded epilogue stores one row past M or one column past N, a
split-k slab indexed one entry too far, a pair tile writing its absent partner -- all land in a margin instead of in a neighbour nobody
looks at.  The split-k / ticket workspace gets the same treatment: exactly bd_gemm_workspace_bytes() usable bytes carved from a larger
poisoned buffer."""
import contextlib

import torch

POISON = 0x7F            # 0x7F7F (bf16 / fp16) and 0x7F7F7F7F (fp32) are large finite values: a stray READ of a margin shows up too
ROW_MARGIN = 256         # the largest tile is 256 x 256
COL_MARGIN = 256         # (a multiple of 8 elements: the view keeps the alignment class of a fresh allocation with the same N)


class CanaryOut:
    """`.view`: a (B, M, N) tensor of `dtype` inside a poisoned (B + 2, M + 2 ROW_MARGIN, N + 2 COL_MARGIN) buffer."""

    def __init__(self, B, M, N, dtype, device="cuda", row_margin=ROW_MARGIN, col_margin=COL_MARGIN):
        self.buf = torch.empty(B + 2, M + 2 * row_margin, N + 2 * col_margin, dtype=dtype, device=device)
        self.bytes = self.buf.view(torch.uint8)
        self.bytes.fill_(POISON)
        self.view = self.buf[1:B + 1, row_margin:row_margin + M, col_margin:col_margin + N]
        self._poison = torch.full((), POISON * (0x0101 if dtype.itemsize == 2 else 0x01010101),
                                  dtype=torch.int16 if dtype.itemsize == 2 else torch.int32, device=device).view(dtype)

    def untouched_outside(self):
        """True when no byte outside the output changed (the output is re-poisoned for the check: call after reading the result)."""
        keep = self.view.clone()
        self.view.copy_(self._poison.expand_as(self.view))
        ok = bool((self.bytes == POISON).all())
        self.view.copy_(keep)
        return ok

    def result(self):
        return self.view.clone()


@contextlib.contextmanager
def canary_workspace(monkeypatch_target):
    """Replace `monkeypatch_target.workspace` (bitdelta_amd.binary_gemm_kernel / serving_ops) so that every launch inside the block gets
    EXACTLY the scratch bytes it asked for -- zero-filled, per the C ABI's contract -- followed by 1 MiB of poison.  Yields a checker:
    `ok()` is True when no launch wrote past its request."""
    made = []
    orig = monkeypatch_target.workspace

    def ws(nbytes, device, zeroed=False):
        if nbytes <= 0:
            return None, 0
        nbytes = int(nbytes)
        pad = (-nbytes) % 256
        buf = torch.empty(nbytes + pad + (1 << 20), dtype=torch.uint8, device=device)
        buf[:nbytes].zero_()
        buf[nbytes:].fill_(POISON)
        made.append((buf, nbytes))
        return buf, nbytes

    monkeypatch_target.workspace = ws
    try:
        yield lambda: all(bool((b[n:] == POISON).all()) for b, n in made), made
    finally:
        monkeypatch_target.workspace = orig
