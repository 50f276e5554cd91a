import torch, time, sys
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from bitdelta_amd import serving_ops as ops
dev = 'cuda'
x = torch.randn(1, 2048, 4096, device=dev).bfloat16()
w = torch.randn(4096, device=dev).bfloat16()
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = lambda: F.rms_norm(x, (4096,), w, 1e-5)
b = lambda: ops.rmsnorm_tenant(x, w[None], 1e-5)
print('torch rms_norm', t(a), 'us; hip rmsnorm_tenant', t(b), 'us; equal', torch.equal(a(), b()), (a().float() - b().float()).abs().max().item())
gu = torch.randn(1, 2048, 22016, device=dev).bfloat16()
print('swiglu_interleaved8', t(lambda: ops.swiglu_interleaved8(gu)), 'us')
qkv = torch.randn(1, 2048, 12288, device=dev).bfloat16()
cos = torch.randn(2048, 128, device=dev).bfloat16(); sin = torch.randn(2048, 128, device=dev).bfloat16()
print('rope q+k one launch', t(lambda: ops.rope_(qkv[..., :8192], cos, sin, 64, 2048, 0)), 'us;  q only', t(lambda: ops.rope_(qkv[..., :4096], cos, sin, 32, 2048, 0)), 'us')
