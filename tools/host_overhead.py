import sys, time, torch
sys.path.insert(0, "/root/repo")
from bitdelta_amd.binary_gemm_kernel import binary_linear, delta_bmm
import cProfile, pstats
dev = "cuda"
x = torch.randn(1, 64, 256, device=dev).bfloat16()
w = torch.randn(256, 256, device=dev).bfloat16()
m = torch.zeros(1, 8, 256, device=dev, dtype=torch.int32)
a = torch.ones(1, 1, device=dev)
for _ in range(50): binary_linear(x, w, m, a)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(2000): binary_linear(x, w, m, a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"binary_linear host: {(t1-t)/2000*1e6:.1f} us per call (enqueue), {(t2-t)/2000*1e6:.1f} us incl. drain")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): binary_linear(x, w, m, a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
