"""One decode step per flag under rocprofv3 --kernel-trace: `python tools/prof_handoff.py <0|1> [tenants]` runs 12 eager decode steps of the
6-tenant Mistral-7B decoder with TenantDecoder.norm_handoff = flag (kernel-trace CSV gives per-kernel durations and start times)."""
import sys
import torch
sys.path.insert(0, ".")
from bitdelta_amd.serving_loop import TenantDecoder

flag, tenants = bool(int(sys.argv[1])), int(sys.argv[2]) if len(sys.argv) > 2 else 6
dec = TenantDecoder.synthetic("mistral-7b", tenants, "cuda", dtype=torch.float16, seed=4321, layers=8)
dec.norm_handoff = flag
cache = dec.new_cache(512 + 64)
st = {"cache": cache, "tok": torch.randint(0, 1000, (tenants, 1), device="cuda"), "pos": torch.tensor([512], device="cuda"),
      "step": torch.zeros(1, dtype=torch.long, device="cuda"), "out": torch.zeros(tenants, 4096, dtype=torch.long, device="cuda"),
      "stopped": torch.zeros(tenants, dtype=torch.bool, device="cuda"), "stop_ids": torch.full((tenants, 8), -1, device="cuda")}
cache["valid"][:, :512] = True
g = None
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        st["pos"].fill_(512); st["step"].zero_(); dec._decode_step(st)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
st["pos"].fill_(512); st["step"].zero_()
with torch.cuda.graph(g, stream=side):
    dec._decode_step(st)
for _ in range(12):
    st["pos"].fill_(512); st["step"].zero_(); g.replay()
torch.cuda.synchronize()
print("done", flag)
