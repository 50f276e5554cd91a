import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bitdelta_amd.serving_loop import TenantDecoder
dec = TenantDecoder.synthetic("mistral-7b", 6, "cuda", dtype=torch.float16, seed=1, layers=8, max_len=600)
g = torch.Generator().manual_seed(1)
prompts = [torch.randint(1, 32000, (512,), generator=g).tolist() for _ in range(6)]
for pers in (True, False):
    dec.persistent = pers
    out, n = dec.generate(prompts, max_new_tokens=12, use_graph=True)
    torch.cuda.synchronize()
print("ok")
