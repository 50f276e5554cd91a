"""Board power / shader clock while the multi-tenant decode step replays back to back (is the step at the power cap?).
   python tools/decode_power.py [tenants] [seconds]"""
import glob, sys, threading, time
import torch
sys.path.insert(0, ".")
from bitdelta_amd.serving_loop import TenantDecoder

tenants = int(sys.argv[1]) if len(sys.argv) > 1 else 6
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
hws = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
samples, stop = [], False


def rd(p):
    try:
        return int(open(p).read())
    except Exception:
        return 0


def sampler():
    while not stop:
        samples.append((time.perf_counter(), [(rd(h + "/power1_input"), rd(h + "/freq1_input"), rd(h + "/freq2_input")) for h in hws]))
        time.sleep(0.05)


th = threading.Thread(target=sampler)
th.start()
time.sleep(0.5)
t_idle = time.perf_counter()
dec = TenantDecoder.synthetic("mistral-7b" if tenants > 1 else "llama-2-7b", tenants, "cuda", dtype=torch.float16, seed=4321)
cache = dec.new_cache(512 + 64)
st = {"cache": cache, "tok": torch.randint(0, 1000, (tenants, 1), device="cuda"), "pos": torch.tensor([512], device="cuda"),
      "step": torch.zeros(1, dtype=torch.long, device="cuda"), "out": torch.zeros(tenants, 4096, dtype=torch.long, device="cuda"),
      "stopped": torch.zeros(tenants, dtype=torch.bool, device="cuda"), "stop_ids": torch.full((tenants, 8), -1, device="cuda")}
cache["valid"][:, :512] = True
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
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        st["pos"].fill_(512); st["step"].zero_(); g.replay()
    torch.cuda.synchronize()
    n += 20
t1 = time.perf_counter()
stop = True
th.join()
# the card whose power rose most
best, bd = None, -1
for i in range(len(hws)):
    idle = [s[1][i][0] for s in samples if s[0] < t_idle]
    run = [s[1][i][0] for s in samples if t0 + 1.0 < s[0] < t1]
    if idle and run and (sum(run) / len(run) - sum(idle) / len(idle)) > bd:
        bd, best = sum(run) / len(run) - sum(idle) / len(idle), i
run = [s[1][best] for s in samples if t0 + 1.0 < s[0] < t1]
idle = [s[1][best] for s in samples if s[0] < t_idle]
print(f"decode step, {tenants} tenant(s): {n} replays in {t1 - t0:.2f} s = {(t1 - t0) / n * 1e3:.3f} ms/step; idle {sum(x[0] for x in idle) / len(idle) / 1e6:.0f} W -> "
      f"avg {sum(x[0] for x in run) / len(run) / 1e6:.0f} W (max {max(x[0] for x in run) / 1e6:.0f}) over {len(run)} samples, "
      f"avg sclk {sum(x[1] for x in run) / len(run) / 1e6:.0f} MHz, avg mclk {sum(x[2] for x in run) / len(run) / 1e6:.0f} MHz")
