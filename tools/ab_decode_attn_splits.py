import sys, os, subprocess, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for name in ("libbitdelta_hip.so", "libbd_split2.so", "libbd_split8.so"):
    code = f"""
import sys; sys.path.insert(0, {root!r})
import os
from bitdelta_amd import _lib
_lib.LIB_PATH = os.path.join({root!r}, 'bitdelta_amd', 'lib', {name!r})
sys.argv = ['bench.py', '--workload', 'mt-decode', '--steps', '20', '--warmup', '3', '--no-cpu-baseline']
import runpy; runpy.run_path(os.path.join({root!r}, 'bench.py'), run_name='__main__')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=400)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(name, "ms/step", round(d["ms_per_step"], 4), "eager", d["mt_decode"].get("ms_per_step_eager"))
    except Exception as e:
        print(name, "failed", e, r.stderr[-500:])
