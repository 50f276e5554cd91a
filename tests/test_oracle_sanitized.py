"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer ("race detection / sanitizers"): the checker every
parity claim rests on is itself checked for out-of-bounds accesses, signed overflow, misaligned loads and bad shifts, on the reference's
own golden vectors (the whole of tests/test_oracle_golden.py, re-run in a child process that loads the sanitized build)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_suite_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    so = os.path.join(ROOT, "oracle", "_san", "libbd_oracle_san.so")
    assert os.path.exists(so)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env = dict(os.environ)
    env.update(BD_ORACLE_SO=so, LD_PRELOAD=f"{asan} {ubsan}", OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout
    # the child really loaded the sanitized build (not the plain one)
    chk = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); from oracle import bd_oracle as o; o.lib(); "
                          "print(any('libbd_oracle_san' in l for l in open('/proc/self/maps')))"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert chk.stdout.strip().endswith("True"), chk.stdout + chk.stderr
