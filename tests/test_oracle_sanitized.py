"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are unavailable on the
pool): the golden core traces (G2), wrapper traces (G3), arbitrary states (G7), spawns (G1) and the GAE vectors
(G4) are replayed through `make -C oracle san`'s build in a child process with libasan preloaded; any out-of-bounds access,
signed overflow, misaligned load or shift past the width aborts the child."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_vectors_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    san = os.path.join(ROOT, "oracle", "liboracle_inversus_san.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    env = dict(os.environ, INVERSUS_ORACLE_LIB=san, LD_PRELOAD=libasan, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        "-k", "core_trace or wrapper_trace or arbitrary or spawn_golden or gae_golden or dummy_decide"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "passed" in p.stdout and "runtime error" not in p.stdout and "AddressSanitizer" not in p.stdout, p.stdout[-4000:]
