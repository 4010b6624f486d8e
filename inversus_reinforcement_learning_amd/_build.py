"""Build recipe of libinversus_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libinversus_hip.so")
SOURCES = [os.path.join(CSRC, "inv_api.hip"), os.path.join(CSRC, "inv_gae.hip"), os.path.join(CSRC, "inv_policy.hip"),
           os.path.join(CSRC, "inv_optim.hip"), os.path.join(CSRC, "inv_heads.hip"), os.path.join(CSRC, "inv_fc1.hip")]
HEADERS = [os.path.join(CSRC, "inv_device.h"), os.path.join(CSRC, "inv_step4.h"), os.path.join(CSRC, "inv_internal.h"), os.path.join(ROOT, "include", "inversus_hip.h")]

# -ffp-contract=off: the reward is an f64 sum in the reference's order and GAE is strict f32
# (ppo_agent.py:145-152); an FMA contraction would change results.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libinversus_hip.so cannot be built")


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(p) <= t for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build_library(force=False, verbose=False):
    """Compile the HIP kernels + C ABI into inversus_reinforcement_learning_amd/libinversus_hip.so."""
    if not force and up_to_date():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
