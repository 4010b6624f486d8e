#!/usr/bin/env python3
"""Build a variant of libinversus_hip.so with extra -D flags into tools/exp/ (git-ignored, travels to the GPU box):
    python tools/build_variant.py NAME [-DFLAG=VALUE ...]      ->  tools/exp/lib_NAME.so
Used by the A/B tools (INVERSUS_HIP_LIB selects the library a process loads)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inversus_reinforcement_learning_amd import _build as B

name, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "exp", f"lib_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
only = os.environ.get("ONLY_SOURCES")                      # e.g. "inv_api.hip": a step-kernel variant needs nothing else
srcs = [s for s in B.SOURCES if not only or os.path.basename(s) in only.split(",")]
cmd = [B.hipcc()] + B.FLAGS + ["-DINV_TOOLS_BUILD"] + defs + ["-I", os.path.join(ROOT, "include"), "-o", out] + srcs
subprocess.check_call(cmd)
print(out)
