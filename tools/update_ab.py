#!/usr/bin/env python3
"""Same-box A/B of the fused update's scheduling switches (each variant in its own process, alternated twice):
INVERSUS_PG_STREAM (LayerNorm reduction on a side stream) x INVERSUS_FC1_MID (first Linear's weight-gradient GEMM after the trunk's
backward kernel, beside that reduction).  ms per 16 384-sample minibatch from tools/agent_update_profile.py --no-prof."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for pg in ("1", "0"):
        for mid in ("1", "0"):
            env = dict(os.environ, INVERSUS_PG_STREAM=pg, INVERSUS_FC1_MID=mid)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "agent_update_profile.py"), "--no-prof"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(f"PG_STREAM={pg} FC1_MID={mid}", line[0] if line else r.stderr[-300:], flush=True)
