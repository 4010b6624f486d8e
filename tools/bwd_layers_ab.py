#!/usr/bin/env python3
"""Same-box A/B of INVERSUS_BWD_LAYERS (fused_train.py): the per-sample backward kernel + LayerNorm reduction (0) against the
layer-major backward with the sums in registers (1); each variant in its own process, alternated three times.  ms per
16 384-sample minibatch from tools/agent_update_profile.py --no-prof."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(3):
    for v in ("0", "1"):
        env = dict(os.environ, INVERSUS_BWD_LAYERS=v)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "agent_update_profile.py"), "--no-prof"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"BWD_LAYERS={v}", line[0] if line else r.stderr[-300:], flush=True)
