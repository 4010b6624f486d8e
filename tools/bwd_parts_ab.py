#!/usr/bin/env python3
"""Same-box A/B of INVERSUS_BWD_PARTS (fused_train.py: the trunk's backward kernel in 1 / 2 / 3 / 4 launches over sample ranges, each
range's LayerNorm reduction on the side stream beside the next range's backward kernel): each variant in its own process, alternated
three times.  ms per 16 384-sample minibatch from tools/agent_update_profile.py --no-prof."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(3):
    for parts in ("1", "2", "4", "3"):
        env = dict(os.environ, INVERSUS_BWD_PARTS=parts)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "agent_update_profile.py"), "--no-prof"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"BWD_PARTS={parts}", line[0] if line else r.stderr[-300:], flush=True)
