#!/bin/bash
# 15 x 10 trunk forward with two workgroups per CU (shipped) vs one (tools/exp/lib_bandone.so: -DPOL_BAND_ONE_PER_CU), alternated.
cd "$(dirname "$0")/.."
for v in shipped one shipped one; do
  if [ "$v" = shipped ]; then unset INVERSUS_HIP_LIB; else export INVERSUS_HIP_LIB=$PWD/tools/exp/lib_bandone.so; fi
  timeout -k 10 120 python - "$v" <<'PY' || exit 1
import sys, torch
sys.path.insert(0, ".")
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
torch.manual_seed(0)
B = 16384
f = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval())
g = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device="cuda")
for _ in range(3): f.features(g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f.features(g)
e1.record(); torch.cuda.synchronize()
print("variant", sys.argv[1], "15x10 trunk forward ms", round(e0.elapsed_time(e1) / 20, 4), flush=True)
PY
done
