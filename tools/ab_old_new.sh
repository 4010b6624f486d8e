#!/bin/bash
# Same-box A/B of the fused update: an OLDER tree of this repository against the working tree, both in one gpurun call.
#   REV=b84057e tools/ab_old_new.sh prepare     (in the build container: exports the old tree into tools/exp/old and builds its library)
#   gpurun -- tools/ab_old_new.sh               (on the GPU box: alternates the two trees three times)
cd "$(dirname "$0")/.."
if [ "$1" = prepare ]; then
  rm -rf tools/exp/old && mkdir -p tools/exp/old
  git archive ${REV:?set REV} inversus_reinforcement_learning_amd include tools/agent_update_profile.py | tar -x -C tools/exp/old
  (cd tools/exp/old/inversus_reinforcement_learning_amd && python _build.py > /dev/null) && echo "old tree ready: tools/exp/old"
  exit 0
fi
for rep in 1 2 3; do
  (cd tools/exp/old && python tools/agent_update_profile.py --no-prof 2>/dev/null | grep update | sed 's/^/OLD /')
  python tools/agent_update_profile.py --no-prof 2>/dev/null | grep update | sed 's/^/NEW /'
done
