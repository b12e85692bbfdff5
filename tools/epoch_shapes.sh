#!/bin/bash
# learn_epoch timing over the shapes the kernel variants cover: bash tools/epoch_shapes.sh  (hidden minibatch K obs actions)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "64,64 64 4096 3 1" "256 64 4096 3 1" "64,64 64 4096 4 2" "128 64 4096 8 4" "256 32 4096 3 1" "48,32 64 4096 3 4" "256 64 4096 15 2"; do
  timeout -k 10 120 python -u tools/epoch_time.py $cfg 2>&1 | grep "learn_epoch hidden" | cut -c1-120
done
