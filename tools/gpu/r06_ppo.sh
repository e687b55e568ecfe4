#!/bin/bash
# Round 6: PPO training experiments on one MI355X (examples/train_ppo.py).  usage: r06_ppo.sh "<tag>:<args>" ...
mkdir -p gpurun_out/r06
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  ( time python examples/train_ppo.py $args --curve gpurun_out/r06/ppo_curve_$tag.json --out gpurun_out/r06/ppo_model_$tag.npz ) > gpurun_out/r06/ppo_$tag.txt 2>&1
  echo "== $tag: $args"; grep -v "^update" gpurun_out/r06/ppo_$tag.txt | tail -6; grep "^update" gpurun_out/r06/ppo_$tag.txt | grep -v "  -  " | tail -12
done
