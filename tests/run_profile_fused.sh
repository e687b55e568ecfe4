#!/bin/bash
# rocprofv3 kernel trace of the fused C5 rollout (env step + rollout head); summary -> gpurun_out/prof_fused/
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fused
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fused -- python3 $REPO/tests/prof_fused.py > /tmp/prof_fused.log 2>&1
mkdir -p $REPO/gpurun_out/prof_fused
f=$(find /tmp/prof_fused -name "*kernel_stats.csv" | head -1)
cp "$f" $REPO/gpurun_out/prof_fused/kernel_stats.csv
head -12 "$f"
