# the same hunt with the exact invocation of tools/gpu/r04_final.sh (cache provider on, output through a pipe)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flaky2
for i in $(seq 1 ${FLAKY_RUNS:-8}); do
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/flaky2/run_$i.txt | tail -1
  if grep -q "FAILED\|differ" gpurun_out/flaky2/run_$i.txt; then grep -n "differ\|FAILED\|reproduces" gpurun_out/flaky2/run_$i.txt | cut -c1-600 | head -40; break; fi
done
