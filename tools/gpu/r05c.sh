# round 5: the new GPU tests (G5 states through the team kernels, replay-stable windows, suite-context soak) + timing of one soak iteration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "g5" 2>&1 | grep -v Warning | tail -25 > gpurun_out/r05/pytest_g5.txt
cat gpurun_out/r05/pytest_g5.txt
timeout 900 python -m pytest tests/test_obs_log.py tests/test_rollout.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05/pytest_obs_rollout.txt
cat gpurun_out/r05/pytest_obs_rollout.txt
timeout 400 python tests/soak_suite_context.py 240 gpurun_out/r05/soak 2>&1 | grep -v Warn | tail -5 | tee gpurun_out/r05/soak_a.txt
