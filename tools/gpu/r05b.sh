# round 5: GPU test-suite + default bench line of the product build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05/pytest_gpu.txt
cat gpurun_out/r05/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -1 gpurun_out/r05/bench_default.json | cut -c1-1500
