# the round's GPU evidence in ONE gpurun call on the final sources: full GPU test-suite, rocprofv3 summaries (row log + dense), C5 kernel
# trace, fwg_gae trace, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
( time timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider ) 2>&1 | tail -12 > gpurun_out/r06f/pytest_gpu.txt
cat gpurun_out/r06f/pytest_gpu.txt
bash tests/run_profile.sh r06 > gpurun_out/r06f/profile.log 2>&1
bash tests/run_profile.sh r06_dense --obs-layout dense > gpurun_out/r06f/profile_dense.log 2>&1
bash tests/run_profile_fused.sh > gpurun_out/r06f/profile_fused.log 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06f/gae_trace -o trace -- python3 $GRAFT_REPO_ROOT/tests/prof_gae.py > $GRAFT_REPO_ROOT/gpurun_out/r06f/gae.txt 2>&1 )
grep -h "k_gae" gpurun_out/r06f/gae_trace/*/*kernel_stats.csv gpurun_out/r06f/gae_trace/*kernel_stats.csv 2>/dev/null | head -3 >> gpurun_out/r06f/gae.txt
rm -rf gpurun_out/r06f/gae_trace
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06f/bench_default.json 2> gpurun_out/r06f/bench_default.err
timeout 900 python bench.py > gpurun_out/r06f/bench_noflags.json 2> gpurun_out/r06f/bench_noflags.err
timeout 300 python bench.py --workload c5 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r06f/bench_c5.json 2> gpurun_out/r06f/bench_c5.err
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-side > gpurun_out/r06f/bench_2000.json 2> gpurun_out/r06f/bench_2000.err
ls -la gpurun_out/prof_r06 gpurun_out/prof_r06_dense gpurun_out/prof_fused 2>/dev/null | head -30
tail -3 gpurun_out/r06f/gae.txt
tail -1 gpurun_out/r06f/bench_default.json | cut -c1-600
