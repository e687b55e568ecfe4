# the round's GPU evidence in ONE gpurun call on the final sources: full GPU test-suite, rocprofv3 summaries (row log + dense), C5 kernel
# trace, bench lines, per-block timeline, suite-context soak
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05f/pytest_gpu.txt
cat gpurun_out/r05f/pytest_gpu.txt
bash tests/run_profile.sh r05 > gpurun_out/r05f/profile.log 2>&1
bash tests/run_profile.sh r05_dense --obs-layout dense > gpurun_out/r05f/profile_dense.log 2>&1
bash tests/run_profile_fused.sh > gpurun_out/r05f/profile_fused.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05f/bench_default.json 2> gpurun_out/r05f/bench_default.err
timeout 900 python bench.py > gpurun_out/r05f/bench_noflags.json 2> gpurun_out/r05f/bench_noflags.err
timeout 300 python bench.py --workload c5 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r05f/bench_c5.json 2> gpurun_out/r05f/bench_c5.err
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-side > gpurun_out/r05f/bench_2000.json 2> gpurun_out/r05f/bench_2000.err
TL_LIB=libfwgym_final_tl.so TL_WL=c3:log TL_STAGGER=2000 TL_PERM=1 timeout 300 python tools/timeline.py run 2>&1 | grep -v "Warn\|^/\|return fnb" > gpurun_out/r05f/timeline_log.txt
timeout 700 python tests/soak_suite_context.py 600 gpurun_out/r05f/soak 2>&1 | grep -v Warn | tail -5 > gpurun_out/r05f/soak.txt
cat gpurun_out/r05f/soak.txt
ls -la gpurun_out/prof_r05 gpurun_out/prof_r05_dense gpurun_out/prof_fused 2>/dev/null | head -30
tail -1 gpurun_out/r05f/bench_default.json | cut -c1-400
