# the round's final GPU evidence: full GPU test-suite, rocprofv3 summaries (row log + dense), C5 kernel trace, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04f/pytest_gpu.txt
cat gpurun_out/r04f/pytest_gpu.txt
bash tests/run_profile.sh r04 > gpurun_out/r04f/profile.log 2>&1
bash tests/run_profile.sh r04_dense --obs-layout dense > gpurun_out/r04f/profile_dense.log 2>&1
bash tests/run_profile_fused.sh > gpurun_out/r04f/profile_fused.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04f/bench_default.json 2> gpurun_out/r04f/bench_default.err
timeout 300 python bench.py --workload c5 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r04f/bench_c5.json 2> gpurun_out/r04f/bench_c5.err
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-side > gpurun_out/r04f/bench_2000.json 2> gpurun_out/r04f/bench_2000.err
(cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; nproc; grep -c processor /proc/cpuinfo) > gpurun_out/r04f/host_cpus.txt 2>&1
TL_LIB=libfwgym_tl_log.so TL_WL=c3:log TL_STAGGER=2000 TL_PERM=1 timeout 300 python tools/timeline.py run 2>&1 | grep -v Warn > gpurun_out/r04f/timeline_log.txt
TL_LIB=libfwgym_tl_dense.so TL_WL=c3:dense TL_STAGGER=2000 TL_PERM=1 timeout 300 python tools/timeline.py run 2>&1 | grep -v Warn > gpurun_out/r04f/timeline_dense.txt
TL_LIB=libfwgym_tl_c5.so timeout 300 python tools/timeline_rollout.py 2>&1 | grep -v Warn > gpurun_out/r04f/timeline_rollout.txt
ls -la gpurun_out/prof_r04 gpurun_out/prof_r04_dense gpurun_out/prof_fused 2>/dev/null | head -30
tail -3 gpurun_out/r04f/bench_default.json | cut -c1-600
