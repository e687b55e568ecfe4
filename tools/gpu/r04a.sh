cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_rollout.py tests/test_actor.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a/pytest_rollout.txt
cat gpurun_out/r04a/pytest_rollout.txt
for f in 1 0; do
  FWGYM_ROLLOUT_FUSED=$f timeout 300 python bench.py --workload c5 --steps 256 --warmup 20 --no-cpu-baseline 2>gpurun_out/r04a/c5_$f.err | tail -1 > gpurun_out/r04a/c5_fused$f.json
  python -c "
import json;d=json.load(open('gpurun_out/r04a/c5_fused$f.json'));print('fused=$f', d['ms_per_step']*1e3,'us', d['value']/1e9,'G', d['roofline']['kernel'])"
done
