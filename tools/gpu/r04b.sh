cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04b/pytest_gpu.txt
cat gpurun_out/r04b/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench_default.json 2> gpurun_out/r04b/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04b/bench_default.json') if l.startswith('{')][-1])
print('value %.3f G  ms %.3f us frac %.3f'%(d['value']/1e9,d['ms_per_step']*1e3,d['roofline']['frac']))
for k in ('steady_state','obs_delivered','dense_layout','c2','c5','c5_bf16','generic_kernel','integrator_4x64','randomised_aircraft'):
    v=d.get(k,{})
    print(k, {x:(round(v[x],4) if isinstance(v[x],float) else v[x]) for x in v if x in ('ms_per_step','roofline_frac','error','mirror','launches_per_step','steady_state_ms_per_step')})
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','per_process','single_process')})
PY
