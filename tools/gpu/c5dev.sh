# usage: bash tools/gpu/c5dev.sh lib1.so [lib2.so ...]   (paths under gym_fixed_wing/_abl/) -> C5 us per rollout step for each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5dev
for rep in 1 2; do
for lib in "$@"; do
  timeout 300 python bench.py --workload c5 --steps 256 --warmup 20 --no-cpu-baseline --lib $PWD/fixed-wing-gym_amd/gym_fixed_wing/_abl/$lib 2>gpurun_out/c5dev/err.log | tail -1 > gpurun_out/c5dev/line.json
  python -c "
import json;d=json.load(open('gpurun_out/c5dev/line.json'));print('$lib', round(d['ms_per_step']*1e3,2),'us', round(d['value']/1e9,3),'G', d['roofline']['kernel'])" || tail -5 gpurun_out/c5dev/err.log
done
done
