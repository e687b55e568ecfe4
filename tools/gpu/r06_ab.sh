# usage: bash tools/gpu/r06_ab.sh OUT lib1.so [lib2.so ...]  (under gym_fixed_wing/_abl/) -> C3 us per step, alternating, three modes:
# the driver's flags (--steps 20 --warmup 5), 2000 steps fresh, 512 steps steady state
mkdir -p gpurun_out/r06
out=gpurun_out/r06/$1; shift
: > $out
for rep in 1 2 3; do
for lib in "$@"; do
  for mode in "--steps 20 --warmup 5" "--steps 2000 --warmup 20" "--steps 512 --warmup 20 --stagger 2000"; do
  timeout 300 python bench.py --workload c3 $mode --no-cpu-baseline --no-side --lib $PWD/fixed-wing-gym_amd/gym_fixed_wing/_abl/$lib 2>gpurun_out/r06/ab_err.log | tail -1 > gpurun_out/r06/ab_line.json
  python -c "
import json;d=json.load(open('gpurun_out/r06/ab_line.json'));print('$lib', '| $mode |', round(d['ms_per_step']*1e3,2),'us frac', round(d['roofline']['frac'],3), 'hip events', round(d['roofline'].get('kernel_ms_hip_events',0)*1e3,2))" | tee -a $out || tail -5 gpurun_out/r06/ab_err.log
  done
done
done
