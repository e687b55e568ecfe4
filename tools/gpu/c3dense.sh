# usage: bash tools/gpu/c3dense.sh lib1.so [lib2.so ...]  (under gym_fixed_wing/_abl/) -> C3 with the DENSE observation batch: fresh window and steady state
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3dense
for rep in 1 2; do
for lib in "$@"; do
  for mode in "--steps 512" "--steps 512 --stagger 2000"; do
  timeout 300 python bench.py --workload c3 --obs-layout dense $mode --warmup 20 --no-cpu-baseline --no-side --lib $PWD/fixed-wing-gym_amd/gym_fixed_wing/_abl/$lib 2>gpurun_out/c3dense/err.log | tail -1 > gpurun_out/c3dense/line.json
  python -c "
import json;d=json.load(open('gpurun_out/c3dense/line.json'));print('$lib', '$mode', round(d['ms_per_step']*1e3,2),'us frac', round(d['roofline']['frac'],3))" || tail -5 gpurun_out/c3dense/err.log
  done
done
done
