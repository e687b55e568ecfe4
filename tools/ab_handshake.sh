#!/bin/bash
# same-box A/B on the GPU box: a baseline library (built beforehand in the container with tools/ab_build.sh <commit> <name>)
# against the product build; three alternating repetitions of the steady state, a long fresh run and the driver's flags.
# usage: tools/ab_build.sh 382b618 before_handshake && gpurun -- 'bash tools/ab_handshake.sh before_handshake'
#        -> gpurun_out/ab_<name>.txt (round 3's run is kept as profiles/r03_ab_handshake.txt)
set -uo pipefail
name=${1:-before_handshake}
cd "${GRAFT_REPO_ROOT:-.}"
base=fixed-wing-gym_amd/gym_fixed_wing/_abl/libfwgym_${name}.so
prod=fixed-wing-gym_amd/gym_fixed_wing/libfwgym.so
for lib in $base $prod; do
  [ -f "$lib" ] || { echo "ab_handshake: missing $lib (build it with tools/ab_build.sh <commit> $name)" >&2; exit 1; }
done
mkdir -p gpurun_out
OUT=gpurun_out/ab_${name}.txt
: > $OUT
modes=("${AB_MODES:-}")
[ -n "${AB_MODES:-}" ] || modes=("--steps 512 --stagger 2000" "--steps 2000" "--steps 20")
for rep in 1 2 3; do
  for lib in $base $prod; do
    for mode in "${modes[@]}"; do
      line=$(FWGYM_LIB=$PWD/$lib timeout 300 python bench.py --gpus 1 $mode --warmup 5 --no-side --no-cpu-baseline 2>gpurun_out/ab_err.log | tail -1)
      [ -n "$line" ] || { echo "ab_handshake: bench.py failed for $lib $mode:" >&2; tail -20 gpurun_out/ab_err.log >&2; exit 1; }
      echo "$line" | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$rep', '$(basename $lib)', '$mode', 'ms_per_step', round(d['ms_per_step']*1e3,3), 'frac', round(d['roofline']['frac'],4))" >> $OUT
    done
  done
done
cat $OUT
