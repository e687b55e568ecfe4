#!/bin/bash
# same-box A/B: steady state and fresh run, the library before the one-way hand-shake (commit 382b618) against the final build.
# The baseline library is built here beforehand from that commit's csrc/ and include/ (git show 382b618:... into a scratch
# directory, the Makefile's hipcc line with -o gym_fixed_wing/_abl/libfwgym_before_handshake.so); FWGYM_LIB selects the library.
# usage: gpurun -- 'bash tools/ab_handshake.sh'   -> gpurun_out/ab_handshake.txt (kept as profiles/r03_ab_handshake.txt)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_handshake.txt
: > $OUT
for rep in 1 2 3; do
  for lib in fixed-wing-gym_amd/gym_fixed_wing/_abl/libfwgym_before_handshake.so fixed-wing-gym_amd/gym_fixed_wing/libfwgym.so; do
    for mode in "--steps 512 --stagger 2000" "--steps 2000" "--steps 20"; do
      FWGYM_LIB=$PWD/$lib timeout 300 python bench.py --gpus 1 $mode --warmup 5 --no-side --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$rep', '$(basename $lib)', '$mode', 'ms_per_step', round(d['ms_per_step']*1e3,3), 'frac', round(d['roofline']['frac'],4))" >> $OUT
    done
  done
done
cat $OUT
