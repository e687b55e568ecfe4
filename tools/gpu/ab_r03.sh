# full-stack A/B on one box: round 3's final tree (commit 5942968, built under _abl/r03tree by hand) against the working tree
cd $GRAFT_REPO_ROOT
OLD=fixed-wing-gym_amd/gym_fixed_wing/_abl/r03tree
for rep in 1 2 3; do
  for tree in $OLD .; do
    for mode in "--steps 512 --stagger 2000" "--steps 2000" "--steps 20"; do
      (cd $tree && timeout 300 python bench.py --gpus 1 $mode --warmup 5 --no-side --no-cpu-baseline 2>/tmp/ab_err.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$rep', '$tree', '$mode', 'us', round(d['ms_per_step']*1e3,3), 'frac', round(d['roofline']['frac'],4))" || tail -3 /tmp/ab_err.log)
    done
  done
done
