# same-box bisect of the row-log step kernel over this round's commits (developer builds, tools/devlib.py) + round 3's tree
cd $GRAFT_REPO_ROOT
OLD=fixed-wing-gym_amd/gym_fixed_wing/_abl/r03tree
for rep in 1 2; do
  for mode in "--steps 2000" "--steps 512 --stagger 2000"; do
    (cd $OLD && timeout 300 python bench.py --gpus 1 $mode --warmup 20 --no-side --no-cpu-baseline 2>/tmp/ab_err.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r03tree', '$mode', round(d['ms_per_step']*1e3,2),'us')")
  done
  bash tools/gpu/c3dev.sh "$@" | grep -v "^$" | awk -v r=$rep 'NR<=2*'$#'{print}'
done
