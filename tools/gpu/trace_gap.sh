# does the tracer's per-dispatch duration of the step kernel depend on how soon after its predecessor the launch is queued?
# same box: eager launches back to back / every launch on a drained device (--eager-sync); then the graph-replay step time.
# (It does not: 11.76 / 11.74 us, profiles/r05_trace_gap.txt -- while a replayed graph advances one step per 10.35 us on that box:
# a host launch's kernel carries its own end-of-kernel release, a graph node's does not wait for it before the next node starts)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_gap
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-steady-state --no-side --eager --steps 300 --warmup 50"
for mode in queued sync; do
  extra=""; [ $mode = sync ] && extra="--eager-sync"
  timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o trace -- python3 $B $extra > $OUT/bench_$mode.log 2>&1
  python3 - $OUT/$mode $mode <<'PY'
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/trace_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_step2" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-300:]
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
g = np.array([int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows[:-1], rows[1:])]) / 1e3
print("{}: k_step2 n={} mean {:.2f} us median {:.2f} min {:.2f} p90 {:.2f} | gap to the next launch mean {:.2f} median {:.2f} us".format(
    sys.argv[2], len(d), d.mean(), np.median(d), d.min(), np.percentile(d, 90), g.mean(), np.median(g)))
PY
  rm -rf $OUT/$mode
done
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-side 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph replay, 2000 steps: {:.2f} us per step'.format(d['ms_per_step']*1e3))"
