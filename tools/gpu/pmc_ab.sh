# PMC counters of the step kernel for a list of developer libraries (under gym_fixed_wing/_abl/): instruction cache, issue and wait
# usage: bash tools/gpu/pmc_ab.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/pmc
mkdir -p $OUT
for lib in "$@"; do
  B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side --eager --steps 100 --warmup 20 --workload c3 --lib $GRAFT_REPO_ROOT/fixed-wing-gym_amd/gym_fixed_wing/_abl/$lib"
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"; do
    i=$((i+1))
    rm -rf $OUT/raw
    timeout 170 rocprofv3 --pmc $set --output-format csv -d $OUT/raw -o pmc -- python3 $B > $OUT/log_${lib}_$i.txt 2>&1
    python3 - "$lib" $OUT/raw <<'PY'
import sys, glob, csv, collections
lib, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(lib, "  ".join("{}={:.4g}".format(k, sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2)) for k, v in sorted(acc.items())))
PY
  done
done
rm -rf $OUT/raw
