#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box: kernel-trace stats + separate PMC passes (never combined:
# gpurun refuses --pmc together with the trace domains).  Launches are EAGER (one host launch per step): rocprofv3 does not
# attribute kernels inside replayed hipGraphs one by one, and a 250-step graph under the tracer ran into the call's time limit.
# usage: tests/run_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r03}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-steady-state --eager --steps 300 --warmup 50 $*"
T="timeout 170"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $B > $OUT/bench_trace.log 2>&1
# the same launches in the steady state of a long run (episode ages uniform: every step ends ~33 episodes somewhere on the chip)
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_steady -o trace -- python3 $B --stagger 2000 > $OUT/bench_trace_steady.log 2>&1
$T rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc1 -o pmc -- python3 $B > $OUT/bench_pmc1.log 2>&1
$T rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/pmc2 -o pmc -- python3 $B > $OUT/bench_pmc2.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc -- python3 $B > $OUT/bench_pmc3.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc -- python3 $B > $OUT/bench_pmc4.log 2>&1
# instruction cache: the episode-end path is cold code (steady state: --stagger)
timeout 280 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/pmc5 -o pmc -- python3 $B --stagger 2000 > $OUT/bench_pmc5.log 2>&1
python3 $GRAFT_REPO_ROOT/tests/summarize_prof.py $OUT > /dev/null
# keep only the summary (the raw traces are large)
rm -rf $OUT/trace $OUT/trace_steady $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5
ls -la $OUT
