# Round 6: what the host-side wait policy does to the driver's 20-step line (one hipGraph launch + device synchronise inside the clock)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/sync_policy.txt
: > $out
L=$PWD/fixed-wing-gym_amd/gym_fixed_wing/libfwgym.so
run() {  # name, env assignments...
  name=$1; shift
  for rep in 1 2 3; do
    env "$@" timeout 300 python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline --no-side --lib $L 2>gpurun_out/r06/sync_err.log | tail -1 > gpurun_out/r06/sync_line.json
    python -c "
import json;d=json.load(open('gpurun_out/r06/sync_line.json'));print('$name | rep $rep |', round(d['ms_per_step']*1e3,2),'us/step  hip events', round(d['roofline'].get('kernel_ms_hip_events',0)*1e3,2))" | tee -a $out || tail -3 gpurun_out/r06/sync_err.log
  done
}
run default FWG_X=1
run HSA_ENABLE_INTERRUPT=0 HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000 ROC_ACTIVE_WAIT_TIMEOUT=1000
run both HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
run GPU_MAX_HW_QUEUES=1 GPU_MAX_HW_QUEUES=1
run default_again FWG_X=1
