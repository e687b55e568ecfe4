# what the completion wait costs at the driver's flags (one 20-step graph + sync inside the clock): runtime wait settings
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-side --no-cpu-baseline"
run() { for i in 1 2 3; do env "$@" $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   wall {:.2f} us/step, hip events {:.2f}'.format(d['ms_per_step']*1e3, d['roofline']['kernel_ms_hip_events']*1e3))"; done; }
echo "default"; run X=1
echo "HSA_ENABLE_INTERRUPT=0"; run HSA_ENABLE_INTERRUPT=0
echo "ROC_ACTIVE_WAIT_TIMEOUT=1000"; run ROC_ACTIVE_WAIT_TIMEOUT=1000
echo "both"; run HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
echo "GPU_MAX_HW_QUEUES=1"; run GPU_MAX_HW_QUEUES=1
