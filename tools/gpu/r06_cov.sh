#!/bin/bash
# Round 6, oracle coverage on the GPU: the new tests, then the same steady-state tests against the two mutants (kernels with
# round 5's fixes reverted, tools/mutation_check.py hip): both must FAIL.  Output: gpurun_out/r06/cov_*.txt
mkdir -p gpurun_out/r06
M=fixed-wing-gym_amd/gym_fixed_wing/_abl/mut
( time python -m pytest tests/test_gpu_oracle_coverage.py -q -s -p no:cacheprovider ) > gpurun_out/r06/cov_clean.txt 2>&1
echo "clean rc $?" >> gpurun_out/r06/cov_clean.txt
for mut in airdata install_on_failed_last_step; do
  for layout in row_log dense; do
    up=$(echo $layout | tr a-z A-Z)
    ( env FWGYM_MUTANT_LIB_$up=$M/libfwgym_mut_${mut}_${layout}.so python -m pytest tests/test_gpu_oracle_coverage.py -q -s -p no:cacheprovider \
        -k "fail_prone and $layout" ) > gpurun_out/r06/cov_mut_${mut}_${layout}.txt 2>&1
    echo "mutant $mut $layout rc $? (non-zero = caught)" | tee -a gpurun_out/r06/cov_mutants.txt
  done
done
tail -5 gpurun_out/r06/cov_clean.txt
cat gpurun_out/r06/cov_mutants.txt
