#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_builder.py -m gpu -q -x > $OUT/r3_pytest_lat4.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r3_pytest_lat4.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/scripts/lat4_profile.py > $OUT/r3_l4.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db | grep "walk_lat" > $OUT/r3_lat4_kernel_trace.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/p_s1 -o s1 -- python $R/scripts/lat4_profile.py >> $OUT/r3_l4.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/scripts/lat4_profile.py >> $OUT/r3_l4.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_s1/s1_results.db /tmp/p_s2/s2_results.db | grep "walk_lat.*| *256 " > $OUT/r3_lat4_sq_counters.txt
grep "single batch" $OUT/r3_l4.log; cat $OUT/r3_lat4_kernel_trace.txt; cat $OUT/r3_lat4_sq_counters.txt
