#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun): one 256-query batch at a time under rocprofv3
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/scripts/single_batch_trace.py > $OUT/r3_sb.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $OUT/r3_single_batch_kernel_trace.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_s1 -o s1 -- python $R/scripts/single_batch_trace.py >> $OUT/r3_sb.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/scripts/single_batch_trace.py >> $OUT/r3_sb.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_s1/s1_results.db /tmp/p_s2/s2_results.db | grep -v "link_kernel\|evict\|claim" > $OUT/r3_single_batch_sq_counters.txt
grep "single batch" $OUT/r3_sb.log; head -8 $OUT/r3_single_batch_kernel_trace.txt; grep "walk_lat" $OUT/r3_single_batch_sq_counters.txt
