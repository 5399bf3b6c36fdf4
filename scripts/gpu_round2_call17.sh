#!/bin/bash
# GPU call 17 of round 2: SQ counters of the latency walk on single 256-query batches (c2 index, ef 64): wave-cycle split and
# instruction mix, each in its own --pmc pass (kernel trace only)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/scripts/single_batch_trace.py > $O/r2_c17_single.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $O/r2_c17_single_kernel_trace.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_s1 -o s1 -- python $R/scripts/single_batch_trace.py > $O/r2_c17_s1.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_s1/s1_results.db > $O/r2_c17_single_sq_cycles.txt 2>> $O/r2_c17_s1.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/scripts/single_batch_trace.py > $O/r2_c17_s2.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_s2/s2_results.db > $O/r2_c17_single_sq_insts.txt 2>> $O/r2_c17_s2.log
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/p_s3 -o s3 -- python $R/scripts/single_batch_trace.py > $O/r2_c17_s3.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/p_s3/s3_results.db > $O/r2_c17_single_sq_lds.txt 2>> $O/r2_c17_s3.log
cd $R
tail -2 $O/r2_c17_single.log; grep -i "walk" $O/r2_c17_single_kernel_trace.txt | head -4; grep -i "walk_lat" $O/r2_c17_single_sq_cycles.txt | head -6; grep -i "walk_lat" $O/r2_c17_single_sq_insts.txt | head -6; grep -i "walk_lat" $O/r2_c17_single_sq_lds.txt | head -6
