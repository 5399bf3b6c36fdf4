#!/bin/bash
# kernel trace + SQ counters of the c3 scan (query-resident kernel)
O=/root/repo/gpurun_out; mkdir -p $O; R=/root/repo
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/scripts/bench_c3.py --walk-n 0 --reps 3 > $O/r2_c16_kt.json 2> $O/r2_c16_kt.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $O/r2_c16_kt.txt; head -30 $O/r2_c16_kt.txt
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/p_$name -o $name -- python $R/scripts/bench_c3.py --walk-n 0 --reps 2 > $O/r2_c16_$name.json 2> $O/r2_c16_$name.err
  python $R/scripts/rocprof_summary.py /tmp/p_$name/${name}_results.db > $O/r2_c16_$name.txt 2>> $O/r2_c16_$name.err
  grep -E "flat_scan_q2_areg.*\| [A-Z]" $O/r2_c16_$name.txt | head -12; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU
