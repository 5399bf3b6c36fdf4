#!/bin/bash
# GPU call 8 of round 2: SQ / LDS counters of the i8 scan kernel (c3), to see what bounds it.
O=/root/repo/gpurun_out; mkdir -p $O; R=/root/repo
cd /tmp; export TMPDIR=/tmp
true
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/p_$name -o $name -- python $R/scripts/bench_c3.py --walk-n 0 --reps 2 > $O/r2_c10_$name.json 2> $O/r2_c10_$name.err
  python $R/scripts/rocprof_summary.py /tmp/p_$name/${name}_results.db > $O/r2_c10_$name.txt 2>> $O/r2_c10_$name.err
  grep -E "flat_scan_q2_areg" $O/r2_c10_$name.txt | head -12
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run tcp TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
