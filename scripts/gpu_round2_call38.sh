#!/bin/bash
# SQ counters of the f32-storage walk (c2, ef 64)
O=/root/repo/gpurun_out; mkdir -p $O; R=/root/repo
cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/p_$name -o $name -- python $R/scripts/bench_c2_sweep.py --cases clustered:f32 --efs 64 --launches 8 > $O/r2_c38_$name.json 2> $O/r2_c38_$name.err
  python $R/scripts/rocprof_summary.py /tmp/p_$name/${name}_results.db > $O/r2_c38_$name.txt 2>> $O/r2_c38_$name.err
  grep -E "walk_kernel<2, 1, 1.*\| +8192 \| [A-Z]" $O/r2_c38_$name.txt | head -12; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run fetch FETCH_SIZE
run tcp TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum
