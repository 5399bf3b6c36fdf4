#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun): GPU tests, the level-table probe on c2, one SQ pass
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest ${GPU_TESTS:-tests} -m gpu -q -x > $OUT/r04_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_pytest.log
PROBE_COLS=${PROBE_COLS:-0,8192} timeout 300 python scripts/table_probe.py > $OUT/r04_table_probe_c2.jsonl 2> $OUT/r04_table_probe_c2.err; echo "probe rc=$?"; cat $OUT/r04_table_probe_c2.jsonl
cd /tmp; export TMPDIR=/tmp
PROBE_COLS=8192 PROBE_REPS=6 timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/scripts/table_probe.py > $OUT/r04_table_probe_sq.jsonl 2> $OUT/r04_table_probe_sq.err
python $R/scripts/rocprof_summary.py /tmp/p_s2/s2_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/r04_sq_instruction_mix.txt 2>> $OUT/r04_table_probe_sq.err
grep "walk_kernel\|flat_codes" $OUT/r04_sq_instruction_mix.txt | head -12
