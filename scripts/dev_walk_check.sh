#!/bin/bash
# development check of a walk-kernel change on the GPU box: parity suites, the table probe, one SQ pass of the probe
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-dev}
timeout 700 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_walk_order.py tests/test_gpu_dim1024.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_builder.py tests/test_gpu_meta.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.log
PROBE_EFS=64,256 PROBE_COLS=4294967295 PROBE_REPS=24 timeout 400 python scripts/table_probe.py > $OUT/${TAG}_probe.jsonl 2> $OUT/${TAG}.err; echo "probe rc=$?"; cut -c1-420 $OUT/${TAG}_probe.jsonl
cd /tmp; export TMPDIR=/tmp
PROBE_EFS=64 PROBE_COLS=4294967295 PROBE_REPS=6 timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_s -o s -- python $R/scripts/table_probe.py > $OUT/${TAG}_sq.jsonl 2> $OUT/${TAG}_sq.err
python $R/scripts/rocprof_summary.py /tmp/p_s/s_results.db > $OUT/${TAG}_sq_summary.txt
grep -A 14 "by level range" $OUT/${TAG}_sq_summary.txt | grep "32768" | head -12
