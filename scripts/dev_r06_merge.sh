#!/bin/bash
# round 6, first GPU call: parity of the ranked merge / two-key pool, then the table probe on c2 and on the metric's shard with the
# knob variants side by side (walk_r2 = 0 + walk_merge_min = 0 is round 5's kernel).
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_merge}
timeout 900 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_walk_order.py tests/test_gpu_dim1024.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_builder.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.log
V="walk_r2=0,walk_merge_min=0;walk_merge_min=0;base;walk_merge_min=2;walk_merge_min=6;walk_r2=0,walk_merge_min=3"
PROBE_VARIANTS="$V" PROBE_EFS=64,128,256 PROBE_COLS=4294967295 PROBE_REPS=16 timeout 600 python scripts/table_probe.py > $OUT/${TAG}_probe_c2.jsonl 2> $OUT/${TAG}_c2.err; echo "probe c2 rc=$?"; cut -c1-330 $OUT/${TAG}_probe_c2.jsonl
PROBE_N=12500000 PROBE_D=1024 PROBE_M0=256 PROBE_M=64 PROBE_EFC=256 PROBE_VARIANTS="$V" PROBE_EFS=128 PROBE_COLS=4294967295 PROBE_REPS=8 timeout 900 python scripts/table_probe.py > $OUT/${TAG}_probe_c4shard.jsonl 2> $OUT/${TAG}_c4.err; echo "probe c4 rc=$?"; cut -c1-330 $OUT/${TAG}_probe_c4shard.jsonl
