#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_trace2}
timeout 600 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_walk_order.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log
for V in "walk_table_after_sort=0" ""; do
COS_TUNING="$V" COS_BENCH_FULL_RECORD=${TAG}_x.json timeout 900 python bench.py --ef 112 --configs none --no-cpu-baseline --no-hbm-probe --steps 16 --warmup 4 --recall-queries 2048 > $OUT/${TAG}_c4_${V:-default}.json 2>> $OUT/${TAG}_bench.err; echo "c4 [$V] rc=$?"
COS_TUNING="$V" COS_BENCH_FULL_RECORD=${TAG}_x.json timeout 900 python bench.py --workload c2 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --steps 24 --warmup 4 --recall-queries 2048 --ef-sweep 256 > $OUT/${TAG}_c2_${V:-default}.json 2>> $OUT/${TAG}_bench.err; echo "c2 [$V] rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/${TAG}_c*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "value", round(j["value"]), "ms/step", round(j["ms_per_step"], 3), "ef", j["config"]["ef_search"], "sweep", [(e["ef_search"], round(e["qps"])) for e in j.get("ef_sweep", [])])
PY
cd /tmp; export TMPDIR=/tmp
COS_BENCH_FULL_RECORD=${TAG}_full.json timeout 900 rocprofv3 --kernel-trace -d /tmp/p_tl -o tl -- python $R/bench.py --ef 112 --configs none --no-cpu-baseline --no-hbm-probe --steps 12 --warmup 3 --recall-queries 2048 > $OUT/${TAG}_bench.json 2>> $OUT/${TAG}_bench.err; echo "bench rc=$?"
cd $R/scripts; python trace_timeline.py /tmp/p_tl/tl_results.db 3 > $OUT/${TAG}_timeline.txt 2>&1; tail -2 $OUT/${TAG}_timeline.txt
