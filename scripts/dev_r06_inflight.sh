#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_inflight}
for S in 2 3 4; do for V in "walk_table_after_sort=0" ""; do
COS_TUNING="$V" COS_BENCH_FULL_RECORD=${TAG}_x.json timeout 900 python bench.py --inflight $S --ef 112 --configs none --no-cpu-baseline --no-hbm-probe --steps 24 --warmup 6 --recall-queries 2048 > $OUT/${TAG}_c4_S${S}_${V:-default}.json 2>> $OUT/${TAG}_bench.err; echo "c4 S=$S [$V] rc=$?"
done; done
for S in 2 3; do for V in "walk_table_after_sort=0" ""; do
COS_TUNING="$V" COS_BENCH_FULL_RECORD=${TAG}_x.json timeout 900 python bench.py --inflight $S --workload c2 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --steps 24 --warmup 6 --recall-queries 2048 --ef-sweep 256 > $OUT/${TAG}_c2_S${S}_${V:-default}.json 2>> $OUT/${TAG}_bench.err; echo "c2 S=$S [$V] rc=$?"
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/${TAG}_c*.json")):
    j = json.load(open(f))
    p = j["roofline"]["parts"]
    print(f.split("/")[-1].ljust(52), "value", round(j["value"]), "ms/step", round(j["ms_per_step"], 3), "up", round(p["walk_upper"]["ms"], 2), "lo", round(p["walk_lower"]["ms"], 2), "gemm co-run", round(p["level_table_gemm"]["ms_next_to_a_walk"], 2), "sweep", [(e["ef_search"], round(e["qps"])) for e in j.get("ef_sweep", [])])
PY
