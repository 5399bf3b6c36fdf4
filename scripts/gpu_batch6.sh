#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
COS_BENCH_FULL_RECORD=r04_b6_bench_c2_full.json timeout 400 python bench.py --configs none --ef-sweep "" > $OUT/r04_b6_bench_c2.json 2> $OUT/r04_b6_bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b6_bench_c2.json")); print(r["value"], json.dumps(r["host_api_pcie_inclusive"])[:1500]); print(json.dumps(r["roofline"]["parts"])[:1800])
except Exception as e: print("parse", e)
PY
tail -3 $OUT/r04_b6_bench_c2.err
COS_BENCH_FULL_RECORD=r04_b6_bench_c4_m_full.json timeout 900 python bench.py --workload smoke --no-cpu-baseline --no-hbm-probe --ef-sweep "" --configs c4shard_ref_m0_256_m_64,c4shard_ref_m0_256_m_128 > $OUT/r04_b6_bench_c4_m.json 2> $OUT/r04_b6_bench_c4_m.err; echo "c4 rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b6_bench_c4_m_full.json"))
    for k,v in r["configs"].items(): print(k, {x:v.get(x) for x in ("qps","recall_at_10","meets_recall_target","build_seconds","seconds","error")}, v.get("ef_selection"))
except Exception as e: print("c4 parse", e)
PY
tail -3 $OUT/r04_b6_bench_c4_m.err
