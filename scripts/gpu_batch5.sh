#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_walk_table.py -m gpu -q -x > $OUT/r04_b5_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_b5_pytest.log
COS_BENCH_FULL_RECORD=r04_b5_bench_c2_full.json timeout 400 python bench.py --configs none > $OUT/r04_b5_bench_c2.json 2> $OUT/r04_b5_bench_c2.err; echo "bench rc=$?"; head -c 3000 $OUT/r04_b5_bench_c2.json; echo; tail -3 $OUT/r04_b5_bench_c2.err
timeout 400 python scripts/bench_c3.py --n 1000000 --walk-n 1000000 --cpu-seconds 3 > $OUT/r04_b5_c3_walk_1M.json 2> $OUT/r04_b5_c3_walk_1M.err; echo "c3 rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b5_c3_walk_1M.json")); print(json.dumps(r.get("hnsw_walk_quaternary"))[:1500])
except Exception as e: print("c3 parse", e)
PY
COS_BENCH_FULL_RECORD=r04_b5_bench_c4_m0_full.json timeout 900 python bench.py --workload smoke --no-cpu-baseline --no-hbm-probe --ef-sweep "" --configs c4shard_ref,c4shard_ref_m0_128,c4shard_ref_m0_256 > $OUT/r04_b5_bench_c4_m0.json 2> $OUT/r04_b5_bench_c4_m0.err; echo "c4 rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b5_bench_c4_m0.json"))
    for k,v in r["configs"].items(): print(k, {x:v.get(x) for x in ("qps","recall_at_10","meets_recall_target","ef_search","build_seconds","seconds","error")}, v.get("parity_vs_oracle"), (v.get("roofline") or {}).get("frac"))
except Exception as e: print("c4 parse", e)
PY
tail -3 $OUT/r04_b5_bench_c4_m0.err
