#!/bin/bash
# does a wider construction beam let the metric's shard meet the recall target at a smaller ef_search?  (ef_construction: tests/test.py uses 512, config.toml 128; the shard's record used 256)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_efc}
timeout 600 python -m pytest tests/test_gpu_append.py -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log
for EFC in 384 512; do
COS_BENCH_FULL_RECORD=${TAG}_${EFC}_full.json timeout 900 python bench.py --ef-construction $EFC --configs none --no-cpu-baseline --no-hbm-probe --steps 12 --warmup 3 > $OUT/${TAG}_${EFC}.json 2> $OUT/${TAG}_${EFC}.err; echo "efc $EFC rc=$?"
python - <<PY
import json
j = json.load(open("$OUT/${TAG}_${EFC}.json"))
print("efc", $EFC, "value", j["value"], "ef", j["config"]["ef_search"], "recall", j["recall_at_10"], j["recall_lower95"], "build_s", j["build_seconds"], "sel", j["ef_selection"])
PY
done
