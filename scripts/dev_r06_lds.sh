#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_lds}
timeout 900 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_walk_order.py tests/test_gpu_dim1024.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_builder.py tests/test_gpu_append.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.log
V="walk_upper_lds_pad=-1;base"
PROBE_N=12500000 PROBE_D=1024 PROBE_M0=256 PROBE_M=64 PROBE_EFC=256 PROBE_VARIANTS="$V" PROBE_EFS=112,128 PROBE_COLS=4294967295 PROBE_REPS=8 timeout 900 python scripts/table_probe.py > $OUT/${TAG}_probe_c4shard.jsonl 2> $OUT/${TAG}_c4.err; echo "probe c4 rc=$?"
PROBE_VARIANTS="$V" PROBE_EFS=64,128,256 PROBE_COLS=4294967295 PROBE_REPS=16 timeout 600 python scripts/table_probe.py > $OUT/${TAG}_probe_c2.jsonl 2> $OUT/${TAG}_c2.err; echo "probe c2 rc=$?"
TAG=$TAG python - <<'PY'
import json, os
tag = os.environ["TAG"]
for f in (f"gpurun_out/{tag}_probe_c2.jsonl", f"gpurun_out/{tag}_probe_c4shard.jsonl"):
    for l in open(f):
        j = json.loads(l)
        if "variant" in j:
            a = j["alone"]
            print(f[-16:-6], j["variant"].ljust(26), j["ef"], "same", j["ids_identical_to_first_config"], "qps", j["qps_2_in_flight"], "up", a["upper_ms"], "lo", a["lower_ms"], "tab", a["table_ms"])
PY
