#!/bin/bash
# round 6, third GPU call: the automatic level-table rule's constant at the beam widths around the metric's shard's ef, and the adjacency-side norms' policy
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_rule}
V="base;walk_table_rule_c=8;walk_table_rule_c=10;walk_table_rule_c=12;walk_table_rule_c=16"
PROBE_VARIANTS="$V" PROBE_EFS=96,128,256 PROBE_COLS=4294967295 PROBE_REPS=12 timeout 600 python scripts/table_probe.py > $OUT/${TAG}_probe_c2.jsonl 2> $OUT/${TAG}_c2.err; echo "probe c2 rc=$?"
PROBE_N=12500000 PROBE_D=1024 PROBE_M0=256 PROBE_M=64 PROBE_EFC=256 PROBE_VARIANTS="$V" PROBE_EFS=80,96,112,128 PROBE_COLS=4294967295 PROBE_REPS=8 timeout 900 python scripts/table_probe.py > $OUT/${TAG}_probe_c4shard.jsonl 2> $OUT/${TAG}_c4.err; echo "probe c4 rc=$?"
TAG=$TAG python - <<'PY'
import json, os
tag = os.environ["TAG"]
for f in (f"gpurun_out/{tag}_probe_c2.jsonl", f"gpurun_out/{tag}_probe_c4shard.jsonl"):
    for l in open(f):
        j = json.loads(l)
        if "variant" in j:
            a = j["alone"]
            print(f[-16:-6], j["variant"].ljust(22), j["ef"], "same", j["ids_identical_to_first_config"], "qps", j["qps_2_in_flight"], "up", a["upper_ms"], "lo", a["lower_ms"], "tab", a["table_ms"], "cols", j["table_cols"], "lmin", j["table_level_min"])
PY
