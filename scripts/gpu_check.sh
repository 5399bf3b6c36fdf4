#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_wave_reduce.py -m gpu -q -x > $OUT/r3_wave_reduce.log 2>&1; echo "wave_reduce rc=$?"; tail -5 $OUT/r3_wave_reduce.log
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r3_pytest_evalblock.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3_pytest_evalblock.log
COS_BENCH_FULL_RECORD=r3_evalblock_full_record.json timeout 600 python bench.py --configs none --cpu-seconds 4 > $OUT/r3_bench_evalblock.json 2> $OUT/r3_bench_evalblock.err; echo "bench rc=$?"
python - <<'P'
import json
j=json.loads(open("gpurun_out/r3_bench_evalblock.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"])
for k in ("single_batch_qps","single_batch_qps_one_wave_latency_kernel","single_batch_qps_throughput_kernel","recall_at_10","build_seconds"):
    print(k, j.get(k))
print(json.dumps(j.get("ef_sweep"))[:600])
print(json.dumps(j.get("parity"))[:400])
P
