#!/bin/bash
# First GPU call of the next round: the candidates written at the end of round 4 without a device at hand (each is OFF by default,
# each has its parity tests behind COS_CANDIDATES=1).  Run from the repo root; results land in gpurun_out/.
#   1. learned-sparse index, packed posting layout (kernels_sparse.hip sparse_packed_kernel):
#      parity tests, then scripts/bench_sparse.py with the unpacked layout, the packed one, and the packed one with 16 postings per
#      lane and step — kernel time, fraction of the HBM roof, parity 256 / 256 each.
# If the packed layout is green and faster: make it the default in cos_sparse_create (COS_SPARSE_PACKED unset -> packed when the
# collection fits 24-bit ids), drop the `candidates` mark from tests/test_sparse.py, re-run scripts/bench_sparse.py under
# rocprofv3 (kernel trace + SQ_INSTS_VALU) for profiles/.
# usage: bash scripts/round5_candidates.sh [sparse] [host] [walk]   (no argument = all three, ~30 GPU-minutes; one part = 5-15)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
WHAT="${*:-sparse host walk}"
part_sparse() {
COS_CANDIDATES=1 timeout 600 python -m pytest tests/test_sparse.py -m gpu -q > $OUT/cand_sparse_pytest.log 2>&1; echo "sparse candidates pytest rc=$?"; tail -3 $OUT/cand_sparse_pytest.log
timeout 300 python scripts/bench_sparse.py > $OUT/cand_sparse_unpacked.json 2> $OUT/cand_sparse_unpacked.err; echo "unpacked rc=$?"
COS_SPARSE_PACKED=1 timeout 300 python scripts/bench_sparse.py > $OUT/cand_sparse_packed.json 2> $OUT/cand_sparse_packed.err; echo "packed rc=$?"
COS_SPARSE_PACKED=1 COS_SPARSE_SPU=16 timeout 300 python scripts/bench_sparse.py > $OUT/cand_sparse_packed_spu16.json 2> $OUT/cand_sparse_packed_spu16.err; echo "packed spu16 rc=$?"
python - <<'PY'
import json
for f in ("unpacked", "packed", "packed_spu16"):
    try:
        r = json.load(open(f"gpurun_out/cand_sparse_{f}.json"))
        print(f, "kernel_ms", round(r["roofline"]["per_launch"]["avg_ms"], 3), "frac", round(r["roofline"]["frac"], 3), "host_ms", round(r["ms_per_batch_host_api"], 3),
              "parity", r["parity_vs_oracle"])
    except Exception as e:
        print(f, "failed:", e)
PY
}
part_host() {
#   2. host-buffer API, one synchronous caller of 32 768 queries: COS_HOST_STAGE_THREADS = helper threads that stage the call's chunks
#      into pinned memory (engine.hip search_host_pipelined, host_stage.h).  Parity test, then scripts/host_api_sweep.py per setting.
#      If >= 0.9 x the resident rate with some n: make that n the default (stage_pool()), rerun bench.py for host_api_pcie_inclusive.
COS_CANDIDATES=1 timeout 900 python -m pytest tests/test_gpu_host_stage.py -m gpu -q > $OUT/cand_host_stage_pytest.log 2>&1; echo "host stage pytest rc=$?"; tail -3 $OUT/cand_host_stage_pytest.log
for T in 0 2 4 8; do
  COS_HOST_STAGE_THREADS=$T timeout 400 python scripts/host_api_sweep.py > $OUT/cand_host_stage_threads_$T.json 2> $OUT/cand_host_stage_threads_$T.err; echo "threads $T rc=$?"; head -c 700 $OUT/cand_host_stage_threads_$T.json; echo
done
}
part_walk() {
#   3. walk, upper range of the split launch with the four-row-buffer variant (53 VGPRs = 8 waves per SIMD instead of 69 = 7; the range
#      is a latency chain over table levels since level 3 joined the table): COS_WALK_PB_UPPER=4 against the default, ef 64 and 256.
#      PROBE_COLS=4294967295 = the automatic table rule.  If faster: make it walk_pb_policy's default for that range.
for PBU in 0 4; do
  COS_WALK_PB_UPPER=$PBU PROBE_EFS=64,256 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_pb_upper_$PBU.jsonl 2> $OUT/cand_walk_pb_upper_$PBU.err; echo "pb_upper $PBU rc=$?"; cut -c1-400 $OUT/cand_walk_pb_upper_$PBU.jsonl
done
#   4. walk with table values gathered ahead (kernels_walk_spec.hip, walk_kernel.inc COS_WALK_SPEC): the parity
#      suites of the walk under COS_WALK_SPEC_TABLE=1 (two window entries gathered ahead; =4: all four; read once per process), then
#      the same probe as in 3.
#      If green and faster: launch it by default for the launches it covers (launch_walk_r), and give the profile scripts its kernel
#      name (scripts/final_profile.sh, gpu_profile_only.sh, bench.py's roofline.kernel: they look for "walk_kernel<0, 1, 1, true, false, 8>").
for SP in 2 4 6 8; do
  COS_WALK_SPEC_TABLE=$SP COS_WALK_SPEC_WARM=1 timeout 600 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_walk_order.py tests/test_gpu_dim1024.py -m gpu -q -x > $OUT/cand_walk_spec_pytest_$SP.log 2>&1; echo "walk spec $SP (+ warm) pytest rc=$?"; tail -2 $OUT/cand_walk_spec_pytest_$SP.log
done
COS_WALK_SPEC_TABLE=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_builder.py -m gpu -q -x > $OUT/cand_walk_spec_pytest_parity.log 2>&1; echo "walk spec 4, parity / edges / builder suites rc=$?"; tail -2 $OUT/cand_walk_spec_pytest_parity.log
COS_WALK_SPEC_TABLE=1 PROBE_EFS=64,256 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_spec.jsonl 2> $OUT/cand_walk_spec.err; echo "walk spec probe rc=$?"; cut -c1-400 $OUT/cand_walk_spec.jsonl
COS_WALK_SPEC_TABLE=4 PROBE_EFS=64,256 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_spec4.jsonl 2> $OUT/cand_walk_spec4.err; echo "walk spec (4 entries ahead) probe rc=$?"; cut -c1-400 $OUT/cand_walk_spec4.jsonl
COS_WALK_SPEC_TABLE=6 PROBE_EFS=64,256 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_spec6.jsonl 2> $OUT/cand_walk_spec6.err; echo "walk spec (six-entry window, all ahead; 6.1 KB of LDS per wave = 6 waves per SIMD) probe rc=$?"; cut -c1-400 $OUT/cand_walk_spec6.jsonl
COS_WALK_SPEC_TABLE=8 PROBE_EFS=64,256 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_spec8.jsonl 2> $OUT/cand_walk_spec8.err; echo "walk spec (eight-entry window, all ahead; 7.7 KB of LDS per wave = 5 waves per SIMD) probe rc=$?"; cut -c1-400 $OUT/cand_walk_spec8.jsonl
COS_WALK_SPEC_TABLE=1 COS_WALK_PB_UPPER=4 PROBE_EFS=64 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_spec_pbu4.jsonl 2> $OUT/cand_walk_spec_pbu4.err; echo "walk spec + pb_upper 4 rc=$?"; cut -c1-400 $OUT/cand_walk_spec_pbu4.jsonl
# one 256-query batch is the same chain with nothing to hide it behind (~400 table-level expansions per query): the candidate there
for SP in 0 2 4 6 8; do
  COS_WALK_SPEC_TABLE=$SP timeout 300 python scripts/single_batch_probe.py > $OUT/cand_single_batch_spec_$SP.jsonl 2> $OUT/cand_single_batch_spec_$SP.err; echo "single batch, spec $SP rc=$?"; cut -c1-500 $OUT/cand_single_batch_spec_$SP.jsonl
done
# + the top of every query's table row fetched in one instruction before the first level (COS_WALK_SPEC_WARM=1)
COS_WALK_SPEC_TABLE=2 COS_WALK_SPEC_WARM=1 PROBE_EFS=64 PROBE_COLS=4294967295 timeout 400 python scripts/table_probe.py > $OUT/cand_walk_spec_warm.jsonl 2> $OUT/cand_walk_spec_warm.err; echo "walk spec + warm rc=$?"; cut -c1-400 $OUT/cand_walk_spec_warm.jsonl
COS_WALK_SPEC_TABLE=2 COS_WALK_SPEC_WARM=1 PROBE_EFS=64 timeout 300 python scripts/single_batch_probe.py > $OUT/cand_single_batch_spec_warm.jsonl 2> $OUT/cand_single_batch_spec_warm.err; echo "single batch, spec + warm rc=$?"; cut -c1-500 $OUT/cand_single_batch_spec_warm.jsonl
}
for P in $WHAT; do part_$P; done
