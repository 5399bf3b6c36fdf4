#!/bin/bash
# GPU call 5 of round 2: full GPU suite, forced-dist bench through the shard set, final profiles, c5 under rocprofv3, c4shard exact.
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 1200 > $O/r2_c5_pytest.log 2>&1; tail -12 $O/r2_c5_pytest.log
COS_FORCE_DIST=1 COS_SHARDSET_FORCE_RCCL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-probe --ef 64 --ef-sweep "" --recall-queries 2048 > $O/r2_c5_bench_forced_dist.json 2> $O/r2_c5_bench_forced_dist.err; tail -3 $O/r2_c5_bench_forced_dist.err; python -c "
import json; b=json.load(open('$O/r2_c5_bench_forced_dist.json')); print(b['value'], b['config']['exchange'], b['recall_at_10'])"
bash scripts/final_profile.sh
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python /root/repo/scripts/bench_c5.py > /root/repo/$O/r2_c5_c5.json 2> /root/repo/$O/r2_c5_c5.err )
python scripts/rocprof_summary.py /tmp/p_c5/c5_results.db > $O/r2_c5_c5_kt.txt; cat $O/r2_c5_c5.json; grep -E "bm25|rrf" $O/r2_c5_c5_kt.txt
COS_BUILD_PROFILE=1 timeout 1200 python bench.py --workload c4shard --steps 8 --warmup 2 --coalesce 64 --visited exact --build-visited exact --ef-construction 256 --ef-sweep "exact:256,ref:256" --cpu-seconds 8 > $O/r2_c5_c4shard_exact.json 2> $O/r2_c5_c4shard_exact.err
tail -3 $O/r2_c5_c4shard_exact.err; python -c "
import json; b=json.load(open('$O/r2_c5_c4shard_exact.json')); print({k:b[k] for k in ['value','recall_at_10','recall_lower95','build_seconds','parity_vs_oracle','cpu_baseline','ef_selection','ef_sweep']}, b['roofline']['frac'], b['config']['ef_search'])"
