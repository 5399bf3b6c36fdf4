#!/bin/bash
# full GPU suite + default bench (driver flags) + c3 / c5 under rocprofv3 kernel trace
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 1200 > $O/r2_c34_pytest.log 2>&1; tail -6 $O/r2_c34_pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_c34_bench.json 2> $O/r2_c34_bench.err; tail -2 $O/r2_c34_bench.err; python -c "
import json; b=json.load(open('$O/r2_c34_bench.json')); print({k:b[k] for k in ['value','ms_per_step','steps','warmup','recall_at_10','single_batch_qps','parity_vs_oracle']}, b['roofline']['frac'], b['roofline']['traffic'], b['cpu_baseline'])"
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d /tmp/p_c3 -o c3 -- python /root/repo/scripts/bench_c3.py --walk-n 0 > /root/repo/$O/r2_c34_c3.json 2> /root/repo/$O/r2_c34_c3.err )
python scripts/rocprof_summary.py /tmp/p_c3/c3_results.db > $O/r2_c34_c3_kt.txt; cat $O/r2_c34_c3.json; grep -E "flat_scan|flat_codes|flat_select|flat_rerank|expand" $O/r2_c34_c3_kt.txt
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python /root/repo/scripts/bench_c5.py > /root/repo/$O/r2_c34_c5.json 2> /root/repo/$O/r2_c34_c5.err )
python scripts/rocprof_summary.py /tmp/p_c5/c5_results.db > $O/r2_c34_c5_kt.txt; cat $O/r2_c34_c5.json; grep -E "bm25|rrf" $O/r2_c34_c5_kt.txt
