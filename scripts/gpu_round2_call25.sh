O=gpurun_out; mkdir -p $O
timeout 200 python scripts/bench_sparse.py > $O/r2_c25_sparse_before.json 2> $O/r2_c25_sparse_before.err; tail -1 $O/r2_c25_sparse_before.err; cat $O/r2_c25_sparse_before.json
