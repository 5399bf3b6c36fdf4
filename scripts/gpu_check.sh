#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python scripts/bench_meta.py > $OUT/r3_meta_200k.json 2> $OUT/r3_meta.err; echo "meta rc=$?"; cat $OUT/r3_meta_200k.json; tail -3 $OUT/r3_meta.err
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r3_bench_full2.json 2> $OUT/r3_bench_full2.err; echo "bench rc=$?"; tail -3 $OUT/r3_bench_full2.err
