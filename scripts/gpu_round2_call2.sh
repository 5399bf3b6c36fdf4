#!/bin/bash
# GPU call 2 of round 2: full GPU test suite (device link, shardset, walk chain), default bench, launch/PB sweeps, c4shard.
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 1200 > $O/r2_c2_pytest.log 2>&1; tail -25 $O/r2_c2_pytest.log
COS_BUILD_PROFILE=1 python bench.py --steps 20 --warmup 5 --coalesce 128 > $O/r2_c2_bench_c128.json 2> $O/r2_c2_bench_c128.err; tail -2 $O/r2_c2_bench_c128.err; head -c 400 $O/r2_c2_bench_c128.json; echo
python scripts/sweep_launch.py --queries 8192,16384,32768,65536 --inflight 1,2 --ef 64,256 > $O/r2_c2_sweep_chain.jsonl 2>&1; tail -17 $O/r2_c2_sweep_chain.jsonl
COS_WALK_PB=8 python scripts/sweep_launch.py --queries 256,1024,4096,8192,32768 --inflight 1 --ef 64,256 > $O/r2_c2_sweep_pb8.jsonl 2>&1; tail -11 $O/r2_c2_sweep_pb8.jsonl
if python -m pytest tests/test_gpu_builder.py -x -q --timeout 600 > $O/r2_c2_builder_gate.log 2>&1; then
  COS_BUILD_PROFILE=1 timeout 900 python bench.py --workload c4shard --steps 8 --warmup 2 --coalesce 64 --ef-construction 256 --ef-sweep "ref:256,exact:64" --cpu-seconds 8 > $O/r2_c2_c4shard_ref.json 2> $O/r2_c2_c4shard_ref.err
  tail -3 $O/r2_c2_c4shard_ref.err; head -c 600 $O/r2_c2_c4shard_ref.json; echo
else
  tail -20 $O/r2_c2_builder_gate.log
fi
