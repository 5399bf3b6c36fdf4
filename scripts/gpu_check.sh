#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
PROBE_EFS=64 timeout 300 python $R/scripts/order_probe.py > $OUT/r3_order_probe_inflight.jsonl 2> $OUT/r3_order_probe_inflight.err; echo "c2 rc=$?"; cat $OUT/r3_order_probe_inflight.jsonl; tail -2 $OUT/r3_order_probe_inflight.err
