#!/bin/bash
# GPU call 18 of round 2: latency walk after the scalar-instruction diet of its evaluation passes (parity against the throughput
# kernel at every size + timing)
O=gpurun_out; mkdir -p $O
SWEEP_BS=256,2048,4096 SWEEP_LA=4 timeout 300 python scripts/latency_sweep.py > $O/r2_c18_latency_sweep.jsonl 2> $O/r2_c18_latency_sweep.err; tail -2 $O/r2_c18_latency_sweep.err; cut -c1-230 $O/r2_c18_latency_sweep.jsonl
