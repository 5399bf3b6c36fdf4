#!/bin/bash
# GPU call 26 of round 2: the driver's bench command once more (host-API / PCIe-inclusive rate added to the JSON line)
O=gpurun_out; mkdir -p $O
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_c26_bench.json 2> $O/r2_c26_bench.err; tail -1 $O/r2_c26_bench.err
python -c "
import json;d=json.load(open('$O/r2_c26_bench.json'));print({k:d[k] for k in ('value','ms_per_step','single_batch_qps','host_api_pcie_inclusive')}, d['roofline']['frac'], d['parity_vs_oracle'])"
