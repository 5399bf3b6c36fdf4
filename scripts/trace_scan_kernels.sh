#!/bin/bash
# rocprofv3 kernel traces of the two scan probes (round 6's new kernels): flat_scan_u8_areg on 4M x 768 u8 codes, flat_scan_q2_fp4_w8 on 10M x 768
# quaternary codes; summaries (scripts/rocprof_summary.py) -> gpurun_out/r06_scan_kernels_trace_*.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d /tmp/p_u8 -o u8 -- python $R/scripts/u8_scan_probe.py --reps 5 > $OUT/r06_scan_kernels_trace_u8_probe.jsonl 2> $OUT/r06_scan_kernels_trace_u8.err
python $R/scripts/rocprof_summary.py /tmp/p_u8/u8_results.db > $OUT/r06_scan_kernels_trace_u8.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_q2 -o q2 -- python $R/scripts/fp4_w8_probe.py --reps 5 > $OUT/r06_scan_kernels_trace_fp4_probe.jsonl 2> $OUT/r06_scan_kernels_trace_fp4.err
python $R/scripts/rocprof_summary.py /tmp/p_q2/q2_results.db > $OUT/r06_scan_kernels_trace_fp4.txt
grep -i "flat_scan\|flat_codes" $OUT/r06_scan_kernels_trace_u8.txt | head; grep -i "flat_scan" $OUT/r06_scan_kernels_trace_fp4.txt | head
