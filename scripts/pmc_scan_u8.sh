cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/p_a -o a -- python $R/scripts/u8_scan_probe.py --reps 2 > $OUT/r06_u8q_pmc_a.jsonl 2> $OUT/r06_u8q_pmc_a.err
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_b -o b -- python $R/scripts/u8_scan_probe.py --reps 2 > $OUT/r06_u8q_pmc_b.jsonl 2> $OUT/r06_u8q_pmc_b.err
python $R/scripts/pmc_scan_counters.py $(find /tmp/p_a /tmp/p_b -name "*.db") > $OUT/r06_u8q_pmc_counters.jsonl 2> $OUT/r06_u8q_pmc_counters.err
cat $OUT/r06_u8q_pmc_counters.jsonl; for f in counters a b; do tail -n 3 $OUT/r06_u8q_pmc_$f.err; done
