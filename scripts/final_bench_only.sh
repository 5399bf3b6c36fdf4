R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
COS_BENCH_FULL_RECORD=final_bench_all_configs_full_record.json timeout 1700 python bench.py > $OUT/final_bench_all_configs.json 2> $OUT/final_bench_all_configs.err; echo "bench rc=$?"; tail -3 $OUT/final_bench_all_configs.err
python -c "
import json; j=json.load(open('$OUT/final_bench_all_configs.json')); print('value', j['value'], 'total_s', j['total_seconds']); print(json.dumps(j.get('append_and_delete_on_the_resident_graph'), indent=0)[:1500])"
COS_FORCE_DIST=1 COS_BENCH_FULL_RECORD=forced_dist_full.json timeout 900 python bench.py --n 1000000 --configs none --no-cpu-baseline --no-hbm-probe --steps 8 --warmup 2 --recall-queries 2048 > $OUT/final_bench_forced_dist_world_of_one.json 2> $OUT/final_bench_forced_dist.err; echo "forced dist rc=$?"; tail -2 $OUT/final_bench_forced_dist.err; head -c 900 $OUT/final_bench_forced_dist_world_of_one.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
