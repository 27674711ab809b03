R=$GRAFT_REPO_ROOT; OUT=gpurun_out/r3r; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-end-to-end --slam-steps 0 > $R/$OUT/plain_short.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-end-to-end --slam-steps 0 > $R/$OUT/plain_long.json 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -o r3 -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-end-to-end --slam-steps 0 > $R/$OUT/rocprof_long.json 2> $R/$OUT/stats.err
cp $(find $R/$OUT/stats -name "*kernel_stats.csv" | head -1) $R/$OUT/kernel_stats_long.csv
rm -rf $R/$OUT/stats
cd $R
for f in plain_short plain_long rocprof_long; do python3 -c "
import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], r['avg_launch_us'], r['launches_timed'], r['frac'], r['valu']['shader_clock_mhz_during_probe_by_waves'])"; done
grep "k_ens_block" $OUT/kernel_stats_long.csv | cut -c1-200
