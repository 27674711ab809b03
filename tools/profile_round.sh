#!/bin/bash
# Profiles the judge reads (run on the GPU box; writes under gpurun_out/$1, copy the summaries into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the bench command (headline kernel + SLAM leg)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over k_ens_block      -> tools/parse_pmc.py
#   3. the same two passes over every k_round dispatch of SLAM config 3                -> tools/parse_round_pmc.py
# The profiled program is python3 itself (no env / bash -c hop between rocprofv3 and the program).
set -o pipefail
OUT=${1:-gpurun_out/prof}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-end-to-end --slam-cpu-steps 0"
# (the stats pass runs the DRIVER's command shape - 20 timed blocks behind 5 warm-up ones: the first launches of a short run are
#  ~13 % slower, and the average of a 3-block run does not agree with the bench line's)
BS="$R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-end-to-end --slam-cpu-steps 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -o r4 -- python3 $BS --slam-steps 256 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/stats.err || exit 1
find $R/$OUT/stats -name "*kernel_trace*" -delete; find $R/$OUT/stats -name "*agent_info*" -delete
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_ens_block --output-format csv -d $R/$OUT/pmc_fetch -o r4 -- python3 $B --slam-steps 0 > $R/$OUT/pmc_fetch.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_ens_block --output-format csv -d $R/$OUT/pmc_write -o r4 -- python3 $B --slam-steps 0 > $R/$OUT/pmc_write.log 2>&1 || exit 3
echo block pmc done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_round --output-format csv -d $R/$OUT/slam_fetch -o r4 -- python3 $R/tools/experiments/slam_flags.py 0 > $R/$OUT/slam_fetch.log 2>&1 || exit 4
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_round --output-format csv -d $R/$OUT/slam_write -o r4 -- python3 $R/tools/experiments/slam_flags.py 0 > $R/$OUT/slam_write.log 2>&1 || exit 5
echo slam pmc done
cd $R
# summaries (the raw counter files stay on the box: gpurun merges at most 64 MiB back)
KS=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $KS $OUT/kernel_stats.csv
python3 tools/parse_pmc.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $KS $OUT/pmc_traffic.json 5080000000 k_ens_block > $OUT/pmc_traffic.txt
# (time per timestep of the UNPROFILED run: bench.py's slam leg; the profiled passes are slower)
python3 tools/parse_round_pmc.py $(find $OUT/slam_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/slam_write -name "*counter_collection.csv" | head -1) 640 ${2:-114} > $OUT/slam_traffic.txt
cat $OUT/pmc_traffic.txt $OUT/slam_traffic.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/slam_fetch $OUT/slam_write
du -sh $OUT
