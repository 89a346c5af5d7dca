#!/bin/bash
# GPU box: everything profiles/rNN/ holds for the round (run from the repo root): tools/round_profiles.sh r02
set -e
R=$1
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $O/pmc
cd $GRAFT_REPO_ROOT
# 1. the default bench command (with the CPU baseline)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
# 2. the same command under rocprofv3 --kernel-trace --stats (2 steps)
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_batch1024.csv
rm -rf $O/stats
# 3. HBM traffic of the dither kernel (separate --pmc passes, kernel trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 4 8 1 1 > /dev/null 2>&1 )
  python3 tools/pmc_summary.py $(find $O/t_$c -name "*counter_collection.csv" | head -1) gilbert_fast > $O/pmc/${c}_gilbert_fast.txt
  python3 tools/pmc_summary.py $(find $O/t_$c -name "*counter_collection.csv" | head -1) saliency_kernel >> $O/pmc/${c}_gilbert_fast.txt
  grep -E "gilbert_fast|saliency_kernel|Kernel_Name" $(find $O/t_$c -name "*counter_collection.csv" | head -1) | cut -d, -f9-19 | cut -c1-60,200- | head -12 > $O/pmc/${c}_rows.csv || true
  rm -rf $O/t_$c
done
# 4. SQ counters of the dither kernel
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/t_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 4 8 1 1 > /dev/null 2>&1 )
python3 tools/pmc_summary.py $(find $O/t_sq -name "*counter_collection.csv" | head -1) gilbert_fast > $O/pmc/SQ_gilbert_fast.txt
rm -rf $O/t_sq
# 5. the other configurations at N = 1
python3 bench.py --config cfg4 --steps 2 --warmup 1 > $O/bench_cfg4.json 2> /dev/null
python3 bench.py --config cfg5 --steps 2 --warmup 1 > $O/bench_cfg5.json 2> /dev/null
ls -la $O $O/pmc
cat $O/bench_default.json | cut -c1-1500
cat $O/pmc/*.txt
