#!/bin/bash
# GPU box: everything profiles/rNN/ holds for the round (run from the repo root): tools/round_profiles.sh r03 [part]
# part: all (default) | bench | prof | pmc | mergepmc | configs | sweep | latency | rgb | scaling | rehearsal
set -e
R=$1
PART=${2:-all}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $O/pmc
cd $GRAFT_REPO_ROOT
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
if want bench; then
  # 1. the default bench command (with the CPU baseline on the headline image)
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
  cut -c1-400 $O/bench_default.json
fi
if want prof; then
  # 2. the same command under rocprofv3 --kernel-trace --stats (2 steps, no CPU leg)
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_batch1536.csv
  rm -rf $O/stats
  head -8 $O/kernel_stats_batch1536.csv | cut -c1-160
fi
if want pmc; then
  # 3. HBM traffic of the dither kernel and of the lookup pass (separate --pmc passes, kernel trace only); raw rows kept whole
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 4 8 1 1 > /dev/null 2>&1 )
    f=$(find $O/t_$c -name "*counter_collection.csv" | head -1)
    python3 tools/pmc_summary.py $f gilbert_fast > $O/pmc/${c}_gilbert_fast.txt
    python3 tools/pmc_summary.py $f saliency_kernel >> $O/pmc/${c}_gilbert_fast.txt
    python3 - "$f" > $O/pmc/${c}_rows.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Dispatch_Id", "Counter_Name", "Counter_Value"])
for r in rows:
    if "gilbert_fast" in r["Kernel_Name"] or "saliency_kernel" in r["Kernel_Name"]:
        w.writerow([r["Kernel_Name"].split("(")[0], r.get("Dispatch_Id", ""), r["Counter_Name"], r["Counter_Value"]])
PY
    rm -rf $O/t_$c
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/l_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 4 8 1 0 2 > /dev/null 2>&1 )
    python3 tools/pmc_summary.py $(find $O/l_$c -name "*counter_collection.csv" | head -1) fast_lookup_pass1 > $O/pmc/${c}_lookup_pass1.txt
    python3 tools/pmc_summary.py $(find $O/l_$c -name "*counter_collection.csv" | head -1) fast_lookup_pass2 > $O/pmc/${c}_lookup_pass2.txt
    rm -rf $O/l_$c
  done
  # 4. SQ counters of the dither kernel and of the lookup pass
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/t_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 4 8 1 1 > /dev/null 2>&1 )
  python3 tools/pmc_summary.py $(find $O/t_sq -name "*counter_collection.csv" | head -1) gilbert_fast > $O/pmc/SQ_gilbert_fast.txt
  rm -rf $O/t_sq
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/l_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 4 8 1 0 2 > /dev/null 2>&1 )
  python3 tools/pmc_summary.py $(find $O/l_sq -name "*counter_collection.csv" | head -1) fast_lookup_pass1 > $O/pmc/SQ_lookup_pass1.txt
  rm -rf $O/l_sq
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/l_st -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 8 8 1 0 2 > /dev/null 2>&1 )
  cp $(find $O/l_st -name "*kernel_stats.csv" | head -1) $O/kernel_stats_lookup_only.csv
  rm -rf $O/l_st
  cat $O/pmc/*.txt
fi
if want mergepmc; then
  # 4b. SQ counters of the batch's merge kernel (the dense 128-thread variant: 1536 loops, six per CU)
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/m_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/batch_rate.py 1536 1 > /dev/null 2>&1 )
  python3 tools/pmc_summary.py $(find $O/m_sq -name "*counter_collection.csv" | head -1) merge_kernel > $O/pmc/SQ_merge_batch1536.txt
  rm -rf $O/m_sq
  cat $O/pmc/SQ_merge_batch1536.txt
fi
if want configs; then
  # 5. the other configurations at N = 1
  python3 bench.py --config cfg4 --steps 3 --warmup 1 > $O/bench_cfg4.json 2> /dev/null
  python3 bench.py --config cfg5 --steps 3 --warmup 1 > $O/bench_cfg5.json 2> /dev/null
  python3 bench.py --config cfg5 --no-dither --steps 2 --warmup 1 > $O/bench_cfg5_nodither.json 2> /dev/null
  cut -c1-300 $O/bench_cfg4.json $O/bench_cfg5.json $O/bench_cfg5_nodither.json
fi
if want sweep; then
  # 6. batch size against throughput (the headline needs every image of a batch resident)
  for b in 64 128 256 512 1024 1280 1536; do
    python3 bench.py --batch $b --steps 2 --warmup 1 --cpu-sample 0 --no-extras 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('batch %4d: %8.1f Mpx/s  %.3f ms per image  amortised %s' % (d['config']['batch'], d['value'], d['config']['ms_per_image'], {k.split(' ')[0]: v for k, v in d['amortised_ms_per_image'].items()}))"
  done > $O/batch_sweep.txt
  cat $O/batch_sweep.txt
fi
if want latency; then
  # 7. one image alone: the merge team against the single workgroup
  for h in 0 1 3 7; do
    echo "== NQ_MERGE_HELPERS=$h, production (unstamped) merge kernel"; NQ_MERGE_STATS=0 NQ_MERGE_HELPERS=$h python3 tools/latency.py 4096 1 2>&1 | grep "^latency\|palette sha"
    echo "== NQ_MERGE_HELPERS=$h, stamped build (NQ_MERGE_STATS=1: phase ticks)"; NQ_MERGE_HELPERS=$h python3 tools/latency.py 4096 1 2>&1 | grep -v amdgpu.ids
  done > $O/latency_merge_team.txt
  cat $O/latency_merge_team.txt
fi
if want rgb; then
  # 8. the RGB quantizer
  { python3 tools/batch_rate.py 1536 0 2>&1 | grep -E "^kind|batch phases|gave up" | tail -3; python3 tools/latency.py 4096 0 2>&1 | grep -v amdgpu.ids; python3 tools/latency.py 4096 0 uniform 2>&1 | grep -v amdgpu.ids; } > $O/rgb_kind.txt
  cat $O/rgb_kind.txt
fi
if want scaling; then
  # 9. the share ONE rank of an N-GPU run has, measured alone on this GPU (cfg 4: 64 / N frames; cfg 5: one band + the whole histogram)
  python3 tools/scaling_predict.py 2> /dev/null | grep -v amdgpu.ids > $O/scaling_predict.txt
  cat $O/scaling_predict.txt
fi
if want rehearsal; then
  # 10. REHEARSAL ONLY: two ranks on this ONE GPU under gloo (both share the card: the rates mean nothing, the code path is the point)
  for cfg in cfg3 cfg4 cfg5; do
    extra=""; [ $cfg = cfg3 ] && extra="--batch 128 --cpu-sample 0"
    NQ_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 2 --warmup 1 --config $cfg $extra > $O/rehearsal_gloo_2ranks_$cfg.json 2> $O/rehearsal_$cfg.err || echo "rehearsal $cfg failed"
    cut -c1-260 $O/rehearsal_gloo_2ranks_$cfg.json
  done
fi
ls -la $O $O/pmc
