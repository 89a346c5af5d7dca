#!/bin/bash
# GPU box: time the fast dither kernel with stages left out (NQ_FAST_DEBUG bit mask; results are wrong, timing only).
# Needs a library built with NQ_BUILD_KNOCKOUT=1.
for m in "$@"; do
  export NQ_FAST_DEBUG=$m
  mkdir -p gpurun_out/ko
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ko/m$m -o p -- python3 $GRAFT_REPO_ROOT/tools/dither_only.py 4096 5 8 1 1 > $GRAFT_REPO_ROOT/gpurun_out/ko/m$m.log 2>&1 )
  echo "mask $m: $(grep gilbert_fast gpurun_out/ko/m$m/p_kernel_stats.csv | awk -F, '{print $(NF-4)}') ns avg"
done
