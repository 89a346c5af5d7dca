#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof.sh <name> <python script + args...>
# rocprofv3 --kernel-trace --stats of the command; CSV summary under gpurun_out/<name>/
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 "$@" > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "== $f"
head -14 $f | cut -c1-200
