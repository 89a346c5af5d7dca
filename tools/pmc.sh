#!/bin/bash
# Usage (GPU box, repo root): tools/pmc.sh <name> <kernel substring> "<counters>" <python script + args...>
# one rocprofv3 --pmc pass (kernel trace only), summed per kernel
set -e
name=$1; sub=$2; ctrs=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o p -- python3 "$@" > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*counter_collection.csv" | head -1)
echo "== $f"
python3 tools/pmc_summary.py $f $sub
