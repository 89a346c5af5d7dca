# virtual-merge speculation against the same tree without it (libnquant_hip.novm.so: NQ_BUILD_TAG=novm NQ_BUILD_DEFS=-DNQ_TEAM_VM=0), same box
NOVM=$GRAFT_REPO_ROOT/nquant.android_amd/libnquant_hip.novm.so
for rep in 1 2; do
  echo "vm   $(NQ_MERGE_STATS=0 python3 tools/latency.py 4096 1 2>&1 | grep '^latency\|palette sha' | cut -c1-28 | tr '\n' ' ')"
  echo "novm $(NQ_LIB=$NOVM NQ_MERGE_STATS=0 python3 tools/latency.py 4096 1 2>&1 | grep '^latency\|palette sha' | cut -c1-28 | tr '\n' ' ')"
done
for h in 3; do
  echo "helpers $h vm   $(NQ_MERGE_HELPERS=$h NQ_MERGE_STATS=0 python3 tools/latency.py 4096 1 2>&1 | grep '^latency' | cut -c1-24)"
  echo "helpers $h novm $(NQ_MERGE_HELPERS=$h NQ_LIB=$NOVM NQ_MERGE_STATS=0 python3 tools/latency.py 4096 1 2>&1 | grep '^latency' | cut -c1-24)"
done
python3 tools/latency.py 4096 1 2>&1 | grep "^team\|^control\|virtual" | cut -c1-260
