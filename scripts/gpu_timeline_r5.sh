#!/bin/bash
# rocprofv3 kernel trace of the PRODUCT path at cfg1 / cfg3 on 2 and 3 lanes: how many launches are in flight (timeline_concurrency.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/lanes"; mkdir -p "$OUT"; : > "$OUT/timeline_lanes.txt"
for cfg in cfg1 cfg3; do
for lanes in 2 3; do
  D=/tmp/trace_${cfg}_$lanes
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o trace -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --no-plugin-path --no-secondary --streams $lanes --regions 1 --sustain 0 > /dev/null 2>&1 )
  echo "== $cfg, product path, $lanes lanes (the longest stretches are the spin-up and the 200 timed steps; single-stream passes of the same run show 1.00 in flight)" >> "$OUT/timeline_lanes.txt"
  python scripts/timeline_concurrency.py $D >> "$OUT/timeline_lanes.txt" 2>&1
  rm -rf $D
done
done
cat "$OUT/timeline_lanes.txt"
