#!/bin/bash
# Overlap mode with up to four lanes: the product path at 2 / 3 / 4 lanes per configuration, with the runtime's default number
# of hardware queues and with GPU_MAX_HW_QUEUES=8, same box, alternating (-> gpurun_out/lanes/lanes_ab.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/lanes; mkdir -p $OUT; : > $OUT/lanes_ab.txt
for rep in 1 2; do
for cfg in "--config cfg3" "--config cfg1"; do
  for q in default 8; do
  for lanes in 2 3 4; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    echo "== [$cfg] lanes=$lanes hw_queues=$q rep=$rep" >> $OUT/lanes_ab.txt
    timeout 300 python bench.py $cfg --no-cpu-baseline --no-plugin-path --no-secondary --streams $lanes --regions 5 --sustain 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','host_us_per_call')}))" >> $OUT/lanes_ab.txt
  done
  done
done
done
unset GPU_MAX_HW_QUEUES
cat $OUT/lanes_ab.txt
