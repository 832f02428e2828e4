#!/bin/bash
# 2 against 3 lanes of the overlap mode at sizes the first pass (gpu_lanes_r5.sh) did not cover (-> gpurun_out/lanes/lanes_ab3.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/lanes; mkdir -p $OUT; : > $OUT/lanes_ab3.txt
for rep in 1 2; do
for cfg in "" "--steps 20 --warmup 5" "--sr 44100 --envs 128 --steps 60 --warmup 5" "--config cfg4 --steps 100"; do
  for lanes in 2 3; do
    echo "== [$cfg] lanes=$lanes rep=$rep" >> $OUT/lanes_ab3.txt
    timeout 300 python bench.py $cfg --no-cpu-baseline --no-plugin-path --streams $lanes --sustain 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'host_us_per_call':d['host_us_per_call'],'spectral':d['spectral_bank']['value']}))" >> $OUT/lanes_ab3.txt
  done
done
done
cat $OUT/lanes_ab3.txt
