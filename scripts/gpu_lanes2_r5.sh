#!/bin/bash
# 2 against 3 lanes of the overlap mode at the sizes the first pass (gpu_lanes_r5.sh) did not cover, product path incl. the spectral
# bank's line: small steps -> gpurun_out/lanes/lanes_ab2.txt, chip-filling steps -> lanes_ab3.txt (same box, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/lanes; mkdir -p $OUT; : > $OUT/lanes_ab2.txt; : > $OUT/lanes_ab3.txt
run() {  # $1 = output file, $2 = extra flags, $3 = bench arguments
  for lanes in 2 3; do
    echo "== [$3] lanes=$lanes rep=$rep" >> $1
    timeout 300 python bench.py $3 --no-cpu-baseline --no-plugin-path --streams $lanes $2 --sustain 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'host_us_per_call':d['host_us_per_call'],'spectral':d['spectral_bank']['value']}))" >> $1
  done
}
for rep in 1 2; do
  for cfg in "--sr 44100 --envs 10 --steps 100 --warmup 10" "--sr 44100 --envs 5 --steps 100 --warmup 10" "--envs 64" "--envs 8"; do run $OUT/lanes_ab2.txt "--regions 5" "$cfg"; done
  for cfg in "" "--steps 20 --warmup 5" "--sr 44100 --envs 128 --steps 60 --warmup 5" "--config cfg4 --steps 100"; do run $OUT/lanes_ab3.txt "" "$cfg"; done
done
cat $OUT/lanes_ab2.txt $OUT/lanes_ab3.txt
