#!/bin/bash
# Randomised parity sweeps of the product path against the oracle on the GPU box (scripts/gpu_fuzz.py: SoundSpacesSim steps,
# SoundSpaces 2.0 steps, multi-step engine runs with eviction; scripts/gpu_fuzz_features.py: the waveform-side entry points;
# scripts/gpu_fuzz_plugin.py: simulators that walk, served eager / deferred / batched through the plugin boundary).
# usage: gpurun -- bash scripts/gpu_fuzz.sh [first seed]   -> gpurun_out/fuzz/*.txt (summaries: profiles/r6/fuzz/)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
s0=${1:-10}
out=gpurun_out/fuzz; mkdir -p $out
for k in 0 1 2; do s=$((s0 + k))
  timeout 900 python scripts/gpu_fuzz.py --mode sim --trials 400 --seed $s --out $out/sim_seed$s.txt 2>&1 | grep -v " ok " | tail -6
done
for k in 0 1; do s=$((s0 + k))
  timeout 900 python scripts/gpu_fuzz.py --mode continuous --trials 300 --seed $s --out $out/continuous_seed$s.txt 2>&1 | grep -v " ok " | tail -6
  timeout 900 python scripts/gpu_fuzz.py --mode engine --trials 300 --seed $s --out $out/engine_seed$s.txt 2>&1 | grep -v " ok " | tail -6
  timeout 900 python scripts/gpu_fuzz_features.py --trials 200 --seed $s --out $out/features_seed$s.txt 2>&1 | grep -v " ok " | tail -6
  timeout 900 python scripts/gpu_fuzz_plugin.py --trials 60 --seed $s --out $out/plugin_seed$s.txt 2>&1 | grep -v " ok \|WARNING" | tail -6
  timeout 900 python scripts/gpu_fuzz_plugin.py --mode continuous --trials 40 --seed $s --out $out/plugin_continuous_seed$s.txt 2>&1 | grep -v " ok \|WARNING" | tail -6
done
grep -h "^#" $out/*_seed*.txt > $out/SUMMARY.txt
cat $out/SUMMARY.txt
