#!/bin/bash
# pass E: tests; the loader bench after the 4-syscall reader + the agent-like walk with / without the azimuth prefetch; SS2.0
# deferred three times (box noise); 44.1 kHz small steps with the capped split (product library).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5e"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 900 python scripts/bench_loader.py --out "$OUT/loader.json" > "$OUT/loader.log" 2>&1; echo "loader rc=$?"; grep -v amdgpu "$OUT/loader.log" | cut -c1-330
for i in 1 2 3; do timeout 300 python scripts/bench_deferred_continuous.py 2>/dev/null | cut -c1-260; done
timeout 300 python scripts/bench_deferred_continuous.py --profile 2>/dev/null | head -24
timeout 300 python scripts/kbench.py --sr 44100 --raw --only fused --sizes 1,5,10,16,32 --reps 200 --bank-mib 512 2>/dev/null
timeout 300 python scripts/kbench.py --sr 44100 --raw --only fused --spectral --sizes 1,5,10,16,32 --reps 200 --bank-mib 512 2>/dev/null
