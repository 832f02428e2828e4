#!/bin/bash
# Round 5, pass I: the miss path's scatter entry (ss_bank_scatter_rows_f32) - tests, breakdown of a miss step, SS2.0 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r5i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wav_loader.py tests/test_context.py tests/test_deferred_columns.py tests/test_plugin_api.py tests/test_deferred.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
for r in 0.01 0.05 0.25; do timeout 200 python scripts/miss_breakdown.py --rate $r 2>&1 | grep -v amdgpu | tail -15 >> $OUT/miss_breakdown.txt; done
cat $OUT/miss_breakdown.txt
for rep in 1 2; do
  timeout 200 python scripts/bench_deferred_continuous.py 2>/dev/null | tail -1 >> $OUT/deferred_continuous.jsonl
  timeout 200 python scripts/bench_deferred_continuous.py --scatter-copy 2>/dev/null | tail -1 >> $OUT/deferred_continuous.jsonl
done
cat $OUT/deferred_continuous.jsonl
