cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/miss; mkdir -p $OUT
timeout 600 python -m pytest tests/test_wav_loader.py tests/test_gpu_parity.py -m gpu -q -x -k "pose_misses or deferred or gpu_store" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout 900 python scripts/bench_loader.py --out $OUT/loader.json > $OUT/loader.log 2>&1; echo "loader rc=$?"
grep -h miss_rate $OUT/loader.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['mode'],'|',d['reader'],'|',d['miss_rate'],d['trainer_half_us_per_step_median'],d['poses_loaded_inside_the_call'], d['store'][:12])"
tail -3 $OUT/loader.log | cut -c1-300
