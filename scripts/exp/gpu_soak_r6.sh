cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for i in 1 2; do timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -1; done
timeout 900 python scripts/leak_check.py 2>&1 | grep -v amdgpu | tail -6
timeout 900 python scripts/gpu_soak.py 1 15 2>&1 | tail -2
timeout 900 python scripts/gpu_soak_ctx.py 1 8 2>&1 | tail -2
timeout 600 python scripts/soak_ctx.py 400 2>&1 | tail -2
