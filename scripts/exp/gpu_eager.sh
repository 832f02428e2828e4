cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 600 python scripts/prof_eager.py 2>&1 | grep "^eager" | tail -5
