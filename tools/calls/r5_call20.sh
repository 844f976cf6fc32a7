cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/dbg_graphed_tol.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tail -12 | cut -c1-700
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
