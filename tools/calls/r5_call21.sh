cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/dbg_graphed_step2.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | tail -8 | cut -c1-900
