cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in "nodet" "eagergeo" ""; do
  echo "== flags: $f"
  timeout 600 python tools/dbg_c5.py $f 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | tail -4 | cut -c1-300
done
