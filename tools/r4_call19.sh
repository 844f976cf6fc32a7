cd $GRAFT_REPO_ROOT
timeout 600 python tools/dump_flush_plan.py 2>&1 | tail -40
