cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c31
timeout 1500 python tools/loss_gap_probe.py --steps 200 --reps 2 --out gpurun_out/c31/gap_probe.json 2>&1 | grep -v "Warning\|warn" > gpurun_out/c31/log.txt
tail -30 gpurun_out/c31/log.txt
