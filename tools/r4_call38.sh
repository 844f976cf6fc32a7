cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c38
timeout 1500 python tools/loss_curve.py --steps 200 --out gpurun_out/c38/r04_loss_curve.json 2>&1 | grep -v "Warning\|warn" > gpurun_out/c38/log.txt
cat gpurun_out/c38/log.txt
