cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c32
{
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_configs_gpu.py tests/test_convergence_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -25
timeout 900 python tools/loss_gap_probe.py --steps 200 --reps 3 --variants fp32,bf16 --out gpurun_out/c32/gap_after.json 2>&1 | grep -v "Warning\|warn"
} > gpurun_out/c32/log.txt 2>&1
tail -40 gpurun_out/c32/log.txt
