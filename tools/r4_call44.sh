cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c44
{
for c in 8192 4096 16384 32768; do
  bash tools/rebuild_with.sh adamw -DBQ_ADAMW_CHUNK=$c
  echo "chunk $c: $(timeout 200 python tools/bench_adamw.py 2>&1 | grep -v Warn | tail -2 | tr '\n' ' ')"
done
bash tools/rebuild_with.sh adamw
} > gpurun_out/c44/log.txt 2>&1
cat gpurun_out/c44/log.txt
