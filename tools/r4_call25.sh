cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c25
timeout 900 python tools/who_launches.py x 400 > gpurun_out/c25/who.txt 2> gpurun_out/c25/who.err
tail -3 gpurun_out/c25/who.err; head -5 gpurun_out/c25/who.txt
