cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
timeout 600 python tools/who_launches.py > gpurun_out/c3/who.txt 2> gpurun_out/c3/who.err; echo "who rc=$?"
bash tools/run_step_profile.sh c3/prof > gpurun_out/c3/prof.log 2>&1; tail -3 gpurun_out/c3/prof.log
