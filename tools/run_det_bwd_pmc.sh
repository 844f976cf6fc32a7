# PMC passes over tools/det_bwd_once.py (durations, FETCH_SIZE, WRITE_SIZE) for the fused SharedMLP backward and its reductions.
# usage: bash tools/run_det_bwd_pmc.sh <outdir-under-gpurun_out>
OUT=${1:-det_bwd_pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/pmc_trace -- python3 $R/tools/det_bwd_once.py > /dev/null 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_b -- python3 $R/tools/det_bwd_once.py > /dev/null 2>&1; echo "pmc b rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_c -- python3 $R/tools/det_bwd_once.py > /dev/null 2>&1; echo "pmc c rc=$?"
cd $R
python tools/pmc_summary.py gpurun_out/$OUT/pmc_trace gpurun_out/$OUT/pmc_b gpurun_out/$OUT/pmc_c > gpurun_out/$OUT/all.txt 2>&1
grep -A3 "sa_bwd_kernel\|bn_bwd_reduce" gpurun_out/$OUT/all.txt > gpurun_out/$OUT/det_bwd_pmc_summary.txt
find gpurun_out/$OUT -name "*.csv" -size +30M -delete
find gpurun_out/$OUT -name "*.db" -delete
rm -rf gpurun_out/$OUT/pmc_trace gpurun_out/$OUT/pmc_b gpurun_out/$OUT/pmc_c
cut -c1-220 gpurun_out/$OUT/det_bwd_pmc_summary.txt
