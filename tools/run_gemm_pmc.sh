# PMC passes over tools/gemm_once.py (durations, MFMA / VALU, FETCH_SIZE, WRITE_SIZE, waits + LDS), summarised.  usage: bash tools/run_gemm_pmc.sh <outdir-under-gpurun_out>
OUT=${1:-gemm_pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/pmc_trace -- python3 $R/tools/gemm_once.py 3 > /dev/null 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$OUT/pmc_a -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc a rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_b -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc b rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_c -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc c rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/$OUT/pmc_d -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc d rc=$?"
cd $R
python tools/pmc_summary.py gpurun_out/$OUT/pmc_trace gpurun_out/$OUT/pmc_a gpurun_out/$OUT/pmc_b gpurun_out/$OUT/pmc_c gpurun_out/$OUT/pmc_d --match gemm > gpurun_out/$OUT/gemm_pmc_summary.txt 2>&1
find gpurun_out/$OUT -name "*.csv" -size +30M -delete
find gpurun_out/$OUT -name "*.db" -delete
rm -rf gpurun_out/$OUT/pmc_trace gpurun_out/$OUT/pmc_a gpurun_out/$OUT/pmc_b gpurun_out/$OUT/pmc_c gpurun_out/$OUT/pmc_d
grep -A2 "gemm256_kernel<true, false, 3\|gemm256_kernel<false, false, 2" gpurun_out/$OUT/gemm_pmc_summary.txt | cut -c1-200
