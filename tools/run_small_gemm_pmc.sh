OUT=${1:-small_gemm_pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp
P=$R/tools/small_gemm_once.py
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/trace -- python3 $P > /dev/null 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$OUT/pmc_a -- python3 $P > /dev/null 2>&1; echo "pmc a rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/$OUT/pmc_d -- python3 $P > /dev/null 2>&1; echo "pmc d rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_b -- python3 $P > /dev/null 2>&1; echo "pmc b rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/$OUT/pmc_e -- python3 $P > /dev/null 2>&1; echo "pmc e rc=$?"
cd $R
python tools/pmc_summary.py gpurun_out/$OUT/trace gpurun_out/$OUT/pmc_a gpurun_out/$OUT/pmc_d gpurun_out/$OUT/pmc_b gpurun_out/$OUT/pmc_e --match gemm64 > gpurun_out/$OUT/summary.txt 2>&1
find gpurun_out/$OUT -name "*.db" -delete
cat gpurun_out/$OUT/summary.txt
