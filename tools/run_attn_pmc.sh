# rocprofv3 passes over tools/attn_once.py (plain ViT-shaped attention, fwd + bwd): kernel trace, then PMC passes
# (separate runs: --pmc is never combined with other tracing).  usage: bash tools/run_attn_pmc.sh <outdir-under-gpurun_out>
OUT=${1:-attn_pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/trace -- python3 $R/tools/attn_once.py > /dev/null 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$OUT/pmc_a -- python3 $R/tools/attn_once.py > /dev/null 2>&1; echo "pmc a rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/$OUT/pmc_d -- python3 $R/tools/attn_once.py > /dev/null 2>&1; echo "pmc d rc=$?"
cd $R
python tools/pmc_summary.py gpurun_out/$OUT/trace gpurun_out/$OUT/pmc_a gpurun_out/$OUT/pmc_d --match attn > gpurun_out/$OUT/attn_pmc_summary.txt 2>&1
find gpurun_out/$OUT -name "*.db" -delete
cat gpurun_out/$OUT/attn_pmc_summary.txt
