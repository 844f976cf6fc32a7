# PMC passes over tools/gemm_once.py, ONE launch form per process (durations, MFMA / VALU, FETCH_SIZE, WRITE_SIZE, waits +
# LDS), summarised per form.  usage: bash tools/run_gemm_pmc.sh <outdir-under-gpurun_out>
OUT=${1:-gemm_pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
: > $R/gpurun_out/$OUT/gemm_pmc_summary.txt
: > $R/gpurun_out/$OUT/gemm_pmc.jsonl
for NAME in fwd_qkv fwd_proj fwd_fc1_gelu fwd_fc2 dx_qkv dx_proj dx_fc1 dx_fc2_dgelu dw_grouped48 text_fc1_gelu; do
  cd /tmp
  D=$R/gpurun_out/$OUT/$NAME
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/t -- python3 $R/tools/gemm_once.py 4 --only $NAME > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $D/a -- python3 $R/tools/gemm_once.py 2 --only $NAME > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/b -- python3 $R/tools/gemm_once.py 2 --only $NAME > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/c -- python3 $R/tools/gemm_once.py 2 --only $NAME > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $D/d -- python3 $R/tools/gemm_once.py 2 --only $NAME > /dev/null 2>&1
  cd $R
  echo "== $NAME" >> gpurun_out/$OUT/gemm_pmc_summary.txt
  python tools/pmc_summary.py $D/t $D/a $D/b $D/c $D/d --match gemm --json gpurun_out/$OUT/gemm_pmc.jsonl --label $NAME >> gpurun_out/$OUT/gemm_pmc_summary.txt 2>&1
  rm -rf $D
done
cat gpurun_out/$OUT/gemm_pmc_summary.txt | cut -c1-250
