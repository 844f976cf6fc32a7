mkdir -p gpurun_out/r2j
timeout 2000 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/r2j/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r2j/gpu_tests.log | tail -12
timeout 600 python bench.py > gpurun_out/r2j/bench_default.json 2> gpurun_out/r2j/bench_default.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/r2j/bench_default.json
BQ_FUSION_NO_FORK=1 BQ_TWO_SEGMENT_KV=1 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r2j/bench_nf1_ts1.json 2> gpurun_out/r2j/bench_nf1_ts1.err; echo "nf1 ts1 rc=$?"; cut -c1-200 gpurun_out/r2j/bench_nf1_ts1.json
BQ_FUSION_NO_FORK=1 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r2j/bench_nf1_ts0.json 2> gpurun_out/r2j/bench_nf1_ts0.err; echo "nf1 ts0 rc=$?"; cut -c1-200 gpurun_out/r2j/bench_nf1_ts0.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2j/pmc_trace -- python3 $R/tools/gemm_once.py 3 > /dev/null 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/r2j/pmc_a -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc a rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2j/pmc_b -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc b rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2j/pmc_c -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc c rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/r2j/pmc_d -- python3 $R/tools/gemm_once.py 2 > /dev/null 2>&1; echo "pmc d rc=$?"
cd $R
python tools/pmc_summary.py gpurun_out/r2j/pmc_trace gpurun_out/r2j/pmc_a gpurun_out/r2j/pmc_b gpurun_out/r2j/pmc_c gpurun_out/r2j/pmc_d --match gemm > gpurun_out/r2j/gemm_pmc_summary.txt 2>&1; head -60 gpurun_out/r2j/gemm_pmc_summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2j/prof -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r2j/prof_bench.json 2> $R/gpurun_out/r2j/prof_bench.err; echo "prof rc=$?"
cd $R
for f in $(find gpurun_out/r2j/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/r2j/kernel_stats.csv; done
find gpurun_out/r2j -name "*.csv" -size +30M -delete
find gpurun_out/r2j -name "*.db" -delete
du -sh gpurun_out/r2j
