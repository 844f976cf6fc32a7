# SA1's ball query under rocprofv3: trace pass + FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section),
# summarised into profiles-style JSON for bench.py's roofline_ballquery.traffic.  usage: bash tools/run_ballquery_pmc.sh <outdir-under-gpurun_out>
OUT=${1:-bq_pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/t -- python3 $R/tools/time_ball_query.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/b -- python3 $R/tools/time_ball_query.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/c -- python3 $R/tools/time_ball_query.py > /dev/null 2>&1
cd $R
rm -f $D/bq.jsonl
python tools/pmc_summary.py $D/t $D/b $D/c --match ball_query_grid --json $D/bq.jsonl --label ball_query_sa1 > $D/ballquery_pmc.txt 2>&1
rm -f $D/gb.jsonl
python tools/pmc_summary.py $D/t $D/b $D/c --match grid_build --json $D/gb.jsonl --label grid_build_sa1 >> $D/ballquery_pmc.txt 2>&1
python - <<PY
import json
rs = [json.loads(l) for l in open("$D/bq.jsonl")]   # (the trace pass and the counter passes report the grid in different units: two records)
pick = lambda k: next((r[k] for r in rs if r.get(k) is not None), None)
gb = [json.loads(l) for l in open("$D/gb.jsonl")]
gpick = lambda k: next((r[k] for r in gb if r.get(k) is not None), None)
out = {"B": 16, "N": 40000, "M": 2048, "avg_us": pick("avg_us"), "hbm_read_bytes": pick("fetch_bytes"), "hbm_write_bytes": pick("write_bytes"),
       "grid_build": {"avg_us": gpick("avg_us"), "hbm_read_bytes": gpick("fetch_bytes"), "hbm_write_bytes": gpick("write_bytes"),
                      "note": "the launch that bins the scene's points in front of the query kernel (one workgroup per scene); avg_us above is the QUERY kernel alone, bench.py's 'alone' time covers both"},
       "file": "profiles/r06_ballquery_pmc.json",
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/time_ball_query.py; read = 2 x FETCH_SIZE x 1024 "
                 "(gfx950), write = WRITE_SIZE x 1024; fabric side, per launch"}
json.dump(out, open("$D/ballquery_pmc.json", "w"))
print(out)
PY
rm -rf $D/t $D/b $D/c
