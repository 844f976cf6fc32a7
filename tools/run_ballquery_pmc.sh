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
for k in grid_box grid_cellid grid_chunk grid_build; do
  rm -f $D/$k.jsonl
  python tools/pmc_summary.py $D/t $D/b $D/c --match $k --json $D/$k.jsonl --label ${k}_sa1 >> $D/ballquery_pmc.txt 2>&1
done
python - <<PY
import json
def rec(path):
    rs = [json.loads(l) for l in open(path)]   # (the trace pass and the counter passes report the grid in different units: two records)
    pick = lambda k: next((r[k] for r in rs if r.get(k) is not None), None)
    return {"avg_us": pick("avg_us"), "hbm_read_bytes": pick("fetch_bytes"), "hbm_write_bytes": pick("write_bytes")}
q = rec("$D/bq.jsonl")
parts = {k: rec("$D/%s.jsonl" % k) for k in ("grid_box", "grid_cellid", "grid_chunk")}
tot = lambda f: (sum(p[f] for p in parts.values()) if all(p[f] is not None for p in parts.values()) else None)
out = {"B": 16, "N": 40000, "M": 2048, "avg_us": q["avg_us"], "hbm_read_bytes": q["hbm_read_bytes"], "hbm_write_bytes": q["hbm_write_bytes"],
       "grid_build": {"avg_us": tot("avg_us"), "hbm_read_bytes": tot("hbm_read_bytes"), "hbm_write_bytes": tot("hbm_write_bytes"),
                      "kernels": parts,
                      "note": "the three launches that bin the scene's points in front of the query kernel (round 6: box, cell ids, 16 cell-chunk "
                              "workgroups per scene); avg_us above is the QUERY kernel alone, bench.py's 'alone' time covers all four"},
       "grid_build_one_workgroup": dict(rec("$D/grid_build.jsonl"), note="round 5's build, bq_ball_query_grid_build_mode(0): one workgroup per scene"),
       "file": "profiles/r06_ballquery_pmc.json",
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/time_ball_query.py; read = 2 x FETCH_SIZE x 1024 "
                 "(gfx950), write = WRITE_SIZE x 1024; fabric side, per launch"}
json.dump(out, open("$D/ballquery_pmc.json", "w"))
print(out)
PY
rm -rf $D/t $D/b $D/c
