"""BASELINE.md section 2's planned CPU baseline: the c3 hot path (fwd + bwd + clip + AdamW) over the CPU oracle at a given
thread count and batch; one JSON line.  One process per configuration (the OpenMP runtime reads its thread count once).

    python tools/cpu_baseline_full.py THREADS SCENES        (THREADS = 0: os.cpu_count())
tools/run_r6_profiles.sh runs (all cores, 16 scenes), (32, 16) and (32, 2) -> profiles/r06_cpu_baseline_full.json
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
threads, scenes = int(sys.argv[1]) or os.cpu_count(), int(sys.argv[2])
os.environ["OMP_NUM_THREADS"] = str(threads)
import bench

sys.argv = ["bench.py"]
args = bench.parse()
for k, v in (("batch", 16), ("points", 40000), ("image", 512)):
    if getattr(args, k, None) is None:
        setattr(args, k, v)
args.cpu_threads, args.cpu_scenes = threads, scenes
r = bench.cpu_baseline(args, "c3")
r.pop("deviation_from_BASELINE_md_2", None)
r["scenes"] = scenes
print(json.dumps(r), flush=True)
