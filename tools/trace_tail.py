"""Extract ONE step (the k-th from the end) of a rocprofv3 kernel-trace CSV; steps are segmented by the END of the
optimizer launch (kernel-name pattern argv[4], default adamw_kernel): a step = everything after the previous mark up to
and including this one.   python tools/trace_tail.py '<glob of *_kernel_trace.csv>' out.csv k [pattern]"""
import csv, sys, glob
src = glob.glob(sys.argv[1])[0]; dst = sys.argv[2]; k = int(sys.argv[3])
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[4] if len(sys.argv) > 4 else "adamw_kernel"
marks = [int(r["End_Timestamp"]) for r in rows if pat in r["Kernel_Name"]]
t0, t1 = marks[-k - 1], marks[-k]
keep = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
with open(dst, "w") as f:
    w = csv.writer(f)
    w.writerow(["name", "start_us", "dur_us", "queue", "stream"])
    for r in keep:
        w.writerow([r["Kernel_Name"][:100], (int(r["Start_Timestamp"]) - t0) / 1e3,
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", ""), r.get("Stream_Id", "")])
print("steps seen", len(marks), "kept", len(keep), "span ms", (t1 - t0) / 1e6)
