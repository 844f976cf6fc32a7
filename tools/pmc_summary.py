"""Summarise rocprofv3 output directories (one per pass) into per-kernel averages.

    python tools/pmc_summary.py <dir> [<dir> ...] [--match gemm] [--json FILE --label NAME]

--json appends one record per (kernel, grid) group to FILE (a JSON-lines file): label, kernel, grid, launches, avg_us,
mfma_util, valu_busy, fetch_bytes (2 x FETCH_SIZE), write_bytes -- what bench.py reads back for `roofline.traffic`.

Reads every *_kernel_trace.csv (durations) and *_counter_collection.csv (PMC values) below the directories, groups
dispatches by (kernel name [first 70 chars], grid size) and prints the mean duration and mean counter values.  Derived
figures (MI355X_MICROARCH.md): MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024);
HBM bytes = 2 * FETCH_SIZE (gfx950 counts half of a wide read) + WRITE_SIZE, both reported in KiB by rocprofv3."""
import collections
import csv
import glob
import os
import sys


def short(name):
    for cut in ("(", ):
        if cut in name:
            name = name[:name.index(cut)]
    return name[-90:]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = jfile = label = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    if "--json" in sys.argv:
        jfile = sys.argv[sys.argv.index("--json") + 1]
        label = sys.argv[sys.argv.index("--label") + 1] if "--label" in sys.argv else ""
        args = [a for a in args if a not in (jfile, label)]
    dur = collections.defaultdict(list)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))
                ctr[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    keys = sorted(set(dur) | set(ctr))
    for key in keys:
        if match and match not in key[0]:
            continue
        c = {k: sum(v) / len(v) for k, v in ctr[key].items()}
        line = "%s  grid %s" % key
        if dur[key]:
            line += "  n=%d  avg %.1f us" % (len(dur[key]), sum(dur[key]) / len(dur[key]))
        print(line)
        if c:
            print("   " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(c.items())))
            der = []
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                der.append("MFMA utilisation %.1f %%" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
            if "GRBM_GUI_ACTIVE" in c and dur[key]:
                # (MI355X_MICROARCH.md, DVFS give-back: the chip lowers its clock under MFMA load; rocprofv3 sums the 8 XCDs;
                # reads high on dispatches under ~0.3 ms)
                der.append("effective clock %.2f GHz" % (c["GRBM_GUI_ACTIVE"] / 8 / (sum(dur[key]) / len(dur[key])) / 1e3))
            if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
                der.append("VALU busy %.1f %%" % (100 * 4 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
            if "SQ_INSTS_VALU" in c and c.get("SQ_INSTS_MFMA"):
                der.append("VALU instr / MFMA %.1f" % (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]))
            if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
                der.append("HBM bytes/launch: read %.1f MB (2 x FETCH_SIZE), write %.1f MB"
                           % (2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6))
            if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
                der.append("LDS conflict cycles %.1f %%" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
            if der:
                print("   => " + "   ".join(der))
        if jfile:
            import json
            rec = dict(label=label, kernel=key[0].strip(), grid=key[1], launches=len(dur[key]),
                       avg_us=round(sum(dur[key]) / len(dur[key]), 2) if dur[key] else None)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                rec["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
                if dur[key]:
                    rec["effective_clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / (sum(dur[key]) / len(dur[key])) / 1e3, 3)
            if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
                rec["valu_busy"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
            if "FETCH_SIZE" in c:
                rec["fetch_bytes"] = round(2 * c["FETCH_SIZE"] * 1024)
            if "WRITE_SIZE" in c:
                rec["write_bytes"] = round(c["WRITE_SIZE"] * 1024)
            with open(jfile, "a") as f:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
