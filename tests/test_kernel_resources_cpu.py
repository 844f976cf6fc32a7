"""Every HIP kernel of libbqhip.so is held to the LDS and scratch size recorded in tests/golden/kernel_resources.json (from
hipcc's -Rpass-analysis=kernel-resource-usage remarks, written by bridgeqa_amd/build.py at every compile).

Why: in round 4 an edit to the GEMM descriptor made hipcc promote a 16-byte per-lane address table of gemm64_kernel to LDS --
4 KB more LDS per workgroup, an LDS round trip in front of every fragment read, every small-tile launch of the step 10-20 us
slower (0.8 ms per c3 step) -- without a warning, a failing test or a changed result.  Scratch (spills) and LDS growth are
the two silent performance cliffs of this toolchain; both are compile-time facts, so they are checked without a GPU.
Regenerate the fixture deliberately (python -m bridgeqa_amd.build --force, then this file's __main__) when a kernel's
footprint is MEANT to change."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")


def test_no_kernel_grew_its_lds_or_started_to_spill():
    from bridgeqa_amd import build
    build.build()
    res = build.kernel_resources()
    if res is None:
        pytest.skip("objects were not compiled by this checkout's build.py (prebuilt library)")
    gold = json.load(open(GOLD))
    assert len(res) >= 150
    grown = [(k, gold[k], {"lds": v.get("lds", 0), "scratch": v.get("scratch", 0)}) for k, v in res.items()
             if k in gold and (v.get("lds", 0) > gold[k]["lds"] or v.get("scratch", 0) > gold[k]["scratch"])]
    assert not grown, "kernels whose LDS / scratch grew against tests/golden/kernel_resources.json: %s" % grown[:5]
    missing = [k for k in gold if k not in res]
    assert len(missing) <= len(gold) // 10, "fixture out of date: %d recorded kernels no longer exist" % len(missing)


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ROOT)
    from bridgeqa_amd import build
    r = build.kernel_resources()
    json.dump({k: {"lds": v.get("lds", 0), "scratch": v.get("scratch", 0)} for k, v in sorted(r.items())}, open(GOLD, "w"),
              indent=0, sort_keys=True)
    print("wrote", GOLD, len(r))
