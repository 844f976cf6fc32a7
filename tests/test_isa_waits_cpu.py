"""The weight-gradient / forward forms of gemm256_kernel keep their look-ahead: hipcc must not fence the K loop's fragment reads
with its own s_waitcnt vmcnt (tools/isa_waits.py; round 6 found vmcnt(0) in front of every phase's reads -- the kernel's counted
waits were decoration -- and moved the reads to inline asm).  Compiles csrc/gemm.hip device-only to assembly (~15 s)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_gemm256_k_loop_has_no_compiler_fence_in_front_of_its_fragment_reads():
    import isa_waits
    ks = isa_waits.scan(os.path.join(ROOT, "bridgeqa_amd", "csrc", "gemm.hip"))
    seen = 0
    for name, lines in ks.items():
        if "gemm256_kernel" not in name:
            continue
        seen += 1
        # the K loop = everything up to the last v_mfma; the epilogue's output image is read back with compiler-visible loads
        last_mfma = max(i for i, x in enumerate(lines) if "v_mfma" in x)
        bad = [h for h in isa_waits.compiler_waits(lines[:last_mfma]) if h[1].startswith("ds_read")]
        assert not bad, (name, bad)
        assert isa_waits.lds_dmas(lines) >= 16, name
    assert seen >= 8
