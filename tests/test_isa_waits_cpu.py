"""The kernels that stage operands by LDS-DMA keep their look-ahead: hipcc must not fence the tile loop's LDS reads with its own
s_waitcnt vmcnt (tools/isa_waits.py).  Round 6 found vmcnt(0) in front of every phase's fragment reads of gemm256_kernel -- the
kernel's counted waits were decoration -- and the same fence in the contraction-major forms of gemm128 / gemm64, in wgrad_rows
and on every LDS access of sa_bwd; the reads (or the DMAs) are inline asm since.  Compiles three sources device-only to
assembly (~15 s each)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HAVE_HIPCC = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not HAVE_HIPCC, reason="no hipcc")
@pytest.mark.parametrize("src,families,at_least", [("gemm.hip", ("gemm256_kernel", "gemm64_kernel", "wgrad_rows_kernel", "pwconv64"), 40),
                                                   ("gemm_mid.hip", ("gemm128_kernel",), 10),
                                                   ("detbwd.hip", ("sa_bwd_kernel",), 20)])
def test_tile_loops_have_no_compiler_fence_in_front_of_their_lds_reads(src, families, at_least):
    import isa_waits
    ks = isa_waits.scan(os.path.join(ROOT, "bridgeqa_amd", "csrc", src))
    seen = 0
    for name, lines in ks.items():
        if not any(f in name for f in families) or not isa_waits.lds_dmas(lines):
            continue
        mf = [i for i, x in enumerate(lines) if "v_mfma" in x]
        if not mf:
            continue
        seen += 1
        # the tile loop = everything up to the last v_mfma (epilogues read their output images back with compiler-visible loads)
        bad = [h for h in isa_waits.compiler_waits(lines[:mf[-1]]) if h[1].startswith("ds_read")]
        assert not bad, (name, bad)
    assert seen >= at_least, seen
