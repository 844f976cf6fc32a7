"""Where does hipcc put its OWN s_waitcnt vmcnt in the kernels that stage operands by LDS-DMA?

The waitcnt pass knows that `buffer_load ... lds` writes LDS and, without alias information, fences an LDS read that follows one
with s_waitcnt vmcnt(<=N) -- N = the vector-memory operations issued AFTER the last DMA, usually 0.  A software pipeline whose
look-ahead lives in counted waits (inline asm: invisible to the pass) silently loses it: in gemm256_kernel every phase's fragment
reads waited for ALL the DMAs in flight (rounds 1-5; found in round 6 by reading the ISA, not in a profile).

    python tools/isa_waits.py [file.hip ...] [--match SUBSTRING]

compiles the sources device-only to assembly and lists, per kernel with LDS-DMAs, the compiler's vmcnt waits (those outside
#ASMSTART / #ASMEND) and the instruction each one guards.  `ds_read*` right behind `vmcnt(0..1)` inside a K loop is the pattern.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bridgeqa_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-S"]


def kernels(asm_text):
    """{mangled name: [lines]} of every kernel of a device assembly file"""
    out, name, buf = {}, None, []
    for line in asm_text.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, buf = m.group(1), []
            continue
        if name:
            buf.append(line)
            if "s_endpgm" in line:
                out[name] = buf
                name = None
    return out


def compiler_waits(lines):
    """[(vmcnt value, mnemonic of the next instruction)] for the s_waitcnt vmcnt the COMPILER inserted (not inline asm)"""
    hits, inasm = [], False
    for i, x in enumerate(lines):
        if "#ASMSTART" in x:
            inasm = True
        if "#ASMEND" in x:
            inasm = False
        m = re.search(r"s_waitcnt vmcnt\((\d+)\)", x)
        if m and not inasm:
            nxt = [y.strip() for y in lines[i + 1:i + 4] if y.strip() and not y.strip().startswith(";")]
            hits.append((int(m.group(1)), nxt[0].split()[0] if nxt else ""))
    return hits


def lds_dmas(lines):
    return sum(1 for x in lines if "buffer_load" in x and x.rstrip().endswith("lds"))


def scan(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC] + FLAGS + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
        return kernels(open(out).read())


def main():
    args = sys.argv[1:]
    match = ""
    if "--match" in args:
        match = args[args.index("--match") + 1]
        args = [a for a in args if a not in ("--match", match)]
    srcs = args or [os.path.join(CSRC, f) for f in ("gemm.hip", "gemm_mid.hip", "attn.hip", "attn_persist.hip", "detbwd.hip")]
    for src in srcs:
        print("==", os.path.relpath(src, ROOT))
        for name, lines in scan(src).items():
            if match not in name or not lds_dmas(lines):
                continue
            w = [h for h in compiler_waits(lines) if h[1].startswith("ds_read")]
            print("%-100s LDS-DMAs %3d   compiler vmcnt waits in front of LDS reads: %s" % (name[:100], lds_dmas(lines), w or "none"))


if __name__ == "__main__":
    main()
