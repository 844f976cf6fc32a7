"""Builds bridgeqa_amd/lib/libbqhip.so (the C-ABI library of include/bqhip.h) with hipcc for gfx950.

In-tree on purpose: the .so travels with the repo snapshot to the GPU box.  `python -m bridgeqa_amd.build`.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libbqhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the index-exact kernels round every product/sum separately.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _resources(text):
    """{mangled kernel name: {"lds", "scratch", "vgprs", "occupancy"}} from hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"    VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            # (the resource remarks go to <obj>.res: tests/test_kernel_resources_cpu.py holds every kernel to its recorded LDS
            # and scratch size -- round 4 lost 0.8 ms per step to a 16-byte table that hipcc silently promoted to LDS)
            cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stderr=open(obj + ".res", "w"))))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        for s_ in failed:
            sys.stderr.write(open(os.path.join(objdir, os.path.basename(s_) + ".o.res")).read()[-4000:])
        raise RuntimeError("hipcc failed for: %s" % ", ".join(failed))
    for src, _ in procs:   # compiler diagnostics other than the remarks still reach the terminal
        for line in open(os.path.join(objdir, os.path.basename(src) + ".o.res")):
            if "warning:" in line or "error:" in line:
                sys.stderr.write(line)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        # a kernel whose body does not compile in the HOST pass loses its launch stub without a diagnostic (deferred
        # device-code errors) and the library then fails at dlopen on the GPU box: catch it here
        syms = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True).stdout
        lost = [l.split()[-1] for l in syms.splitlines() if "__device_stub__" in l]
        if lost:
            os.remove(LIB)
            raise RuntimeError("kernels without a host launch stub (host-pass compile error in their body): %s" % ", ".join(lost))
    return LIB


def kernel_resources():
    """resource usage of every kernel of the last build (None when an object was not built by this build.py)"""
    objdir = os.path.join(PKG, "build")
    res = {}
    for src in sources():
        path = os.path.join(objdir, os.path.basename(src) + ".o.res")
        if not os.path.exists(path):
            return None
        res.update(_resources(open(path).read()))
    return res


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
