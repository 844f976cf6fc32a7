"""A/B of module-level switches inside the whole step: runs bench.main() with attributes set first.

    python tools/ab_bench.py fusion_state.TRANSPOSED_DX[0]=False pipeline._T_REFRESH_WGS[0]=256 -- --steps 30 --warmup 8

Each NAME=VALUE names an attribute of a bridgeqa_amd module (list-valued switches take NAME[0]=VALUE); what follows `--`
goes to bench.py.  Prints bench.py's JSON line.  (These switches are constants of the product path; this tool exists so
that an experiment does not need an environment variable in the library.)"""
import ast
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    argv = sys.argv[1:]
    cut = argv.index("--") if "--" in argv else len(argv)
    for item in argv[:cut]:
        if item.startswith("call:"):   # call:_ext.attn_set_persistent(0) -- a setter of the library instead of an attribute
            name, args = item[5:].split("(", 1)
            mod, fn = name.split(".", 1)
            getattr(importlib.import_module("bridgeqa_amd." + mod), fn)(*ast.literal_eval("(" + args.rstrip(")") + ",)"))
            continue
        name, value = item.split("=", 1)
        mod, attr = name.split(".", 1)
        m = importlib.import_module("bridgeqa_amd." + mod)
        v = ast.literal_eval(value)
        if attr.endswith("[0]"):
            getattr(m, attr[:-3])[0] = v
        else:
            setattr(m, attr, v)
    import bench
    sys.argv = ["bench.py"] + argv[cut + 1:]
    bench.main()


if __name__ == "__main__":
    main()
