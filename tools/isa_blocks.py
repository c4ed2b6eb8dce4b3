#!/usr/bin/env python3
"""Basic-block statistics of ONE kernel of a HIP source file, from the gfx950 assembly the build's own flags produce: per block the
vector / binary64 / compare / scalar / LDS / matrix instruction counts and the blocks it branches back to (its loops).  Static counts
times the trip counts the launch geometry fixes give the instruction mix by phase; SQ_INSTS_VALU of a profiled launch is the check
(profiles/r05_fused_kernel_instruction_mix.txt).

usage: isa_blocks.py <file.hip> <substring of the mangled kernel name> [min instructions per block to print, default 15]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def assembly(src):
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
    flags = [f for f in G.FLAGS if f != "-fPIC"]
    subprocess.check_call([G.HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), src, "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernel_lines(lines, pattern):
    names = [m.group(1) for m in (re.match(r"^(_Z\w+):", l) for l in lines) if m and pattern in m.group(1)]
    if not names:
        sys.exit(f"no kernel whose mangled name contains {pattern!r}")
    if len(names) > 1:
        print("matches:", *names, sep="\n  ", file=sys.stderr)
    name = names[0]
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return name, lines[start:end]


def main():
    src, pattern = sys.argv[1], sys.argv[2]
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    name, body = kernel_lines(assembly(src), pattern)
    blocks, cur = [], None
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = {"name": m.group(1) if m else "entry", "line": i, "valu": 0, "f64": 0, "cmp": 0, "salu": 0, "lds": 0, "mfma": 0, "vmem": 0, "to": []}
            blocks.append(cur)
            if m:
                continue
        t = l.strip()
        op = t.split()[0] if t else ""
        if op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
            cur["cmp"] += op.startswith("v_cmp")
            cur["f64"] += "_f64" in op
        elif op.startswith("s_"):
            cur["salu"] += 1
            b = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if b:
                cur["to"].append(b.group(1))
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
    index = {b["name"]: k for k, b in enumerate(blocks)}
    total = {k: sum(b[k] for b in blocks) for k in ("valu", "f64", "cmp", "salu", "lds", "mfma", "vmem")}
    print(name)
    print("whole kernel (static):", total)
    for k, b in enumerate(blocks):
        back = sorted({t for t in b["to"] if index.get(t, 1 << 30) <= k})
        if b["valu"] + b["mfma"] >= least or back:
            print(f"{b['name']:12s} line {b['line']:5d}  valu {b['valu']:4d} (f64 {b['f64']:3d}, cmp {b['cmp']:2d})  salu {b['salu']:3d}  lds {b['lds']:3d}  "
                  f"mfma {b['mfma']:2d}  vmem {b['vmem']:2d}" + (f"  loops back to {', '.join(back)}" if back else ""))


if __name__ == "__main__":
    main()
