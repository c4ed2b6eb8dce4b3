#!/usr/bin/env python3
"""Instruction counts per SAMPLE of the carrier-loop kernels (pm_loops.hip: loop_kernel<MODE>), from the compiler's gfx950 assembly:
the innermost loop with float64 work is the per-sample body (one lane per loop, strictly sequential).
    python tools/loop_isa_count.py [out.json]      (hipcc cross-compiles; no GPU needed)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"0": "costas_bpsk (psk.py:173-189)", "1": "pll_afsk (afsk_pll.py:153-165)", "2": "mpsk (psk.py:734-747)", "3": "costas_qpsk (psk.py:436-466)"}


def main():
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "loops.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "pymodem_amd", "csrc", "pm_loops.hip")],
                              stderr=subprocess.DEVNULL)
        txt = open(asm).read().split("\n")
    out = {}
    for si, line in enumerate(txt):
        m = re.match(r"^(_ZN.*loop_kernelILi(\d+)EEE.*):", line)
        if not m:
            continue
        mode = m.group(2)
        end = next(j for j in range(si, len(txt)) if "s_endpgm" in txt[j])
        body = txt[si:end]
        labels = {mm.group(1): k for k, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        best = None
        for k, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                blk = body[labels[mm.group(1)]:k + 1]
                ops = [x.split()[0] for x in blk if re.match(r"^\s+[a-z]", x)]
                # the per-sample body is the smallest loop around the NCO's table read (one ds_read_b128 per sample); the rare
                # `while phase >= 2 pi` loops nested inside it are counted with it (a handful of instructions, never taken)
                if "ds_read_b128" in ops and (best is None or len(ops) < best[0]):
                    best = (len(ops), ops)
        ops = best[1]
        cnt = {}
        for o in ops:
            cnt[o] = cnt.get(o, 0) + 1
        out[MODES[mode]] = {
            "instructions": len(ops), "valu": sum(v for k, v in cnt.items() if k.startswith("v_")),
            "valu_f64": sum(v for k, v in cnt.items() if k.startswith("v_") and "f64" in k),
            "lds": sum(v for k, v in cnt.items() if k.startswith("ds_")), "salu": sum(v for k, v in cnt.items() if k.startswith("s_")),
            "by_opcode": dict(sorted(cnt.items(), key=lambda t: -t[1]))}
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
