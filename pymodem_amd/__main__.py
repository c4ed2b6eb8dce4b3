"""python -m pymodem_amd <config json> <sound file> -- the reference's command line (pymodem.py:25-183) on the GPU path.
Same JSON-lines configs, same exit codes (1 wrong Python, 2 wrong argument count, 3 config unreadable, 4 audio unreadable),
same report text.  Chains run as one group on one GPU (pymodem_amd.chain_execute.process_chains_device) instead of one
forked process each, and are de-duplicated in config order (the reference's order depends on which process ends first)."""
import json
import struct
import sys
import time

import numpy as np


def read_wav(path):
    """Minimal RIFF reader: PCM16 mono (what the reference's recordings are) -> (rate, int16 ndarray)."""
    with open(path, "rb") as f:
        b = f.read()
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, rate = 12, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt, channels, rate, _, _, bits = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
            if fmt != 1 or bits != 16 or channels != 1:
                raise ValueError("only PCM16 mono is supported")
        elif cid == b"data":
            if rate is None:
                raise ValueError("data chunk before fmt chunk")
            return rate, np.frombuffer(b[pos + 8:pos + 8 + size], dtype="<i2").copy()
        pos += 8 + size + (size & 1)
    raise ValueError("no data chunk")


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if sys.version_info < (3, 0):
        print("Python version should be 3.x, exiting")
        return 1
    if len(argv) != 3:
        print("Not enough arguments. Usage: python3 -m pymodem_amd <config json> <sound file>")
        return 2
    try:
        with open(argv[1], "r") as f:
            plan = [json.loads(line) for line in f]
    except Exception:
        print("Unable to open config json file.")
        return 3
    try:
        rate, audio = read_wav(argv[2])
    except Exception:
        print("Unable to open audio file.")
        return 4

    from . import chain_builder as cb, chain_execute as ce
    from .packet_meta import PacketMetaArray, ReportStyle
    from .report import raw_bad_text, report_text

    print("Building processing stacks from config json")
    chains, reports = [], []
    for n, line in enumerate(plan, 1):
        kind = line.get("object_type")
        print(f"Found object_type: {kind}")
        if kind == "demod_chain":
            if "object_name" not in line:
                print(f"Missing 'object_name' in {argv[1]} line {n}, skipping this chain.")
                continue
            print(f"Line {n}: {line['object_name']}")
            chains.append(cb.build_chain(rate, line))         # a bad modem section is fatal here too (pymodem.py:79)
        elif kind == "report":
            if "object_name" not in line:
                print(f"Missing 'object_name' in {argv[1]} line {n}, skipping this reporter.")
                continue
            print(f"Line {n}: {line['object_name']}")
            reports.append((line["object_name"], ReportStyle(line.get("options", {}))))

    print("Executing demod stack plan.")
    t0 = time.time()
    results = PacketMetaArray()
    for packets in ce.process_chains_device(chains, audio):
        results.add(packets)
    print("Correlating results.")
    results.CalcCRCs()
    results.Correlate(address_distance=rate / 40)
    for name, style in reports:
        print(f"Generating {name}")
        print(raw_bad_text(results))
        print(report_text(results, style))
    print(f"Elapsed time: {round(time.time() - t0, 2)} seconds.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
