"""Diagnostic: a few recordings through the native pipelined executor with PM_PIPE_TRACE, next to the group executor's counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PM_PIPE_TRACE", "1")
import json

import pymodem_amd
from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "tests", "golden", "configs", "afsk_1200_ax25_super_opt.json")) as f:
    lines = [l for l in (json.loads(s) for s in f if s.strip()) if l.get("object_type") == "demod_chain"]
audio = siggen.recording("afsk1200_ax25", 48000, packets=6, seed=11, noise_sigma=800.0, payload_len=(20, 80))[0]
st = {}
rows = ce.process_chains_device([cb.build_chain(48000, l) for l in lines], audio, stages=st, _rows=True)
print("group executor: slicer bytes / packets per chain", [(len(s), len(r)) for s, r in zip(st["sliced"], rows)])
ctx = pymodem_amd.Context.default()
d = ctx.upload(audio)
ctx.sync()
pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], len(audio), 1200.0, ctx=ctx)
for t in [pipe.submit(d) for _ in range(3)]:
    tb = pipe.table(t)
    print("native:", tb.counts, tb.CountGood(), tb.latency_ms)
pipe.close()
