"""examples/chain_demo.c: a plain C host drives one AFSK chain through the C ABI (pm_chain_run + pm_lfsr_unscramble + pm_codec_*);
its packets must be the Python path's."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def build(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    exe = tmp_path / "chain_demo"
    libdir = os.path.join(ROOT, "pymodem_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "chain_demo.c"), "-o", str(exe), "-L", libdir, "-lpymodem_amd", f"-Wl,-rpath,{libdir}"])
    return exe


def test_c_example_builds(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_c_example_decodes_like_the_python_path(tmp_path):
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    exe = build(tmp_path)
    line = {"object_name": "AFSK 1200", "object_type": "demod_chain", "modem": {"type": "afsk", "config": "1200", "options": {}},
            "slicer": {"type": "binary", "config": "1200", "options": {}},
            "stream": {"type": "lfsr", "options": {"poly": "0x3", "invert": "True"}}, "codec": {"type": "ax25"}}
    audio, frames = siggen.recording("afsk1200_ax25", 48000, packets=5, seed=9, noise_sigma=400.0, payload_len=(20, 60))
    chain = cb.build_chain(48000, line)
    m, s = chain[1], chain[2]
    with open(tmp_path / "taps.bin", "wb") as f:
        f.write(struct.pack("<iiii", len(m.input_bpf), len(m.mark_correlator_i), len(m.output_lpf), 0))
        f.write(struct.pack("<dd", s.samples_per_symbol, s.lock_rate))
        for v in (m.input_bpf, m.mark_correlator_i, m.mark_correlator_q, m.space_correlator_i, m.space_correlator_q, m.output_lpf):
            f.write(np.ascontiguousarray(v, dtype="<f8").tobytes())
    audio.astype("<i2").tofile(tmp_path / "audio.s16")
    out = subprocess.check_output([str(exe), str(tmp_path / "taps.bin"), str(tmp_path / "audio.s16")], text=True).splitlines()
    pkts = ce.process_chain(chain, audio)
    for p in pkts:
        p.CalcCRC()
    want = [f"packet {p.streamaddress} {len(p.data)} {int(p.ValidCRC)} {p.CalculatedCRC}" for p in pkts]
    assert out[0].startswith("bytes ") and out[1:] == want
    assert sum(int(p.ValidCRC) for p in pkts) == len(frames) == 5
