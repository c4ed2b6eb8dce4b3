"""Report text (pymodem_amd/report.py) against text captured from the reference (tests/golden/reports.json), with the packets
rebuilt from the goldens (CPU) and, on the GPU box, the whole command line end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REPORTS = json.load(open(os.path.join(GOLDEN, "reports.json")))


def packets_from_golden(g, prefix, name):
    from pymodem_amd.packet_meta import PacketMeta
    lens, data, addr, corr = g[prefix + "_pkt_len"], g[prefix + "_pkt_data"], g[prefix + "_pkt_addr"], g[prefix + "_pkt_corrected"]
    out, pos = [], 0
    for k in range(len(lens)):
        out.append(PacketMeta.from_bytes(bytes(data[pos:pos + lens[k]]), addr[k], name, corr[k]))
        pos += int(lens[k])
    return out


@pytest.mark.parametrize("cfg,npz,prefix", [("afsk_300.json", "wav_chains", "afsk_300"), ("afsk_300_ax25.json", "wav_chains", "afsk_300_ax25"),
                                            ("afsk_1200.json", "signal_chains", "afsk1200_ax25__afsk_1200__48000"),
                                            ("qpsk_2400.json", "signal_chains", "qpsk2400_il2p__qpsk_2400__48000")])
def test_report_text_matches_the_reference(golden, config_lines, cfg, npz, prefix):
    from pymodem_amd.packet_meta import PacketMetaArray, ReportStyle
    from pymodem_amd.report import raw_bad_text, report_text
    g = golden(npz)
    arr = PacketMetaArray()
    for ci, line in enumerate(config_lines(cfg)):
        arr.add(packets_from_golden(g, f"{prefix}__c{ci}", line["object_name"]))
    arr.CalcCRCs()
    arr.Correlate(address_distance=REPORTS[cfg]["rate"] / 40)
    assert raw_bad_text(arr) == REPORTS[cfg]["raw_bad"]
    assert report_text(arr, ReportStyle({"style": "decoded_headers"})) == REPORTS[cfg]["report"]


def test_cli_exit_codes(tmp_path):
    run = lambda *a: subprocess.run([sys.executable, "-m", "pymodem_amd", *a], cwd=ROOT, capture_output=True, text=True)
    assert run().returncode == 2
    assert run("/nonexistent.json", "x.wav").returncode == 3
    bad = tmp_path / "bad.json"
    bad.write_text("{not json\n")
    assert run(str(bad), "x.wav").returncode == 3
    ok = os.path.join(GOLDEN, "configs", "afsk_300.json")
    assert run(ok, "/nonexistent.wav").returncode == 4
    notwav = tmp_path / "x.wav"
    notwav.write_bytes(b"hello")
    assert run(ok, str(notwav)).returncode == 4


@pytest.mark.gpu
def test_cli_end_to_end_on_bundled_recording():
    """python -m pymodem_amd configs/afsk_300.json afsk_300_il2pc_noise.wav prints the reference's report (49 / 6)."""
    r = subprocess.run([sys.executable, "-m", "pymodem_amd", os.path.join(GOLDEN, "configs", "afsk_300.json"),
                        os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert REPORTS["afsk_300.json"]["raw_bad"] in r.stdout
    assert REPORTS["afsk_300.json"]["report"] in r.stdout
    assert "Unique, valid packets:  49" in r.stdout
