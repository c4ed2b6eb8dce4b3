"""The C-ABI library loads without a GPU and exports every symbol include/pymodem_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pymodem_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import pymodem_amd
    from pymodem_amd import _native
    path = pymodem_amd.library_path()
    assert os.path.exists(path), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert set(_native.EXPORTS) == set(names), set(_native.EXPORTS) ^ set(names)


def test_loading_does_not_need_or_touch_a_gpu():
    import pymodem_amd
    lib = pymodem_amd.lib()
    assert lib.pm_version() == 100
    assert lib.pm_device_count() >= 0
    if lib.pm_device_count() == 0:
        with pytest.raises(pymodem_amd.NativeError):          # no CPU fallback: the product path fails loudly
            pymodem_amd.Context(0)
        from pymodem_amd import chain_builder as cb
        import numpy as np
        m = cb.ModemConfigurator(48000, {"type": "fsk", "config": "9600", "options": {}})
        with pytest.raises(pymodem_amd.NativeError):
            m.demod(np.zeros(1000, dtype=np.int16))


def test_header_is_plain_c_and_ctypes_mirrors_match(tmp_path):
    """include/pymodem_amd.h compiles as C11 (-pedantic) and links against the library from a C program; the struct sizes and field
    offsets the C compiler sees are the ones the ctypes mirrors in pymodem_amd/_native.py use."""
    import ctypes
    import shutil
    import subprocess
    from pymodem_amd import _native as N
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    structs = {"pm_packet": N.Packet, "pm_loop": N.Loop, "pm_agc_params": N.AGCParams, "pm_slicer_params": N.SlicerParams,
               "pm_slicer_state": N.SlicerState, "pm_slice_job": N.SliceJob, "pm_chain_desc": N.ChainDesc, "pm_host_job": N.HostJob, "pm_afsk_tones": N.AfskTones,
               "pm_afsk_sweep_desc": N.AfskSweepDesc, "pm_lbatch_desc": N.LBatchDesc,
               "pm_pipe_chain": N.PipeChain, "pm_pipe_desc": N.PipeDesc, "pm_pipe_result": N.PipeResult}
    fields = [("pm_packet", "data"), ("pm_packet", "correlated_count"), ("pm_loop", "bb0"), ("pm_loop", "sy0"), ("pm_slice_job", "h_state"),
              ("pm_slice_job", "count"), ("pm_slicer_state", "streamaddress"), ("pm_chain_desc", "slicer"), ("pm_chain_desc", "loop"),
              ("pm_chain_desc", "pd_table"), ("pm_chain_desc", "agc"), ("pm_slicer_params", "demap"), ("pm_afsk_sweep_desc", "h_tones"),
              ("pm_afsk_sweep_desc", "lpf_abs_sum"), ("pm_lbatch_desc", "agc"), ("pm_lbatch_desc", "pd_table"), ("pm_lbatch_desc", "output_fir"),
              ("pm_pipe_chain", "lfsr_poly"), ("pm_pipe_chain", "source_decoder"), ("pm_pipe_desc", "x_bound"), ("pm_pipe_desc", "max_samples"),
              ("pm_pipe_desc", "address_distance"), ("pm_pipe_result", "h_corr_decoders"), ("pm_pipe_result", "ms_to_done")]
    src = ['#include "pymodem_amd.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {",
           '    printf("version %d\\n", pm_version());']
    for name in structs:
        src.append(f'    printf("sizeof {name} %zu\\n", sizeof({name}));')
    for s, f in fields:
        src.append(f'    printf("offsetof {s} {f} %zu\\n", offsetof({s}, {f}));')
    src += ['    printf("head %zu %zu\\n", sizeof(pm_packet_head), offsetof(pm_packet, data));', "    return pm_device_count() < 0;", "}"]
    c = tmp_path / "abi.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "abi"
    libdir = os.path.join(ROOT, "pymodem_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe),
                           "-L", libdir, "-lpymodem_amd", f"-Wl,-rpath,{libdir}"])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    assert out[0].startswith("version 1")
    seen = {}
    for line in out[1:]:
        parts = line.split()
        if parts[0] == "sizeof":
            assert ctypes.sizeof(structs[parts[1]]) == int(parts[2]), line
        elif parts[0] == "offsetof":
            assert getattr(structs[parts[1]], parts[2]).offset == int(parts[3]), line
        seen[parts[0]] = True
    assert seen.get("sizeof") and seen.get("offsetof")
    assert out[-1] == "head 40 40"


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under pymodem_amd/ or tools/ may import or load it (only tests/, __graft_entry__'s
    build()/smoke() and bench.py's cpu_baseline leg do)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for path in glob.glob(os.path.join(root, "pymodem_amd", "**", "*"), recursive=True) + glob.glob(os.path.join(root, "tools", "*")):
        if not os.path.isfile(path) or not path.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
            continue
        text = open(path, errors="replace").read()
        code = re.sub(r"//.*|#.*", "", text)                  # comments may cite the oracle; code may not use it
        if re.search(r"^\s*(from|import)\s+oracle\b|pm_oracle|pmo_|oracle/_ref|liboracle", code, re.M):
            bad.append(os.path.relpath(path, root))
    assert bad == []
