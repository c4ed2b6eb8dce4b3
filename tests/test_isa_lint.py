"""Properties of the generated gfx950 code that the slicer's speed depends on and that a harmless-looking source change loses
("The loop's memory operations"): checked on the device assembly, no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def slicer_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "pm_slicer.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-o", str(out), os.path.join(ROOT, "pymodem_amd", "csrc", "pm_slicer.hip")],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_Z\w*slice_walk_kernel\w*):.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        kernels[m.group(1)] = m.group(0)
    assert kernels, "no slice_walk_kernel in the assembly"
    return kernels


def test_walkers_use_global_not_flat_memory_operations(slicer_asm):
    # the bitmap pointers come out of the job table; taken as generic pointers their loads are `flat`, may return out of order and
    # put a full wait (the previous word's stores included) in front of every use
    for name, body in slicer_asm.items():
        assert "flat_load" not in body and "flat_store" not in body, name


def test_walk_loop_does_not_reload_the_job_table(slicer_asm):
    # every hand-scheduled word is two inline-assembly blocks; between the second block of one word and the first of the next
    # there are the word's own loads (sign bits in phase / quadrature, checkpoint) and nothing else -- three re-loads of
    # jobs[j].bi, the word behind it and jobs[j].n used to sit there, each behind its own wait
    checked = 0
    for name, body in slicer_asm.items():
        steps = [m.start() for m in re.finditer(r";;#ASMSTART\n(?:(?!;;#ASMEND).)*v_fma_f64", body, re.S)]
        if len(steps) < 2:
            continue                      # the compiled (non-assembly) step forms
        header = body[:steps[0]].rfind("Loop Header: Depth=1")   # the word loop (the partial last word has an inner one)
        assert header >= 0, name
        loads = re.findall(r"global_load_\w+", body[header:steps[0]])
        assert 1 <= len(loads) <= 3, (name, loads)
        checked += 1
    assert checked >= 1
