"""The host codecs (pymodem_amd/csrc/pm_codec.cpp: LFSR, AX.25 with its 64-bit skim, IL2P + Reed-Solomon, CRC) under AddressSanitizer and
UndefinedBehaviorSanitizer on the CPU -- the only place sanitizers run (there is none for the GPU build on this pool): tests/codec_fuzz.cpp
feeds 400 random streams (eight bit densities, planted flags) to both decoders in calls of random sizes down to one byte, every call from
an exact-size heap copy so that a read past a piece is a read past an allocation, and prints a digest of every packet that came out.
Run twice: the AX.25 decoder with its skim and with every byte through the table-driven machine (PM_AX25_SKIM=0) -- same packets.
tests/codec_threads.cpp under ThreadSanitizer: four threads, each taking recordings of eight chains through pm_host_decode_batch,
pm_codec_fetch_batch and pm_correlate on the library's shared worker pool -- the native executor's host stage without the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_codecs_are_clean_under_asan_and_ubsan_and_the_skim_changes_nothing(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "codec_fuzz")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-mpopcnt", "-pthread",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "codec_fuzz.cpp"), os.path.join(ROOT, "pymodem_amd", "csrc", "pm_codec.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "asan" in built.stderr.lower():
        pytest.skip("no sanitizer runtime for g++ here")
    assert built.returncode == 0, built.stderr[-2000:]
    out = []
    for skim in ("1", "0"):
        env = dict(os.environ, PM_AX25_SKIM=skim, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
        env.pop("LD_PRELOAD", None)
        r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (skim, r.stdout[-500:], r.stderr[-3000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        out.append(r.stdout.split())
    assert out[0] == out[1], out
    assert int(out[0][0]) > 10000                              # packets came out: the decoders were exercised, not idle


def test_the_host_stage_s_worker_pool_is_clean_under_tsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "codec_threads")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-mpopcnt", "-pthread", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "codec_threads.cpp"), os.path.join(ROOT, "pymodem_amd", "csrc", "pm_codec.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "tsan" in built.stderr.lower():
        pytest.skip("no sanitizer runtime for g++ here")
    assert built.returncode == 0, built.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:exitcode=66")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow here")
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert all(int(x) > 500 for x in r.stdout.split()), r.stdout
