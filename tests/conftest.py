import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress -- a kernel that never ends, a stream waiting for an event nobody records -- must fail with the
    test's name and every thread's stack, not sit there until whoever runs the suite gives up (the longest test takes 90 s; pytest-timeout's
    thread method dumps the stacks and ends the process, so the device context goes with it)."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(360, method="thread"))


def pytest_sessionstart(session):
    """The in-tree libraries are git-ignored build products: build them when they are missing (hipcc cross-compiles gfx950
    without a GPU; ~40 s once).  On the GPU box they arrive prebuilt with the snapshot."""
    lib = os.path.join(ROOT, "pymodem_amd", "libpymodem_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    """Lazy loader for the committed .npz fixtures (tests/golden/make_goldens.py made them)."""
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def config_lines():
    def load(name):
        with open(os.path.join(GOLDEN, "configs", name)) as f:
            return [l for l in (json.loads(s) for s in f if s.strip()) if l.get("object_type") == "demod_chain"]
    return load


import contextlib

_TUNE_DEFAULTS = {"loop_wide": -1, "lbatch_tail": -1, "fir8": 1, "bpf8_max": 1, "lbatch_loop_cus": -1, "loop_agc": 1, "loop_vec": 1, "agc_rows_prio": 2}


@contextlib.contextmanager
def tuned(ctx, **switches):
    """ctx.tune(**switches) for the body, the defaults again afterwards (the contexts of the tests are shared)."""
    ctx.tune(**switches)
    try:
        yield ctx
    finally:
        ctx.tune(**{k: _TUNE_DEFAULTS.get(k, 0) for k in switches})


def noise_i16(n, seed=1234, sigma=8000.0):
    """The synthetic buffer BASELINE.md prescribes."""
    x = np.random.default_rng(seed).standard_normal(n) * sigma
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def read_wav_pcm16(path):
    """Minimal RIFF/PCM16 mono reader (tests only)."""
    import struct
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    rate = None
    while pos < len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt, ch, rate, _, _, bits = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
            assert fmt == 1 and ch == 1 and bits == 16
        elif cid == b"data":
            return rate, np.frombuffer(b[pos + 8:pos + 8 + sz], dtype="<i2").copy()
        pos += 8 + sz + (sz & 1)
    raise ValueError("no data chunk")


def oracle_chains(jobs, rate=48000, workers=8):
    """[(config line, int16 audio), ...] -> futures of O.run_chain(O.build_chain(rate, line), audio, canon=True), on a thread pool: the
    oracle's filters, loops and slicers are C behind ctypes (no interpreter lock while they run), so the checker's minute of one host core
    per full-size chain lies beside the GPU run it checks and beside the other chains' -- the suite's wall time, not its coverage."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.fir_canon(np.ones(4), np.ones(2))                       # the library and its tables are loaded before any thread asks for them
    pool = ThreadPoolExecutor(max_workers=max(1, min(workers, len(jobs))))
    futures = [pool.submit(lambda l=line, a=audio: O.run_chain(O.build_chain(rate, l), a, canon=True)) for line, audio in jobs]
    pool.shutdown(wait=False)
    return futures
