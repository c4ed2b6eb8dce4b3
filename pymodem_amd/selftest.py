"""Small end-to-end check used by __graft_entry__.smoke(): one AFSK-1200 chain and one QPSK-2400 chain on
cuda:0 through the C ABI, compared with the oracle (the oracle is only the checker here)."""
import json
import os

import numpy as np


def smoke():
    from . import chain_builder as cb, lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import oracle as O
    assert lib().pm_device_count() >= 1, "smoke() needs a GPU"
    x = np.random.default_rng(1234).standard_normal(60000) * 8000
    audio = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    for name, idx in [("afsk_1200.json", 0), ("qpsk_2400.json", 1)]:
        with open(os.path.join(root, "tests", "golden", "configs", name)) as f:
            line = [json.loads(s) for s in f if s.strip()][idx]
        chain = cb.build_chain(48000, line)
        demod = chain[1].demod(audio)
        sliced = chain[2].slice(demod)
        pkts = chain[4].decode(chain[3].stream_unscramble_8bit(sliced))
        want = O.run_chain(O.build_chain(48000, line), audio, canon=True)
        got = [demod.i_data, demod.q_data] if hasattr(demod, "i_data") else [demod]
        ref = list(want["demod"]) if isinstance(want["demod"], tuple) else [want["demod"]]
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), f"{name}: demodulated stream differs from the oracle"
        assert np.array_equal(sliced.data, want["slice_data"]) and np.array_equal(sliced.address, want["slice_addr"]), name
        assert len(pkts) == len(want["packets"])
        print(f"smoke {name}: {len(demod.i_data) if hasattr(demod, 'i_data') else len(demod)} samples, {len(sliced)} bytes, ok")
