"""pm_packets_pack / pm_packets_unpack: the wire form of packet rows used by the multi-GPU exchange (host-only code)."""
import ctypes

import numpy as np
import pytest

from pymodem_amd._native import NativeError, check, lib, packet_dtype


def make_rows(n, seed):
    rng = np.random.default_rng(seed)
    rows = np.zeros(n, dtype=packet_dtype())
    rows["streamaddress"] = np.sort(rng.integers(0, 1 << 40, n))
    rows["len"] = rng.choice([0, 1, 2, 17, 255, 1023, 1280], n)
    rows["bytes_corrected"] = rng.integers(0, 17, n)
    rows["calculated_crc"] = rng.integers(0, 65536, n)
    rows["carried_crc"] = rng.integers(0, 65536, n)
    rows["valid_crc"] = rng.integers(0, 2, n)
    rows["valid_header"] = rng.integers(0, 2, n)
    rows["source_decoder"] = rng.integers(0, 64, n)
    for k in range(n):
        rows[k]["data"][:rows[k]["len"]] = rng.integers(0, 256, rows[k]["len"])
    return rows


@pytest.mark.parametrize("n", [0, 1, 7, 300])
def test_pack_unpack_round_trip(n):
    rows = make_rows(n, n)
    need = check(lib().pm_packets_pack(rows.ctypes.data_as(ctypes.c_void_p), n, None, 0))
    assert need == 40 * n + int(rows["len"].sum())
    buf = np.full(need + 8, 0xAA, dtype=np.uint8)
    assert check(lib().pm_packets_pack(rows.ctypes.data_as(ctypes.c_void_p), n, buf.ctypes.data_as(ctypes.c_void_p), need)) == need
    assert np.all(buf[need:] == 0xAA)                                   # nothing written past the need
    out = np.empty(n, dtype=packet_dtype())
    out.view(np.uint8)[:] = 0x55                                        # uninitialised on purpose
    assert check(lib().pm_packets_unpack(buf.ctypes.data_as(ctypes.c_void_p), need, out.ctypes.data_as(ctypes.c_void_p), n)) == n
    assert np.array_equal(out, rows)


def test_pack_reports_need_when_it_does_not_fit_and_unpack_rejects_garbage():
    rows = make_rows(5, 3)
    need = check(lib().pm_packets_pack(rows.ctypes.data_as(ctypes.c_void_p), 5, None, 0))
    small = np.zeros(need - 1, dtype=np.uint8)
    assert check(lib().pm_packets_pack(rows.ctypes.data_as(ctypes.c_void_p), 5, small.ctypes.data_as(ctypes.c_void_p), need - 1)) == need
    assert not small.any()
    buf = np.zeros(need, dtype=np.uint8)
    check(lib().pm_packets_pack(rows.ctypes.data_as(ctypes.c_void_p), 5, buf.ctypes.data_as(ctypes.c_void_p), need))
    out = np.empty(5, dtype=packet_dtype())
    with pytest.raises(NativeError):                                   # truncated stream
        check(lib().pm_packets_unpack(buf.ctypes.data_as(ctypes.c_void_p), need - 3, out.ctypes.data_as(ctypes.c_void_p), 5))
    with pytest.raises(NativeError):                                   # more rows than the caller has room for
        check(lib().pm_packets_unpack(buf.ctypes.data_as(ctypes.c_void_p), need, out.ctypes.data_as(ctypes.c_void_p), 4))
    bad = rows.copy()
    bad["len"][2] = 5000
    with pytest.raises(NativeError):
        check(lib().pm_packets_pack(bad.ctypes.data_as(ctypes.c_void_p), 5, None, 0))
