"""Boundary types of the chain (data_classes.py:7-17 of the reference) plus their device/array forms."""
import numpy as np


class AddressedData:
    """One sliced byte and the 1-based stream address of the sample that completed it."""
    __slots__ = ("data", "address")

    def __init__(self, data, address, *args):
        self.data = data
        self.address = address


class IQData:
    def __init__(self):
        self.i_data = []
        self.q_data = []


class DeviceIQ:
    """IQData whose two streams are DeviceBuffers in HBM (MPSKModem.demod(device_out=True))."""
    def __init__(self, i_data, q_data):
        self.i_data = i_data
        self.q_data = q_data


class SignBits:
    """What a slicer needs of a demodulated stream: the (x >= 0) bitmap(s) in HBM and the sample count.  Produced by
    modem.demod_signs(), where the chain's last FIR writes the bitmap instead of the float64 stream."""
    def __init__(self, bits_i, bits_q, n):
        self.bits_i, self.bits_q, self.n = bits_i, bits_q, int(n)


class AddressedArray:
    """list[AddressedData] stored as two NumPy arrays.  Behaves like the reference's list for len(), indexing and
    iteration (items are materialised on demand), while the native stages read `.data` / `.address` directly."""

    def __init__(self, data, address):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self._address = np.ascontiguousarray(address, dtype=np.int64)
        self._steps = None
        assert self.data.shape == self._address.shape

    @classmethod
    def from_steps(cls, data, steps, first):
        """The slicer's compact form (pm_slice_compact): address[i] = first + steps[0] + ... + steps[i] with steps[0] = 0.  The
        native host stages take it as it is; `.address` expands it when somebody asks."""
        self = cls.__new__(cls)
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self._steps = (np.ascontiguousarray(steps, dtype=np.uint16), int(first))
        self._address = None
        assert self.data.shape == self._steps[0].shape
        return self

    @property
    def address(self):
        if self._address is None:
            steps, first = self._steps
            self._address = np.cumsum(steps, dtype=np.int64) + first
        return self._address

    @property
    def address_steps(self):
        """(uint16 steps, first address) if the addresses are still in compact form, else None."""
        return self._steps if self._address is None else None

    @classmethod
    def coerce(cls, seq):
        if isinstance(seq, cls):
            return seq
        return cls(np.fromiter((int(s.data) for s in seq), dtype=np.uint8, count=len(seq)),
                   np.fromiter((int(s.address) for s in seq), dtype=np.int64, count=len(seq)))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return AddressedArray(self.data[k], self.address[k])
        return AddressedData(int(self.data[k]), int(self.address[k]))

    def __iter__(self):
        for d, a in zip(self.data.tolist(), self.address.tolist()):
            yield AddressedData(d, a)
