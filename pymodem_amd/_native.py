"""ctypes binding of libpymodem_amd.so (declared in include/pymodem_amd.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  Loading is
lazy and does not touch HIP: the reference forks one process per chain after building the stage
objects (pymodem.py:144-151), so nothing here may initialise the GPU at import time.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class NativeError(RuntimeError):
    pass


def library_path():
    # (PYMODEM_AMD_LIB: another build of the same library -- a kernel variant compiled with different constants -- for A/B measurements
    # on one box; tools/build_variant.py makes them)
    return os.environ.get("PYMODEM_AMD_LIB") or os.path.join(_HERE, "libpymodem_amd.so")


class AGCParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("attack_rate", "decay_rate", "sustain_time", "sample_rate", "target_amplitude")]


class Loop(ctypes.Structure):
    """pm_loop"""
    _fields_ = [(k, ctypes.c_double) for k in (
        "phase_scaling", "index_scaling", "set_frequency", "b0", "b1", "a1", "p_rate", "i_rate", "i_limit", "gain",
        "phase", "control", "sine", "cosine", "x0", "x1", "y0", "integral", "proportional",
        "bb0", "bb1", "ba1", "cx0", "cx1", "cy0", "sx0", "sx1", "sy0")]


class AfskSweepDesc(ctypes.Structure):
    """pm_afsk_sweep_desc"""
    _fields_ = [("d_mark_i", ctypes.c_void_p), ("d_mark_q", ctypes.c_void_p), ("d_unit_i", ctypes.c_void_p), ("d_unit_q", ctypes.c_void_p),
                ("d_space", ctypes.c_void_p), ("h_gains", ctypes.c_void_p), ("groups", ctypes.c_int32), ("m", ctypes.c_int32),
                ("d_lpf", ctypes.c_void_p), ("ml", ctypes.c_int32), ("reserved", ctypes.c_int32), ("lpf_abs_sum", ctypes.c_double),
                ("h_bits", ctypes.c_void_p), ("h_tones", ctypes.c_void_p)]


class SlicerParams(ctypes.Structure):
    _fields_ = [("samples_per_symbol", ctypes.c_double), ("lock_rate", ctypes.c_double),
                ("bits_per_symbol", ctypes.c_int32), ("state_mask", ctypes.c_int32), ("demap", ctypes.c_int32 * 16)]


class SlicerState(ctypes.Structure):
    """pm_slicer_state: what a slicer object carries from one slice() call to the next"""
    _fields_ = [("phase_clock", ctypes.c_double), ("last_i_negative", ctypes.c_int32), ("last_q_negative", ctypes.c_int32),
                ("working_byte", ctypes.c_int32), ("working_bits", ctypes.c_int32), ("state_register", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("streamaddress", ctypes.c_int64)]


class RowSliceRec(ctypes.Structure):
    """pm_rowslice_rec: a stream's record of a sliced engine run (pm_lbatch_run_sliced)"""
    _fields_ = [("count", ctypes.c_int64), ("first_addr", ctypes.c_int64), ("last_addr", ctypes.c_int64), ("seen", ctypes.c_int64),
                ("clk", ctypes.c_double), ("li_neg", ctypes.c_int32), ("lq_neg", ctypes.c_int32), ("wbyte", ctypes.c_int32),
                ("wbits", ctypes.c_int32), ("sreg", ctypes.c_int32), ("flags", ctypes.c_int32)]


def rowslice_dtype():
    import numpy as np
    return np.dtype([("count", "<i8"), ("first_addr", "<i8"), ("last_addr", "<i8"), ("seen", "<i8"), ("clk", "<f8"), ("li_neg", "<i4"),
                     ("lq_neg", "<i4"), ("wbyte", "<i4"), ("wbits", "<i4"), ("sreg", "<i4"), ("flags", "<i4")])


class SliceJob(ctypes.Structure):
    """pm_slice_job"""
    _fields_ = [("d_bits_i", ctypes.c_void_p), ("d_bits_q", ctypes.c_void_p), ("n", ctypes.c_int64), ("params", SlicerParams),
                ("d_data", ctypes.c_void_p), ("d_addr", ctypes.c_void_p), ("cap", ctypes.c_int64), ("count", ctypes.c_int64),
                ("h_state", ctypes.POINTER(SlicerState))]


class AfskTones(ctypes.Structure):
    """pm_afsk_tones"""
    _fields_ = [("mark_rot", ctypes.c_double * 2), ("mark_end", ctypes.c_double * 2), ("space_rot", ctypes.c_double * 2),
                ("space_end", ctypes.c_double * 2), ("tap_dev", ctypes.c_double)]


class HostJob(ctypes.Structure):
    """pm_host_job"""
    _fields_ = [("codec", ctypes.c_void_p), ("h_data", ctypes.c_void_p), ("h_addr", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("lfsr_poly", ctypes.c_uint64), ("lfsr_state", ctypes.c_uint64), ("pending", ctypes.c_int64),
                ("lfsr_invert", ctypes.c_int32), ("status", ctypes.c_int32),
                ("h_addr_delta", ctypes.c_void_p), ("addr_first", ctypes.c_int64), ("h_plain", ctypes.c_void_p)]


class ChainDesc(ctypes.Structure):
    """pm_chain_desc"""
    _dp, _i32 = ctypes.POINTER(ctypes.c_double), ctypes.c_int32
    _fields_ = [("modem", _i32), ("flags", _i32),
                ("input_fir", _dp), ("n_input_fir", _i32),
                ("mark_i", _dp), ("mark_q", _dp), ("space_i", _dp), ("space_q", _dp), ("n_corr", _i32),
                ("hilbert", _dp), ("n_hilbert", _i32), ("hilbert_delay", _i32),
                ("output_fir", _dp), ("n_output_fir", _i32),
                ("use_agc", _i32), ("agc", AGCParams),
                ("loop", Loop), ("wavetable", _dp), ("pd_table", ctypes.POINTER(ctypes.c_int32)),
                ("quadrature", _i32), ("slicer", SlicerParams)]


class LBatchDesc(ctypes.Structure):
    """pm_lbatch_desc"""
    _dp, _i32 = ctypes.POINTER(ctypes.c_double), ctypes.c_int32
    _fields_ = [("modem", _i32), ("recordings", _i32), ("chains", _i32), ("chunk", _i32),
                ("input_fir", _dp), ("n_input_fir", _i32),
                ("hilbert", _dp), ("n_hilbert", _i32), ("hilbert_delay", _i32),
                ("output_fir", _dp), ("n_output_fir", _i32),
                ("agc", AGCParams), ("loops", ctypes.POINTER(Loop)), ("wavetable", _dp), ("pd_table", ctypes.POINTER(ctypes.c_int32))]


class PipeChain(ctypes.Structure):
    """pm_pipe_chain"""
    _i32 = ctypes.c_int32
    _fields_ = [("sweep", _i32), ("slot", _i32), ("slicer", SlicerParams), ("lfsr_poly", ctypes.c_uint64), ("lfsr_invert", _i32),
                ("codec_kind", _i32), ("crc", _i32), ("disable_rs", _i32), ("min_dist", _i32), ("sync_tol", _i32), ("source_decoder", _i32)]


class PipeFir(ctypes.Structure):
    """pm_pipe_fir"""
    _fields_ = [("d_taps", ctypes.c_void_p), ("m", ctypes.c_int32), ("flags", ctypes.c_int32)]


class PipeDesc(ctypes.Structure):
    """pm_pipe_desc"""
    _i32 = ctypes.c_int32
    _fields_ = [("d_bpf", ctypes.c_void_p), ("mb", _i32), ("nsweeps", _i32), ("x_bound", ctypes.c_double),
                ("sweeps", ctypes.POINTER(AfskSweepDesc)), ("chains", ctypes.POINTER(PipeChain)), ("nchains", _i32), ("slots", _i32),
                ("slice_workers", _i32), ("slice_group", _i32), ("slice_min_group", _i32), ("demod_streams", _i32), ("host_threads", _i32), ("decode_threads", _i32),
                ("address_distance", ctypes.c_double), ("max_samples", ctypes.c_int64),
                ("firs", ctypes.POINTER(PipeFir)), ("nfirs", _i32), ("keep_slices", _i32)]


class PipeResult(ctypes.Structure):
    """pm_pipe_result"""
    _fields_ = [("ticket", ctypes.c_int64), ("status", ctypes.c_int32), ("reserved", ctypes.c_int32), ("rows", ctypes.c_int64),
                ("h_rows", ctypes.c_void_p), ("h_counts", ctypes.POINTER(ctypes.c_int64)), ("unique", ctypes.c_int64),
                ("h_unique_idx", ctypes.c_void_p), ("h_corr_decoders", ctypes.c_void_p),
                ("ms_to_demod_done", ctypes.c_double), ("ms_to_sliced", ctypes.c_double), ("ms_to_done", ctypes.c_double),
                ("done_at_ms", ctypes.c_double)]


MODEM_AFSK, MODEM_FSK, MODEM_BPSK, MODEM_MPSK, MODEM_AFSK_PLL, MODEM_QPSK = range(6)
CHAIN_INVERT = 1
CHAIN_CARRY_HISTORY = 2
KERNEL_CLASSES = ("fir_i16", "fir_f64", "afsk_correlate", "signs", "slice_iter", "slice_emit", "agc", "loop")
PKT_MAX = 1280


class Packet(ctypes.Structure):
    """pm_packet"""
    _fields_ = [("streamaddress", ctypes.c_int64), ("len", ctypes.c_int32), ("bytes_corrected", ctypes.c_int32),
                ("calculated_crc", ctypes.c_int32), ("carried_crc", ctypes.c_int32), ("valid_crc", ctypes.c_int32),
                ("valid_header", ctypes.c_int32), ("source_decoder", ctypes.c_int32), ("correlated_count", ctypes.c_int32),
                ("data", ctypes.c_uint8 * PKT_MAX)]


def packet_dtype():
    """NumPy view of pm_packet (same layout as the ctypes Packet: 40-byte header + PKT_MAX data bytes)."""
    import numpy as np
    dt = np.dtype([("streamaddress", "<i8"), ("len", "<i4"), ("bytes_corrected", "<i4"), ("calculated_crc", "<i4"), ("carried_crc", "<i4"),
                   ("valid_crc", "<i4"), ("valid_header", "<i4"), ("source_decoder", "<i4"), ("correlated_count", "<i4"),
                   ("data", "u1", (PKT_MAX,))])
    assert dt.itemsize == ctypes.sizeof(Packet)
    return dt


def head_dtype():
    """NumPy view of pm_packet_head: the 40 bytes every pm_packet starts with."""
    import numpy as np
    dt = np.dtype([("streamaddress", "<i8"), ("len", "<i4"), ("bytes_corrected", "<i4"), ("calculated_crc", "<i4"), ("carried_crc", "<i4"),
                   ("valid_crc", "<i4"), ("valid_header", "<i4"), ("source_decoder", "<i4"), ("correlated_count", "<i4")])
    assert dt.itemsize == 40
    return dt


_vp, _i64, _int, _dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
_SIGS = {
    "pm_version": ([], _int),
    "pm_device_count": ([], _int),
    "pm_last_error": ([ctypes.c_char_p, ctypes.c_size_t], _int),
    "pm_ctx_create": ([_int, ctypes.POINTER(_vp)], _int),
    "pm_ctx_create_prio": ([_int, _int, ctypes.POINTER(_vp)], _int),
    "pm_ctx_tune": ([_vp, ctypes.c_char_p, _i64], _int),
    "pm_d2d": ([_vp, _vp, _vp, ctypes.c_size_t], _int),
    "pm_host_pin": ([_vp, _vp, ctypes.c_size_t], _int),
    "pm_host_unpin": ([_vp], _int),
    "pm_ctx_scratch": ([_vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)], _int),
    "pm_event_query": ([_vp], _int),
    "pm_event_sync": ([_vp], _int),
    "pm_event_sync_relaxed": ([_vp, _int], _int),
    "pm_ctx_create_cumask": ([_int, ctypes.POINTER(ctypes.c_uint32), _int, ctypes.POINTER(_vp)], _int),
    "pm_device_cus": ([_int], _int),
    "pm_event_record": ([_vp, ctypes.POINTER(_vp)], _int),
    "pm_event_wait": ([_vp, _vp], _int),
    "pm_event_destroy": ([_vp], _int),
    "pm_ctx_destroy": ([_vp], _int),
    "pm_ctx_sync": ([_vp], _int),
    "pm_ctx_stream": ([_vp], _vp),
    "pm_malloc": ([_vp, ctypes.c_size_t, ctypes.POINTER(_vp)], _int),
    "pm_free": ([_vp, _vp], _int),
    "pm_h2d": ([_vp, _vp, _vp, ctypes.c_size_t], _int),
    "pm_d2h": ([_vp, _vp, _vp, ctypes.c_size_t], _int),
    "pm_memset": ([_vp, _vp, _int, ctypes.c_size_t], _int),
    "pm_timer_start": ([_vp], _int),
    "pm_timer_stop": ([_vp, ctypes.POINTER(ctypes.c_float)], _int),
    "pm_prof_enable": ([_vp, _int], _int),
    "pm_prof_read": ([_vp, _int, ctypes.POINTER(_dbl), ctypes.POINTER(_i64)], _int),
    "pm_prof_work": ([_vp, _int, ctypes.POINTER(_dbl), ctypes.POINTER(_dbl)], _int),
    "pm_fir_valid_i16": ([_vp, _vp, _i64, _vp, _int, _vp, _int], _int),
    "pm_fir_valid_i16_limbs": ([_vp, _vp, _i64, _vp, _int, _vp, _vp], _int),
    "pm_fir_valid_f64": ([_vp, _vp, _i64, _vp, _int, _vp, _int], _int),
    "pm_fir_signs_i16": ([_vp, _vp, _i64, _vp, _int, _vp, _int], _int),
    "pm_fir_signs_f64": ([_vp, _vp, _i64, _vp, _int, _vp, _int], _int),
    "pm_fir_signs_f64_batch": ([_vp, _int, ctypes.POINTER(_vp), ctypes.POINTER(_i64), _vp, _int, ctypes.POINTER(_vp), _int], _int),
    "pm_fir_rows_i16": ([_vp, _vp, _i64, _int, _i64, _vp, _int, _vp, _i64, _int], _int),
    "pm_fir_rows_i16_ptrs": ([_vp, _vp, _i64, _int, _int, _i64, _vp, _int, _vp, _i64, _int], _int),
    "pm_fir_rows_f64": ([_vp, _vp, _i64, _int, _i64, _vp, _int, _vp, _i64, _int], _int),
    "pm_fir_rows_signs_f64": ([_vp, _vp, _i64, _int, _i64, _vp, _int, _vp, _i64, _int], _int),
    "pm_ubench_sqrt_f32": ([_vp, _int, ctypes.POINTER(_i64)], _int),
    "pm_matrix_digit_pairs": ([_int], _int),
    "pm_bpf8_rows_max_i16": ([_vp, _vp, _i64, _int, _i64, _vp, _int, _vp, ctypes.POINTER(_i64)], _int),
    "pm_fir8_rows_signs_f64": ([_vp, _vp, _i64, _int, _i64, _vp, _int, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_afsk_correlate": ([_vp, _vp, _i64, _vp, _vp, _vp, _vp, _int, _vp], _int),
    "pm_afsk_correlate_group": ([_vp, _vp, _i64, _vp, _vp, _vp, _int, _int, _vp, _i64], _int),
    "pm_afsk_sweep_signs": ([_vp, _vp, _i64, _dbl, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_dbl), _int, _int, _vp, _int, _dbl,
                            ctypes.POINTER(_vp)], _int),
    "pm_afsk_sweep_signs_tones": ([_vp, _vp, _i64, _dbl, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_dbl), _int, _int, _vp, _int, _dbl,
                            ctypes.POINTER(_vp)] + [ctypes.POINTER(AfskTones)], _int),
    "pm_afsk_magnitudes": ([_vp, _vp, _i64, _dbl, _vp, _vp, _vp, _vp, _int, ctypes.POINTER(AfskTones), _vp, _vp, ctypes.POINTER(_dbl)], _int),
    "pm_afsk_sweep_last": ([_vp, ctypes.POINTER(_i64)], _int),
    "pm_afsk_group_run": ([_vp, _vp, _i64, _vp, _int, _vp, ctypes.c_double, _vp, _int, ctypes.POINTER(_i64)], _int),
    "pm_afsk_sweep_mode": ([_vp, _int], _int),
    "pm_afsk_sweep_ticket": ([_vp, ctypes.POINTER(_i64)], _int),
    "pm_afsk_sweep_results": ([_vp, ctypes.POINTER(_i64), _int, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)], _int),
    "pm_afsk_sweep_result": ([_vp, _i64, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)], _int),
    "pm_signs_f64": ([_vp, _vp, _i64, _vp], _int),
    "pm_agc_apply": ([_vp, _vp, _i64, ctypes.POINTER(AGCParams), ctypes.POINTER(_dbl)], _int),
    "pm_agc_rows_apply": ([_vp, _vp, _i64, _vp, _i64, _int, _i64, ctypes.POINTER(AGCParams), ctypes.POINTER(_dbl), ctypes.POINTER(_dbl)], _int),
    "pm_rows_max_f64": ([_vp, _vp, _i64, _int, _i64, ctypes.POINTER(_dbl)], _int),
    "pm_lbatch_create": ([_vp, ctypes.POINTER(LBatchDesc), ctypes.POINTER(_vp)], _int),
    "pm_lbatch_geometry": ([_vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64)], _int),
    "pm_lbatch_run_sliced": ([_vp, ctypes.POINTER(_vp), _int, _i64, _vp, _int, _vp, _vp, _i64, _vp, ctypes.POINTER(_i64)], _int),
    "pm_rows_gather": ([_vp, _vp, _vp, _vp, _i64, _i64, _int, _vp, ctypes.c_size_t], _int),
    "pm_lbatch_run": ([_vp, ctypes.POINTER(_vp), _int, _i64, _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_lbatch_front_ctx": ([_vp], _vp),
    "pm_lbatch_tail_ctx": ([_vp], _vp),
    "pm_lbatch_loop_ctx": ([_vp], _vp),
    "pm_lbatch_slice_ctx": ([_vp], _vp),
    "pm_lbatch_destroy": ([_vp], _int),
    "pm_prof_intervals": ([_vp, _int, _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_pipe_create": ([_vp, ctypes.POINTER(PipeDesc), ctypes.POINTER(_vp)], _int),
    "pm_pipe_submit": ([_vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_pipe_submit_many": ([_vp, _vp, _vp, _int, _vp], _int),
    "pm_pipe_promise": ([_vp, _int, _vp], _int),
    "pm_pipe_wait": ([_vp, _i64, ctypes.POINTER(PipeResult)], _int),
    "pm_pipe_release": ([_vp, _i64], _int),
    "pm_pipe_slices": ([_vp, _i64, _int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)], _int),
    "pm_pipe_bitmap": ([_vp, _i64, _int, _vp, _i64], _int),
    "pm_pipe_slots": ([_vp], _int),
    "pm_pipe_drain": ([_vp], _int),
    "pm_pipe_stats": ([_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)], _int),
    "pm_pipe_side_ctx": ([_vp, _int], _vp),
    "pm_pipe_demod_ctx": ([_vp, _int], _vp),
    "pm_pipe_destroy": ([_vp], _int),
    "pm_costas_bpsk": ([_vp, ctypes.POINTER(Loop), _int, _vp, _vp, _i64, _i64, _vp, _i64], _int),
    "pm_costas_qpsk": ([_vp, ctypes.POINTER(Loop), _int, _vp, _vp, _i64, _i64, _vp, _vp, _i64], _int),
    "pm_pll_afsk": ([_vp, ctypes.POINTER(Loop), _int, _vp, _vp, _i64, _i64, _vp, _i64], _int),
    "pm_mpsk_loop": ([_vp, ctypes.POINTER(Loop), _int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64], _int),
    "pm_slice_binary": ([_vp, _vp, _i64, ctypes.POINTER(SlicerParams), _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_slice_quadrature": ([_vp, _vp, _vp, _i64, ctypes.POINTER(SlicerParams), _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_slice_batch": ([_vp, ctypes.POINTER(SliceJob), _int], _int),
    "pm_slicer_tune": ([_vp, _i64], _int),
    "pm_slicer_limits": ([_vp, _i64], _int),
    "pm_slice_compact": ([_vp, _vp, _int, _vp, ctypes.c_size_t, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_size_t)], _int),
    "pm_slicer_stats": ([_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_i64)], _int),
    "pm_chain_create": ([_vp, ctypes.POINTER(ChainDesc), ctypes.POINTER(_vp)], _int),
    "pm_chain_run": ([_vp, _vp, _i64, _int, _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_chain_fetch": ([_vp, _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_chain_reset": ([_vp], _int),
    "pm_chain_destroy": ([_vp], _int),
    "pm_lfsr_unscramble": ([_vp, _i64, ctypes.c_uint64, _int, ctypes.POINTER(ctypes.c_uint64), _vp], _int),
    "pm_codec_create": ([_int, _int, _int, _int, _int, _int, ctypes.POINTER(_vp)], _int),
    "pm_codec_destroy": ([_vp], _int),
    "pm_codec_set_source": ([_vp, ctypes.c_int32], _int),
    "pm_codec_decode": ([_vp, _vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_codec_fetch": ([_vp, _vp, _i64, ctypes.POINTER(_i64)], _int),
    "pm_host_decode_batch": ([ctypes.POINTER(HostJob), _int, _int], _int),
    "pm_codec_fetch_batch": ([ctypes.POINTER(_vp), ctypes.POINTER(_i64), _int, _vp, _int], _int),
    "pm_packets_pack": ([_vp, _i64, _vp, _i64], _i64),
    "pm_packets_unpack": ([_vp, _i64, _vp, _i64], _i64),
    "pm_packets_index": ([_vp, _i64, _vp, _vp, _i64], _i64),
    "pm_correlate_strided": ([_vp, _i64, ctypes.POINTER(_i64), _int, _dbl, _vp, _vp, _i64], _i64),
    "pm_crc16_ccitt": ([_vp, _i64], _int),
    "pm_correlate": ([_vp, ctypes.POINTER(_i64), _int, _dbl, _vp, _vp, _i64], _i64),
}
EXPORTS = tuple(_SIGS)


def lib():
    """The loaded library.  Raises NativeError when it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise NativeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). pymodem_amd has no CPU fallback.")
        # (takes effect when this is what loads the HIP runtime: streams of several libraries in one process -- RCCL, torch, ours -- share
        # the default four hardware queues and serialise)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        handle = ctypes.CDLL(path)
        for name, (args, res) in _SIGS.items():
            fn = getattr(handle, name)      # AttributeError here = header and library out of sync
            fn.argtypes = args
            fn.restype = res
        _LIB = handle
    return _LIB


_QUICK = None
QUICK_CALLS = ("pm_afsk_group_run", "pm_codec_create", "pm_codec_destroy", "pm_codec_set_source", "pm_event_record", "pm_event_wait", "pm_event_query", "pm_last_error")


def quick():
    """The same library through ctypes.PyDLL, for the entry points in QUICK_CALLS only: calls of a few microseconds that never wait.
    ctypes.CDLL drops the interpreter lock around every call and has to get it back afterwards -- with a dozen busy threads that
    costs 0.1-0.2 ms per call (measured: 0.19 ms per pm_codec_destroy, eight per recording), a hundred times the call itself."""
    global _QUICK
    if _QUICK is None:
        lib()
        handle = ctypes.PyDLL(library_path())
        for name in QUICK_CALLS:
            fn = getattr(handle, name)
            fn.argtypes, fn.restype = _SIGS[name]
        _QUICK = handle
    return _QUICK


def check(rc):
    if rc < 0:
        buf = ctypes.create_string_buffer(512)
        lib().pm_last_error(buf, 512)
        raise NativeError(f"pymodem_amd error {rc}: {buf.value.decode(errors='replace')}")
    return rc
