"""Chain execution (chain_execute.py:6-52 of the reference): demod -> slice -> stream -> codec.
`process_chain` has the reference's signature and stage-by-stage host hand-offs; `process_chain_device`
keeps the demodulated stream in HBM between modem and slicer (only the sliced bytes come back)."""
from .device import Context, DeviceBuffer


def process_chain(chain, input_audio):
    demod_audio = chain[1].demod(input_audio)
    sliced_data = chain[2].slice(demod_audio)
    descrambled_data = chain[3].stream_unscramble_8bit(sliced_data)
    return chain[4].decode(descrambled_data)


def multiprocess_chain(chain, input_audio, queue):
    queue.put(process_chain(chain, input_audio))


def process_chain_device(chain, input_audio, stages=None):
    """Same result as process_chain; `input_audio` may already be a DeviceBuffer.  If `stages` is a dict it
    receives the slicer output and the descrambled stream (for parity checks)."""
    demod_audio = chain[1].demod(input_audio, device_out=True)
    sliced_data = chain[2].slice(demod_audio)
    descrambled_data = chain[3].stream_unscramble_8bit(sliced_data)
    if stages is not None:
        stages["sliced"], stages["descrambled"] = sliced_data, descrambled_data
    return chain[4].decode(descrambled_data)
