"""Chain assembly with the reference's four factory signatures (chain_builder.py:17-69): the same
{"type", "config", "options"} dicts give the GPU-backed stage objects.  Unknown types return [] like there."""
from . import codecs, lfsr, modems, slicer

_MODEMS = {'qpsk': modems.QPSKModem, 'mpsk': modems.MPSKModem, 'bpsk': modems.BPSKModem, 'fsk': modems.FSKModem, 'afsk': modems.AFSKModem,
           'afsk_pll': modems.AFSKPLLModem}
_SLICERS = {'quadrature': slicer.QuadratureSlicer, 'binary': slicer.BinarySlicer}


def ModemConfigurator(arg_sample_rate, input_args):
    new_object = []
    cls = _MODEMS.get(input_args.get('type'))
    if cls is not None:
        new_object = cls(sample_rate=arg_sample_rate, config=input_args['config'])
        new_object.StringOptionsRetune(input_args['options'])
    return new_object


def SlicerConfigurator(arg_sample_rate, input_args):
    new_object = []
    cls = _SLICERS.get(input_args.get('type'))
    if cls is not None:
        new_object = cls(sample_rate=arg_sample_rate, config=input_args['config'])
        new_object.StringOptionsRetune(input_args['options'])
    return new_object


def StreamConfigurator(input_args):
    new_object = []
    if input_args.get('type') == 'lfsr':
        new_object = lfsr.LFSR()
        new_object.StringOptionsRetune(input_args['options'])
    return new_object


def CodecConfigurator(input_args, name):
    new_object = []
    kind = input_args['type'].lower()
    if kind == 'il2p':
        new_object = codecs.IL2PCodec(ident=name)
        new_object.StringOptionsRetune(input_args['options'])
    elif kind == 'ax25':
        new_object = codecs.AX25Codec(ident=name)
    return new_object


def build_chain(sample_rate, line):
    """One 'demod_chain' config line -> [name, modem, slicer, stream, codec], as pymodem.py:67-115 does."""
    modem = ModemConfigurator(sample_rate, line['modem'])
    slicer_rate = getattr(modem, 'output_sample_rate', sample_rate)
    return [line['object_name'], modem, SlicerConfigurator(slicer_rate, line['slicer']),
            StreamConfigurator(line['stream']), CodecConfigurator(line['codec'], line['object_name'])]
