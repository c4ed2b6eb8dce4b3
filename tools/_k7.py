import numpy as np, sys
sys.path.insert(0,'.')
from pymodem_amd.slicer import BinarySlicer, slice_batch
from oracle import oracle as O
n=200000
x=np.where((np.arange(n)//31000)%2==0,1.0,-1.0)
o=O.BinarySlicer(48000,"9600",{})
o.retune(symbol_rate=5.0) if hasattr(o,'retune') else None
print('oracle attrs', [a for a in dir(o) if not a.startswith('_')][:20])
for t in range(4):
    s=BinarySlicer(sample_rate=48000, config="9600"); s.retune(symbol_rate=5.0)
    got=s.slice(x)
    st=s._state
    print(t, list(got.data), list(got.address), st.phase_clock, st.working_byte, st.working_bits, st.streamaddress, s.last_stats)
