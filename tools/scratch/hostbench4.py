import sys, time, numpy as np, pickle, threading, ctypes
sys.path.insert(0, '.')
import bench
from pymodem_amd import chain_builder as CB, chain_execute as CE
from pymodem_amd.slicer import AddressedArray
factory = bench.WORKLOADS["afsk_1200_super_opt"][0]
sl = pickle.load(open('tools/scratch/sliced.pkl','rb'))
chains = [CB.build_chain(48000, factory(c)) for c in range(8)]
sliced = [AddressedArray(sl[c % 2][0].copy(), sl[c % 2][1].copy()) for c in range(8)]
un = [ch[3].stream_unscramble_8bit(s) for ch,s in zip(chains,sliced)]
[ch[4].decode_pending(u) for ch,u in zip(chains,un)]
bar = threading.Barrier(9)
res=[0]*8
def timed(i):
    for rep in range(3):
        bar.wait()
        t=time.perf_counter(); chains[i][4].decode_pending(un[i]); res[i]=((t-T0)*1e3, (time.perf_counter()-T0)*1e3)
        bar.wait()
ths=[threading.Thread(target=timed,args=(i,)) for i in range(8)]
[t.start() for t in ths]
for rep in range(3):
    T0=time.perf_counter()
    bar.wait(); bar.wait()
    print("total %.3f" % ((time.perf_counter()-T0)*1e3), [(round(a,2),round(b,2)) for a,b in res])
