import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from moira_amd.engine import Engine
n, L, stride = 10_000_000, 300, 320
def setup(eng, n, first):
    d_q = eng.alloc(n*stride); d_ee=eng.alloc(n*8); d_ns=eng.alloc(n*4); d_pass=eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2, first_read=first)
    return d_q,d_ee,d_ns,d_pass
def run(engs, bufs, nper, steps=5):
    prm = engs[0].params()
    for _ in range(2):
        for e,b in zip(engs,bufs): e.filter_device(b[0], nper, stride, fixed_len=L, d_ee=b[1], d_ns=b[2], d_pass=b[3], params=prm, want_counts=False)
    for e in engs: e.synchronize()
    t=time.perf_counter()
    for _ in range(steps):
        for e,b in zip(engs,bufs): e.filter_device(b[0], nper, stride, fixed_len=L, d_ee=b[1], d_ns=b[2], d_pass=b[3], params=prm, want_counts=False)
    for e in engs: e.synchronize()
    dt=(time.perf_counter()-t)/steps
    return dt
for k in (1,2,4,8):
    engs=[Engine(0) for _ in range(k)]
    bufs=[setup(e, n//k, i*(n//k)) for i,e in enumerate(engs)]
    dt=run(engs,bufs,n//k)
    print("contexts=%d  ms/step=%.3f  reads/s=%.3e"%(k, dt*1e3, n/dt), flush=True)
    for e,b in zip(engs,bufs):
        for x in b: x.free()
        e.close()
