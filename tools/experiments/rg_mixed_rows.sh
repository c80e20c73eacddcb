for i in 1 2 3; do
python - <<'PY'
import os,sys,time
sys.path.insert(0,os.getcwd())
import numpy as np
from moira_amd.engine import Engine
n,stride=5_000_000,640
with Engine(0) as eng:
    d_q,d_len,d_ee,d_ns,d_pass=eng.alloc(n*stride),eng.alloc(n*4),eng.alloc(n*8),eng.alloc(n*4),eng.alloc(n)
    eng.synth_fill(d_q,n,stride,min_len=50,max_len=600,d_len=d_len,seed=6,profile=1)
    out=[]
    for name,kw in (("R3",dict(narrow_rows=3)),("R3 split 22",dict(narrow_rows=3,narrow_split=22)),("R3 split 24",dict(narrow_rows=3,narrow_split=24)),("choice",{})):
        prm=eng.params(**kw)
        run=lambda: eng.filter_device(d_q,n,stride,d_len=d_len,d_ee=d_ee,d_ns=d_ns,d_pass=d_pass,params=prm,want_counts=False)
        for _ in range(30): run()
        eng.synchronize(); t=time.perf_counter()
        for _ in range(40): run()
        eng.synchronize(); ms=(time.perf_counter()-t)/40*1e3
        p=eng.last_path(); out.append("%s %.3f ms (split %d, back %d)"%(name,ms,p["narrow_split"],p["n_fallback"]))
    print("  ".join(out),flush=True)
PY
done
