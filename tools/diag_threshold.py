"""Diagnostic (GPU box): the at-threshold cases of tests/test_gpu_filter_adversarial.py one by one."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_gpu_filter_adversarial as t
from oracle import pyoracle
from mjpl_amd import engine as eng_mod
for ta,tb in t.PAIRS[:3]:
    rng=np.random.default_rng(t._seed(ta,tb))
    model, make_q, lo, hi = t._two_pod_case(rng, ta, tb, 0.06, (0.3,-0.2,0.5), 96)
    orc=pyoracle.Oracle(model)
    qv,qc=t._bisect(orc, make_q, lo, hi)
    rows=[]; offs=[]
    for off in t.OFFSETS:
        for base,sgn in ((qv,-1.0),(qc,1.0)):
            rows.append(base+sgn*off*np.sign(hi-lo)); offs.append(np.full(len(base), sgn*off))
    x=np.concatenate(rows); offs=np.concatenate(offs)
    Q=make_q(x, reps=len(rows))
    want=orc.valid_configs(Q,nthreads=8)
    e=eng_mod.Engine(model)
    got=e.check_configs(Q)
    e.set_filter(False)
    got64=e.check_configs(Q)
    bad=np.flatnonzero(got!=want); bad64=np.flatnonzero(got64!=want)
    print(ta,tb,'filter mismatches at offsets',offs[bad],' f64 mismatches at', offs[bad64], e.info()['filter_err_a'], e.info()['filter_max_coord'])
