"""The band repair of the rounding alone (kao_lp_round_host mode 2) under the sanitizers: rounded assignments of three drifted topics with
1-11 replicas moved inside their racks, repaired by the sanitized copy and by the shipped library -- same rows, no report."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tools','analysis'))
import kao_oracle as ko, kao_lp as kl
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import solver as sv
from conftest import to_product_topic
from regret import topic
lib = C.CDLL(os.environ.get('KAO_ROUND_ASAN_LIB', '/tmp/kao_round_asan/libround_asan.so'))
rng=np.random.default_rng(7)
n=bad=0
for (B,R,P,seed) in [(60,6,400,1),(100,10,1000,2),(300,10,2000,1)]:
    t=topic(B,R,P,seed=seed); pt=to_product_topic(t)
    r=kl.port_solve(t, tol=1e-8, maxit=200, primal=True, pert=kl.default_pert(t))
    A,_=kl.round_primal(t,*kl.primal_blocks(t,r['x'],r['xg']))
    rack=np.asarray(t.rack_of)
    for trial in range(40):
        A2=A.copy()
        for _ in range(int(rng.integers(1,12))):      # move a replica to another broker of the same rack that is not in the row
            p=int(rng.integers(P)); k=int(rng.integers(3)); b=int(A2[p,k])
            cands=[x for x in range(B) if rack[x]==rack[b] and x not in A2[p]]
            if cands: A2[p,k]=cands[int(rng.integers(len(cands)))]
        ref=kao.lp_repair_host(pt,A2)
        ct=sv._CTopics([pt]); a=np.ascontiguousarray(A2.reshape(-1),dtype=np.uint16).copy(); rep=(C.c_int32*4)()
        rc=lib.round_asan(ct.ptr(0), None, None, 2, a.ctypes.data_as(C.POINTER(C.c_uint16)), rep); assert rc==0
        n+=1; bad+= a.reshape(P,3).tolist()!=ref.tolist()
print('repairs',n,'different from the shipped library',bad)
