"""The host rounding compiled with -fsanitize=address,undefined (libround_asan.so) on the iterates of the parity sweeps: same rows as the
shipped library, no sanitizer report."""
import os, sys, time, pickle, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tools','analysis'))
import kao_oracle as ko, kao_lp as kl
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import solver as sv
from conftest import to_product_topic, load_golden
from regret import topic
import test_lp_oracle as tl
lib = C.CDLL(os.environ.get('KAO_ROUND_ASAN_LIB', '/tmp/kao_round_asan/libround_asan.so'))
def asan_round(pt, q, zq, fallback=None, mode=None):
    ct = sv._CTopics([pt])
    n = pt.n_partitions * pt.rf
    q = np.ascontiguousarray(q, dtype=np.uint8); zq = np.ascontiguousarray(zq, dtype=np.int32)
    a = np.zeros(n, dtype=np.uint16) if fallback is None else np.ascontiguousarray(np.asarray(fallback).reshape(-1), dtype=np.uint16).copy()
    rep = (C.c_int32 * 4)()
    lib.round_asan.restype=C.c_int
    rc = lib.round_asan(ct.ptr(0), q.ctypes.data_as(C.POINTER(C.c_uint8)), zq.ctypes.data_as(C.POINTER(C.c_int32)), 0 if fallback is None else 1, a.ctypes.data_as(C.POINTER(C.c_uint16)), rep)
    assert rc == 0
    return a.reshape(pt.n_partitions, pt.rf), [int(x) for x in rep]
n=bad=0
def check(ot, pt, blocks, fb=None):
    global n, bad
    q,zq=tl._pack(*blocks)
    d=kao.lp_round_host(pt,q,zq,fallback=fb)
    a,rep=asan_round(pt,q,zq,fallback=fb)
    n+=1; bad += a.tolist()!=d['assignment'].tolist()
# golden families, at a vertex and five iterations in
cases = [ko.random_case_rf(c["seed"]) for c in load_golden("random_rf.json")["cases"] if c["status"] == "optimal"][::3]
cases += [ko.topic_from_dict(c["topic"]) for c in load_golden("random_medium.json")["cases"] if c["status"] == "optimal"][::3]
for t in cases:
    for maxit in (150,5):
        r=kl.port_solve(t, tol=1e-8, maxit=maxit, primal=True, pert=kl.default_pert(t))
        check(t, to_product_topic(t), kl.primal_blocks(t,r['x'],r['xg']))
print('golden', n, bad, flush=True)
# drifted topics: tight and loose tolerances, with and without fallback
for (B,R,P,seed) in [(60,6,400,1),(100,10,1000,2),(300,10,2000,1),(200,8,3000,4),(400,8,6000,2)]:
    t=topic(B,R,P,seed=seed); pt=to_product_topic(t)
    for salt in (1,2,4):
        for tol in (1e-8,1e-5,1e-4,1e-2):
            r=kl.port_solve(t, tol=tol, maxit=200, primal=True, pert=min(1e-2,100.0/(P*3)), salt=salt)
            blocks=kl.primal_blocks(t,r['x'],r['xg'])
            check(t,pt,blocks)
            check(t,pt,blocks,fb=np.tile(np.arange(3),(P,1)))
    print(B,P,n,bad,flush=True)
# the huge iterates pickled earlier
from kafka_assignment_optimizer_amd import synthetic as sy
for (B,R,P,drift,name) in [(1000,20,100_000,0.2,'/tmp/exp/blocks100k_d1.pkl'),(1000,20,100_000,0.4,'/tmp/exp/blocks_1000_100000_0.4.pkl'),(2000,40,100_000,0.2,'/tmp/exp/blocks_2000_100000_0.2.pkl')]:
    if not os.path.exists(name): continue
    pt=sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), drift, 1)[0]
    check(None, pt, pickle.load(open(name,'rb')))
    print(name, n, bad, flush=True)
print('iterates', n, 'different from the shipped library', bad)
