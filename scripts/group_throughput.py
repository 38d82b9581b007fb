"""Throughput of pl_ransac_batch on one bench workload:  python scripts/group_throughput.py {p3p|rel|fund|hom} <group size>
<groups in flight> <problems>  (100 000-iteration problems on device-resident correspondences, as bench.py runs them)"""
import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import poselib_amd as P
from poselib_amd import synth
from concurrent.futures import ThreadPoolExecutor
name=sys.argv[1]; G=int(sys.argv[2]); T=int(sys.argv[3]); NP=int(sys.argv[4])
kind={'p3p':0,'rel':1,'fund':2,'hom':3}[name]
n=5000 if kind<2 else 10000
if kind==0:
    d=synth.absolute_pose_scene(n,0.7,1001); a,b=(d["p2d"]-500.)/1000.,d["p3d"]; thr=12/1000.
else:
    gen={1:synth.relative_pose_scene,2:synth.fundamental_scene,3:synth.homography_scene}[kind]
    d=gen(n,0.5,1002); a,b=(d["x1"]-500.)/1000.,(d["x2"]-500.)/1000.; thr=1/1000.
probs=[P.Problem(kind,a,b) for _ in range(NP)]
opts=[{"max_error":thr,"ransac":{"max_iterations":100000,"min_iterations":100000,"seed":s}} for s in range(NP)]
rb=P.RansacBatch(probs,opts)
rb.run(T,G)
t0=time.perf_counter(); rb.run(T,G); rb.run(T,G); dt=(time.perf_counter()-t0)/2
hyp=sum(st.hypotheses for st in rb.stats())
print(f"{name} group {G} threads {T} problems {NP}: {hyp/dt:.4g} hyp/s, {dt/NP*1e3:.3f} ms/problem")
