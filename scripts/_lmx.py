import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import poselib_amd as P, oracle_lib as O
from poselib_amd import synth
rs=np.random.RandomState(3)
for kind,name in ((0,'abs'),(1,'rel'),(2,'fund'),(3,'hom')):
    eq=tot=0; worst=0.0; iters_eq=0
    for trial in range(60):
        n=int(rs.randint(8,250)); outl=float(rs.uniform(0.1,0.5))
        for loss in ("TRUNCATED","CAUCHY","TRIVIAL"):
            bo={"loss_type":loss,"loss_scale":1e-3 if kind else 1.2e-2,"max_iterations":25}
            if kind==0:
                d=synth.absolute_pose_scene(n,outl,900+trial); par=d["camera"]["params"]
                x=(np.asarray(d["p2d"])-np.array(par[-2:]))/par[0]; X=np.asarray(d["p3d"])
                q=np.asarray(d["q_gt"])+1e-3*rs.randn(4); q/=np.linalg.norm(q); t=np.asarray(d["t_gt"])+1e-3*rs.randn(3)
                pr=P.Problem(0,x,X); got,it=pr.refine(P.CameraPose(q,t),bo); pr.close()
                want,st=O.bundle_adjust(x,X,{"model":"NULL" ,"width":0,"height":0,"params":[]} if False else {"model":"SIMPLE_PINHOLE","width":0,"height":0,"params":[1.0,0.0,0.0]},np.r_[q,t],bo)
                g=np.r_[got.q,got.t]; w=want
            else:
                gen={1:synth.relative_pose_scene,2:synth.fundamental_scene,3:synth.homography_scene}[kind]
                d=gen(n,outl,900+trial); x1=(np.asarray(d["x1"])-500.)/1000.; x2=(np.asarray(d["x2"])-500.)/1000.
                pr=P.Problem(kind,x1,x2)
                if kind==1:
                    q=np.asarray(d["q_gt"])+1e-3*rs.randn(4); q/=np.linalg.norm(q); t=np.asarray(d["t_gt"])+1e-3*rs.randn(3)
                    got,it=pr.refine(P.CameraPose(q,t),bo); want,st=O.refine("relpose",x1,x2,np.r_[q,t],bo); g=np.r_[got.q,got.t]; w=want
                else:
                    if kind==2:
                        m,info=P.ransac_fundamental(x1,x2,{"max_error":1e-3,"ransac":{"seed":trial,"max_iterations":200}})
                    else:
                        m,info=P.ransac_homography(x1,x2,{"max_error":1e-3,"ransac":{"seed":trial,"max_iterations":200}})
                    M=m+1e-4*np.abs(m).max()*rs.randn(3,3)
                    got,it=pr.refine(M,bo); want,st=O.refine("fundamental" if kind==2 else "homography",x1,x2,M,bo); g=np.ravel(got); w=np.ravel(want)
                pr.close()
            tot+=1; same=bool((g==w).all()); eq+=same; iters_eq+=int(it==st.iterations)
            worst=max(worst,float(np.abs(g-w).max()))
    print(f"{name}: {eq}/{tot} refined models bit-identical to the oracle's, iterations equal {iters_eq}/{tot}, worst |diff| {worst:.2e}")
