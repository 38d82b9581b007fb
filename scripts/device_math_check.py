"""The device kernels' scalar math against the host libm (pl_debug_device_math): cube of the LM's Nielsen update vs pow(x, 3),
sqrt, reciprocal.  Run on a GPU box."""
import sys, math, numpy as np
sys.path.insert(0,'/root/repo')
import poselib_amd as P
rs=np.random.RandomState(0)
x=np.concatenate([rs.uniform(-1,1,2000000), rs.uniform(-3,3,500000), [0.0,1.0,-1.0,0.5,1/3,2/3,1e-300,1e-160,-1e-160,1e100,-1e100,np.inf,-np.inf,np.nan]])
got=P.device_math(0,x)
want=np.array([math.pow(v,3) if np.isfinite(v) else (np.float64(v)**3) for v in x])
bad=~((got==want)|(np.isnan(got)&np.isnan(want)))
print("cube (x in [-3,3]): mismatches vs host pow", int(bad.sum()), "of", x.size, "first", x[bad][:3], got[bad][:3], want[bad][:3])
xs=np.abs(x[np.isfinite(x)])
print("sqrt mismatches", int((P.device_math(1,xs)!=np.sqrt(xs)).sum()), "recip", int((P.device_math(2,xs[xs>0])!=1.0/xs[xs>0]).sum()))
