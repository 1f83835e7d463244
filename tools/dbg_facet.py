import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, os
import oracle
from test_gpu_parity import _random_feature_scene, DEV, n
from conftest import rel_l2
from artist_amd import trace_rays
H,P,R,facet=100,10000,24,2500
sc=_random_feature_scene(21,H,P,R)
f32=lambda x: np.ascontiguousarray(x.detach().cpu().numpy())
dv=lambda x: x.to(DEV)
tix=sc["target_idx"]%2
res=(128,128)
both=dv(sc["both"])
w=torch.rand((H,res[1],res[0]), generator=torch.Generator().manual_seed(21)).to(DEV)
def run(ppf):
    prims={k:dv(v) for k,v in sc["prims"].items()}
    prims["corners"].requires_grad_(True)
    o,nn_=dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
    args=(o,nn_,dv(sc["incident"]),both[...,0],both[...,1],dv(tix),dv(sc["planes"]["centers"]),dv(sc["planes"]["normals"]),dv(sc["planes"]["dims"]))
    flux,fac,_=trace_rays(*args,ray_magnitude=0.7,extinction=0.05,reflectivity=0.9,resolution=res,blocking=dict(prims,lbvh_compat=False),points_per_facet=ppf)
    (flux*w).sum().backward(); torch.cuda.synchronize()
    return n(flux),n(fac),n(o.grad),n(nn_.grad)
a=run(facet); b=run(0)
sel=[0,1,2,3,17,50,99]
sub=lambda x: f32(x[sel])
oargs=(sub(sc["origins"]),sub(sc["normals"]),sub(sc["incident"]),sub(sc["both"][...,0]),sub(sc["both"][...,1]),sub(tix),f32(sc["planes"]["centers"]),f32(sc["planes"]["normals"]),f32(sc["planes"]["dims"]),res)
okw=dict(blocking=dict({k:(sub(v) if k=="owner" else f32(v)) for k,v in sc["prims"].items()},lbvh_compat=False))
of,ofac=oracle.trace_fwd(*oargs,0.7,0.05,0.9,**okw)[:2]
go,gn=oracle.trace_bwd(*oargs,f32(w[sel]),0.7,0.05,0.9,**okw)[:2]
for i,s in enumerate(sel):
    print(s,"fac",a[1][:,s],ofac[:,i],"flux",rel_l2(a[0][s],of[i]),"go facet",rel_l2(a[2][s],go[i]),"go nofacet",rel_l2(b[2][s],go[i]),"gn",rel_l2(a[3][s],gn[i]), "facet vs nofacet", rel_l2(a[2][s],b[2][s]))
    d=np.abs(a[2][s]-go[i]).sum(axis=1); k=np.argsort(-d)[:3]; print("   worst points",k,d[k],np.abs(go[i]).sum(axis=1)[k])
