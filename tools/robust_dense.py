import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import eks_oracle as orc
from eks_amd import hip_ops, synth
from eks_amd.core import run_kalman_smoother
rng=np.random.default_rng(0)
# dense multicam-like: D=3,O=4,K=4,T=3000
T,K,D,O=3000,4,3,4
lat=np.cumsum(rng.normal(size=(K,T,D))*0.7,axis=1)
C=np.linalg.qr(rng.normal(size=(K,O,D)))[0]
ev=(0.25*rng.gamma(2,1,size=(T,K,O))).clip(1e-3)
for name,mod in (('base',None),('clip',lambda e: e.__setitem__((slice(None,None,7),slice(None),1),0.0)),
                 ('tiny',lambda e: e.__setitem__((slice(None,None,5),slice(None),slice(None)),1e-8)),
                 ('mixed',lambda e: e.__setitem__((slice(3,None,13),slice(None),2),1e9))):
    e=ev.copy()
    if mod: mod(e)
    y=np.einsum('kod,ktd->kto',C,lat)+rng.normal(size=(K,T,O))*np.sqrt(np.maximum(np.swapaxes(e,0,1),1e-12))
    y=y.astype(np.float32).astype(np.float64); e=e.astype(np.float32).astype(np.float64)
    L=rng.normal(size=(K,D,D))*0.3; Q=L@np.swapaxes(L,1,2)+0.2*np.eye(D)
    m0=np.zeros((K,D)); S0=np.tile(4*np.eye(D),(K,1,1)); A=np.tile(np.eye(D),(K,1,1))
    for sp in (10.0, 1e-3):
        s,ms,Vs=run_kalman_smoother(y,m0,S0,A,C,Q,e,smooth_param=sp)
        mo,Vo,_=orc.kalman_smoother(y,m0,S0,A,C,Q,np.full(K,sp),np.maximum(np.swapaxes(e,0,1),1e-12))
        em=(np.abs(ms-mo)/np.abs(mo).max(axis=(1,2),keepdims=True)).max(); eV=(np.abs(Vs-Vo)/np.abs(Vo).max(axis=(1,),keepdims=True)).max()
        print(name,sp,'smooth err ms',em,'Vs',eV, flush=True)
    # dense nll grid
    from eks_amd.core import _DeviceProblem
    P=_DeviceProblem(y,m0,S0,A,C,Q,e)
    rc=hip_ops.const_r(P.var,1e-4)
    cand=torch.exp(torch.linspace(-8,8,9,dtype=torch.float64,device=P.dev))
    nll,g=hip_ops.nll(P.y,rc,*P.params,cand,want_grad=False,flags=P.flags),None
    Rc=rc.cpu().numpy()
    no=np.stack([orc.filter_nll(y,m0,S0,A,C,Q,np.full(K,float(c)),Rc) for c in cand.cpu().numpy()],axis=1)
    print(name,'nll rel err',(np.abs(nll.cpu().numpy()-no)/np.abs(no)).max(), flush=True)
# scalar-chain (singlecam) path with clipped variances
T,K=3000,5
mk=synth.singlecam_markers(T,K,seed=5)
arrs=orc.singlecam_arrays(mk)
e=arrs['ensemble_vars'].copy(); e[::7,:,0]=0.0; e[3::11,:,:]=1e-9
y=arrs['ys'].astype(np.float32).astype(np.float64); e=e.astype(np.float32).astype(np.float64)
for sp in (10.0,1e-3,2980.0):
    s,ms,Vs=run_kalman_smoother(y,arrs['m0s'],arrs['S0s'],arrs['As'],arrs['Cs'],arrs['Qs'],e,smooth_param=sp)
    mo,Vo,_=orc.kalman_smoother(y,arrs['m0s'],arrs['S0s'],arrs['As'],arrs['Cs'],arrs['Qs'],np.full(K,sp),np.maximum(np.swapaxes(e,0,1),1e-12))
    print('diag clip',sp,(np.abs(ms-mo)/np.abs(mo).max(axis=(1,2),keepdims=True)).max(),(np.abs(Vs-Vo)/np.abs(Vo).max(axis=1,keepdims=True)).max(),flush=True)
# pupil final smooth with clipped variances
from eks_amd.ibl_pupil_smoother import run_pupil_kalman_smoother
ys,ev,m0,S0,lv=synth.pupil_observations(3000,seed=2)
ev[::7,3]=0.0; ev[5::11,:]*=1e-6
ev=ev.astype(np.float32).astype(np.float64)
for sp in ([0.99,0.99],[0.999,0.5]):
    s,ms,Vs=run_pupil_kalman_smoother(ys,m0,S0,orc.PUPIL_C,ev,*lv,smooth_params=sp)
    so,mo,Vo,_=orc.run_pupil_kalman_smoother(ys,m0,S0,orc.PUPIL_C,ev,lv,smooth_params=sp)
    print('pupil clip',sp,(np.abs(ms-mo)/np.abs(mo).max(axis=0)).max(),(np.abs(Vs-Vo)/np.abs(Vo).max(axis=0)).max(),flush=True)
