import sys, os; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import c_oracle
import test_gpu_kernels as tg
from eks_amd import hip_ops
T,K=40000,32
arrs,y_tk,var_tk=tg._singlecam_problem(T,K,seed=5+T,unit=True)
cand=np.exp(np.linspace(-8,8,64))
flags=hip_ops.model_flags(arrs['S0s'],arrs['As'],arrs['Cs'],arrs['Qs'])
rconst=hip_ops.const_r(tg._dev(var_tk),1e-4)
nll=hip_ops.nll(tg._dev(y_tk),rconst,*tg._params_dev(arrs),tg._dev(cand),flags=flags).cpu().numpy()
ref=c_oracle.nll_grid(arrs['ys'],rconst.cpu().numpy(),arrs['m0s'],arrs['S0s'],arrs['As'],arrs['Cs'],arrs['Qs'],cand)
err=np.abs(nll-ref)/np.abs(ref)
print(os.environ.get('EKS_NLL_EXACT_ENTRY'), 'max per candidate (first 12):', np.round(err.max(axis=0)[:12]*1e6,2), 'overall', err.max())
print('signed mean err cand0..3', ((nll-ref)/np.abs(ref)).mean(axis=0)[:4])
print('rconst range', rconst.cpu().numpy().min(), rconst.cpu().numpy().max(), 'y abs max', np.abs(y_tk).max())
