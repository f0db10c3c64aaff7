"""The kernels' lane bodies (the headers the GPU kernels are built from, driven by tests/host_sim/diag_sim.cpp) under
AddressSanitizer + UBSan on the CPU: the grid search's lag form and lean form, the general loss kernel's three modes, the
gradient's converged-entry chunk, the scalar-chain smoother - on chunk geometries with ragged ends.
usage (the sanitizer runtimes have to be loaded first):
   LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
       python tools/host_asan/lane_bodies_asan.py"""
import os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(tempfile.gettempdir(), 'libdiag_sim_asan.so')
subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-shared', '-fPIC', '-fsanitize=address,undefined', '-fno-omit-frame-pointer',
                '-I', os.path.join(ROOT, 'eks_amd', 'csrc'), os.path.join(ROOT, 'tests', 'host_sim', 'diag_sim.cpp'), '-o', LIB], check=True)
import sys, ctypes, numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_host_sim as th
from oracle import eks_oracle as orc
_p=th._p
sim=ctypes.CDLL(LIB)
f,d=ctypes.c_float,ctypes.c_double
for (T,B0,BN,unit) in [(9001,1024,1600,1),(4000,512,1600,1),(7000,1024,512,0),(3333,1024,1056,1)]:
    K=3
    arrs,y,var,ys64,ev64=th._problem(T,K,seed=7)
    Rc=orc.constant_R_from_timevarying(orc.build_R_from_vars(ev64)); rconst=np.ascontiguousarray(Rc.reshape(-1))
    cand=np.exp(np.linspace(-8,8,64)); nll=np.zeros((K,64)); n=ctypes.c_int(0)
    sim.sim_diag_nll_lag(T,2*K,2,B0,BN,unit,_p(y,f),_p(rconst,d),_p(arrs['m0s'],d),_p(arrs['S0s'],d),_p(arrs['As'],d),_p(arrs['Cs'],d),_p(arrs['Qs'],d),_p(cand,d),64,_p(nll,d),ctypes.byref(n))
    sim.sim_diag_nll_lean(T,2*K,2,B0,BN,unit,_p(y,f),_p(rconst,d),_p(arrs['m0s'],d),_p(arrs['S0s'],d),_p(arrs['As'],d),_p(arrs['Cs'],d),_p(arrs['Qs'],d),_p(cand,d),64,_p(nll,d),ctypes.byref(n))
    for grad in (0,1,2):
        dn=np.zeros((K,64))
        sim.sim_diag_nll(T,2*K,2,BN,unit,grad,_p(y,f),_p(rconst,d),_p(arrs['m0s'],d),_p(arrs['S0s'],d),_p(arrs['As'],d),_p(arrs['Cs'],d),_p(arrs['Qs'],d),_p(cand,d),64,0,_p(nll,d),_p(dn,d))
    s=np.exp(np.linspace(-5,5,K)); n1=np.zeros(K); g1=np.zeros(K)
    sim.sim_diag_nll_conv_grad(T,2*K,2,392,unit,_p(y,f),_p(rconst,d),_p(arrs['m0s'],d),_p(arrs['S0s'],d),_p(arrs['As'],d),_p(arrs['Cs'],d),_p(arrs['Qs'],d),_p(s,d),_p(n1,d),_p(g1,d))
    ms=np.empty((T,2*K),np.float32); Vd=np.empty((T,2*K),np.float32)
    sim.sim_diag_smooth(T,2*K,2,32,unit,_p(y,f),_p(var,f),_p(arrs['m0s'],d),_p(arrs['S0s'],d),_p(arrs['As'],d),_p(arrs['Cs'],d),_p(arrs['Qs'],d),_p(s,d),_p(ms,f),_p(Vd,f))
    print('ok',T,B0,BN,unit,flush=True)
print('lane bodies clean under ASan/UBSan')
