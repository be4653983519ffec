import ctypes as C, os, sys
import numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0]=[ROOT, os.path.join(ROOT,"pigeons.jl_amd")]
from pigeons_amd import _lib
_lib.LIB_PATH=os.path.join(ROOT,"build_variants","libpte_waves.so")
import pigeons_amd as P
from pigeons_amd.pt import reduce_recorders, adapt
for N in (512, 256, 768, 1024):
    pt=P.PT(P.Inputs(target=P.toy_mvn_target(1024), n_chains=N, n_rounds=10, explorer=P.SliceSampler(), show_report=False, record=[P.round_trip, P.log_sum_ratio]), debug_kernel=_lib.KERNEL_TWO_LAUNCHES)
    e=pt.replicas; e.run_scans(1,8); adapt(pt, reduce_recorders(pt))
    L=_lib.load(); L.pte_debug_wave_profile.argtypes=[C.c_void_p, C.POINTER(C.c_double)]
    e.run_scans(1,1)
    out=np.zeros(4*N); assert L.pte_debug_wave_profile(e.h, out.ctypes.data_as(C.POINTER(C.c_double)))==0
    o=out.reshape(N,4)[1:]
    hw=o[:,2].astype(np.int64); xcc=o[:,3].astype(np.int64)&15
    simd=(hw>>4)&3; cu=(hw>>8)&15; se=(hw>>13)&7
    key=((xcc*8+se)*16+cu)*4+simd
    cnt=np.bincount(key, minlength=8*8*16*4)
    dur=(o[:,1]-o[:,0])/100.0
    shared=cnt[key]>1
    # which SIMDs of a CU hold the waves, and how long such waves run
    cukey=key//4
    pats={}
    for ck in np.unique(cukey):
        sel=cukey==ck
        pat=tuple(sorted(simd[sel].tolist()))
        pats.setdefault(pat,[]).append(dur[sel].mean())
    print("N=%d SIMD patterns per CU: %s" % (N, {k:(len(v), round(float(np.mean(v)),1)) for k,v in sorted(pats.items(), key=lambda kv:-len(kv[1]))[:8]}))
    print("N=%d: waves per SIMD histogram (SIMDs with 1, 2, 3.. waves): %s; mean duration of waves alone on their SIMD %.1f us, sharing a SIMD %.1f us (%d waves); per-CU wave counts %s"%(N, np.bincount(cnt[cnt>0])[1:], dur[~shared].mean(), dur[shared].mean() if shared.any() else float('nan'), shared.sum(), np.bincount(np.bincount(key//4)[np.bincount(key//4)>0])[1:]))
