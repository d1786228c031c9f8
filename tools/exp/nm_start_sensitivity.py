import ctypes as C, numpy as np, sys
sys.path.insert(0,'/root/repo')
lib=C.CDLL('/root/repo/oracle/liboracle.so')
class Obj(C.Structure):
    _fields_=[("kind",C.c_int),("manifold",C.c_int),("D",C.c_int),("solve_b",C.c_int),("z",C.c_double*3),("other",C.c_double*3),("rmask",C.c_int)]
import iif_amd_loader; iif=iif_amd_loader.load(); abi=iif.abi
lib.orc_nelder_mead.argtypes=[C.POINTER(Obj),C.c_int,C.POINTER(C.c_double),C.POINTER(C.c_int)]
rng=np.random.default_rng(0)
for n,man in ((2,abi.EUCLID2),(3,abi.EUCLID3)):
    amps=[]; its=[]
    for t in range(2000):
        o=Obj(); o.kind=abi.F_LINREL; o.manifold=man; o.D=n; o.solve_b=1; o.rmask=0
        z=rng.normal(size=3)+1; oth=rng.normal(size=3)
        for k in range(3): o.z[k]=z[k]; o.other[k]=oth[k]
        x0=rng.normal(size=3)*0.5+ (oth+z)  # start near the root (+-0.5)
        xa=(C.c_double*3)(*x0); it=C.c_int()
        lib.orc_nelder_mead(C.byref(o),n,xa,C.byref(it))
        x1=x0.copy(); x1[0]=np.nextafter(x1[0],10)  # 1 ulp
        xb=(C.c_double*3)(*x1)
        lib.orc_nelder_mead(C.byref(o),n,xb,None)
        d=max(abs(xa[k]-xb[k]) for k in range(n))
        amps.append(d); its.append(it.value)
    amps=np.array(amps)
    print(n,"D: iterations median",np.median(its),"| output diff for a 1-ulp start perturbation: quantiles 50/90/99/max",np.quantile(amps,[.5,.9,.99,1.0]), "share > 1e-12:",(amps>1e-12).mean())
