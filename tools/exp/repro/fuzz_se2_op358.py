# the proposal the fuzz found (tools/exp/fuzz_proposals.py seed 19, op 358 of 400 plain SE(2) proposals, N = 300), alone, by number of inflation cycles
# usage (GPU box): GPU=1 python tools/exp/repro/fuzz_se2_op358.py
import os
os.environ['FUZZ_SHORT'] = '0'  # the generator's stream of the run that found it
import sys, os
sys.argv=['x']
src=open('/root/repo/tests/fuzz_proposals.py').read().replace("\nmain()\n","\n")
g={'__file__':'/root/repo/tests/fuzz_proposals.py','__name__':'fz'}
exec(compile(src,'fz','exec'),g)
import numpy as np
abi=g['abi']; iif=g['iif']; OracleBackend=g['OracleBackend']
seed, N, B, which = 19400, 300, 400, g['KINDS'][4]
rng=np.random.default_rng(seed)
descs=[];writes={};s=0
for j in range(B):
    name,kind,man=which
    ins=list(range(s,s+4)); d,used=g['make_case'](rng,name,kind,man,ins,s+4,j*N,N,True)
    for k in range(used): writes[ins[k]]=(man,g['belief'](rng,man,N))
    descs.append(d); s+=5
j=358
d=descs[j]
A=writes[d.var_slot[0]][1]; Bp=writes[d.var_slot[1]][1]
def th(p): return np.arctan2(p[:,3],p[:,2]) if p.shape[1]==6 else p[:,2]
print("desc sfidx",d.sfidx,"cycles",d.inflate_cycles,"inflation",d.inflation,"comp",[d.comp[0][i] for i in range(13)])
for nm,p in (("A",A),("B",Bp)):
    t=th(p); print(nm,p.shape,"xy mean",p[:,:2].mean(0),"xy std",p[:,:2].std(0),"theta range",t.min(),t.max(),"theta std",t.std())
np.save('/tmp/op358_A.npy',A); np.save('/tmp/op358_B.npy',Bp)
if len(sys.argv)>1 or os.environ.get("GPU"):
    for cyc in (1,2,3):
        outs=[]
        for make in (lambda: OracleBackend(N,4,N,threads=4), lambda: iif.HipBackend(N,4,side_ints=N)):
            be=make()
            be.slot_write(0,abi.SE2,A); be.slot_write(1,abi.SE2,Bp)
            import copy
            dd=g['relative_factor_desc'](abi.F_SE2,abi.SE2,2,d.sfidx,[0,1],2,d.seed,[d.comp[0][1],d.comp[0][2],d.comp[0][3]],[d.comp[0][4],d.comp[0][8],d.comp[0][12]],cycles=cyc,inflation=d.inflation,mhidx_out=0)
            be.run_proposals([dd]); outs.append(be.slot_read(2,abi.EUCLID3)); be.close()
        dp=np.abs(outs[0][0]-outs[1][0])
        print("cycles",cyc,"particles differing",int((dp>0).any(axis=1).sum()),"max",dp.max(),"bw diff",np.abs(np.asarray(outs[0][1])-np.asarray(outs[1][1])).max())
