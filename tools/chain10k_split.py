"""per-kernel split (HIP events) of the 10 000-variable north-star chain, native host compile"""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import iif_amd_loader
iif = iif_amd_loader.load()
from iif_amd import native_host
fg = iif.generateChainEuclid(10000, vardims=2, priorEvery=100, N=200)
mk = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
iif.initAll(fg, backend=mk, seed=0)
g = native_host.NativeGraph.from_fg(fg)
nt = g.build_tree(g.order_nested_dissection())
be = mk(200, nt.plan_slots(True))
prog = nt.compile(be, 1)
for v in fg.ls():
    var = fg.getVariable(v); be.slot_write(nt.snap[v], var.varType.manifold, var.val, var.bw)
prog.run(); be.synchronize()
be.timing_enable(True); be.timing_read()
t=time.perf_counter()
for k in range(3):
    prog.reseed(k); prog.run()
be.synchronize()
dt=(time.perf_counter()-t)/3
tim=be.timing_read()
print("ms/step", dt*1e3, {k:(round(v[0]/3,2), v[1]//3) for k,v in tim.items()}, nt.stats()["stages"])
