#!/usr/bin/env python3
"""the CPU baseline's scaling over host threads on a smaller chain: ops of a stage only (flat) against ops + the inner level
(tasks), under two OpenMP environments.  usage: python tools/exp/cpu_baseline_sweep.py [nvars=300]   (one process per environment)"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, R)
    import iif_amd_loader
    iif = iif_amd_loader.load()
    from oracle import oracle_backend as ob
    ob.use_native_build(f"/tmp/liboracle_native_{os.getpid()}.so")
    from oracle.oracle_backend import OracleBackend
    nvars = int(sys.argv[2])
    fg = iif.generateChainEuclid(nvars, vardims=2, priorEvery=100, N=200)
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    iif.initAll(fg, backend=lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=32), seed=0)
    tp = iif.TreeProgram(fg, tree, seed=1)
    for nested in (False, True):
        for threads in (8, 16, 32, 64, 128):
            be = OracleBackend(200, tp.n_slots, 0, threads=threads, nested=nested)
            for v in fg.ls():
                var = fg.getVariable(v)
                be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
            prog = be.program(tp.stages)
            t0 = time.perf_counter(); prog.run(); dt = time.perf_counter() - t0
            print(f"   {'ops + inner level (tasks)' if nested else 'ops of a stage only     '} {threads:4d} threads: {tp.n_messages / dt:8.1f} messages/s ({dt:.2f} s)", flush=True)
    sys.exit(0)
nvars = sys.argv[1] if len(sys.argv) > 1 else "300"
for name, env in (("default OpenMP environment", {}), ("OMP_PROC_BIND=spread OMP_PLACES=threads OMP_WAIT_POLICY=active", {"OMP_PROC_BIND": "spread", "OMP_PLACES": "threads", "OMP_WAIT_POLICY": "active"}),
                  ("OMP_PROC_BIND=close OMP_PLACES=cores OMP_WAIT_POLICY=passive", {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores", "OMP_WAIT_POLICY": "passive"})):
    print(f"{name} ({nvars}-variable chain):", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", nvars], env=dict(os.environ, **env))
