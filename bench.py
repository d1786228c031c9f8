#!/usr/bin/env python
"""bench.py -- clique-messages/s of the MI355X-native nonparametric belief-propagation solve.

A "step" is one full up+down solveTree pass (every clique up-solved and down-solved once) over one BASELINE.json
configuration; the default is config 2, the configuration the metric is quoted on: ContinuousEuclid(2) 1000-variable
odometry chain with periodic priors, N = 200 particles, nested-dissection elimination order.  `--config 2p|3|4|5`
runs the other configurations at full size (3, 4, 5 are 8-GPU targets for 4 and 5: the graph grows with --gpus).
Beliefs are resident in HBM when the timed region starts (the initial beliefs are restored from a device-side
snapshot at the start of every step).

Timed region: K steps between barriers, nothing else -- no per-kernel events, the staged program replayed as a
hipGraph.  The per-kernel split, the roofline figures and the counters come from a separate profiling pass of the
same steps with HIP events around every launch (on the library's stream), which is slower and is NOT what `value` is
computed from.

Prints ONE JSON line (see the driver contract).  value = clique messages (one per tree edge and direction,
CliqueStateMachine.jl:590-593/900-903) per second over all ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (MI355X_MICROARCH.md): 256 CUs x 128 flop/clk x 2.4 GHz
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector FP32 (same table): packed FMA, 2 x 2 flop per lane and issue
# what the single-precision pair loop of the fits issues (neg_loo_ll_f32): v_sub_f32, v_mul_f32, v_add_f32 -- plain (unpacked)
# instructions, one result per lane, 16 lanes per cycle and SIMD = a quarter of the packed-FMA flop peak -- and v_exp_f32, a
# quarter-rate (transcendental) instruction: 16 cycles per wave (tools/exp/lcv_f32_proto.hip: 128 of the loop's 216 cycles per
# eight pairs are its eight v_exp_f32)
FP32_PLAIN_OPS_PER_S = FP32_VALU_PEAK_TFLOPS * 1e12 / 4
FP32_TRANS_OPS_PER_S = FP32_PLAIN_OPS_PER_S / 4


def f32_issue_seconds(ordered_pairs):
    """least time the chip needs to issue the single-precision evaluations' pair loops: three plain operations and one
    transcendental per ordered pair"""
    return ordered_pairs * (3.0 / FP32_PLAIN_OPS_PER_S + 1.0 / FP32_TRANS_OPS_PER_S)
HBM_PEAK_FALLBACK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="2", choices=["2", "2p", "3", "4", "5"], help="BASELINE.json configuration (default 2: the metric's)")
    ap.add_argument("--nvars", type=int, default=None, help="override the per-GPU size of the configuration (variables / poses / lattice rows)")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="several GPUs: strong = the configuration's graph as BASELINE.json defines it, its cliques sharded over the ranks "
                         "(auto: configs 4 and 5, which BASELINE quotes on 8 GPUs); weak = the graph grows with the ranks (auto: 2, 2p, 3)")
    ap.add_argument("--fused-min", type=int, default=None,
                    help="rounds with at least this many variable updates run as ONE launch of the fused update kernel "
                         "(NBP_FUSED_MIN; default: never -- the fused form moves a ninth of the bytes and is slower, DESIGN.md 3)")
    ap.add_argument("--pipeline-min", type=int, default=None,
                    help="rounds with at least this many variable updates run as two halves on two streams, one launch apart "
                         "(NBP_PIPELINE_MIN; same posteriors bit for bit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-vars", type=int, default=1000, help="variables of the CPU baseline's chain (default: the whole config-2 graph)")
    ap.add_argument("--no-10k", action="store_true", help="skip the secondary 10 000-variable north-star measurement")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the per-kernel profiling pass (roofline fields become null)")
    ap.add_argument("--python-host", action="store_true", help="build tree and schedule with the Python mirror instead of the native host")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: take the sharded multi-GPU code path (process group, torch-owned arena) even with one rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="testing: process-group backend.  gloo lets several ranks SHARE one GPU (RCCL refuses two ranks on one "
                         "device), separator slots then travel host-staged: every line of the sharded path but the RCCL calls")
    return ap.parse_args()


def kernel_sources_sha():
    """sha1 over the device-code sources of libnbp: what a committed PMC profile must have been taken on to be quoted"""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "incrementalinference.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".h") or (f.startswith("nbp_k_") and f.endswith(".hip")):  # device code; nbp_api.hip is host code
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def hbm_peak_gbps(device):
    """peak HBM bandwidth from the device properties (memory clock x bus width, double data rate); the datasheet
    figure of MI355X_MICROARCH.md when the runtime does not report them"""
    import torch
    try:
        p = torch.cuda.get_device_properties(device)
        clk, width = getattr(p, "memory_clock_rate", 0), getattr(p, "memory_bus_width", 0)
        gbps = 2.0 * clk * 1e3 * width / 8 / 1e9
        if 2000.0 < gbps < 20000.0:
            return gbps, f"device properties: {clk / 1e3:.0f} MHz x {width} bit x 2"
    except Exception:  # noqa: BLE001
        pass
    return HBM_PEAK_FALLBACK_GBPS, "MI355X_MICROARCH.md (the runtime reports no memory clock / bus width)"


def cpu_baseline(iif, nvars, N, thread_counts):
    """the CPU restatement (oracle/, kind="port") on the config-2 chain shape: one full up+down solve per thread count,
    OpenMP over the independent ops of a stage; the graph is initialised once.  -> [(messages/s, seconds, messages, threads)]"""
    from oracle.oracle_backend import OracleBackend
    fg = iif.generateChainEuclid(nvars, vardims=2, priorEvery=100, N=N)
    order = iif.nestedDissectionOrder(fg)
    tree = iif.buildTreeReset(fg, order)
    iif.initAll(fg, backend=lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=min(32, max(thread_counts))), seed=0)
    tp = iif.TreeProgram(fg, tree, seed=1)
    out = []
    for threads in thread_counts:
        # OpenMP over the independent ops of a stage.  (An inner level -- particles, product samples and likelihood rows of an op as
        # taskloops of the same team, `nested=True`: the serial results bit for bit -- was built and measured in round 6 and is NOT
        # used: on the 256-thread host libgomp's task queue costs more than the idle threads bring from 32 threads on,
        # profiles/r06_cpu_baseline_scaling.txt.  What caps the port at 16-32 threads is the tree: its last levels hold one to six
        # cliques, and a clique's Gibbs steps depend on each other.)
        be = OracleBackend(N, tp.n_slots, 0, threads=threads, nested=False)
        for v in fg.ls():
            var = fg.getVariable(v)
            be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
        prog = be.program(tp.stages)
        t0 = time.perf_counter()
        prog.run()
        dt = time.perf_counter() - t0
        be.close()
        out.append((tp.n_messages / dt, dt, tp.n_messages, threads))
    return out


def timed_steps(rs, steps, warmup, barrier):
    for w in range(warmup):
        rs.step(1000 + w)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        rs.step(k)
    barrier()
    return time.perf_counter() - t0


def launch_ranks_if_asked():
    """`python bench.py --gpus N` from a bare environment (no launcher: WORLD_SIZE unset) starts its own N ranks: the process
    re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on the loopback, one rank per
    GPU; rank 0 prints the one JSON line to the inherited stdout.  Under a launcher (the driver's torchrun line) nothing
    happens here: RANK / WORLD_SIZE are taken from the environment as before.  Through round 5 `--gpus` was parsed and never
    read -- the first scaling run invoked this way would have executed one rank and printed n_gpus: 1."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    gpus = 1
    argv = sys.argv[1:]
    for i, tok in enumerate(argv):
        if tok == "--gpus" and i + 1 < len(argv):
            gpus = int(argv[i + 1])
        elif tok.startswith("--gpus="):
            gpus = int(tok.split("=", 1)[1])
    if gpus <= 1:
        return
    import socket
    with socket.socket() as so:  # a free port on the loopback
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    launch_ranks_if_asked()
    # stdout carries ONE JSON line: whatever libraries print there (RCCL's version banner, for one) goes to stderr
    # (fd 1 stays redirected until the process ends: RCCL prints its banner from a destructor, after main returns)
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = os.environ.get("RANK", "0")
    try:
        _main(real_stdout)
    except BaseException as e:  # noqa: BLE001 -- every rank says what killed it before the launcher's table buries it
        if isinstance(e, SystemExit) and not e.code:
            raise
        import traceback
        for line in traceback.format_exc().rstrip().splitlines():
            sys.stderr.write(f"[bench rank {rank}] {line}\n")
        sys.stderr.flush()
        os._exit(1)  # no destructors: a rank that failed must not sit in a collective the others never reach


def _main(real_stdout):
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:  # (a launcher's world size wins; said once, on stderr)
        print(f"[bench] --gpus {a.gpus} but the launcher started {world} rank(s): running {world}", file=sys.stderr, flush=True)
    import torch
    import iif_amd_loader
    iif = iif_amd_loader.load()
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        if "RANK" not in os.environ:  # --force-dist outside a launcher: a world of one on the loopback
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if a.dist_backend == "gloo":
            local %= torch.cuda.device_count()
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    from bench_support import RankSolve, workloads
    wl = workloads(iif)[a.config]
    N = a.particles or wl.N
    size = a.nvars or wl.size
    if a.fused_min is not None:
        os.environ["NBP_FUSED_MIN"] = str(a.fused_min)
    if a.pipeline_min is not None:
        os.environ["NBP_PIPELINE_MIN"] = str(a.pipeline_min)
    scaling = a.scaling if a.scaling != "auto" else ("strong" if a.config in ("4", "5") else "weak")
    rs = RankSolve(iif, wl, size, N, rank, world, local, dist, python_host=a.python_host, scaling=scaling)
    rs.prepare()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        rs.be.synchronize()

    # ---- timed region ------------------------------------------------------------------------------------------------
    rs.be.timing_enable(False)
    dt = timed_steps(rs, a.steps, a.warmup, barrier)
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    try:
        rs.check_posteriors()
    finally:
        if world > 1 or os.environ.get("NBP_BENCH_SHA"):
            print(f"[bench rank {rank}] posterior_max_mean_err={rs.posterior_max_mean_err} mode_share={rs.posterior_mode_share} "
                  f"sha={rs.posterior_sha()} ({len(rs.mine)} variables of this rank)", file=sys.stderr, flush=True)
    msgs_total = rs.global_messages
    value = msgs_total * a.steps / dt
    st = rs.stats

    # ---- profiling pass (not timed): HIP events around every launch, counters ------------------------------------------
    prof = None
    psteps = min(a.steps, 5)
    if not a.no_profile_pass:
        rs.be.timing_enable(True)
        rs.be.timing_read()
        rs.be.diag(reset=True)
        tprof = timed_steps(rs, psteps, 0, barrier)
        tim = rs.be.timing_read()
        diag = rs.be.diag()
        rs.be.timing_enable(False)
        prof = (tim, diag, tprof)

    peak, peak_src = hbm_peak_gbps(local)
    out = {
        "metric": "clique-messages/sec", "value": value, "unit": "messages/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl.name.format(size=rs.size_total) + f", N={N} particles, nested-dissection order, full up+down solveTree",
                   "baseline_config": a.config, f"{wl.unit_name}_per_gpu": rs.size_total / world, "variables_per_gpu": len(rs.fg.ls()) // world,
                   "particles": N, "cliques": st["cliques_global"], "messages_per_step": msgs_total,
                   "variable_updates_per_step": st["updates_global"],
                   "launch": "staged program replayed as a hipGraph; per-kernel events only in the separate profiling pass",
                   "fused_update_min": int(os.environ["NBP_FUSED_MIN"]) if os.environ.get("NBP_FUSED_MIN") else None,
                   "two_stream_round_min": int(os.environ["NBP_PIPELINE_MIN"]) if os.environ.get("NBP_PIPELINE_MIN") else None,
                   "parallelism": (f"cliques sharded over {world} GPU(s), separator exchange: "
                                   f"{getattr(getattr(rs, 'impl', None), 'transport', 'none')}") if world > 1 else "single GPU"},
        # how the separator slots travelled between the ranks, and how many ranks RCCL itself counts in the library's communicator
        # (ncclCommCount through nbp_comm_info; null where no RCCL communicator exists: one rank, or the gloo test transport)
        "exchange_transport": getattr(getattr(rs, "impl", None), "transport", "none") if world > 1 else "none",
        "rccl_ranks": getattr(getattr(rs, "impl", None), "rccl_ranks", None),
        "solve_wall_s": dt / a.steps, "posterior_max_mean_err": rs.posterior_max_mean_err,
        "posterior_baseline5_band_share": getattr(rs, "posterior_baseline5_band_share", None),
        "posterior_mode_share_min_median": rs.posterior_mode_share,
        "posterior_alias_share_min_median": getattr(rs, "posterior_alias_share", None), "host_setup": rs.host_setup,
    }
    hs = out["host_setup"]
    if hs:
        tb = hs["elimination_order_s"] + hs["tree_build_s"] + hs["schedule_compile_s"]
        out["value_incl_tree_build"] = msgs_total / (dt / a.steps + tb)
        # solveTree! runs initAll! first when graphinit = true (SolverAPI.jl:368-377): the end-to-end rate of one solve
        out["value_incl_graph_init"] = msgs_total / (dt / a.steps + tb + hs["graph_init_s"])
    if prof is not None:
        tim, diag, tprof = prof
        per_step = {k: v[0] / psteps for k, v in tim.items()}
        dominant = max(per_step, key=per_step.get)
        launches = tim[dominant][1] / psteps
        avg_ms = tim[dominant][0] / max(tim[dominant][1], 1)
        # algorithmic bytes = B_upd of every variable update (this rank's), all of it charged to the
        # launches of the dominant kernel: B_upd x (updates per launch) / (average launch duration of that kernel)
        alg_rank = float(sum(st["alg_bytes"].values()))
        bytes_per_launch = alg_rank / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic cannot be counted from inside this process (PMC passes need rocprofv3 around it): the figures of the
        # committed PMC run of this very command (tools/pmc_quick.sh -> profiles/rNN_pmc_traffic.json) are quoted, for the
        # configuration they were taken on only
        traffic, traffic_src, traffic_step, traffic_ratio = None, None, None, None
        fused_on = os.environ.get("NBP_FUSED_MIN") is not None and os.environ.get("NBP_NO_FUSED_UPDATE") is None
        # (the newest round's file; it is quoted only while it was taken on the kernel sources of this build, below)
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic_fused.json" if fused_on else "r[0-9][0-9]_pmc_traffic.json")))
        pmc = cands[-1] if cands else os.path.join(ROOT, "profiles", "none")
        traffic_stale = None
        if world == 1 and a.config == "2" and size == 1000 and N == 200 and os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic_src = os.path.relpath(pmc, ROOT)
                # the counters were taken on ONE build of the kernels: quoted only while the kernel sources are the ones
                # that build had (tools/pmc_quick.sh stamps the file with kernel_sources_sha())
                traffic_stale = pj.get("kernel_sources_sha") != kernel_sources_sha()
                if not traffic_stale:
                    traffic_step, traffic_ratio = pj.get("hbm_bytes_per_step"), pj.get("traffic_over_algorithmic")
                    traffic = next(v["hbm_bytes_per_launch"] for k, v in pj["kernels"].items() if k.startswith(dominant))
            except Exception:  # noqa: BLE001
                pass
        kern_s = sum(per_step.values()) * 1e-3
        pairs = N * (N - 1) / 2
        lcv_flop = diag["lcv_evals"] * pairs * 25.0
        # (with fused rounds some of the fits run inside nbp_update_kernel: its time is counted in full, so the figure is a
        #  lower bound then)
        prep_s = (tim["nbp_prep_kernel"][0] + tim["nbp_bandwidth_kernel"][0] + tim.get("nbp_update_kernel", (0.0, 0))[0]) * 1e-3
        valu = lcv_flop / prep_s / 1e12 if prep_s > 0 else 0.0
        # the bracketing evaluations of the searches run in single precision (neg_loo_ll_f32): N(N-1) ordered pairs, 4 operations
        # each (v_sub, v_mul, v_exp_f32, v_add), priced at the rate the chip issues each kind (f32_issue_seconds); the kernel's
        # time is the sum of both kinds
        f32_flop = diag.get("lcv_evals_f32", 0) * (2 * pairs) * 4.0
        valu_mixed = (lcv_flop / (FP64_VALU_PEAK_TFLOPS * 1e12) + f32_issue_seconds(diag.get("lcv_evals_f32", 0) * 2 * pairs)) / prep_s if prep_s > 0 else 0.0
        valu_flop_peaks = (lcv_flop / (FP64_VALU_PEAK_TFLOPS * 1e12) + f32_flop / (FP32_VALU_PEAK_TFLOPS * 1e12)) / prep_s if prep_s > 0 else 0.0
        out["roofline"] = {
            "bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
            "traffic": traffic, "traffic_source": traffic_src, "traffic_source_is_stale": traffic_stale, "traffic_per_step": traffic_step,
            "traffic_over_algorithmic": traffic_ratio, "peak_source": peak_src,
            "alg_bytes_per_launch": bytes_per_launch, "alg_bytes_per_step": alg_rank, "avg_launch_ms": avg_ms,
            "launches_per_step": launches, "whole_solve_GBps": alg_rank / (dt / a.steps) / 1e9,
            "kernel_ms_per_step": per_step, "profiling_pass_ms_per_step": tprof / psteps * 1e3,
            # (measured quantities only: HBM bytes per second against the HBM peak, FP64 flop per second against the FP64 vector peak)
            "within_2x_of_a_ceiling": "neither" if (achieved / peak < 0.5 and valu / FP64_VALU_PEAK_TFLOPS < 0.5) else
                                      ("hbm" if achieved / peak >= 0.5 else "fp64_valu"),
            "note": "SURVEY 8(d) formula: B_upd of every update of the step charged to the launches of the dominant kernel.  HBM is "
                    "the roofline the north star names; the path is FP64-VALU / latency bound by construction (~13 KB algorithmic "
                    "bytes per variable update against ~1e6 FP64 kernel pairs): see roofline_valu"}
        # secondary, honest roofline: FP64 vector rate of the leave-one-out likelihood evaluations, which dominate
        # nbp_prep_kernel.  One LCV evaluation = N(N-1)/2 kernel pairs, 25 FP64 flop per pair (16 FP64 instructions, 9 of
        # them FMA: counted in the ISA of the inner loop).
        out["roofline_valu"] = {"bound": "fp64_valu", "kernel": "nbp_prep_kernel", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS,
                                # frac = achieved / peak, FP64 alone: the definition of rounds 1-4 again (round 5 had put a modelled
                                # issue time of the single-precision bracketing loops into it -- that figure is frac_issue_model now)
                                "achieved": valu, "frac": valu / FP64_VALU_PEAK_TFLOPS, "frac_fp64_only": valu / FP64_VALU_PEAK_TFLOPS,
                                "frac_issue_model": valu_mixed,
                                "frac_flop_peaks": valu_flop_peaks,  # every operation a flop against its precision's packed-FMA peak (the stricter reading)
                                "frac_note": "frac = FP64 flop of the fits' double-precision evaluations over their kernels' time, against the FP64 "
                                             "vector peak (the bracketing evaluations that run in single precision are NOT counted in it); "
                                             "frac_issue_model = the same plus the least issue time of the single-precision pair loops (three plain "
                                             "operations at a quarter of the packed-FMA peak, one v_exp_f32 at a sixteenth): a model, not a measurement",
                                "lcv_evals_per_step": diag["lcv_evals"] / psteps, "lcv_evals_f32_per_step": diag.get("lcv_evals_f32", 0) / psteps,
                                "residual_evals_per_step": diag["residual_evals"] / psteps,
                                "nonconverged_solves": diag["nonconverged"], "nan_results": diag["nan_results"]}
    else:
        out["roofline"] = {"bound": "hbm", "kernel": None, "achieved": None, "peak": peak, "unit": "GB/s", "frac": None, "traffic": None,
                           "peak_source": peak_src}
    if rank == 0 and not a.no_cpu_baseline and world == 1 and dist is None:
        # the OpenMP port parallelises over the independent ops of a stage; with hundreds of threads it is oversubscribed
        # (measured on the MI355X host: 16-32 threads is the sweet spot), so the baseline is the best of a few thread counts
        # and `cores` is the count actually used
        ncpu = os.cpu_count() or 1
        # the port is compiled for THIS host before it is timed (-O3 -march=native; the stock test library is -O2 generic);
        # -ffp-contract=off stays, so its results are the oracle's
        from oracle import oracle_backend as _ob
        # threads pinned and spread over the cores (read by the OpenMP runtime when the port's library is loaded)
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "threads")
        os.environ.setdefault("OMP_WAIT_POLICY", "active")
        flags = _ob.use_native_build(f"/tmp/liboracle_native_{os.getpid()}.so") or "-O2 (stock test library: the native build failed)"
        sweep = cpu_baseline(iif, a.cpu_sample_vars, 200, sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128)}))
        v, secs, m, threads = max(sweep)
        out["cpu_baseline"] = {"value": v, "unit": "messages/s", "cores": threads, "kind": "port", "build": flags,
                               "thread_sweep": {str(t): round(val, 1) for val, _, _, t in sweep},
                               "sample": f"the config-2 chain with {a.cpu_sample_vars} variables, N=200, one full up+down solve "
                                         f"({m} messages) in {secs:.1f} s, OpenMP over the ops of a stage, threads pinned and spread over the "
                                         f"cores; the port stops scaling at 16-32 threads because the last levels of the tree hold one to six cliques "
                                         f"whose Gibbs steps depend on each other (an inner level of parallelism was measured and is slower on this host: "
                                         f"profiles/r06_cpu_baseline_scaling.txt); best of the thread sweep on a "
                                         f"{ncpu}-thread host, every bandwidth fit made (the port has no dead-fit elimination); "
                                         f"the restatement baseline, not the Julia package (no Julia on the box)"}
        # also the single-thread rate of the same restatement (smaller sample: it is slow)
        v1, secs1, m1, _ = cpu_baseline(iif, max(40, a.cpu_sample_vars // 16), 200, [1])[0]
        out["cpu_baseline"]["single_thread"] = {"value": v1, "unit": "messages/s", "cores": 1,
                                                "sample": f"{max(40, a.cpu_sample_vars // 16)}-variable chain, {m1} messages in {secs1:.1f} s"}
        if a.config in ("2", "2p"):
            out["vs_cpu_baseline_lazy_fits"] = value / v  # (the device leaves out the fits nothing reads; like for like below)
    out["lazy_bandwidth"] = True  # NBP_OPT_LAZY_BANDWIDTH: fits whose result nothing reads are not made (same posteriors, bit for bit)
    if world == 1 and dist is None and not a.no_profile_pass:
        # the same solve with every bandwidth fit the reference makes (manikde! on every setBelief!): what the option saves
        rs.close()
        os.environ["NBP_NO_LAZY_BANDWIDTH"] = "1"
        try:
            rse = RankSolve(iif, wl, size, N, 0, 1, local, None, python_host=a.python_host, scaling=scaling)
            rse.prepare()
            dte = timed_steps(rse, min(a.steps, 5), 1, lambda: rse.be.synchronize()) / min(a.steps, 5)
            rse.be.diag(reset=True)
            rse.step(999)
            rse.be.synchronize()
            out["ms_per_step_every_fit"] = dte * 1e3
            out["value_every_fit"] = msgs_total / dte
            out["lcv_evals_per_step_every_fit"] = rse.be.diag()["lcv_evals"]
            if "cpu_baseline" in out and a.config in ("2", "2p"):
                # like for like: both sides make every bandwidth fit the reference makes (FactorGraph.jl:250-263)
                out["vs_cpu_baseline"] = out["value_every_fit"] / out["cpu_baseline"]["value"]
            rse.close()
        finally:
            os.environ.pop("NBP_NO_LAZY_BANDWIDTH", None)
    if world == 1 and dist is None and not a.no_10k and a.config == "2":
        # BASELINE.md config 2': the same chain with 10 000 variables is the graph the north-star target (>= 20x the CPU
        # baseline) is stated on; measured the same way (same --steps / --warmup), reported beside the headline value
        if not getattr(rs, "_closed", False):
            rs.close()
        rs10 = RankSolve(iif, workloads(iif)["2p"], 10000, N, 0, 1, local, None, python_host=a.python_host)
        rs10.prepare()
        dt10 = timed_steps(rs10, a.steps, a.warmup, lambda: rs10.be.synchronize()) / a.steps
        rs10.check_posteriors()
        valu10 = None
        if not a.no_profile_pass:  # FP64 vector rate of the leave-one-out evaluations over the prep kernel's time, as above
            rs10.be.timing_enable(True)
            rs10.be.timing_read()
            rs10.be.diag(reset=True)
            timed_steps(rs10, 2, 0, lambda: rs10.be.synchronize())
            tim10, diag10 = rs10.be.timing_read(), rs10.be.diag()
            rs10.be.timing_enable(False)
            prep10 = (tim10["nbp_prep_kernel"][0] + tim10["nbp_bandwidth_kernel"][0]) * 1e-3
            tf10 = diag10["lcv_evals"] * (N * (N - 1) / 2) * 25.0 / prep10 / 1e12 if prep10 > 0 else 0.0
            sf10 = f32_issue_seconds(diag10.get("lcv_evals_f32", 0) * (N * (N - 1))) / prep10 if prep10 > 0 else 0.0
            valu10 = {"bound": "fp64_valu", "kernel": "nbp_prep_kernel", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS,
                      "achieved": tf10, "frac": tf10 / FP64_VALU_PEAK_TFLOPS, "frac_issue_model": tf10 / FP64_VALU_PEAK_TFLOPS + sf10,
                      "frac_flop_peaks": tf10 / FP64_VALU_PEAK_TFLOPS + (diag10.get("lcv_evals_f32", 0) * (N * (N - 1)) * 4.0 / prep10 / 1e12 if prep10 > 0 else 0.0) / FP32_VALU_PEAK_TFLOPS,
                      "frac_fp64_only": tf10 / FP64_VALU_PEAK_TFLOPS, "lcv_evals_per_step": diag10["lcv_evals"] / 2,
                      "lcv_evals_f32_per_step": diag10.get("lcv_evals_f32", 0) / 2,
                      "kernel_ms_per_step": {k: v[0] / 2 for k, v in tim10.items()}}
        rs10.close()
        os.environ["NBP_NO_LAZY_BANDWIDTH"] = "1"
        try:  # the same graph with every fit the reference makes: the like-for-like figure against the CPU port
            rs10e = RankSolve(iif, workloads(iif)["2p"], 10000, N, 0, 1, local, None, python_host=a.python_host)
            rs10e.prepare()
            dt10e = timed_steps(rs10e, min(a.steps, 3), 1, lambda: rs10e.be.synchronize()) / min(a.steps, 3)
            rs10e.close()
        finally:
            os.environ.pop("NBP_NO_LAZY_BANDWIDTH", None)
        out["north_star_10k"] = {"workload": f"ContinuousEuclid(2) 10000-variable chain, N={N}", "value": rs10.global_messages / dt10,
                                 "unit": "messages/s", "ms_per_step": dt10 * 1e3, "messages_per_step": rs10.global_messages,
                                 "steps": a.steps, "warmup": a.warmup, "posterior_max_mean_err": rs10.posterior_max_mean_err,
                                 "graph_init_s": rs10.host_setup["graph_init_s"],
                                 "ms_per_step_every_fit": dt10e * 1e3, "value_every_fit": rs10.global_messages / dt10e,
                                 "vs_cpu_baseline": (rs10.global_messages / dt10e) / out["cpu_baseline"]["value"]
                                 if "cpu_baseline" in out else None, "target_vs_cpu_baseline": 20.0,
                                 "roofline_valu": valu10}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
