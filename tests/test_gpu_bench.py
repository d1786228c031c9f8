"""-m gpu: bench.py's contract (one JSON line, required fields) on a small chain, single process and
through torch.distributed.run with the sharded code path (RCCL process group, torch-owned arena)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _why(out):
    """what the ranks themselves said (bench.py prefixes every line of a rank's traceback), then the launcher's tail"""
    said = [l for l in out.stderr.splitlines() if "[bench rank" in l]
    return "\n".join(said[-80:]) + "\n---- launcher tail ----\n" + out.stderr[-1500:]


def _check(line):
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "clique-messages/sec" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["config"]["messages_per_step"] == 2 * (d["config"]["cliques"] - 1)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    return d


def test_bench_single_process():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--nvars", "200", "--cpu-sample-vars", "40"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, _why(out)
    d = _check(out.stdout.strip().splitlines()[-1])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    ns = d["north_star_10k"]  # BASELINE config 2': the graph the >= 20x target is stated on
    assert ns["messages_per_step"] > 19000 and ns["vs_cpu_baseline"] > 20.0


def test_bench_sharded_path_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "2",
                          "--warmup", "1", "--nvars", "200", "--force-dist", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, _why(out)
    _check(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("config,size,scaling,world", [("2", "300", "weak", 2), ("4", "6", "strong", 2), ("4", "8", "strong", 8), ("5", "400", "strong", 4)])
def test_bench_ranks_sharing_one_gpu(config, size, scaling, world):
    """the driver's multi-GPU command line with 2 / 4 / 8 ranks on a 1-GPU box: every rank opens cuda:0, the process group is gloo
    (RCCL refuses two ranks on one device) and separator slots travel host-staged.  Everything of the sharded leg but the
    RCCL calls themselves runs: partition, per-rank compile, exchange segments, barriers, max-over-ranks timing, one JSON
    line from rank 0 only."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29555 + world), "bench.py", "--gpus", str(world), "--steps", "2",
                          "--warmup", "1", "--config", config, "--nvars", size, "--dist-backend", "gloo", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, _why(out)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = _check(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == scaling and "staged" in d["config"]["parallelism"]
    if config == "2":
        assert d["config"]["variables_per_gpu"] == 300 and d["posterior_max_mean_err"] < 1.5


def test_bench_launches_its_own_ranks_from_a_bare_environment():
    """`python bench.py --gpus 2` with NO launcher around it (no RANK / WORLD_SIZE in the environment): bench.py re-executes
    itself under torch.distributed.run with two ranks and rank 0 prints the one JSON line.  Through round 5 `--gpus` was parsed
    and never read: invoked this way a scaling run executed one rank and printed n_gpus: 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--nvars", "300",
                          "--dist-backend", "gloo", "--no-cpu-baseline", "--no-10k"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, _why(out)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = _check(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["exchange_transport"] == "staged" and d["rccl_ranks"] is None  # gloo ranks sharing one GPU: no RCCL communicator
    # and the one-rank line is what it was: no launcher, n_gpus 1, no exchange
    out1 = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--nvars", "200", "--no-cpu-baseline", "--no-10k"],
                          cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out1.returncode == 0, _why(out1)
    d1 = _check(out1.stdout.strip().splitlines()[-1])
    assert d1["n_gpus"] == 1 and d1["exchange_transport"] == "none" and d1["rccl_ranks"] is None


def _shas(out):
    """{rank: posterior sha} from the `[bench rank N] ... sha=...` lines"""
    import re
    return {int(m.group(1)): m.group(2) for m in re.finditer(r"\[bench rank (\d+)\] posterior_max_mean_err=.*?sha=([0-9a-f]+)", out.stderr)}


def test_ranks_in_lock_step_are_deterministic_and_equal_to_the_sequential_fits():
    """Two ranks driving ONE device in lock step: the posteriors of a rank are the same run after run, with the speculative
    bandwidth fits and without them (round 3's red gate: the rendezvous areas of the speculative fits were blanked with
    hipMemsetAsync, which was not ordered with the fit kernel under a second process -- stale values of the previous launch
    were consumed and every run differed)"""
    def run(port, extra_env):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "2", "--nvars", "300",
                              "--dist-backend", "gloo", "--no-cpu-baseline", "--no-profile-pass"],
                             cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, _why(out)
        sh = _shas(out)
        assert set(sh) == {0, 1}, out.stderr[-2000:]
        return sh
    a, b, c = run(29571, {}), run(29572, {}), run(29573, {"NBP_NO_SPECULATIVE_FITS": "1"})
    assert a == b, (a, b)
    assert a == c, (a, c)


def test_bench_two_ranks_if_two_gpus():
    """cliques sharded over 2 GPUs with RCCL point-to-point separator exchange; skipped on 1-GPU boxes
    (the world-2 logic itself is covered on CPU by tests/test_dist_gloo.py)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29544", "bench.py", "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--nvars", "300"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, _why(out)
    d = _check([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["variables_per_gpu"] == 300
    assert d["posterior_max_mean_err"] < 1.5
