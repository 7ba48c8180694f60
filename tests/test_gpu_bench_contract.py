"""bench.py prints ONE JSON line with the fields the driver and the judge read; a small run must keep that contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "16",
                          "--cpu-sample", "2", "--cpu-frames", "3"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]          # whole-job rate = sequences / step time
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["parity"]["traj_rmse_vs_oracle"] <= 1e-4          # BASELINE.json's tolerance
    # the parity leg follows EVERY frame the engine stepped (warm-up + timed + the bracketed extra steps) on a few sequences
    p = d["parity"]
    assert p["frames"] >= d["steps"] + d["warmup"] and p["covers_every_timed_frame"] is True
    assert p["full_length"]["traj_rmse"] <= 1e-12 and p["full_length"]["final_state_maxabs"] <= 1e-12
    assert p["full_length"]["final_covariance_rel_fro"] <= 1e-11
    # profiling scopes are kernel symbols (they join with rocprofv3's kernel_stats.csv); the search roofline carries its
    # matrix-core floor next to the HBM fraction
    assert {"k_syrk", "k_build_AS", "k_chol_left", "k_search_mfma", "k_search_score"} <= set(d["kernels"])
    rs = d["roofline_search"]
    assert rs["kernel"] == "k_search_mfma" and rs["mfma_floor_us"] > 0 and "valu_insts_per_search" in rs


def test_bench_mapping_line_is_checked_against_the_oracle():
    """bench.py --mapping (the reference's default workload): every sampled sequence is followed over EVERY stepped frame by the
    oracle, one worker process per usable physical core (oracle/cpu_baseline.py), and the CPU baseline is that same code
    (kind "port")."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mapping", "--steps", "8", "--warmup", "8", "--batch", "16",
                          "--cpu-sample", "8"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    p, c = d["parity"], d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["cores"] <= c["cores_usable"] <= c["affinity_cpus"]
    assert "oracle" in p["checker"] and p["sequences"] == 8 and p["frames"] >= 16 and p["covers_every_timed_frame"] is True
    assert p["maps_equal"] is True and p["traj_rmse_vs_oracle"] <= 1e-11 and p["final_state_maxabs"] <= 1e-10
    assert p["per_sequence_mean"]["features_initialised"] > 0.0


def test_bench_self_spawns_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself.  On a 1-GPU box the two ranks share the device
    through the gloo test hook (the code path - rendezvous, barrier, MAX-reduce, gather - is the one RCCL runs on N GPUs)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SL2_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "16"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gathered_states"] == 32 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 16 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["cpu_baseline"] is None                      # rank 0 at N = 1 only
    # ... but every rank checks a sample of its own sequences against the oracle: values, not shapes
    assert d["parity"]["ranks_checked"] == 2 and d["parity"]["traj_rmse_vs_oracle"] <= 1e-12


def test_bench_rccl_path_single_rank():
    """The collectives of the multi-GPU bench over RCCL itself (backend "nccl"), as far as a one-GPU box allows: a launcher-style
    environment with WORLD_SIZE = 1 and SL2_BENCH_FORCE_DIST=1 creates the process group bound to the rank's device, and the
    barrier, the MAX / SUM reductions and the all-gather run on device tensors."""
    env = dict(os.environ)
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731",
               SL2_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SL2_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "16",
                          "--cpu-sample", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["gathered_states"] == 16
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]


def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`.  Two ranks on the one-GPU box through the gloo hook; stdout must carry
    exactly the one JSON record of rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SL2_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "2", "--batch", "16"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gathered_states"] == 32 and d["steps"] == 3 and d["warmup"] == 2
    assert d["parity"]["ranks_checked"] == 2 and d["parity"]["traj_rmse_vs_oracle"] <= 1e-12
    assert abs(d["value"] - 2 * 16 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
