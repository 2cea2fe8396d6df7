"""GPU (-m gpu): bench.py itself as the driver runs it — fresh child processes, the JSON line on stdout — including its N > 1 control flow
(torch.distributed.run, row strips, per-rank statistics, the reductions, rank 0's line) with two ranks sharing cuda:0 and gloo standing in for
RCCL (PT_BENCH_REHEARSAL=1: RCCL refuses two ranks on one device; the record says "REHEARSAL").  RCCL with more than one rank needs more
than one GPU and is the driver's to run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, env_extra=None, launcher=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_single_gpu():
    d = _bench(["--steps", "1", "--warmup", "1", "--spp", "4"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["unit"] == "Mray/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["metric"].startswith("Mray/s at 1920x1080, 4 spp") and d["config"]["parallelism"] == "rows/1"
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "rays_per_launch", "algorithmic_bytes_per_ray"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    # achieved = algorithmic bytes per launch / launch duration, all three on the line
    assert abs(r["achieved"] - r["algorithmic_bytes_per_ray"] * r["rays_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mray/s" and c["cores"] >= 1 and "samples 0..63 of 256 (fixed)" in c["sample"]


def test_bench_two_ranks_rehearsal_matches_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (the driver's launch line) against the one-rank run of the same
    frame: same whole-job ray and path counts (the frame does not depend on how its rows are dealt), n_gpus / parallelism on the line."""
    args = ["--steps", "1", "--warmup", "0", "--spp", "4", "--no-cpu-baseline", "--no-kernel-ms"]
    one = _bench(args)
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541"]
    two = _bench(["--gpus", "2"] + args, env_extra={"PT_BENCH_REHEARSAL": "1"}, launcher=launcher)
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "rows/2" and two["scaling"] == "strong" and "REHEARSAL" in two["data"]
    assert one["n_gpus"] == 1 and one["data"] == "synthetic"
    assert two["config"]["job_per_step"] == one["config"]["job_per_step"]
    assert one["config"]["job_per_step"]["paths"] == 1920 * 1080 * 4
