"""The multi-rank path of bench.py on one GPU: two (four) ranks share device 0, gather their
bands through gloo and host memory, and rank 0 checks the gathered frame against one
renderer doing the same samples alone (bench.py exits non-zero if a pixel or the
segment count differs).  On a multi-GPU node the same code runs with one GPU per rank
and RCCL instead."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks,steps,extra", [(2, 3, ["--spp-per-step", "8"]), (4, 2, [])])
def test_ranks_compose_the_single_gpu_frame(ranks, steps, extra):
    """two ranks rendering 8 samples per step, and four ranks with the default step, a whole frame"""
    env = dict(os.environ, RT_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # bench.py starts its own ranks when asked for more than one GPU
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--steps", str(steps), "--warmup", "1", "--workload", "c2"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "equals the single-renderer frame" in out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == ranks and rec["steps"] == steps and rec["scaling"] == "strong"
    assert rec["value"] > 0 and "cpu_baseline" not in rec
    assert "interleaved 8-row bands" in rec["config"]["partition"]
    assert rec["config"]["spp_per_step"] == (8 if extra else 64)
    # what a shortfall in scaling would be read from: per-rank kernel time, segments and fan-in time
    per_rank = rec["config"]["ranks"]
    assert per_rank["kernel_ms_per_step"]["min"] > 0 and per_rank["fan_in_ms_per_step"]["max"] > 0
    assert per_rank["segments_per_step"]["min"] <= per_rank["segments_per_step"]["max"]
    assert abs(per_rank["segments_per_step"]["mean"] * ranks - rec["config"]["segments_per_step"]) <= ranks
