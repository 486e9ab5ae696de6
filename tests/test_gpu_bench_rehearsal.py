"""The multi-rank path of bench.py on one GPU: two ranks share device 0, gather their
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


def test_two_ranks_compose_the_single_gpu_frame():
    env = dict(os.environ, RT_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # bench.py starts its own ranks when asked for more than one GPU
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "c2", "--spp-per-step", "8"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "equals the single-renderer frame" in out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "strong"
    assert rec["value"] > 0 and "cpu_baseline" not in rec
    assert "interleaved 8-row bands" in rec["config"]["partition"]
