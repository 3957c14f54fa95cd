"""bench.py's REAL N > 1 path (run_leg with world = 2: shards of the global block range, the kernels, the double-buffered asynchronous
result gather, fences, max-over-ranks timing, rank 0's checks and line) executed on the one GPU a test box has: both ranks use cuda:0
and the gather runs over gloo (CW_BENCH_REHEARSE; RCCL refuses two ranks on one device).  The gathered digests of the two shards must
equal those of one rank over the whole range."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(r):
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_two_ranks_on_one_gpu_equal_one_rank():
    nb = 4096   # per rank: enough for the sliced Skein launches and the span scan (256 MiB of input per rank)
    common = ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-legs"]
    env = dict(os.environ, CW_BENCH_REHEARSE="gloo-one-gpu", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--blocks-per-gpu", str(nb)] + common,
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    a = _line(two)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks-per-gpu", str(2 * nb)] + common,
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    b = _line(one)
    assert a["n_gpus"] == 2 and "rehearsal" in a and "rehearsal" not in b
    assert a["gathered"]["digests"] == b["gathered"]["digests"] == 2 * nb
    assert a["gathered"]["sha256"] == b["gathered"]["sha256"]
    assert a["compression_ratio"] == b["compression_ratio"]
    assert a["parity"]["ok"] and a["roundtrip"]["ok"]
    assert a["scaling"] == "weak" and a["value"] > 0
