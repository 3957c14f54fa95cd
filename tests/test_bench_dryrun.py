"""bench.py's N > 1 control flow rehearsed on CPU (gloo, two ranks, the oracle standing in for the device): process group
set-up, contiguous shards, the double-buffered asynchronous gather, the barrier + max-over-ranks timing and rank 0's line --
so that the first run on a multi-GPU node is not the first execution of that code."""
import hashlib
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_dry_run_two_gloo_ranks(oracle):
    nb, bs = 6, 4096
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--blocks-per-gpu", str(nb), "--block-bytes", str(bs), "--dry-run-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["blocks"] == 2 * nb and line["steps"] == 3
    data = oracle.gen_random_blocks(0xC0FFEE, 0, 2 * nb, bs)        # rank order == block order
    _, dig, sizes, _ = oracle.hash_and_compress(data, bs, oracle.HASH_SKEIN512, oracle.COMP_LZ4, threads=2, want_payload=False)
    assert line["bytes_out"] == int(sizes.sum())
    assert line["digests_sha256"] == hashlib.sha256(dig.tobytes()).hexdigest()


def test_bench_refuses_mismatched_world():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
