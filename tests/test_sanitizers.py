"""CPU sanitizer builds (the reference has none, src/hashandcompress/Makefile:31): the oracle and the
host programs under AddressSanitizer + UBSan and under ThreadSanitizer.  The host programs are linked against
tests/stub/cwhc_stub.c -- the C ABI answered by the oracle on the CPU, test infrastructure only -- so that their worker
threads, unit queue, start barrier, per-device shards and buffers really run; with CW_STUB_DEVICES=N they drive N FAKE
devices.  (The library's own host code -- HIP streams, pinned staging, the offload thread -- needs a device and is covered by
the -m gpu tests; GPU-side sanitizers do not exist on this pool.)"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, corpus_file

STUB = os.path.join(ROOT, "tests", "stub")
B = os.path.join(STUB, "_build")
ENV = {**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1",
       "TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"}


@pytest.fixture(scope="module", autouse=True)
def build_sanitizer_binaries():
    r = subprocess.run(["make", "-C", STUB, "-j4", "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _run(exe, args, devices=1):
    r = subprocess.run([os.path.join(B, exe)] + args, capture_output=True, text=True, timeout=600, env={**ENV, "CW_STUB_DEVICES": str(devices), "CW_DRIVER_ALL_THREADS": "1"})
    assert r.returncode == 0, (exe, r.stdout[-1000:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout.strip().splitlines()


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_oracle_selftest_under_sanitizers(kind):
    lines = _run(f"{kind}_oracle_selftest", [])
    assert re.fullmatch(r"oracle selftest ok [0-9a-f]{16}", lines[-1])


def _expected(oracle, files, bs, rbf, hash_alg, comp):
    unit, out_total, fold, nblocks = bs * rbf, 0, np.uint64(0), 0
    for f in files:
        data = corpus_file(f)
        for u in range(len(data) // unit):
            for b in range(rbf):
                blk = data[u * unit + b * bs: u * unit + (b + 1) * bs]
                c = oracle.lz4_compress(blk) if comp == "lz4" else oracle.lzf_compress(blk)
                out_total += len(c) if c else bs
                d = {"skein": lambda x: oracle.skein256(x, 128), "skein512": lambda x: oracle.skein512(x, 512), "sha256mb": oracle.sha256}[hash_alg](blk)
                fold ^= np.bitwise_xor.reduce(np.frombuffer(d, dtype="<u8"))
                nblocks += 1
    return nblocks, out_total, int(fold)


@pytest.mark.parametrize("kind", ["asan", "tsan"])
@pytest.mark.parametrize("offload,devices,threads", [("true", 2, 5), ("true", 1, 2), ("false", 1, 3)])
def test_driver_threads_and_fake_devices_under_sanitizers(oracle, kind, offload, devices, threads):
    """hashandcompress.c with 2 FAKE devices: contiguous shards, one queue per device, workers round-robin over devices,
    totals summed "over the devices" -- and the same totals as the oracle computes directly."""
    files = ["kennedy.xls", "ptt5", "sum"] if offload == "true" else ["fields.c", "sum"]
    paths = [os.path.join(GOLDEN, "corpus", "canterbury", f) for f in files]
    bs, rbf, h, c = (65536, 1, "skein512", "lz4") if offload == "true" else (4096, 8, "skein", "lzf")
    lines = _run(f"{kind}_hashandcompress", ["-v", "-g", offload, "--devices", str(devices), "-c", str(threads), "-r", str(rbf), f"--block-size={bs}",
                                            "-H", h, "-C", c] + paths, devices)
    nblocks, out, fold = _expected(oracle, files, bs, rbf, h, c)
    assert re.fullmatch(rf"{h}\|{c}\|\d+\|\d+", lines[0])
    assert lines[1] == f"blocks={nblocks} in={nblocks * bs} out={out} fold={fold:016x}"
    assert lines[2] == f"devices={devices} in={nblocks * bs} out={out} (ncclAllReduce over the per-device totals)"


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_mgpu_stream_three_fake_devices_under_sanitizers(oracle, kind):
    bs, nb, G = 4096, 7, 3
    lines = _run(f"{kind}_mgpu_stream", ["--devices", str(G), "--blocks-per-gpu", str(nb), "--block-size", str(bs), "--steps", "2", "--warmup", "1",
                                        "--data", "mixed", "-H", "skein512", "-C", "lz4"], G)
    rep = json.loads(lines[-1])
    data = oracle.gen_mixed_blocks(0xC0FFEE, 0, nb * G, bs)       # rank order == block order: one contiguous stream
    out, fold = 0, np.uint64(0)
    for i in range(nb * G):
        blk = data[i * bs:(i + 1) * bs].tobytes()
        out += len(oracle.lz4_compress(blk))
        fold ^= np.bitwise_xor.reduce(np.frombuffer(oracle.skein512(blk, 512), dtype="<u8"))
    assert rep["n_gpus"] == G and rep["bytes_out"] == out and int(rep["digest_fold"], 16) == int(fold) and rep["all_devices_agree"]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_mgpu_stream_eight_fake_devices_uneven_shards_under_sanitizers(oracle, kind):
    """The shape an 8-GPU node will use (BASELINE configs[4]) with a stream the devices do not divide: 8 rank threads, shards of 4 and 5
    blocks ([g*N/8, (g+1)*N/8)), every rank's digests padded to the largest shard for the all-gather, the fold over the padded
    layout equal on every device and equal to the oracle's over the whole stream."""
    bs, total, G = 4096, 37, 8
    lines = _run(f"{kind}_mgpu_stream", ["--devices", str(G), "--blocks", str(total), "--block-size", str(bs), "--steps", "2", "--warmup", "1",
                                        "--data", "mixed", "-H", "skein512", "-C", "lz4"], G)
    rep = json.loads(lines[-1])
    data = oracle.gen_mixed_blocks(0xC0FFEE, 0, total, bs)
    out, fold = 0, np.uint64(0)
    for i in range(total):
        blk = data[i * bs:(i + 1) * bs].tobytes()
        out += len(oracle.lz4_compress(blk))
        fold ^= np.bitwise_xor.reduce(np.frombuffer(oracle.skein512(blk, 512), dtype="<u8"))
    assert rep["n_gpus"] == G and rep["blocks"] == total and rep["bytes_out"] == out
    assert int(rep["digest_fold"], 16) == int(fold) and rep["all_devices_agree"]


@pytest.mark.parametrize("kind", ["tsan"])
def test_driver_eight_fake_devices_under_tsan(oracle, kind):
    """hashandcompress --devices 8 with 9 worker threads over units the devices do not divide (TSan)."""
    files = ["kennedy.xls", "ptt5", "sum"]
    paths = [os.path.join(GOLDEN, "corpus", "canterbury", f) for f in files]
    lines = _run(f"{kind}_hashandcompress", ["-v", "-g", "true", "--devices", "8", "-c", "9", "-r", "1", "--block-size=65536", "-H", "skein512", "-C", "lz4"] + paths, 8)
    nblocks, out, fold = _expected(oracle, files, 65536, 1, "skein512", "lz4")
    assert nblocks % 8 != 0
    assert lines[1] == f"blocks={nblocks} in={nblocks * 65536} out={out} fold={fold:016x}"
    assert lines[2] == f"devices=8 in={nblocks * 65536} out={out} (ncclAllReduce over the per-device totals)"


def test_more_devices_than_present_fails_early_and_loudly():
    """--devices beyond cw_device_count(): an error message and a non-zero exit before any work."""
    for exe, args in (("asan_mgpu_stream", ["--devices", "9", "--blocks-per-gpu", "4", "--block-size", "4096"]),
                      ("asan_hashandcompress", ["-g", "true", "--devices", "9", os.path.join(GOLDEN, "corpus", "canterbury", "sum")])):
        r = subprocess.run([os.path.join(B, exe)] + args, capture_output=True, text=True, timeout=120, env={**ENV, "CW_STUB_DEVICES": "8"})
        assert r.returncode != 0 and "9" in r.stderr and ("usable" in r.stderr or "device" in r.stderr), (exe, r.stderr[-500:])


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_perf_harnesses_under_sanitizers(kind, tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    (d / "a.bin").write_bytes(corpus_file("fields.c")[:8192] + corpus_file("sum")[:8192])
    lines = _run(f"{kind}_hashing_perf", ["--verify", str(d)])
    assert any("|Skein256|" in ln for ln in lines) and any("|Sha256MB|" in ln for ln in lines) and sum(ln.startswith("verify|") for ln in lines) == 3
    lines = _run(f"{kind}_compression_perf", ["-4", "-f", str(d / "a.bin")])
    assert lines and all(ln.split("|")[0] in ("lz4", "lzf") for ln in lines if "|" in ln)
