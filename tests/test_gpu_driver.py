"""The C host driver (compute_war_amd/host/hashandcompress.c, the reference's main() re-stated over the C ABI)
and the hashing_perf harness, run as subprocesses on the GPU box and checked against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, corpus_file

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "compute_war_amd", "host", "hashandcompress")
PERF = os.path.join(ROOT, "compute_war_amd", "host", "hashing_perf")


@pytest.fixture(scope="module", autouse=True)
def build_host():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "compute_war_amd", "host")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _expected(oracle, files, bs, rbf, hash_alg, comp):
    unit = bs * rbf
    out_total, fold, nblocks = 0, np.uint64(0), 0
    for f in files:
        data = corpus_file(f)
        for u in range(len(data) // unit):          # a partial last read unit is dropped (HashAndCompress.cpp:207-210)
            for b in range(rbf):
                blk = data[u * unit + b * bs: u * unit + (b + 1) * bs]
                c = oracle.lz4_compress(blk) if comp == "lz4" else oracle.lzf_compress(blk)
                out_total += len(c) if c else bs
                d = {"skein": lambda x: oracle.skein256(x, 128), "skein512": lambda x: oracle.skein512(x, 512),
                     "sha256mb": oracle.sha256}[hash_alg](blk)
                fold ^= np.bitwise_xor.reduce(np.frombuffer(d, dtype="<u8"))
                nblocks += 1
    return nblocks, out_total, int(fold)


@pytest.mark.parametrize("offload", ["true", "false"])
@pytest.mark.parametrize("hash_alg,comp,bs,rbf", [("skein", "lz4", 4096, 1), ("sha256mb", "lzf", 4096, 8),
                                                  ("skein512", "lz4", 65536, 1)])
def test_driver_report_and_totals(oracle, offload, hash_alg, comp, bs, rbf):
    files = ["alice29.txt", "fields.c", "sum"] if offload == "false" else ["alice29.txt", "kennedy.xls", "ptt5", "sum"]
    paths = [os.path.join(GOLDEN, "corpus", "canterbury", f) for f in files]
    r = subprocess.run([EXE, "-v", f"--gpu-offload={offload}", "--c-threads=3", f"--read-blocks={rbf}", f"--block-size={bs}",
                        f"--hash-alg={hash_alg}", f"--comp-alg={comp}"] + paths, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    # the reference's report line: hash|comp|totalTimeMS|throughputMBPS (HashAndCompress.cpp:409-412)
    assert re.fullmatch(rf"{hash_alg}\|{comp}\|\d+\|\d+", lines[0]), lines[0]
    m = re.fullmatch(r"blocks=(\d+) in=(\d+) out=(\d+) fold=([0-9a-f]{16})", lines[1])
    nblocks, out_total, fold = _expected(oracle, files, bs, rbf, hash_alg, comp)
    assert (int(m[1]), int(m[2]), int(m[3]), int(m[4], 16)) == (nblocks, nblocks * bs, out_total, fold)


def test_driver_usage_errors():
    r = subprocess.run([EXE, "--hash-alg=md5", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "invalid hashing algorithm" in r.stderr


def _fold(digests: bytes) -> int:
    import numpy as np
    return int(np.bitwise_xor.reduce(np.frombuffer(digests, dtype="<u8"))) if digests else 0


def test_hashing_perf_log_format(tmp_path, oracle):
    d = tmp_path / "data"
    d.mkdir()
    data = corpus_file("alice29.txt")[:5 * 4096 + 100]
    (d / "a.bin").write_bytes(data)
    r = subprocess.run([PERF, "--verify", str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    # --verify: the XOR fold of every digest the harness computed, against the oracle's digests of the same blocks
    blocks = [data[i * 4096:(i + 1) * 4096] for i in range(5)]
    sk = b"".join(oracle.skein256(b, 128) for b in blocks)
    sha = b"".join(oracle.sha256(b) for b in blocks)
    mb = b"".join(b"".join(oracle.sha256(b) for b in blocks[: 5 // w * w]) for w in range(1, 65))
    ver = {l.split("|")[1]: l.split("|")[2:] for l in lines if l.startswith("verify|")}
    assert ver["Skein256"] == ["5", "%016x" % _fold(sk)]
    assert ver["Sha256"] == ["5", "%016x" % _fold(sha)]
    assert ver["Sha256MB"] == [str(sum(5 // w * w for w in range(1, 65))), "%016x" % _fold(mb)]
    lines = [l for l in lines if not l.startswith("verify|")]
    sb = [l for l in lines if "|Skein256|" in l or "|Sha256|" in l]
    mb = [l for l in lines if "|Sha256MB|" in l]
    assert len(sb) == 10                                   # 5 whole blocks x 2 algorithms (test.cpp:19-23)
    assert [l.split("|")[1] for l in sb] == [str(i) for i in range(10)]
    assert len(mb) == sum(5 // w for w in range(1, 65))    # windows 1..64 (test.cpp:87-90)
    assert all(re.fullmatch(r".+\|\d+\|Sha256MB\|\d+\|\d+\|", l) for l in mb)


COMP_PERF = os.path.join(ROOT, "compute_war_amd", "host", "compression_perf")


@pytest.mark.parametrize("mode", [[], ["--batch"]])
def test_compression_perf_lines_match_oracle(oracle, tmp_path, mode):
    """N3: alg|csize|comp_us|decomp_us|file|block for every full 4 KiB block (experiment.cpp:105-125,243-267)."""
    d = tmp_path / "data"
    d.mkdir()
    files = {"a.txt": corpus_file("alice29.txt")[:7 * 4096 + 5], "b.bin": os.urandom(2 * 4096), "c.xls": corpus_file("kennedy.xls")[:3 * 4096]}
    for k, v in files.items():
        (d / k).write_bytes(v)
    r = subprocess.run([COMP_PERF, "-f", "-4"] + mode + [str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [l.split("|") for l in r.stdout.strip().splitlines()]
    want = []
    for name in sorted(files):
        data = files[name]
        for i in range(len(data) // 4096):
            b = data[i * 4096:(i + 1) * 4096]
            want.append(("lzf", len(oracle.lzf_compress(b)), str(d / name), i + 1))
            want.append(("lz4", len(oracle.lz4_compress(b)), str(d / name), i + 1))
    assert [(g[0], int(g[1]), g[4], int(g[5])) for g in got] == want
    assert all(g[2].isdigit() and g[3].isdigit() and len(g) == 6 for g in got)
    # --best: one line per block, lz4 only when strictly smaller than lzf's result (experiment.cpp:115,261,507)
    r = subprocess.run([COMP_PERF, "-f", "-4", "-B"] + mode + [str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    best = [l.split("|") for l in r.stdout.strip().splitlines()]
    exp = [(f if f[1] <= z[1] else z) for f, z in zip(want[0::2], want[1::2])]
    assert [(g[0], int(g[1]), g[4], int(g[5])) for g in best] == exp


def test_compression_perf_refuses_other_codecs():
    r = subprocess.run([COMP_PERF, "--gzip", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "not part of this build" in r.stderr
