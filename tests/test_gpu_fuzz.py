"""Structured fuzzing of the codecs against the oracle: synthetic blocks that exercise the match / literal machinery
far more densely than the corpora do (motif blocks with random mutations), at block sizes on both
sides of every kernel-selection threshold (LDS-staged vs global, link-chain vs table, ragged sizes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cw():
    import compute_war_amd as cw
    cw.init(0)
    return cw


def _block(rng, n):
    kind = rng.integers(0, 6)
    if kind == 0:      # motif repeated with 1/16 of the bytes mutated
        m = rng.integers(0, 256, int(rng.integers(3, 200)), dtype=np.uint8)
        b = np.resize(m, n).copy()
        idx = rng.integers(0, n, n // 16)
        b[idx] = rng.integers(0, 256, len(idx), dtype=np.uint8)
        return b
    if kind == 1:      # small alphabet: dense short matches, many hash collisions
        return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8)
    if kind == 2:      # long runs with sparse noise: long matches, length-byte chains
        b = np.full(n, rng.integers(0, 256), dtype=np.uint8)
        idx = rng.integers(0, n, max(1, n // int(rng.integers(50, 2000))))
        b[idx] = rng.integers(0, 256, len(idx), dtype=np.uint8)
        return b
    if kind == 3:      # pieces copied from earlier in the block at random distances (far and near references)
        b = rng.integers(0, 256, n, dtype=np.uint8)
        pos = int(rng.integers(8, 64))
        while pos < n:
            ln = int(rng.integers(3, 300))
            src = int(rng.integers(0, pos))
            ln = min(ln, n - pos)
            for i in range(ln):      # overlapping copies allowed
                b[pos + i] = b[src + i]
            pos += ln + int(rng.integers(0, 40))
        return b
    if kind == 4:      # text-like: words from a small dictionary
        words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) + b" " for _ in range(60)]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, len(words)))]
        return np.frombuffer(bytes(out[:n]), dtype=np.uint8)
    return rng.integers(0, 256, n, dtype=np.uint8)   # incompressible


@pytest.mark.parametrize("n,count", [(4096, 192), (65536, 24), (1000, 128), (12345, 48), (16384, 48), (16385, 32), (20000, 32),
                                     (32768, 24), (333, 128), (65535, 12)])
def test_codecs_match_oracle_on_synthetic_mix(cw, oracle, n, count):
    rng = np.random.default_rng(n * 7919 + count)
    data = np.concatenate([_block(rng, n) for _ in range(count)]).tobytes()
    for alg, comp in (("lz4", oracle.lz4_compress), ("lzf", oracle.lzf_compress)):
        sizes, payload = cw.compress_blocks(alg, data, n)
        for i in range(count):
            want = comp(data[i * n:(i + 1) * n])
            assert sizes[i] == len(want) and payload[i, :sizes[i]].tobytes() == want, (alg, n, i, int(sizes[i]), len(want))
        out, status = cw.decompress_blocks(alg, sizes, payload, n)
        for i in range(count):
            if sizes[i]:
                assert status[i] == 0 and out[i].tobytes() == data[i * n:(i + 1) * n], (alg, n, i)


def test_lane_kernels_match_oracle_on_synthetic_mix():
    """The same synthetic mix through the lane-per-block kernels, which the launch policy only uses for large batches: forced on
    by their thresholds (read once per process, hence a subprocess) -- LZ4 lanes (ring form above 4 KiB, tagged tables below,
    beside the wavefront parser), LZF lanes (on their own above 4 KiB with hand-back of blocks that do not compress; beside
    rounds of 7 blocks below), lane decoders -- every block against the oracle's bytes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, compute_war_amd as cw, oracle\n"
        "from test_gpu_fuzz import _block\n"
        "cw.init(0)\n"
        "bad = 0\n"
        "for n, count in ((4096, 300), (65536, 40), (1000, 200), (12345, 60), (16385, 40), (2048, 256), (65535, 20), (333, 150)):\n"
        "    rng = np.random.default_rng(n * 104729 + count)\n"
        "    data = np.concatenate([_block(rng, n) for _ in range(count)]).tobytes()\n"
        "    for alg, comp in (('lz4', oracle.lz4_compress), ('lzf', oracle.lzf_compress)):\n"
        "        sizes, payload = cw.compress_blocks(alg, data, n)\n"
        "        for i in range(count):\n"
        "            want = comp(data[i * n:(i + 1) * n])\n"
        "            if sizes[i] != len(want) or payload[i, :sizes[i]].tobytes() != want:\n"
        "                bad += 1; print('MISMATCH', alg, n, i, int(sizes[i]), len(want))\n"
        "        out, status = cw.decompress_blocks(alg, sizes, payload, n)\n"
        "        for i in range(count):\n"
        "            if sizes[i] and (status[i] != 0 or out[i].tobytes() != data[i * n:(i + 1) * n]):\n"
        "                bad += 1; print('DECODE', alg, n, i)\n"
        "        print('case', alg, n, count, cw.profile_kernels()['codec'])\n"
        "print('bad', bad)\n" % (root, root))
    env = {**os.environ, "CW_LZ4_LANES": "1", "CW_LZF_LANES": "1", "CW_LZF_ROUND": "7", "CW_LANES_RESERVE": "9", "CW_DECODE_LANES": "1"}
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert "bad 0" in lines, "\n".join(ln for ln in lines if not ln.startswith("case"))[:3000]
    cases = [ln for ln in lines if ln.startswith("case")]
    assert len(cases) == 16 and all("lanes" in ln for ln in cases if " 333 " not in ln), cases
