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
