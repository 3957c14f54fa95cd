"""Pin the LZ4 / LZF oracles.  The reference holds no codec vectors (its LZ4 1.8.2 and liblzf exist
only as prebuilt archives, never run here), so the pins are the reference outputs the survey recorded
(tests/golden/survey_anchors.json) -- "parity unpinned by the reference itself" beyond those --
plus round trips through independent decoders and this repo's committed regression values."""
import ctypes
import hashlib

import numpy as np
import pytest

from conftest import anchor_input, corpus_file, corpus_large_file, corpus_large_names, corpus_names, load_golden


def test_survey_anchor_sizes_and_digests(oracle):
    for a in load_golden("survey_anchors.json")["anchors"]:
        d = anchor_input(a["input"], a["n"])
        c4, cf = oracle.lz4_compress(d), oracle.lzf_compress(d)
        assert len(c4) == a["lz4"] and len(cf) == a["lzf"], a
        if "lz4_sha256" in a:
            assert hashlib.sha256(c4).hexdigest() == a["lz4_sha256"]
            assert hashlib.sha256(cf).hexdigest() == a["lzf_sha256"]


@pytest.mark.parametrize("bs", [4096, 65536])
def test_canterbury_ratios(oracle, bs):
    want = next(r for r in load_golden("survey_anchors.json")["corpus_ratios"]
                if r["corpus"] == "canterbury" and r["block"] == bs)
    tin = t4 = tf = 0
    for name in corpus_names():
        data = corpus_file(name)
        whole = len(data) // 65536 * 65536
        for o in range(0, whole, bs):
            b = data[o:o + bs]
            c4, cf = oracle.lz4_compress(b), oracle.lzf_compress(b)
            assert oracle.lz4_decompress(c4, bs) == b
            if cf:
                assert oracle.lzf_decompress(cf, bs) == b
            tin += bs
            t4 += len(c4)
            tf += len(cf) if cf else bs
    assert round(tin / t4, 4) == want["lz4"]
    assert round(tin / tf, 4) == want["lzf"]
    tot = next(t for t in load_golden("codec_regress.json")["totals"]
               if t["corpus"] == "canterbury" and t["block"] == bs)
    assert (tin, t4, tf) == (tot["in"], tot["lz4"], tot["lzf"])


@pytest.mark.parametrize("bs", [4096, 65536])
def test_canterbury_large_ratios(oracle, bs):
    """dataset/canterbury-large (BASELINE.json configs[3]): corpus ratios of SURVEY.md 8(d) [liblz4.a / liblzf.a probe]."""
    want = next(r for r in load_golden("survey_anchors.json")["corpus_ratios"]
                if r["corpus"] == "canterbury-large" and r["block"] == bs)
    tin = t4 = tf = 0
    for name in corpus_large_names():
        data = corpus_large_file(name)
        whole = len(data) // 65536 * 65536
        for o in range(0, whole, bs):
            b = data[o:o + bs]
            c4, cf = oracle.lz4_compress(b), oracle.lzf_compress(b)
            tin += bs
            t4 += len(c4)
            tf += len(cf) if cf else bs
    assert round(tin / t4, 4) == want["lz4"]
    assert round(tin / tf, 4) == want["lzf"]
    tot = next(t for t in load_golden("codec_regress.json")["totals"]
               if t["corpus"] == "canterbury-large" and t["block"] == bs)
    assert (tin, t4, tf) == (tot["in"], tot["lz4"], tot["lzf"])


def test_regression_blocks(oracle):
    for r in load_golden("codec_regress.json")["blocks"]:
        b = corpus_file(r["file"])[r["offset"]:r["offset"] + r["n"]]
        c4, cf = oracle.lz4_compress(b), oracle.lzf_compress(b)
        assert (len(c4), hashlib.sha256(c4).hexdigest()) == (r["lz4"], r["lz4_sha256"])
        assert (len(cf), hashlib.sha256(cf).hexdigest()) == (r["lzf"], r["lzf_sha256"])


def test_random_block_is_stored_raw(oracle):
    """a uniform-random 64 KiB block compresses to exactly 65,794 B with LZ4 (one token,
    257 length bytes, 65,536 literals) and does not fit n-1 with LZF (returns 0)."""
    d = oracle.gen_random_blocks(0xC0FFEE, 5, 2, 65536).tobytes()
    for k in range(2):
        b = d[k * 65536:(k + 1) * 65536]
        c = oracle.lz4_compress(b)
        assert len(c) == load_golden("survey_anchors.json")["random_64k_lz4_size"]
        assert c[0] == 0xF0 and c[258:] == b
        assert oracle.lzf_compress(b) == b""


def _edge_inputs():
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 31, 32, 33, 34, 63, 64, 65, 255, 256,
              270, 271, 272, 273, 300, 1000, 4095, 4096, 4097, 65535, 65536, 65546):
        for alphabet in (1, 2, 3, 16, 256):
            yield rng.integers(0, alphabet, n, dtype=np.uint8).tobytes()
    # long runs / period patterns: exercise length-byte chains (255 boundaries) and LZF max_ref 264
    for period in (1, 2, 3, 4, 7, 8, 255, 256, 263, 264, 265, 8191, 8192, 8193):
        base = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
        yield (base * (70000 // period + 1))[:65536]
        yield (base * (70000 // period + 1))[:4096]


def test_round_trips_on_edge_cases(oracle):
    for d in _edge_inputs():
        c = oracle.lz4_compress(d)
        assert len(c) <= oracle.lz4_bound(len(d))
        assert oracle.lz4_decompress(c, len(d)) == d
        cf = oracle.lzf_compress(d)
        if cf:
            assert len(cf) <= len(d) - 1
            assert oracle.lzf_decompress(cf, len(d)) == d


def test_lz4_agrees_with_system_liblz4(oracle):
    """Corroboration only: the image's liblz4.so.1 is a newer release (1.9.x), not the reference's
    pinned 1.8.2; on the <64 KiB "fast" path the two are expected to emit identical bytes."""
    try:
        L = ctypes.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("no system liblz4")
    L.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    L.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    for d in list(_edge_inputs()) + [corpus_file("alice29.txt")[:65536], corpus_file("kennedy.xls")[:65536]]:
        if not d:
            continue
        cap = max(2 * len(d), len(d) + len(d) // 255 + 16)
        out = ctypes.create_string_buffer(cap)
        c = L.LZ4_compress_default(d, out, len(d), cap)
        mine = oracle.lz4_compress(d)
        assert out.raw[:c] == mine
        back = ctypes.create_string_buffer(len(d))
        assert L.LZ4_decompress_safe(mine, back, len(mine), len(d)) == len(d) and back.raw == d


def test_cpu_hashandcompress_worker_loop(oracle):
    """The oracle's multi-threaded worker phase == per-block calls (ProcessBlock, :231-261)."""
    data = np.frombuffer(corpus_file("alice29.txt")[:16 * 4096], dtype=np.uint8)
    secs, dig, sizes, payload = oracle.hash_and_compress(data, 4096, oracle.HASH_SKEIN256_128, oracle.COMP_LZ4, threads=3)
    assert secs > 0
    for i in range(16):
        b = data[i * 4096:(i + 1) * 4096].tobytes()
        assert dig[i].tobytes() == oracle.skein256(b, 128)
        c = oracle.lz4_compress(b)
        assert sizes[i] == len(c) and payload[i, :len(c)].tobytes() == c
    _, dig, sizes, payload = oracle.hash_and_compress(data, 4096, oracle.HASH_SHA256, oracle.COMP_LZF, threads=2)
    for i in range(16):
        b = data[i * 4096:(i + 1) * 4096].tobytes()
        assert dig[i].tobytes() == hashlib.sha256(b).digest()
        c = oracle.lzf_compress(b)
        assert sizes[i] == len(c) and payload[i, :len(c)].tobytes() == c


def test_probe_count_tool_emulates_the_parser(oracle):
    """tools/lz_probe_count.py (work counts quoted in DESIGN.md 4.3) is a pure-Python walk of the LZ4 parser: its output sizes
    must be the oracle's."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lz_probe_count", os.path.join(root, "tools", "lz_probe_count.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    data = corpus_file("lcet10.txt")[:3 * 65536] + corpus_file("kennedy.xls")[:65536]
    for bs in (65536, 4096):
        for i in range(0, len(data), bs * 5):
            blk = data[i:i + bs]
            probes, seqs, out = mod.lz4_counts(blk)
            assert out == len(oracle.lz4_compress(blk)) and probes >= seqs > 0
