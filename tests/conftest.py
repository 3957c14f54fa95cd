"""Shared fixtures.  GPU parity tests are marked ``gpu``; everything else runs on CPU."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def seeded_block(seed, n, kind):
    """Must stay identical to tools/make_golden.py:seeded_block (inputs of skein_ref_blocks.json)."""
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "text":
        return rng.integers(97, 101, n, dtype=np.uint8).tobytes()
    if kind == "zero":
        return bytes(n)
    if kind == "pattern":
        return bytes(((i * 131 + (i >> 8)) & 0xFF) for i in range(n))
    raise ValueError(kind)


def corpus_file(name):
    with open(os.path.join(GOLDEN, "corpus", "canterbury", name), "rb") as f:
        return f.read()


def corpus_names():
    return sorted(os.listdir(os.path.join(GOLDEN, "corpus", "canterbury")))


def corpus_large_names():
    return sorted(f[:-3] for f in os.listdir(os.path.join(GOLDEN, "corpus", "canterbury-large")) if f.endswith(".xz"))


def corpus_large_file(name):
    """dataset/canterbury-large of the reference (bible.txt, world192.txt), committed xz-compressed (6.3 MB raw)."""
    import lzma
    with lzma.open(os.path.join(GOLDEN, "corpus", "canterbury-large", name + ".xz"), "rb") as f:
        return f.read()


def anchor_input(kind, n):
    if kind == "alice29":
        return corpus_file("alice29.txt")[:n]
    return seeded_block(0, n, kind)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O
