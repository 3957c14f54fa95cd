"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the header declares,
sizes/enums behave, and compute entry points fail loudly (never fall back) without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def cwlib():
    import compute_war_amd as cw
    if not os.path.exists(cw.lib_path()):
        subprocess.run(["make", "-C", os.path.join(ROOT, "compute_war_amd", "csrc"), "-j8"], check=True, capture_output=True)
    return cw


def _header_functions():
    text = open(os.path.join(ROOT, "include", "cw_hashcompress.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cw_[a-z0-9_]+)\s*\(", text)) - {"cw_offload_t"})


def test_library_exports_every_declared_symbol(cwlib):
    from compute_war_amd import _lib
    declared = _header_functions()
    assert declared == sorted(_lib.ABI_SYMBOLS), "header and binding list differ"
    out = subprocess.run(["nm", "-D", "--defined-only", cwlib.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (cw_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    L = cwlib.lib()
    for s in declared:
        assert hasattr(L, s)


def test_sizes_and_bounds(cwlib):
    assert cwlib.digest_bytes("skein512") == 64 and cwlib.digest_bytes("skein") == 16 and cwlib.digest_bytes("sha256mb") == 32
    assert cwlib.compress_bound("lz4", 65536) == 65536 + 65536 // 255 + 16 == 65809
    assert cwlib.compress_bound("lz4", 4096) == 4128 and cwlib.compress_bound("lzf", 4096) == 4096
    cwlib.set_block_size(65536)
    assert cwlib.lib().cw_get_block_size() == 65536
    cwlib.set_block_size(4096)


def test_no_gpu_means_error_not_fallback(cwlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert cwlib.lib().cw_device_count() == 0
    with pytest.raises(cwlib.CwError):
        cwlib.init(0)
    with pytest.raises(cwlib.CwError):
        cwlib.hash_blocks("skein512", np.zeros(4096, np.uint8), 4096)
    with pytest.raises(cwlib.CwError):
        cwlib.compress_blocks("lz4", np.zeros(4096, np.uint8), 4096)
    with pytest.raises(cwlib.CwError):
        cwlib.HashOffload(4, "skein", 4096)


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "compute_war_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "cw_oracle" not in text, f


def test_shard_range_partitions_index_space():
    from compute_war_amd.shard import shard_range
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for w in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1


def test_host_driver_builds_and_refuses_without_gpu(cwlib):
    import torch
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "compute_war_amd", "host")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = os.path.join(ROOT, "compute_war_amd", "host", "hashandcompress")
    bad = subprocess.run([exe, "--comp-alg=zip", "x"], capture_output=True, text=True)
    assert bad.returncode == 1 and "invalid compression algorithm" in bad.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(ROOT, "tests/golden/corpus/canterbury/alice29.txt")], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


def test_shard_range_from_c_matches_python(cwlib):
    """cw_shard_range (the C side of SURVEY 8e's contiguous g*N/G shards) against compute_war_amd.shard.shard_range,
    incl. two "fake devices": no GPU is needed for the sharding logic."""
    import ctypes as C
    from compute_war_amd.shard import shard_range
    L = cwlib.lib()
    for n in (0, 1, 7, 136, 1000, (1 << 20) + 3, 1 << 40):
        for G in (1, 2, 3, 8):
            prev = 0
            for g in range(G):
                a, b = C.c_size_t(), C.c_size_t()
                L.cw_shard_range(n, g, G, C.byref(a), C.byref(b))
                assert (a.value, b.value) == shard_range(n, g, G) and a.value == prev
                prev = b.value
            assert prev == n


def test_multi_device_entry_points_fail_cleanly_without_gpu(cwlib):
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = cwlib.lib()
    devs = (C.c_int * 2)(0, 1)
    assert not L.cw_mgpu_create(devs, 2) and b"no HIP device" in L.cw_mgpu_last_error()
    assert L.cw_set_device(1) != 0 and L.cw_get_device() == -1
    assert not L.cw_dev_alloc(16) and not L.cw_host_alloc(16)
    exe = os.path.join(ROOT, "compute_war_amd", "host", "mgpu_stream")
    subprocess.run(["make", "-C", os.path.dirname(exe)], capture_output=True, text=True, check=True)
    r = subprocess.run([exe, "--devices", "2"], capture_output=True, text=True)
    assert r.returncode == 2 and "0 usable" in r.stderr
