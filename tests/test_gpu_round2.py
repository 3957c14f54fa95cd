"""Round-2 additions, on the GPU: the pipelined host batch path (slots and packed stream, pinned and pageable buffers,
several chunks in flight), the compressible synthetic mix, two host threads on ONE stream, HashOffload's failure state,
digests at any alignment, decoders fed impossible sizes, the alternative LZ4 parser, per-device contexts and the C
multi-GPU harness with the one device the box has."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, corpus_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cw():
    import compute_war_amd as cw
    cw.init(0)
    return cw


def _mixed_corpus(nbytes):
    t = corpus_file("lcet10.txt") + corpus_file("kennedy.xls")[:300000] + bytes(70000) + corpus_file("ptt5")[:200000]
    noise = np.random.default_rng(7).integers(0, 256, 200000, dtype=np.uint8).tobytes()
    t = t + noise
    return (t * (nbytes // len(t) + 1))[:nbytes]


def _check_packed(oracle, data, bs, hash_alg, comp, digests, sizes, offsets, packed):
    hfn = {"skein512": lambda b: oracle.skein512(b, 512), "skein": lambda b: oracle.skein256(b, 128), "sha256mb": oracle.sha256}[hash_alg]
    cfn = oracle.lz4_compress if comp == "lz4" else oracle.lzf_compress
    n = len(data) // bs
    assert offsets[0] == 0 and len(offsets) == n + 1
    for i in range(n):
        b = data[i * bs:(i + 1) * bs]
        want = cfn(b)
        assert sizes[i] == len(want), (i, sizes[i], len(want))
        assert offsets[i + 1] - offsets[i] == len(want)
        assert packed[int(offsets[i]):int(offsets[i + 1])].tobytes() == want, i
        assert digests[i].tobytes() == hfn(b), i
    assert int(offsets[n]) == len(packed)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("hash_alg,comp,bs", [("skein512", "lz4", 65536), ("sha256mb", "lzf", 4096), ("skein", "lz4", 4096)])
def test_host_pipeline_packed_matches_oracle(cw, oracle, hash_alg, comp, bs, pinned):
    data = _mixed_corpus(3 * 1024 * 1024 + (bs if bs == 4096 else 0))
    data = data[: len(data) // bs * bs]
    d, sizes, offsets, packed = cw.hash_and_compress_packed(hash_alg, comp, data, bs, pinned=pinned)
    _check_packed(oracle, data, bs, hash_alg, comp, d, sizes, offsets, packed)


def test_host_pipeline_many_small_chunks_subprocess(oracle):
    """CW_HOST_CHUNK_MB=1 (read once per process): a 9 MiB batch becomes 9 chunks through the three slots -- slot reuse,
    the packed stream's running offset and the slot form's scatter are all exercised; results against the oracle."""
    prog = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, hashlib, compute_war_amd as cw\n"
        "from test_gpu_round2 import _mixed_corpus\n"
        "cw.init(0)\n"
        "data = _mixed_corpus(9 * 1024 * 1024)\n"
        "for comp, bs in (('lz4', 65536), ('lzf', 4096)):\n"
        "    d, sizes, offsets, packed = cw.hash_and_compress_packed('skein512', comp, data, bs)\n"
        "    d2, s2, payload = cw.hash_and_compress_blocks('skein512', comp, data, bs)\n"
        "    assert np.array_equal(d, d2) and np.array_equal(sizes, s2)\n"
        "    for i in range(len(sizes)): assert payload[i, :sizes[i]].tobytes() == packed[int(offsets[i]):int(offsets[i+1])].tobytes(), i\n"
        "    print(comp, bs, hashlib.sha256(d.tobytes() + sizes.tobytes() + packed.tobytes()).hexdigest())\n" % (ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env={**os.environ, "CW_HOST_CHUNK_MB": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    import hashlib
    data = _mixed_corpus(9 * 1024 * 1024)
    lines = r.stdout.strip().splitlines()
    for line, (comp, bs) in zip(lines, (("lz4", 65536), ("lzf", 4096))):
        cfn = oracle.lz4_compress if comp == "lz4" else oracle.lzf_compress
        h = hashlib.sha256()
        n = len(data) // bs
        blocks = [data[i * bs:(i + 1) * bs] for i in range(n)]
        outs = [cfn(b) for b in blocks]
        h.update(b"".join(oracle.skein512(b, 512) for b in blocks))
        h.update(np.array([len(o) for o in outs], dtype=np.uint32).tobytes())
        h.update(b"".join(outs))
        assert line.split()[-1] == h.hexdigest(), (comp, bs)


def test_host_pipeline_growing_chunks_subprocess(oracle):
    """Chunks of compressible blocks > 4 KiB grow once the first results are in (page-locked buffers only).  With CW_HOST_CHUNK_MB=1 and
    CW_HOST_BIG_CHUNK_MB=3 a 17 MiB batch of corpus text goes 1, 1, 3, 3, 3, 3, 3 MiB; one that starts with 4 MiB of noise keeps 1 MiB
    chunks until the text arrives (LZF: its raw-stored blocks count as not compressed).  Results against the oracle."""
    prog = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, hashlib, compute_war_amd as cw\n"
        "from conftest import corpus_file\n"
        "cw.init(0)\n"
        "text = (corpus_file('lcet10.txt') + corpus_file('kennedy.xls') + corpus_file('ptt5')) * 12\n"
        "noise = np.random.default_rng(3).integers(0, 256, 4 << 20, dtype=np.uint8).tobytes()\n"
        "for name, data in (('text', text[:17 << 20]), ('noise_then_text', noise + text[:9 << 20]), ('odd_tail', text[:(7 << 20) + 65536 * 5])):\n"
        "    for comp, bs in (('lz4', 65536), ('lzf', 16384)):\n"
        "        d, sizes, offsets, packed = cw.hash_and_compress_packed('skein512', comp, data, bs, pinned=True)\n"
        "        print(name, comp, bs, hashlib.sha256(d.tobytes() + sizes.tobytes() + packed.tobytes()).hexdigest())\n" % (ROOT, ROOT))
    outs = []
    for env in ({"CW_HOST_CHUNK_MB": "1", "CW_HOST_BIG_CHUNK_MB": "3", "CW_DEBUG_HOST": "1"}, {"CW_HOST_CHUNK_MB": "1"}, {}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(r.stdout.strip().splitlines())
        if "CW_DEBUG_HOST" in env:
            chunks = [int(ln.split("=")[1].split()[0]) for ln in r.stderr.splitlines() if ln.startswith("cw host pipeline: chunk")]
            # first call: 17 MiB of text at 64 KiB: 16, 16, then 48-block chunks (the last two halve what is left)
            assert chunks[:4] == [16, 16, 48, 48] and sum(chunks[:7]) == 272, chunks[:12]
            assert 192 in chunks   # LZF at 16 KiB: 3 MiB chunks too
            # noise first: the 1 MiB chunks stay while what has been seen does not compress (a run of small chunks that text alone never shows)
            assert "16,16,16,16,16" in ",".join(map(str, chunks)), chunks
    assert len(outs[0]) == 6 and outs[0] == outs[1] == outs[2], outs
    import hashlib
    from conftest import corpus_file
    text = (corpus_file("lcet10.txt") + corpus_file("kennedy.xls") + corpus_file("ptt5")) * 12
    data = text[: (7 << 20) + 65536 * 5]   # the smallest case against the oracle itself
    for line, (comp, bs) in zip(outs[0][4:], (("lz4", 65536), ("lzf", 16384))):
        cfn = oracle.lz4_compress if comp == "lz4" else oracle.lzf_compress
        blocks = [data[i * bs:(i + 1) * bs] for i in range(len(data) // bs)]
        cs = [cfn(b) for b in blocks]
        h = hashlib.sha256(b"".join(oracle.skein512(b, 512) for b in blocks))
        h.update(np.array([len(o) for o in cs], dtype=np.uint32).tobytes())
        h.update(b"".join(cs))
        assert line.split()[-1] == h.hexdigest(), (comp, bs)


def test_gen_mixed_matches_host_twin_and_compresses(cw, oracle):
    import torch
    s = torch.cuda.current_stream().cuda_stream
    for bs, nb, first in ((65536, 10, 0), (4096, 33, 5)):
        buf = torch.zeros(nb * bs, dtype=torch.uint8, device="cuda")
        cw.dev_gen_mixed(0xC0FFEE, first, nb, bs, buf.data_ptr(), s)
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        assert np.array_equal(got, oracle.gen_mixed_blocks(0xC0FFEE, first, nb, bs))
        # even blocks are the random stream, odd ones compress (motif + 1/16 mutations)
        assert np.array_equal(got[:bs] if first % 2 == 0 else got[bs:2 * bs],
                              oracle.gen_random_blocks(0xC0FFEE, first + first % 2, 1, bs))
        sizes, payload = cw.compress_blocks("lz4", got, bs)
        for i in range(nb):
            want = oracle.lz4_compress(got[i * bs:(i + 1) * bs].tobytes())
            assert sizes[i] == len(want) and payload[i, :sizes[i]].tobytes() == want
            assert (sizes[i] < bs // 2) == ((first + i) % 2 == 1)


def test_two_host_threads_on_one_stream(cw, oracle):
    """scratch is keyed by stream; two threads calling the fused entry point on the SAME stream must not
    interleave their launch sequences (per-(device, stream) launch mutex)."""
    import torch
    bs, nb = 65536, 24
    stream = torch.cuda.Stream()
    s = stream.cuda_stream
    stride = (cw.compress_bound("lz4", bs) + 15) // 16 * 16
    datas = [(_mixed_corpus(bs * nb + 977 * t))[977 * t: 977 * t + bs * nb] for t in range(2)]
    bufs = []
    for d in datas:
        bufs.append(dict(src=torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda(), dig=torch.zeros(nb * 64, dtype=torch.uint8, device="cuda"),
                         dst=torch.zeros(nb * stride, dtype=torch.uint8, device="cuda"), sizes=torch.zeros(nb, dtype=torch.int32, device="cuda")))
    torch.cuda.synchronize()
    errs = []

    def work(t):
        try:
            b = bufs[t]
            for _ in range(20):
                cw.dev_hash_and_compress("skein512", "lz4", b["src"].data_ptr(), bs, nb, b["dig"].data_ptr(), b["dst"].data_ptr(), stride,
                                         b["sizes"].data_ptr(), s)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errs, errs
    for t in range(2):
        dig, dst, sizes = (bufs[t][k].cpu().numpy() for k in ("dig", "dst", "sizes"))
        for i in range(nb):
            blk = datas[t][i * bs:(i + 1) * bs]
            want = oracle.lz4_compress(blk)
            assert sizes[i] == len(want) and dst[i * stride:i * stride + len(want)].tobytes() == want, (t, i)
            assert dig[64 * i:64 * i + 64].tobytes() == oracle.skein512(blk, 512), (t, i)


def test_offload_failure_is_reported_not_swallowed(cw):
    """Start() on an object whose Reset() gave no buffers must not leave it hOffloaded; through the offload
    thread the waiter is still woken and finds the failure on the object."""
    h = cw.HashOffload(8, "skein", 4096)
    assert cw.lib().cw_offload_reset(h._h, None, None, cw._lib.ON_COMPLETE(lambda _a: None), None) == 0
    h.Enqueue()
    with pytest.raises(cw.CwError):
        h.Start()
    assert h.state == h.hFailed and h.error != 0 and not h.Completed()
    with pytest.raises(cw.CwError):
        h.Complete()
    # through the consumer thread: the callback fires, the object says why
    ev = threading.Event()
    cb = cw._lib.ON_COMPLETE(lambda _a: ev.set())
    assert cw.lib().cw_offload_reset(h._h, None, None, cb, None) == 0
    assert h.state == h.hInit and h.error == 0
    assert cw.lib().cw_offload_thread_start() == 0
    h.Submit()
    assert ev.wait(30)
    cw.lib().cw_offload_thread_stop()
    assert h.state == h.hFailed and not h.Completed()
    # and the object works again after a proper Reset
    data = np.random.default_rng(1).integers(0, 256, 8 * 4096, dtype=np.uint8)
    res = np.zeros(8 * 16, dtype=np.uint8)
    h.Reset(data, res)
    h.Enqueue()
    h.DoOffload()
    assert h.Completed() and res.any()
    h.close()


def test_digests_at_any_alignment(cw, oracle):
    import torch
    s = torch.cuda.current_stream().cuda_stream
    bs, nb = 4096, 70
    raw = np.random.default_rng(11).integers(0, 256, bs * nb, dtype=np.uint8)
    src = torch.from_numpy(raw).cuda()
    for alg, db in (("skein512", 64), ("skein", 16), ("sha256mb", 32)):
        for shift in (1, 8):
            dig = torch.zeros(nb * db + 32, dtype=torch.uint8, device="cuda")
            cw.dev_hash(alg, src.data_ptr(), bs, nb, dig.data_ptr() + shift, s)
            torch.cuda.synchronize()
            hd = dig.cpu().numpy()
            want = {"skein512": lambda b: oracle.skein512(b, 512), "skein": lambda b: oracle.skein256(b, 128), "sha256mb": oracle.sha256}[alg]
            for i in (0, 1, 63, 64, nb - 1):
                assert hd[shift + i * db: shift + (i + 1) * db].tobytes() == want(raw[i * bs:(i + 1) * bs].tobytes()), (alg, shift, i)


def test_decoders_reject_sizes_beyond_the_slot(cw, oracle):
    import torch
    s = torch.cuda.current_stream().cuda_stream
    for bs in (4096, 65536):          # staged and unstaged decoder variants
        blk = corpus_file("alice29.txt")[:bs]
        for alg, comp in (("lz4", oracle.lz4_compress(blk)), ("lzf", oracle.lzf_compress(blk))):
            stride = (cw.compress_bound(alg, bs) + 15) // 16 * 16
            nb = 4
            slots = np.zeros((nb, stride), dtype=np.uint8)
            slots[:, : len(comp)] = np.frombuffer(comp, dtype=np.uint8)
            d_slots = torch.from_numpy(slots.reshape(-1)).cuda()
            sizes = np.array([len(comp), stride + 1, 0xFFFFFFF0, 1 << 25], dtype=np.uint32)   # ok, > slot, absurd, absurd
            d_sizes = torch.from_numpy(sizes.view(np.int32)).cuda()
            out = torch.zeros(nb * bs, dtype=torch.uint8, device="cuda")
            st = torch.full((nb,), 7, dtype=torch.int32, device="cuda")
            cw.dev_decompress(alg, d_slots.data_ptr(), stride, d_sizes.data_ptr(), nb, out.data_ptr(), bs, st.data_ptr(), s)
            torch.cuda.synchronize()
            assert st.cpu().tolist() == [0, 1, 1, 1], (alg, bs)
            assert out[:bs].cpu().numpy().tobytes() == blk


def test_alternative_lz4_parser_is_exact():
    """CW_LZ4_PARSE=fp (the fingerprint parser for blocks read from global memory; read once per process)."""
    prog = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, compute_war_amd as cw\n"
        "from conftest import corpus_file\n"
        "cw.init(0)\n"
        "data = corpus_file('lcet10.txt')[:6*65536] + bytes(65536) + corpus_file('kennedy.xls')[:5*65536] + corpus_file('ptt5')[:3*65536]\n"
        "for bs in (8192, 65536):\n"
        "    sizes, payload = cw.compress_blocks('lz4', data, bs)\n"
        "    h = hashlib.sha256(sizes.tobytes())\n"
        "    for i in range(len(sizes)): h.update(payload[i, :sizes[i]].tobytes())\n"
        "    print(bs, int(sizes.sum()), h.hexdigest(), cw.profile_kernels()['codec'])\n" % (ROOT, ROOT))
    outs = []
    for env in ({}, {"CW_LZ4_PARSE": "fp"}, {"CW_LZ4_PARSE": "fp", "CW_LZ4_HEADW": "32"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    for a, b, c in zip(*outs):
        assert a[:3] == b[:3] == c[:3]
        assert "lz4_vtab3_kernel<true>" in " ".join(a) and "lz4_parse_fp_kernel" in " ".join(b)   # (round 3: the scalar-thread parser is the default for these blocks)


def test_lane_per_block_lz4_parser_is_exact():
    """The lane-per-block LZ4 parser only takes over from 10-14 Ki queued blocks on; CW_LZ4_LANES=1 (threshold 1, read once per
    process) sends every queued block through it: corpus blocks at 8 / 16 / 64 KiB, runs, a block with one long tail."""
    prog = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, compute_war_amd as cw\n"
        "from conftest import corpus_file\n"
        "cw.init(0)\n"
        "rng = np.random.default_rng(5)\n"
        "sparse = rng.integers(0, 256, 3 * 65536, dtype=np.uint8); sparse[1000:1016] = sparse[200:216]; sparse[70000:70300] = 7\n"
        "data = corpus_file('lcet10.txt')[:6*65536] + bytes(65536) + corpus_file('kennedy.xls')[:5*65536] + corpus_file('ptt5')[:3*65536] + corpus_file('sum')[:32768] * 2 + sparse.tobytes()\n"
        "for bs in (8192, 16384, 65536, 4096, 1000):\n"
        "    sizes, payload = cw.compress_blocks('lz4', data, bs)\n"
        "    h = hashlib.sha256(sizes.tobytes())\n"
        "    for i in range(len(sizes)): h.update(payload[i, :sizes[i]].tobytes())\n"
        "    print(bs, int(sizes.sum()), h.hexdigest(), cw.profile_kernels()['codec'])\n" % (ROOT, ROOT))
    outs = []
    # blocks > 4 KiB: the ring form (input in an LDS ring, fingerprints in the table; CW_LZ4_LANES_RING=2: two positions per
    # iteration, =0: lz4_lanes_kernel with / without fingerprints, which blocks <= 4 KiB always use in its tagged form)
    alone = {"CW_LZ4_VTAB": "0", "CW_LANES_CONCURRENT": "0"}   # (round 3: by default the lanes share the queue with two other parsers)
    for env in ({}, {"CW_LZ4_LANES": "1", "CW_LZ4_LANES_RING": "1"}, {"CW_LZ4_LANES": "1", "CW_LANES_WPC": "1"}, {"CW_LZ4_LANES": "1", "CW_LZ4_LANES_RING": "2"},
                {"CW_LZ4_LANES": "1", "CW_LZ4_LANES_RING": "4"}, {"CW_LZ4_LANES": "1", "CW_LZ4_LANES_RING": "8"},
                {"CW_LZ4_LANES": "1", "CW_LZ4_LANES_RING": "0"}, {"CW_LZ4_LANES": "1", "CW_LZ4_LANES_RING": "0", "CW_LZ4_LANES_FP": "0"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env={**os.environ, **(alone if env else {}), **env})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    assert len(outs[0]) == 5   # the last two sizes: LDS-staged blocks, the lanes run BESIDE the LDS-resident parser
    for rows in zip(*outs):
        assert all(r[:3] == rows[0][:3] for r in rows), rows
        assert "lanes" not in " ".join(rows[0]) and "lz4_lanes" in " ".join(rows[1])
    assert "lz4_lanes_ring_kernel" in " ".join(outs[1][0]) and "lz4_lanes_kernel" in " ".join(outs[6][0])
    assert "lz4_lanes_ring_kernel<4>" in " ".join(outs[4][0]) and "lz4_lanes_ring_kernel<8>" in " ".join(outs[5][0])


def test_lane_per_block_lzf_parser_is_exact():
    """CW_LZF_LANES=1 sends every block > 4 KiB through the lane-per-block LZF parser (normally only from 12 Ki blocks on):
    corpus blocks, runs, noise that does not fit (returns 0), at 8 / 16 / 64 KiB."""
    prog = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, compute_war_amd as cw\n"
        "from conftest import corpus_file\n"
        "cw.init(0)\n"
        "rng = np.random.default_rng(5)\n"
        "noise = rng.integers(0, 256, 2 * 65536, dtype=np.uint8).tobytes()\n"
        "data = corpus_file('lcet10.txt')[:6*65536] + bytes(65536) + corpus_file('kennedy.xls')[:5*65536] + corpus_file('ptt5')[:3*65536] + noise + corpus_file('sum')[:32768] * 2\n"
        "# mostly noise: the lanes hand such blocks to the chain parser after a look and, once they are the majority, stop parsing\n"
        "data += rng.integers(0, 256, 150 * 65536, dtype=np.uint8).tobytes() + corpus_file('lcet10.txt')[:2*65536] + rng.integers(0, 256, 40 * 65536, dtype=np.uint8).tobytes()\n"
        "for bs in (8192, 16384, 65536):\n"
        "    sizes, payload = cw.compress_blocks('lzf', data, bs)\n"
        "    h = hashlib.sha256(sizes.tobytes())\n"
        "    for i in range(len(sizes)): h.update(payload[i, :sizes[i]].tobytes())\n"
        "    print(bs, int(sizes.sum()), int((sizes == 0).sum()), h.hexdigest(), cw.profile_kernels()['codec'])\n" % (ROOT, ROOT))
    outs = []
    for env in ({}, {"CW_LZF_LANES": "1"}, {"CW_LZF_LANES": "1", "CW_LANES_WPC": "1"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    assert len(outs[0]) == 3
    for a, b, c in zip(*outs):
        assert a[:4] == b[:4] == c[:4] and int(a[2]) > 0
        assert "lzf_lanes_kernel" not in " ".join(a) and "lzf_lanes_kernel" in " ".join(b)
    # small blocks: the lanes run BESIDE the link/chain rounds (from the top of the batch downwards).  Rounds of 16 blocks
    # (CW_LZF_ROUND, test knob) make a 2.6 MiB batch many rounds, so both sides and their meeting point are exercised.
    prog2 = prog.replace("for bs in (8192, 16384, 65536):", "for bs in (4096, 1000, 2048):")
    outs = []
    for env in ({"CW_LZF_LANES": "0"}, {"CW_LZF_LANES": "1", "CW_LZF_ROUND": "16"}, {"CW_LZF_LANES": "1", "CW_LZF_ROUND": "16", "CW_LANES_WPC": "1"},
                {"CW_LZF_LANES": "1", "CW_LZF_ROUND": "5", "CW_LANES_RESERVE": "10"}):
        r = subprocess.run([sys.executable, "-c", prog2], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    assert len(outs[0]) == 3
    for rows in zip(*outs):
        assert all(r[:4] == rows[0][:4] for r in rows), rows
        assert "lanes" not in " ".join(rows[0]) and "lzf_lanes_kernel<true> [side stream]" in " ".join(rows[1])


def test_lane_parsers_odd_sizes_strides_and_alignment(oracle):
    """Both lane-per-block parsers through the device API with block sizes that are not multiples of anything, a source
    stride larger than the block and a base that is not aligned -- the byte-granular loads at the ends of a block must
    stay inside it (the last block ends where the buffer ends)."""
    prog = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, torch, hashlib, compute_war_amd as cw\n"
        "from conftest import corpus_file\n"
        "cw.init(0)\n"
        "s = torch.cuda.current_stream().cuda_stream\n"
        "text = corpus_file('fields.c') + corpus_file('cp.html') + corpus_file('sum') + corpus_file('ptt5')[:90000]\n"
        "for bs, nb, sstride, shift in ((5001, 40, 5001, 3), (12345, 19, 12345 + 7, 1), (65535, 5, 65535, 0), (4100, 70, 4100, 2)):\n"
        "    raw = np.frombuffer((text * (sstride * nb // len(text) + 2))[:sstride * (nb - 1) + bs + shift], dtype=np.uint8).copy()\n"
        "    raw[::5] ^= np.arange(len(raw[::5]), dtype=np.uint8)\n"
        "    dev = torch.from_numpy(raw).cuda()\n"
        "    base = dev.data_ptr() + shift\n"
        "    for comp in ('lz4', 'lzf'):\n"
        "        dstride = cw.compress_bound(comp, bs) + 3\n"
        "        dst = torch.zeros(nb * dstride + 64, dtype=torch.uint8, device='cuda')\n"
        "        sizes = torch.zeros(nb, dtype=torch.int32, device='cuda')\n"
        "        cw.dev_compress(comp, base, bs, nb, dst.data_ptr() + 1, dstride, sizes.data_ptr(), s, src_stride=sstride)\n"
        "        torch.cuda.synchronize()\n"
        "        hz, hd = sizes.cpu().numpy(), dst.cpu().numpy()\n"
        "        h = hashlib.sha256(hz.tobytes())\n"
        "        for i in range(nb): h.update(hd[1 + i * dstride: 1 + i * dstride + hz[i]].tobytes())\n"
        "        print(bs, comp, int(hz.sum()), h.hexdigest(), cw.profile_kernels()['codec'])\n" % (ROOT, ROOT))
    outs = []
    for env in ({"CW_LZ4_LANES": "0", "CW_LZF_LANES": "0"}, {"CW_LZ4_LANES": "1", "CW_LZF_LANES": "1"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln.split() for ln in r.stdout.strip().splitlines()])
    assert len(outs[0]) == 8
    for a, b in zip(*outs):
        assert a[:4] == b[:4], (a, b)
        assert "lanes" not in " ".join(a) and "lanes" in " ".join(b)
    # and the reference run itself against the oracle, for one of the shapes
    text = corpus_file("fields.c") + corpus_file("cp.html") + corpus_file("sum") + corpus_file("ptt5")[:90000]
    bs, nb, sstride, shift = 5001, 40, 5001, 3
    raw = np.frombuffer((text * (sstride * nb // len(text) + 2))[:sstride * (nb - 1) + bs + shift], dtype=np.uint8).copy()
    raw[::5] ^= np.arange(len(raw[::5]), dtype=np.uint8)
    tot = sum(len(oracle.lz4_compress(raw[shift + i * sstride: shift + i * sstride + bs].tobytes())) for i in range(nb))
    assert int(outs[0][0][2]) == tot


def test_devices_and_contexts(cw):
    assert cw.device_count() >= 1 and cw.get_device() == 0
    cw.set_device(0)
    with pytest.raises(cw.CwError):
        cw.set_device(cw.device_count())     # not a silent no-op: out of range is an error
    assert cw.get_device() == 0
    names = cw.profile_kernels()
    assert set(names) == {"codec", "hash"}


def test_mgpu_stream_one_device(oracle):
    """The C multi-GPU harness with the one device this box has: sharding, per-device worker, RCCL gather (one rank)."""
    exe = os.path.join(ROOT, "compute_war_amd", "host", "mgpu_stream")
    r = subprocess.run(["make", "-C", os.path.dirname(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bs, nb = 65536, 300
    r = subprocess.run([exe, "--devices", "1", "--blocks-per-gpu", str(nb), "--block-size", str(bs), "--steps", "2", "--warmup", "1",
                        "--data", "mixed"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()   # RCCL announces itself (versions, host) on stdout: pick our lines by content
    assert any(ln.startswith("skein512|lz4|") for ln in lines)
    rep = json.loads(next(ln for ln in lines if ln.startswith("{")))
    data = oracle.gen_mixed_blocks(0xC0FFEE, 0, nb, bs)
    out = sum(len(oracle.lz4_compress(data[i * bs:(i + 1) * bs].tobytes())) for i in range(nb))
    fold = np.uint64(0)
    for i in range(nb):
        fold ^= np.bitwise_xor.reduce(np.frombuffer(oracle.skein512(data[i * bs:(i + 1) * bs].tobytes(), 512), dtype="<u8"))
    assert rep["n_gpus"] == 1 and rep["bytes_out"] == out and int(rep["digest_fold"], 16) == int(fold) and rep["all_devices_agree"]


def test_driver_devices_flag_and_rccl_totals(oracle):
    exe = os.path.join(ROOT, "compute_war_amd", "host", "hashandcompress")
    files = ["alice29.txt", "kennedy.xls", "ptt5"]
    paths = [os.path.join(GOLDEN, "corpus", "canterbury", f) for f in files]
    r = subprocess.run([exe, "-v", "-g", "true", "--devices", "1", "-c", "3", "-r", "1", "--block-size=65536", "-H", "skein512", "-C", "lz4"] + paths,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()   # RCCL announces itself on stdout: pick our lines by content
    bs, nblocks, out = 65536, 0, 0
    for f in files:
        d = corpus_file(f)
        for i in range(len(d) // bs):
            out += len(oracle.lz4_compress(d[i * bs:(i + 1) * bs]))
            nblocks += 1
    assert any(ln.startswith(f"blocks={nblocks} in={nblocks * bs} out={out} ") for ln in lines)
    assert f"devices=1 in={nblocks * bs} out={out} (ncclAllReduce over the per-device totals)" in lines
    r = subprocess.run([exe, "--devices", "9", paths[0]], capture_output=True, text=True)
    assert r.returncode == 2 and "9 device(s) asked for, 1 usable" in r.stderr   # (round 3: checked before any work)


def test_lane_per_block_decoders():
    """CW_DECODE_LANES=1 sends every batch through the lane-per-block decoders (normally from about 128 MiB of blocks
    on): round trips of the mixed corpus at 64 KiB and at an odd size, oracle-made streams with runs (offset 1 / 3 / 256 matches
    of tens of kilobytes), and the malformed inputs of the wavefront decoders' tests -- truncated, an offset in front of the
    block, sizes beyond the slot -- which must give status 1 and nothing else."""
    prog = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, torch, compute_war_amd as cw\n"
        "from conftest import corpus_file, corpus_names\n"
        "import oracle\n"
        "cw.init(0)\n"
        "s = torch.cuda.current_stream().cuda_stream\n"
        "data = b''.join(corpus_file(n) for n in corpus_names())\n"
        "rng = np.random.default_rng(3)\n"
        "for bs in (65536, 20001, 4100):\n"
        "    nb = 96\n"
        "    a = np.frombuffer((data * 4)[:nb * bs], dtype=np.uint8).copy()\n"
        "    a[5 * bs:6 * bs] = rng.integers(0, 256, bs, dtype=np.uint8); a[7 * bs:8 * bs] = 0; a[9 * bs:10 * bs] = np.tile(np.arange(3, dtype=np.uint8), bs)[:bs]\n"
        "    src = torch.from_numpy(a).cuda()\n"
        "    for comp in ('lz4', 'lzf'):\n"
        "        stride = (cw.compress_bound(comp, bs) + 15) // 16 * 16\n"
        "        dst = torch.zeros(nb * stride, dtype=torch.uint8, device='cuda'); sizes = torch.zeros(nb, dtype=torch.int32, device='cuda')\n"
        "        back = torch.zeros(nb * bs, dtype=torch.uint8, device='cuda'); status = torch.full((nb,), 7, dtype=torch.int32, device='cuda')\n"
        "        cw.dev_compress(comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)\n"
        "        cw.dev_decompress(comp, dst.data_ptr(), stride, sizes.data_ptr(), nb, back.data_ptr(), bs, status.data_ptr(), s)\n"
        "        torch.cuda.synchronize()\n"
        "        fits = sizes != 0\n"
        "        assert bool((status[fits] == 0).all()) and bool((status[~fits] == 1).all()), (comp, bs, status.cpu().tolist())\n"
        "        ok = (back.view(nb, bs) == src.view(nb, bs)).all(dim=1)\n"
        "        assert bool(ok[fits].all()), (comp, bs)\n"
        "        print('roundtrip', comp, bs, int(fits.sum()))\n"
        "bs = 32768\n"
        "blocks = [corpus_file('alice29.txt')[:bs], bytes(bs), bytes(range(256)) * (bs // 256), (b'abc' * bs)[:bs], corpus_file('ptt5')[:bs]]\n"
        "for comp, enc in (('lz4', oracle.lz4_compress), ('lzf', oracle.lzf_compress)):\n"
        "    stride = (cw.compress_bound(comp, bs) + 15) // 16 * 16\n"
        "    k = len(blocks)\n"
        "    buf = np.zeros((k + 5, stride), dtype=np.uint8); sz = np.zeros(k + 5, dtype=np.uint32)\n"
        "    for i, b in enumerate(blocks):\n"
        "        c = enc(b); buf[i, :len(c)] = np.frombuffer(c, dtype=np.uint8); sz[i] = len(c)\n"
        "    c = enc(blocks[0]); buf[k, :len(c) - 7] = np.frombuffer(c[:-7], dtype=np.uint8); sz[k] = len(c) - 7\n"
        "    bad = (bytes([0x10, 65, 0xFF, 0x7F]) + bytes(20)) if comp == 'lz4' else bytes([0, 65, 0xFF, 0xFF, 0]) + bytes(20)\n"
        "    buf[k + 1, :len(bad)] = np.frombuffer(bad, dtype=np.uint8); sz[k + 1] = len(bad)\n"
        "    buf[k + 2:, :len(c)] = np.frombuffer(c, dtype=np.uint8); sz[k + 2:] = [stride + 1, 0xFFFFFFF0, 1 << 25]\n"
        "    d_buf, d_sz = torch.from_numpy(buf).cuda(), torch.from_numpy(sz.view(np.int32)).cuda()\n"
        "    back = torch.zeros((k + 5, bs), dtype=torch.uint8, device='cuda'); status = torch.full((k + 5,), 7, dtype=torch.int32, device='cuda')\n"
        "    cw.dev_decompress(comp, d_buf.data_ptr(), stride, d_sz.data_ptr(), k + 5, back.data_ptr(), bs, status.data_ptr(), s)\n"
        "    torch.cuda.synchronize()\n"
        "    st, hb = status.cpu().tolist(), back.cpu().numpy()\n"
        "    assert st == [0] * k + [1] * 5, (comp, st)\n"
        "    for i, b in enumerate(blocks): assert hb[i].tobytes() == b, (comp, i)\n"
        "    print('streams', comp, st)\n" % (ROOT, ROOT))
    for env in ({"CW_DECODE_LANES": "1"}, {"CW_DECODE_LANES": "1", "CW_LANES_WPC": "1"}, {"CW_DECODE_LANES": "0"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert r.returncode == 0, (env, r.stderr[-2000:])
        assert r.stdout.count("roundtrip") == 6 and r.stdout.count("streams") == 2


def test_side_by_side_parsers_at_default_thresholds_equal_the_plain_ones():
    """At the sizes where the launch policy itself switches the lane parsers on -- 256 Ki and 64 Ki blocks of 4 KiB (lanes BESIDE the
    LDS-resident parsers, real interleavings of the shared queue / the top-down and bottom-up claims), 40 Ki blocks of 64 KiB
    (round 3: lanes with two positions per iteration BESIDE the register-table and the wavefront parser, leaving them 24 Ki blocks),
    112 Ki blocks of 8 KiB and 72 Ki blocks of 16 KiB (above the round-3 thresholds of those sizes) and 32 Ki blocks of 4 KiB (LZF lanes
    beside the rounds with the small reserve) -- the packed output stream and the sizes are identical to those of the parsers without the lanes
    (CW_LZ4_LANES=0 CW_LZF_LANES=0), which the parity tests pin to the oracle.  Compared through a Skein-512 digest per 64 KiB of
    the packed stream, computed on the device."""
    prog = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np, torch, compute_war_amd as cw\n"
        "from conftest import corpus_file, corpus_names\n"
        "cw.init(0)\n"
        "s = torch.cuda.current_stream().cuda_stream\n"
        "data = b''.join(corpus_file(n) for n in corpus_names())\n"
        "rng = np.random.default_rng(11)\n"
        "for bs, nb in ((4096, 262144), (4096, 65536), (65536, 40960), (8192, 114688), (16384, 73728), (4096, 32768)):\n"
        "    a = np.frombuffer((data * (nb * bs // len(data) + 1))[:nb * bs], dtype=np.uint8).copy()\n"
        "    for o in range(0, nb * bs - 65536, 7 * 65536): a[o:o + 65536] = rng.integers(0, 256, 65536, dtype=np.uint8)\n"
        "    src = torch.from_numpy(a).cuda(); del a\n"
        "    for comp in ('lz4', 'lzf'):\n"
        "        stride = (cw.compress_bound(comp, bs) + 15) // 16 * 16\n"
        "        dst = torch.zeros(nb * stride, dtype=torch.uint8, device='cuda'); sizes = torch.zeros(nb, dtype=torch.int32, device='cuda')\n"
        "        for rep in range(2):\n"
        "            cw.dev_compress(comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)\n"
        "        offs = torch.zeros(nb + 1, dtype=torch.int64, device='cuda')\n"
        "        cw.dev_pack(dst.data_ptr(), stride, sizes.data_ptr(), nb, 0, offs.data_ptr(), s)\n"
        "        torch.cuda.synchronize()\n"
        "        total = int(offs[-1].item()); nd = (total + 65535) // 65536\n"
        "        packed = torch.zeros(nd * 65536, dtype=torch.uint8, device='cuda')\n"
        "        cw.dev_pack(dst.data_ptr(), stride, sizes.data_ptr(), nb, packed.data_ptr(), offs.data_ptr(), s)\n"
        "        dig = torch.zeros(nd * 64, dtype=torch.uint8, device='cuda')\n"
        "        cw.dev_hash('skein512', packed.data_ptr(), 65536, nd, dig.data_ptr(), s)\n"
        "        torch.cuda.synchronize()\n"
        "        h = hashlib.sha256(sizes.cpu().numpy().tobytes()); h.update(dig.cpu().numpy().tobytes())\n"
        "        print('out', comp, bs, total, h.hexdigest(), cw.profile_kernels()['codec'])\n"
        "        del dst, packed, dig\n" % (ROOT, ROOT))
    outs = []
    for env in ({}, {"CW_LZ4_LANES": "0", "CW_LZF_LANES": "0"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0, (env, r.stderr[-2000:])
        outs.append([ln.split(None, 5) for ln in r.stdout.splitlines() if ln.startswith("out ")])
    assert len(outs[0]) == 12 and len(outs[1]) == 12
    for i, (a, b) in enumerate(zip(*outs)):
        assert a[:5] == b[:5], (a, b)
        # 64 Ki blocks of 4 KiB: as many blocks as a full grid of LZF lanes (every lane asks at once -- the case a check-then-add
        # protocol got wrong); 56 Ki of them queued for LZ4: just below its lanes' threshold, so the launched lane kernel returns at once
        assert ("lanes" in a[5]) == (i != 10) and "lanes" not in b[5], (a, b)   # (32 Ki blocks of 4 KiB: LZF lanes beside the rounds, no LZ4 lanes yet)
    assert "lz4_lanes_ring_auto_kernel" in outs[0][4][5] and "[side stream]" in outs[0][4][5], outs[0][4]   # 40 Ki blocks of 64 KiB: lanes beside the other two parsers
    assert "[side stream]" in outs[0][0][5] and "[side stream]" in outs[0][1][5] and "[side stream]" in outs[0][3][5]
