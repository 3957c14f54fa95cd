"""Round-3 GPU tests: the launch policy's own large-batch parsers pinned DIRECTLY to the oracle, and cw_prepare
over jobs without a codec or without a hash."""
import hashlib

import numpy as np
import pytest

from conftest import corpus_file, corpus_large_file, corpus_large_names, corpus_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cw():
    import torch  # noqa: F401  (one HIP runtime for torch and libcwhc.so)
    import compute_war_amd as cw
    cw.init(0)
    return cw


def check(rc):
    from compute_war_amd._lib import check as _check
    _check(rc)


def _corpus_bytes(nbytes):
    """The in-tree corpora (whole files, canterbury then canterbury-large) repeated to nbytes, with a block of noise every 37 blocks of 64 KiB so that
    the batch is not purely periodic and the scan's literal-run path, the LZF hand-back path and the parsers all see work."""
    data = b"".join(corpus_file(n) for n in corpus_names()) + b"".join(corpus_large_file(n) for n in corpus_large_names())
    a = np.frombuffer((data * (nbytes // len(data) + 1))[:nbytes], dtype=np.uint8).copy()
    rng = np.random.default_rng(3)
    for o in range(5 * 65536, nbytes - 65536, 37 * 65536):
        a[o:o + 65536] = rng.integers(0, 256, 65536, dtype=np.uint8)
    return a


@pytest.mark.parametrize("comp,hash_name,bs,nb", [("lz4", "skein512", 65536, 16384), ("lz4", "skein", 4096, 65536),
                                                  ("lzf", "sha256mb", 65536, 16384), ("lzf", "sha256mb", 4096, 65536)])
def test_default_policy_large_batches_equal_the_oracle_on_every_block(cw, oracle, comp, hash_name, bs, nb):
    """>= 16 Ki x 64 KiB and >= 64 Ki x 4 KiB corpus batches through the fused entry point at the DEFAULT thresholds (no knob set):
    every block's size, digest and payload against cw_oracle_hash_and_compress run over the WHOLE batch on the host's threads, and the
    library must report a lane-per-block kernel among what it launched -- so "lanes == oracle" is direct, not via the wavefront parsers."""
    import torch
    a = _corpus_bytes(nb * bs)
    h = {"skein512": oracle.HASH_SKEIN512, "skein": oracle.HASH_SKEIN256_128, "sha256mb": oracle.HASH_SHA256}[hash_name]
    c = oracle.COMP_LZ4 if comp == "lz4" else oracle.COMP_LZF
    _, odig, osz, opay = oracle.hash_and_compress(a, bs, h, c, threads=16, want_payload=True)
    s = torch.cuda.current_stream().cuda_stream
    src = torch.from_numpy(a).cuda()
    stride = (cw.compress_bound(comp, bs) + 15) // 16 * 16
    db = cw.digest_bytes(hash_name)
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    dig = torch.zeros((nb, db), dtype=torch.uint8, device="cuda")
    for _ in range(2):   # twice: the second call runs on warm workspaces and lane tables of the first
        cw.dev_hash_and_compress(hash_name, comp, src.data_ptr(), bs, nb, dig.data_ptr(), dst.data_ptr(), stride, sizes.data_ptr(), s)
    torch.cuda.synchronize()
    names = cw.profile_kernels()["codec"]
    assert "lanes" in names or "vtab" in names, names
    hz = sizes.cpu().numpy().astype(np.uint32)
    bad = np.nonzero(hz != osz)[0]
    assert bad.size == 0, (names, bad[:8], hz[bad[:8]], osz[bad[:8]])
    assert np.array_equal(dig.cpu().numpy(), odig)
    # payload: a digest per block over exactly size bytes, device slots against oracle slots
    slots = dst.view(nb, stride).cpu().numpy()
    for i in range(nb):
        z = int(osz[i])
        if z and hashlib.blake2b(slots[i, :z].tobytes(), digest_size=16).digest() != hashlib.blake2b(opay[i, :z].tobytes(), digest_size=16).digest():
            raise AssertionError(f"payload of block {i} differs ({names})")
    if comp == "lzf":
        assert int((osz == 0).sum()) > 0   # the noise blocks: "did not fit" verdicts are part of the comparison


def test_prepare_without_codec_or_without_hash_then_a_normal_batch(cw, oracle):
    """cw_prepare(hash, CW_COMP_NONE) used to run the pack kernels over NULL sizes/offsets; (CW_HASH_NONE, lz4) and
    (NONE, NONE) are valid too.  A normal batch afterwards still equals the oracle."""
    L = cw.lib()
    H = {"skein512": cw.HASH_SKEIN512, "skein": cw.HASH_SKEIN256_128, "sha256mb": cw.HASH_SHA256, "none": cw.HASH_NONE}
    C = {"lz4": cw.COMP_LZ4, "lzf": cw.COMP_LZF, "none": cw.COMP_NONE}
    for hname, cname in (("skein512", "none"), ("none", "lz4"), ("none", "none"), ("sha256mb", "none"), ("none", "lzf")):
        check(L.cw_prepare(H[hname], C[cname], 4096, 2048, 0))
    rng = np.random.default_rng(5)
    blocks = [corpus_file("alice29.txt")[i * 4096:(i + 1) * 4096] for i in range(24)] + [rng.integers(0, 256, 4096, dtype=np.uint8).tobytes() for _ in range(8)]
    data = np.frombuffer(b"".join(blocks), dtype=np.uint8)
    check(L.cw_prepare(H["skein"], C["lz4"], 4096, len(blocks), 0))
    dig, sizes, payload = cw.hash_and_compress_blocks("skein", "lz4", data, 4096)
    _, odig, osz, opay = oracle.hash_and_compress(data, 4096, oracle.HASH_SKEIN256_128, oracle.COMP_LZ4, threads=2, want_payload=True)
    assert np.array_equal(dig, odig) and np.array_equal(sizes, osz)
    for i in range(len(blocks)):
        assert payload[i][: int(osz[i])].tobytes() == opay[i, : int(osz[i])].tobytes()


def test_lzf_lane_share_claim_that_is_never_seen_ends_in_an_exact_batch(cw, oracle, capfd):
    """The rounds' workgroups wait (bounded) for the round's claim to be published (LaneShare, lzf_kernel.hip).  CW_LZF_SHARE_GIVE_UP=1
    makes every workgroup but the claimant give up at once: the batch's `gave up` word must be set, and the final pass must then parse
    every block again -- sizes and payloads equal the oracle's all the same, no hang.  Without the knob nobody gives up."""
    import torch
    bs, nb = 4096, 32768      # LZF lanes BESIDE the link/chain rounds start at 28 Ki blocks of <= 4 KiB
    a = _corpus_bytes(nb * bs)
    _, _, osz, opay = oracle.hash_and_compress(a, bs, oracle.HASH_NONE, oracle.COMP_LZF, threads=16, want_payload=True)
    s = torch.cuda.current_stream().cuda_stream
    src = torch.from_numpy(a).cuda()
    stride = (cw.compress_bound("lzf", bs) + 15) // 16 * 16
    for give_up in (1, 0):
        dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
        with cw.tuned(CW_LZF_SHARE_GIVE_UP=give_up, CW_DEBUG_LZF=1):
            cw.dev_compress("lzf", src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
            torch.cuda.synchronize()
            assert "lzf_lanes_kernel<true> [side stream]" in cw.profile_kernels()["codec"]
        err = capfd.readouterr().err
        assert ("gave up 1" in err) == bool(give_up), err[-400:]
        assert np.array_equal(sizes.cpu().numpy().astype(np.uint32), osz)
        slots = dst.view(nb, stride).cpu().numpy()
        for i in range(0, nb, 7):
            z = int(osz[i])
            assert slots[i, :z].tobytes() == opay[i, :z].tobytes(), i


def _sweep_data():
    rng = np.random.default_rng(5)
    sparse = rng.integers(0, 256, 3 * 65536, dtype=np.uint8)
    sparse[1000:1016] = sparse[200:216]
    sparse[70000:70300] = 7
    noise = rng.integers(0, 256, 2 * 65536, dtype=np.uint8).tobytes()
    return (corpus_file("lcet10.txt")[:6 * 65536] + bytes(65536) + corpus_file("kennedy.xls")[:5 * 65536] + corpus_file("ptt5")[:3 * 65536]
            + corpus_file("sum")[:32768] * 2 + sparse.tobytes() + noise + corpus_file("alice29.txt")[:2 * 65536])


LZ4_KNOBS = [
    dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0),                                                   # the LDS-table scalar-thread parser alone (blocks > 4 KiB)
    dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0, CW_LZ4_LTAB=0),                                    # round 2's wavefront parser alone
    dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=2, CW_LZ4_LANES=0),                                    # register-table parsers, each taking the whole queue
    dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=3, CW_LZ4_LANES=0),
    dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=3, CW_VTAB_WPC=1, CW_LZ4_LANES=0),
    dict(CW_LZ4_VTAB=2, CW_VTAB_GEN=3, CW_LZ4_LANES=0),                                    # beside the wavefront parser (the default regime)
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LZ4_LANES_RING=1),       # lane-per-block parsers, each taking the whole queue
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LZ4_LANES_RING=2),
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LZ4_LANES_RING=4),
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LZ4_LANES_RING=8),
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LZ4_LANES_RING=0),
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LZ4_LANES_RING=0, CW_LZ4_LANES_FP=0),
    dict(CW_LZ4_VTAB=0, CW_LANES_CONCURRENT=0, CW_LZ4_LANES=1, CW_LANES_WPC=1),
    dict(CW_LZ4_LANES=1),                                                                  # all three side by side on one queue, no reserve
    dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0, CW_LZ4_PARSE="fp"),                                # the fingerprint wavefront parser (diagnostic variant)
    dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0, CW_LZ4_PARSE="fp", CW_LZ4_HEADW=32),
    dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0, CW_LZ_FORCE_REDO=1),                               # every block through the first-generation redo pass
]
LZF_KNOBS = [
    dict(CW_LZF_LANES=0),                                                                  # link/chain rounds: blocks > 4 KiB through the scalar-thread parser
    dict(CW_LZF_LANES=0, CW_LZF_STHREAD=0),                                                # ... through the wavefront-wide chain kernel
    dict(CW_LZF_LANES=0, CW_LZF_ST_WPC=1),
    dict(CW_LZF_LANES=1, CW_LANES_CONCURRENT=1, CW_LZF_ROUND=5, CW_LANES_RESERVE=10),      # lanes beside the rounds at every block size
    dict(CW_LZF_LANES=1),
    dict(CW_LZF_LANES=1, CW_LANES_WPC=1),
    dict(CW_LZF_LANES=1, CW_LZF_ROUND=16),
    dict(CW_LZF_LANES=1, CW_LZF_ROUND=5, CW_LANES_RESERVE=10),
    dict(CW_LZF_LANES=0, CW_LZ_FORCE_REDO=1),
    dict(CW_LZF_MODE="table"),
]


@pytest.mark.parametrize("comp,knob_sets", [("lz4", LZ4_KNOBS), ("lzf", LZF_KNOBS)])
def test_every_parser_variant_equals_the_oracle_in_one_process(cw, oracle, comp, knob_sets):
    """Every parser the launch policy can pick, forced through cw_tune_set in ONE process (round 2 needed a subprocess per setting:
    the knobs were read once per process), each compared with the ORACLE -- sizes and payload bytes of every block -- at block sizes on
    both sides of the LDS-staging limit, an odd size included."""
    data = _sweep_data()
    want = {}
    for bs in (65536, 16384, 8192, 4096, 1000):
        blocks = [data[i * bs:(i + 1) * bs] for i in range(len(data) // bs)]
        want[bs] = [oracle.lz4_compress(b) if comp == "lz4" else oracle.lzf_compress(b) for b in blocks]
    seen = set()
    for knobs in knob_sets:
        with cw.tuned(**knobs):
            for bs, exp in want.items():
                sizes, payload = cw.compress_blocks(comp, data[: len(exp) * bs], bs)
                names = cw.profile_kernels()["codec"]
                seen.add(names)
                assert [int(z) for z in sizes] == [len(e) for e in exp], (knobs, bs, names)
                for i, e in enumerate(exp):
                    assert payload[i, : len(e)].tobytes() == e, (knobs, bs, i, names)
    joined = " | ".join(sorted(seen))
    if comp == "lz4":
        for k in ("lz4_vtab2_kernel", "lz4_lanes_ring_auto_kernel", "lz4_lanes_ring_kernel<1>", "lz4_lanes_ring_kernel<2>", "lz4_lanes_ring_kernel<4>",
                  "lz4_lanes_ring_kernel<8>", "lz4_lanes_kernel<0>", "lz4_lanes_kernel<1>", "lz4_lanes_kernel<2>", "lz4_parse_fp_kernel<32>", "lz4_parse_kernel<true>", "lz4_parse_kernel<false>", "lz4_vtab3_kernel<true>", "lz4_vtab3_kernel<false>"):
            assert k in joined, (k, joined)
    else:
        for k in ("lzf_lanes_kernel<true> [side stream]", "lzf_lanes_kernel<false>", "lzf_parse_kernel", "lzf_chain_kernel<true>", "lzf_chain_kernel<false>", "lzf_sthread_kernel"):
            assert k in joined, (k, joined)


def test_tune_set_rejects_foreign_names_and_reset_restores_defaults(cw):
    L = cw.lib()
    assert L.cw_tune_set(b"PATH", b"x") != 0
    cw.tune_set("CW_LZ4_VTAB", 0)
    data = (corpus_file("lcet10.txt") * 12)[:72 * 65536]
    cw.compress_blocks("lz4", data, 65536)
    assert "lz4_vtab3_kernel<false>" not in cw.profile_kernels()["codec"]   # (the register-table form; <true> is the same parser with its table in LDS)
    cw.tune_reset()
    cw.compress_blocks("lz4", data, 65536)
    assert "lz4_vtab3_kernel<false>" in cw.profile_kernels()["codec"]


def test_short_lived_calling_threads_leave_no_device_memory_behind(cw, oracle):
    """Every calling thread has its own context (streams, pipeline slots) and the codecs keep per-stream scratch; a thread that ends
    must take both with it (the scratch of its streams stayed until cw_shutdown, gigabytes once lane tables exist)."""
    import threading

    import torch
    data = (corpus_file("lcet10.txt") * 40)[:256 * 65536]   # 16 MiB: three pipeline chunks, codec workspaces on three slot streams
    want = [len(oracle.lz4_compress(data[i * 65536:(i + 1) * 65536])) for i in range(4)]
    result = {}

    def work(tag):
        d, sizes, _ = cw.hash_and_compress_blocks("skein512", "lz4", data, 65536)
        result[tag] = [int(z) for z in sizes[:4]]

    t = threading.Thread(target=work, args=("warm",)); t.start(); t.join()   # first use may allocate process-wide state
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(4):
        t = threading.Thread(target=work, args=(k,)); t.start(); t.join()
        assert result[k] == want
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)   # four more threads came and went: nothing of theirs is left


@pytest.mark.parametrize("bs", [5001, 65533, 4093, 70])
def test_register_table_parsers_on_block_sizes_that_are_not_multiples_of_four(cw, oracle, bs):
    """The scalar-thread parsers read the block through a buffer descriptor whose range check works on dwords: a block size that is not
    a multiple of 4 (source stride rounded up, so that they accept the batch) must still give the oracle's bytes -- the last, partial dword
    lies behind every position a match may reach."""
    import torch
    nb = 200
    stride_src = (bs + 3) // 4 * 4
    base = _sweep_data()
    rng = np.random.default_rng(bs)
    raw = np.zeros(nb * stride_src, dtype=np.uint8)
    blocks = []
    for i in range(nb):
        o = int(rng.integers(0, len(base) - bs))
        b = np.frombuffer(base[o:o + bs], dtype=np.uint8).copy()
        if i % 5 == 0:
            b[-9:] = b[-18:-9]          # a match that runs into the end of the block
        raw[i * stride_src:i * stride_src + bs] = b
        blocks.append(b.tobytes())
    want = [oracle.lz4_compress(b) for b in blocks]
    s = torch.cuda.current_stream().cuda_stream
    dev = torch.from_numpy(raw).cuda()
    dstride = (cw.compress_bound("lz4", bs) + 15) // 16 * 16
    for knobs in (dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=2, CW_LZ4_LANES=0), dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=3, CW_VTAB_WPC=4, CW_LZ4_LANES=0),
                  dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=3, CW_LZ4_LANES=0), dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0), dict()):
        dst = torch.zeros(nb * dstride, dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
        with cw.tuned(**knobs):
            cw.dev_compress("lz4", dev.data_ptr(), bs, nb, dst.data_ptr(), dstride, sizes.data_ptr(), s, src_stride=stride_src)
            torch.cuda.synchronize()
            names = cw.profile_kernels()["codec"]
        hz, hd = sizes.cpu().numpy(), dst.view(nb, dstride).cpu().numpy()
        for i, e in enumerate(want):
            assert int(hz[i]) == len(e) and hd[i, : len(e)].tobytes() == e, (knobs, bs, i, names)


def test_sequences_at_the_length_field_boundaries(cw, oracle):
    """The scalar-thread parsers write their sequences out 64 at a time, a lane per output byte (emit_batch, lz4_vtab_kernel.hip): blocks built from
    literal runs and matches whose lengths sit on the encoding's boundaries -- 14 / 15 / 16 (the nibble), 15 + 255 k +- 1 (length bytes), 96 / 97
    (where a literal run leaves the per-byte loop), runs of many kilobytes -- in every order, so that batches mix all kinds."""
    rng = np.random.default_rng(31)
    lits = [0, 1, 2, 3, 13, 14, 15, 16, 17, 95, 96, 97, 98, 110, 254, 255, 256, 269, 270, 271, 272, 300, 524, 525, 526, 1000, 5000]
    mats = [4, 5, 6, 17, 18, 19, 20, 21, 22, 272, 273, 274, 275, 276, 527, 528, 529, 530, 531, 2000, 9000]
    bs, nb = 65536, 96
    blocks = []
    for b in range(nb):
        out = bytearray(rng.integers(0, 256, 64, dtype=np.uint8).tobytes())
        while len(out) < bs:
            lit, m = int(rng.choice(lits)), int(rng.choice(mats))
            if b % 3 == 2:                     # short sequences only: 64 to a batch, thousands of batches
                lit, m = int(rng.integers(0, 18)), int(rng.integers(4, 24))
            out += rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
            back = int(rng.integers(1, min(len(out), 65535) + 1))
            for _ in range(m):                 # (byte by byte: the source may overlap what is being written)
                out.append(out[-back])
        blocks.append(bytes(out[:bs]))
    want = [oracle.lz4_compress(b) for b in blocks]
    data = b"".join(blocks)
    for knobs in (dict(CW_LZ4_VTAB=1, CW_VTAB_GEN=3, CW_LZ4_LANES=0), dict(CW_LZ4_VTAB=0, CW_LZ4_LANES=0), dict()):
        with cw.tuned(**knobs):
            sizes, payload = cw.compress_blocks("lz4", data, bs)
            names = cw.profile_kernels()["codec"]
        assert "lz4_vtab3_kernel" in names, names
        for i, e in enumerate(want):
            assert int(sizes[i]) == len(e) and payload[i, : len(e)].tobytes() == e, (knobs, i, names)


@pytest.mark.parametrize("reps", [56, 96])
def test_lzf_big_blocks_lanes_beside_scalar_thread_rounds_at_the_default_policy(cw, oracle, reps):
    """From 52 Ki blocks of more than 16 KiB on the LZF lanes run BESIDE the link / scalar-thread rounds (LaneShare, as for 4 KiB blocks; rounds of
    4 Ki blocks below 96 Ki blocks, of 8 Ki from there on): a tile of 1,024 corpus-and-noise blocks of 64 KiB repeated 56 and 96 times (3.5 and
    6 GiB) at the default policy; every block's size against the oracle's, the payload of the first, a middle and the last tile byte for byte."""
    import torch
    bs, tile = 65536, 1024
    nb = tile * reps
    a = _corpus_bytes(tile * bs)
    _, _, osz, opay = oracle.hash_and_compress(a, bs, oracle.HASH_NONE, oracle.COMP_LZF, threads=16, want_payload=True)
    s = torch.cuda.current_stream().cuda_stream
    src = torch.from_numpy(a).cuda().repeat(reps)
    stride = (cw.compress_bound("lzf", bs) + 15) // 16 * 16
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    cw.dev_compress("lzf", src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
    torch.cuda.synchronize()
    names = cw.profile_kernels()["codec"]
    assert "lzf_lanes_kernel<false> [side stream]" in names and "lzf_sthread_kernel" in names, names
    want = torch.from_numpy(osz.astype(np.int32)).cuda().repeat(reps)
    bad = torch.nonzero(sizes != want)
    assert bad.numel() == 0, (names, bad[:8].flatten().tolist())
    slots = dst.view(nb, stride)
    for t in (0, reps // 2, reps - 1):
        got = slots[t * tile:(t + 1) * tile].cpu().numpy()
        for i in range(tile):
            z = int(osz[i])
            assert got[i, :z].tobytes() == opay[i, :z].tobytes(), (t, i, names)
    del src, dst, sizes, slots
    torch.cuda.empty_cache()


def test_fused_call_gated_and_side_by_side_give_the_same_bytes(cw, oracle):
    """Blocks of up to 4 KiB, 16 Ki blocks or more, a compressible previous call: the fused call enqueues the hash behind the scan and lets the parsers
    wait for it (cw_api.hip, dev_fused).  Forced on, forced off and decided by the call history (three calls in a row), every digest, size and payload
    byte equals the oracle's."""
    import torch
    bs, nb = 4096, 24576
    a = _corpus_bytes(nb * bs)
    _, odig, osz, opay = oracle.hash_and_compress(a, bs, oracle.HASH_SKEIN256_128, oracle.COMP_LZ4, threads=16, want_payload=True)
    s = torch.cuda.current_stream().cuda_stream
    src = torch.from_numpy(a).cuda()
    stride = (cw.compress_bound("lz4", bs) + 15) // 16 * 16
    for knobs, calls in ((dict(CW_FUSED_GATE=1), 1), (dict(CW_FUSED_GATE=0), 1), (dict(), 3)):
        with cw.tuned(**knobs):
            for _ in range(calls):
                dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
                sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
                dig = torch.zeros((nb, 16), dtype=torch.uint8, device="cuda")
                cw.dev_hash_and_compress("skein", "lz4", src.data_ptr(), bs, nb, dig.data_ptr(), dst.data_ptr(), stride, sizes.data_ptr(), s)
                torch.cuda.synchronize()
                assert np.array_equal(sizes.cpu().numpy().astype(np.uint32), osz), knobs
                assert np.array_equal(dig.cpu().numpy(), odig), knobs
                slots = dst.view(nb, stride).cpu().numpy()
                for i in range(0, nb, 5):
                    z = int(osz[i])
                    assert slots[i, :z].tobytes() == opay[i, :z].tobytes(), (knobs, i)


def test_host_pipeline_with_shared_and_with_own_slot_streams(cw, oracle):
    """The host pipeline's slots share one pair of kernel streams for LZ4 jobs (cw_api.hip, shared_lender; decided when a thread's slots are first
    opened, so each setting runs in a thread of its own): both forms, over enough blocks for several chunks, equal the oracle."""
    import threading
    bs, nb = 65536, 4096          # 256 MiB: CW_HOST_CHUNK_MB=32 makes it eight chunks
    a = _corpus_bytes(nb * bs)
    _, odig, osz, opay = oracle.hash_and_compress(a, bs, oracle.HASH_SKEIN512, oracle.COMP_LZ4, threads=16, want_payload=True)
    want = b"".join(opay[i, : int(osz[i])].tobytes() for i in range(nb))
    errors = []

    def run(shared):
        try:
            cw.init(0)
            with cw.tuned(CW_HOST_SHARED_STREAMS=shared, CW_HOST_CHUNK_MB=32):
                for pinned in (False, True):
                    dig, sizes, offsets, packed = cw.hash_and_compress_packed("skein512", "lz4", a, bs, pinned=pinned)
                    assert np.array_equal(dig, odig) and np.array_equal(sizes, osz), (shared, pinned)
                    assert int(offsets[nb]) == len(want) and packed.tobytes() == want, (shared, pinned)
        except BaseException as e:   # noqa: BLE001 (reported in the main thread)
            errors.append((shared, repr(e)))

    for shared in (1, 0):
        t = threading.Thread(target=run, args=(shared,))
        t.start()
        t.join()
    assert not errors, errors
