"""Encode -> decode -> compare on the device: the reference-independent, size-independent check of the compressors
(SURVEY.md 8(f) N2).  The decoders themselves are pinned by decoding streams the CPU oracle produced."""
import numpy as np
import pytest

from conftest import corpus_file, corpus_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cw():
    import compute_war_amd as cw
    cw.init(0)
    return cw


def _mixed_corpus(nbytes):
    """canterbury files concatenated and tiled; every 5th 64 KiB stretch replaced by noise."""
    data = b"".join(corpus_file(n) for n in corpus_names())
    data = (data * (nbytes // len(data) + 1))[:nbytes]
    a = np.frombuffer(data, dtype=np.uint8).copy()
    rng = np.random.default_rng(9)
    for o in range(0, nbytes - 65536, 5 * 65536):
        a[o:o + 65536] = rng.integers(0, 256, 65536, dtype=np.uint8)
    return a


@pytest.mark.parametrize("comp", ["lz4", "lzf"])
@pytest.mark.parametrize("bs", [4096, 65536])
def test_device_round_trip_at_scale(cw, comp, bs):
    import torch
    total = (256 if comp == "lz4" else 48) << 20   # LZF runs one block per CU: keep its share small
    nb = total // bs
    src = torch.from_numpy(_mixed_corpus(total)).cuda()
    stride = (cw.compress_bound(comp, bs) + 15) // 16 * 16
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    back = torch.zeros(nb * bs, dtype=torch.uint8, device="cuda")
    status = torch.full((nb,), 7, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    cw.dev_compress(comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
    cw.dev_decompress(comp, dst.data_ptr(), stride, sizes.data_ptr(), nb, back.data_ptr(), bs, status.data_ptr(), s)
    torch.cuda.synchronize()
    fits = sizes != 0                      # LZF: 0 = did not fit l-1 (stored raw by the caller)
    if comp == "lz4":
        assert bool(fits.all())
    assert bool((status[fits] == 0).all()) and bool((status[~fits] == 1).all())
    ok = (back.view(nb, bs) == src.view(nb, bs)).all(dim=1)
    assert bool(ok[fits].all())
    ratio = total / float(torch.where(fits, sizes, torch.full_like(sizes, bs)).sum().item())
    assert 1.2 < ratio < 3.0


def test_decoders_accept_oracle_streams_and_reject_garbage(cw, oracle):
    import torch
    s = torch.cuda.current_stream().cuda_stream
    blocks = [corpus_file("alice29.txt")[i * 4096:(i + 1) * 4096] for i in range(8)] + [bytes(4096), bytes(range(256)) * 16]
    for comp, enc in (("lz4", oracle.lz4_compress), ("lzf", oracle.lzf_compress)):
        stride = 4128
        buf = np.zeros((len(blocks) + 2, stride), dtype=np.uint8)
        sz = np.zeros(len(blocks) + 2, dtype=np.int32)
        for i, b in enumerate(blocks):
            c = enc(b)
            buf[i, :len(c)] = np.frombuffer(c, dtype=np.uint8)
            sz[i] = len(c)
        # two corrupted streams: truncated, and an offset pointing before the start of the block
        c = enc(blocks[0])
        buf[-2, :len(c) - 7] = np.frombuffer(c[:-7], dtype=np.uint8); sz[-2] = len(c) - 7
        bad = (bytes([0x10, 65, 0xFF, 0x7F]) + bytes(20)) if comp == "lz4" else bytes([0, 65, 0xFF, 0xFF, 0]) + bytes(20)
        buf[-1, :len(bad)] = np.frombuffer(bad, dtype=np.uint8); sz[-1] = len(bad)
        d_buf, d_sz = torch.from_numpy(buf).cuda(), torch.from_numpy(sz).cuda()
        back = torch.zeros((len(sz), 4096), dtype=torch.uint8, device="cuda")
        status = torch.full((len(sz),), 7, dtype=torch.int32, device="cuda")
        cw.dev_decompress(comp, d_buf.data_ptr(), stride, d_sz.data_ptr(), len(sz), back.data_ptr(), 4096, status.data_ptr(), s)
        torch.cuda.synchronize()
        st, hb = status.cpu().numpy(), back.cpu().numpy()
        assert list(st[:len(blocks)]) == [0] * len(blocks) and list(st[-2:]) == [1, 1]
        for i, b in enumerate(blocks):
            assert hb[i].tobytes() == b


@pytest.mark.parametrize("alg", ["lz4", "lzf"])
def test_host_decompress_blocks_and_slots(alg):
    """cw_decompress_blocks / cw_decompress_lz4 / cw_decompress_lzf invert the compressors (host buffers)."""
    import compute_war_amd as cw
    cw.init(0)
    bs = 4096
    data = (corpus_file("lcet10.txt")[:40 * bs] + bytes(bs) + corpus_file("ptt5")[:23 * bs])
    sizes, payload = cw.compress_blocks(alg, data, bs)
    out, status = cw.decompress_blocks(alg, sizes, payload, bs)
    for i in range(len(data) // bs):
        if sizes[i] == 0:      # lzf: did not fit
            continue
        assert status[i] == 0 and out[i].tobytes() == data[i * bs:(i + 1) * bs]
    cw.set_block_size(bs)
    blk = data[3 * bs:4 * bs]
    comp = payload[3, :sizes[3]].tobytes()
    assert cw.do_decompression(alg, comp, 2 * bs) == blk
    assert cw.do_decompression(alg, comp[:-3], 2 * bs) == b""          # truncated slot = malformed
    assert cw.do_decompression(alg, comp, bs - 1) == b""               # capacity below the block size


@pytest.mark.parametrize("nb", [1, 63, 4096, 4097, 70001])
def test_dev_pack_stream_and_index(cw, nb):
    """N4: packed stream + u64 block index from fixed-stride slots (ragged sizes, zeros, several scan tiles)."""
    import torch
    rng = np.random.default_rng(nb)
    stride = 272
    sizes = rng.integers(0, stride + 1, nb, dtype=np.uint32)
    sizes[rng.integers(0, nb, max(1, nb // 7))] = 0          # "did not fit" blocks occupy nothing
    slots = rng.integers(0, 256, (nb, stride), dtype=np.uint8)
    d_slots, d_sizes = torch.from_numpy(slots).cuda(), torch.from_numpy(sizes.view(np.int32)).cuda()
    d_off = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
    d_out = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    cw.dev_pack(d_slots.data_ptr(), stride, d_sizes.data_ptr(), nb, d_out.data_ptr(), d_off.data_ptr(), s)
    torch.cuda.synchronize()
    want_off = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    assert np.array_equal(d_off.cpu().numpy(), want_off)
    want = np.concatenate([slots[i, :sizes[i]] for i in range(nb)]) if sizes.sum() else np.zeros(0, np.uint8)
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:len(want)], want) and not got[len(want):].any()
    # index only
    d_off.zero_()
    cw.dev_pack(0, stride, d_sizes.data_ptr(), nb, 0, d_off.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.array_equal(d_off.cpu().numpy(), want_off)
