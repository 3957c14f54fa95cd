"""Host-side mirror of the reference's operator interface over the C ABI.

Names follow the reference: ``do_hashing(src, dst, count)`` / ``do_compression(src, dst, len)`` are the two
function slots of src/hashandcompress/HashAndCompress.cpp:111,119; ``HashOffload`` is HashOffload.h:13-64.
The ``dev_*`` functions take raw device pointers (ints, e.g. ``torch.Tensor.data_ptr()``) and a HIP stream
handle; torch is only ever used by callers for memory and streams, never here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import COMP_LZ4, COMP_LZF, COMP_NONE, HASH_NONE, HASH_SKEIN256_128, HASH_SKEIN512, HASH_SHA256, check, lib

_HASH_NAMES = {"skein": HASH_SKEIN256_128, "skein512": HASH_SKEIN512, "sha256mb": HASH_SHA256, "sha256": HASH_SHA256}
_COMP_NAMES = {"lz4": COMP_LZ4, "lzf": COMP_LZF}


def _hash_id(alg) -> int:
    return _HASH_NAMES[alg] if isinstance(alg, str) else int(alg)


def _comp_id(alg) -> int:
    return _COMP_NAMES[alg] if isinstance(alg, str) else int(alg)


def init(device: int = 0) -> None:
    """initializeGpu() (HashAndCompress.cpp:95-98)."""
    check(lib().cw_init(device))


def shutdown() -> None:
    lib().cw_shutdown()


def set_device(device: int) -> None:
    """The calling thread's device for its next calls (initialises it if needed)."""
    check(lib().cw_set_device(device))


def get_device() -> int:
    return int(lib().cw_get_device())


def device_count() -> int:
    return int(lib().cw_device_count())


def set_block_size(n: int) -> None:
    lib().cw_set_block_size(n)


def digest_bytes(alg) -> int:
    return int(lib().cw_digest_bytes(_hash_id(alg)))


def compress_bound(alg, block_bytes: int) -> int:
    return int(lib().cw_compress_bound(_comp_id(alg), block_bytes))


def _np_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data.reshape(-1).view(np.uint8))
    return np.frombuffer(bytes(data), dtype=np.uint8)


# ---- the two slots -------------------------------------------------------------------------------
def do_hashing(alg, src, count: int, block_bytes: int | None = None) -> bytes:
    """doHashing(src, dst, count): `count` consecutive blocks -> `count` consecutive digests."""
    a = _np_u8(src)
    if block_bytes is not None:
        set_block_size(block_bytes)
    hid = _hash_id(alg)
    out = np.zeros(count * digest_bytes(hid), dtype=np.uint8)
    fn = {HASH_SKEIN256_128: lib().cw_hash_skein, HASH_SKEIN512: lib().cw_hash_skein512,
          HASH_SHA256: lib().cw_hash_sha256mb}[hid]
    if a.size < count * int(lib().cw_get_block_size()):
        raise ValueError("src shorter than count * block size")
    fn(a.ctypes.data, out.ctypes.data, count)
    return out.tobytes()


def do_compression(alg, src) -> bytes:
    """doCompression(src, dst, len) with the reference's dst capacity (2*len for lz4, len-1 for lzf);
    b'' when the codec reports 0 (did not fit)."""
    a = _np_u8(src)
    cid = _comp_id(alg)
    out = np.zeros(max(2 * a.size, 16), dtype=np.uint8)
    fn = lib().cw_compress_lz4 if cid == COMP_LZ4 else lib().cw_compress_lzf
    n = fn(a.ctypes.data, out.ctypes.data, a.size)
    return out[:n].tobytes()


# ---- batched host API ----------------------------------------------------------------------------
def hash_blocks(alg, src, block_bytes: int) -> np.ndarray:
    a = _np_u8(src)
    n = a.size // block_bytes if block_bytes else 0
    hid = _hash_id(alg)
    out = np.zeros((n, digest_bytes(hid)), dtype=np.uint8)
    check(lib().cw_hash_blocks(hid, a.ctypes.data, block_bytes, n, out.ctypes.data))
    return out


def hash_tree_blocks(alg, src, block_bytes: int, leaf: int, node: int, max_level: int):
    """Skein tree digests of every block of `src` (cw_hash_tree_blocks): [n, digest_bytes] uint8."""
    a = _np_u8(src)
    n = a.size // block_bytes if block_bytes else 0
    hid = _hash_id(alg)
    digests = np.zeros((n, digest_bytes(hid)), dtype=np.uint8)
    check(lib().cw_hash_tree_blocks(hid, a.ctypes.data, block_bytes, n, leaf, node, max_level, digests.ctypes.data))
    return digests


def dev_hash_tree(alg, d_src: int, block_bytes: int, nblocks: int, leaf: int, node: int, max_level: int, d_digests: int,
                  stream: int = 0, src_stride: int = 0) -> None:
    check(lib().cw_dev_hash_tree(_hash_id(alg), d_src, block_bytes, src_stride or block_bytes, nblocks, leaf, node, max_level,
                                 d_digests, stream))


def compress_blocks(alg, src, block_bytes: int):
    """Returns (sizes[n] uint32, payload[n, stride] uint8)."""
    a = _np_u8(src)
    n = a.size // block_bytes
    cid = _comp_id(alg)
    stride = compress_bound(cid, block_bytes)
    payload = np.zeros((n, stride), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    check(lib().cw_compress_blocks(cid, a.ctypes.data, block_bytes, n, payload.ctypes.data, stride, sizes.ctypes.data))
    return sizes, payload


def decompress_blocks(alg, sizes, payload, block_bytes: int):
    """Inverse of compress_blocks: (blocks[n, block_bytes] uint8, status[n] uint32; 0 = ok)."""
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    n, stride = payload.shape
    out = np.zeros((n, block_bytes), dtype=np.uint8)
    status = np.ones(n, dtype=np.uint32)
    check(lib().cw_decompress_blocks(_comp_id(alg), payload.ctypes.data, stride, sizes.ctypes.data, n, out.ctypes.data,
                                     block_bytes, status.ctypes.data))
    return out, status


def do_decompression(alg, comp: bytes, cap: int) -> bytes:
    """LZ4_decompress_safe / lzf_decompress slot (experiment.cpp:118,256): b"" on a malformed slot."""
    src = np.frombuffer(comp, dtype=np.uint8)
    dst = np.zeros(cap, dtype=np.uint8)
    if _comp_id(alg) == 0:
        n = lib().cw_decompress_lz4(src.ctypes.data, dst.ctypes.data, len(comp), cap)
    else:
        n = lib().cw_decompress_lzf(src.ctypes.data, len(comp), dst.ctypes.data, cap)
    return dst[:n].tobytes() if n > 0 else b""


def hash_and_compress_blocks(hash_alg, comp_alg, src, block_bytes: int):
    """ProcessBlock (:231-261) over every block of `src`: (digests, sizes, payload)."""
    a = _np_u8(src)
    n = a.size // block_bytes
    hid, cid = _hash_id(hash_alg), _comp_id(comp_alg)
    stride = compress_bound(cid, block_bytes)
    digests = np.zeros((n, digest_bytes(hid)), dtype=np.uint8)
    payload = np.zeros((n, stride), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    check(lib().cw_hash_and_compress_blocks(hid, cid, a.ctypes.data, block_bytes, n, digests.ctypes.data,
                                            payload.ctypes.data, stride, sizes.ctypes.data))
    return digests, sizes, payload


def hash_and_compress_packed(hash_alg, comp_alg, src, block_bytes: int, pinned: bool = False):
    """cw_hash_and_compress_packed: (digests[n, db], sizes[n], offsets[n + 1], packed[total] uint8).  pinned=True takes the
    input and the packed output through page-locked buffers (cw_host_alloc), the zero-copy form of the pipeline."""
    a = _np_u8(src)
    n = a.size // block_bytes
    hid, cid = _hash_id(hash_alg), _comp_id(comp_alg)
    cap = max(n * compress_bound(cid, block_bytes), 1)
    digests = np.zeros((n, digest_bytes(hid)), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    if pinned:
        hp, hs = lib().cw_host_alloc(cap), lib().cw_host_alloc(max(a.size, 1))
        if not hp or not hs:
            raise _lib.CwError(-5, lib().cw_last_error().decode())
        try:
            C.memmove(hs, a.ctypes.data, a.size)
            check(lib().cw_hash_and_compress_packed(hid, cid, hs, block_bytes, n, digests.ctypes.data, hp, cap, offsets.ctypes.data,
                                                    sizes.ctypes.data))
            packed = np.ctypeslib.as_array((C.c_uint8 * int(offsets[n])).from_address(hp)).copy() if offsets[n] else np.zeros(0, np.uint8)
        finally:
            lib().cw_host_free(hp)
            lib().cw_host_free(hs)
    else:
        buf = np.zeros(cap, dtype=np.uint8)
        check(lib().cw_hash_and_compress_packed(hid, cid, a.ctypes.data, block_bytes, n, digests.ctypes.data, buf.ctypes.data, cap,
                                                offsets.ctypes.data, sizes.ctypes.data))
        packed = buf[: int(offsets[n])].copy()
    return digests, sizes, offsets, packed


# ---- device-resident API (raw pointers) ------------------------------------------------------------
def dev_hash(alg, d_src: int, block_bytes: int, nblocks: int, d_digests: int, stream: int = 0,
             src_stride: int | None = None) -> None:
    check(lib().cw_dev_hash(_hash_id(alg), d_src, block_bytes, src_stride or block_bytes, nblocks, d_digests, stream))


def dev_compress(alg, d_src: int, block_bytes: int, nblocks: int, d_dst: int, dst_stride: int, d_sizes: int,
                 stream: int = 0, src_stride: int | None = None) -> None:
    check(lib().cw_dev_compress(_comp_id(alg), d_src, block_bytes, src_stride or block_bytes, nblocks, d_dst, dst_stride,
                                d_sizes, stream))


def dev_hash_and_compress(hash_alg, comp_alg, d_src: int, block_bytes: int, nblocks: int, d_digests: int, d_dst: int,
                          dst_stride: int, d_sizes: int, stream: int = 0, src_stride: int | None = None) -> None:
    check(lib().cw_dev_hash_and_compress(_hash_id(hash_alg), _comp_id(comp_alg), d_src, block_bytes,
                                         src_stride or block_bytes, nblocks, d_digests, d_dst, dst_stride, d_sizes, stream))


def dev_decompress(alg, d_comp: int, comp_stride: int, d_sizes: int, nblocks: int, d_dst: int, block_bytes: int,
                   d_status: int, stream: int = 0) -> None:
    check(lib().cw_dev_decompress(_comp_id(alg), d_comp, comp_stride, d_sizes, nblocks, d_dst, block_bytes, d_status, stream))


def dev_pack(d_slots: int, slot_stride: int, d_sizes: int, nblocks: int, d_packed: int, d_offsets: int, stream: int = 0) -> None:
    """Packed stream + u64 block index (nblocks + 1 offsets) from fixed-stride slots; d_packed = 0: index only."""
    check(lib().cw_dev_pack(d_slots, slot_stride, d_sizes, nblocks, d_packed or None, d_offsets, stream))


def dev_gen_random(seed: int, first_block: int, nblocks: int, block_bytes: int, d_dst: int, stream: int = 0) -> None:
    check(lib().cw_dev_gen_random(seed, first_block, nblocks, block_bytes, d_dst, stream))


def dev_gen_mixed(seed: int, first_block: int, nblocks: int, block_bytes: int, d_dst: int, stream: int = 0) -> None:
    """SURVEY 8(d)'s compressible mix: even blocks random, odd blocks a 64-byte motif with 1/16 of the bytes mutated."""
    check(lib().cw_dev_gen_mixed(seed, first_block, nblocks, block_bytes, d_dst, stream))


def dev_sum_sizes(d_sizes: int, nblocks: int, raw_bytes: int, d_totals: int, stream: int = 0) -> None:
    check(lib().cw_dev_sum_sizes(d_sizes, nblocks, raw_bytes, d_totals, stream))


def profile_enable(on: bool = True) -> None:
    lib().cw_profile_enable(1 if on else 0)


def profile_read(reset: bool = True) -> dict:
    """{'codec': (ms_sum, launches), 'hash': ..., 'other': ...} from the library's own HIP events."""
    ms = (C.c_double * 3)()
    cnt = (C.c_uint * 3)()
    check(lib().cw_profile_read(ms, cnt, 1 if reset else 0))
    return {k: (ms[i], cnt[i]) for i, k in enumerate(("codec", "hash", "other"))}


def tune_set(key: str, value=None) -> None:
    """cw_tune_set: a knob's value for the calls that follow (None removes the override; the environment stays the default)."""
    check(lib().cw_tune_set(key.encode(), None if value is None else str(value).encode()))


def tune_reset() -> None:
    lib().cw_tune_reset()


class tuned:
    """with tuned(CW_LZ4_LANES=1, ...): the knobs hold inside the block and are removed afterwards."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        for k, v in self.knobs.items():
            tune_set(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.knobs:
            tune_set(k, None)
        return False


def profile_kernels() -> dict:
    """Names of the kernels the calling thread's latest codec / hash launch used (as rocprofv3 prints them)."""
    out = {}
    for i, k in enumerate(("codec", "hash")):
        buf = C.create_string_buffer(256)
        check(lib().cw_profile_kernels(i, buf, 256))
        out[k] = buf.value.decode()
    return out


# ---- HashOffload ------------------------------------------------------------------------------------
class HashOffload:
    """HashOffload.h:13-64: Reset(data, results, onComplete) -> Enqueue() -> Start() -> Complete()."""

    hInit, hQueued, hOffloaded, hComplete, hFailed = 0, 1, 2, 3, 4

    def __init__(self, n_blocks: int, alg="skein", block_bytes: int = 4096):
        self._h = lib().cw_offload_create(_hash_id(alg), n_blocks, block_bytes)
        if not self._h:
            raise _lib.CwError(-2, lib().cw_last_error().decode())
        self.n_blocks, self.block_bytes, self.alg = n_blocks, block_bytes, _hash_id(alg)
        self._keep = None

    def Reset(self, data: np.ndarray, results: np.ndarray, on_complete=None) -> None:
        cb = _lib.ON_COMPLETE((lambda _arg: on_complete()) if on_complete else (lambda _arg: None))
        self._keep = (data, results, cb)
        check(lib().cw_offload_reset(self._h, data.ctypes.data, results.ctypes.data, cb, None))

    def Enqueue(self) -> None:
        check(lib().cw_offload_enqueue(self._h))

    def Start(self) -> None:
        check(lib().cw_offload_start(self._h))

    def Complete(self) -> None:
        check(lib().cw_offload_complete(self._h))

    def Completed(self) -> bool:
        return bool(lib().cw_offload_completed(self._h))

    def DoOffload(self) -> None:
        check(lib().cw_offload_do(self._h))

    def Submit(self) -> None:
        """Enqueue + hand to the offload thread (hashing_offload_entry_point, :160-183)."""
        check(lib().cw_offload_submit(self._h))

    @property
    def state(self) -> int:
        return int(lib().cw_offload_state(self._h))

    @property
    def error(self) -> int:
        """CW_OK, or the status code that put the object into hFailed."""
        return int(lib().cw_offload_error(self._h))

    def close(self) -> None:
        if self._h:
            lib().cw_offload_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
