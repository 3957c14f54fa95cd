"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package -- as the checker / reported baseline, never as the product path.
The product (``compute_war_amd``) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcw_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libskein_ref.so")

HASH_SKEIN512, HASH_SKEIN256_128, HASH_SHA256, HASH_NONE = 0, 1, 2, 3
COMP_LZ4, COMP_LZF, COMP_NONE = 0, 1, 2
DIGEST_BYTES = {HASH_SKEIN512: 64, HASH_SKEIN256_128: 16, HASH_SHA256: 32, HASH_NONE: 0}


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or not os.path.exists(_REF_PATH):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.c_void_p
        L.cw_oracle_skein512.argtypes = [u8p, C.c_size_t, C.c_uint, u8p]
        L.cw_oracle_skein256.argtypes = [u8p, C.c_size_t, C.c_uint, u8p]
        L.cw_oracle_skein_iv.argtypes = [C.c_int, C.c_uint, u8p]
        L.cw_oracle_skein_tree.argtypes = [C.c_int, u8p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint, u8p]
        L.cw_oracle_skein_tree.restype = C.c_int
        L.cw_oracle_threefish_trace.argtypes = [C.c_int, u8p, u8p, u8p, u8p]
        L.cw_oracle_sha256.argtypes = [u8p, C.c_size_t, u8p]
        for f in (L.cw_oracle_lz4_compress, L.cw_oracle_lzf_compress):
            f.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
            f.restype = C.c_size_t
        for f in (L.cw_oracle_lz4_decompress, L.cw_oracle_lzf_decompress):
            f.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
            f.restype = C.c_long
        L.cw_oracle_lz4_bound.argtypes = [C.c_size_t]
        L.cw_oracle_lz4_bound.restype = C.c_size_t
        L.cw_oracle_gen_random_blocks.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, u8p]
        L.cw_oracle_gen_mixed_blocks.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, u8p]
        L.cw_oracle_digest_bytes.argtypes = [C.c_int]
        L.cw_oracle_digest_bytes.restype = C.c_size_t
        L.cw_oracle_hash_and_compress.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                                  u8p, u8p, C.c_size_t, u8p]
        L.cw_oracle_hash_and_compress.restype = C.c_double
        _lib = L
    return _lib


def _buf(data) -> np.ndarray:
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def skein512(data, hash_bits: int = 512, msg_bits: int | None = None) -> bytes:
    a = _buf(data)
    out = np.zeros((hash_bits + 7) // 8, dtype=np.uint8)
    lib().cw_oracle_skein512(_ptr(a), a.size * 8 if msg_bits is None else msg_bits, hash_bits, _ptr(out))
    return out.tobytes()


def skein256(data, hash_bits: int = 128, msg_bits: int | None = None) -> bytes:
    a = _buf(data)
    out = np.zeros((hash_bits + 7) // 8, dtype=np.uint8)
    lib().cw_oracle_skein256(_ptr(a), a.size * 8 if msg_bits is None else msg_bits, hash_bits, _ptr(out))
    return out.tobytes()


def skein_tree(state_bits: int, data, hash_bits: int, leaf: int, node: int, max_level: int) -> bytes:
    """Skein tree hashing (skein_test.c:616-680): state_bits 256 or 512, byte-granular message."""
    a = _buf(data)
    out = np.zeros((hash_bits + 7) // 8, dtype=np.uint8)
    rc = lib().cw_oracle_skein_tree(state_bits // 64, _ptr(a), a.size, hash_bits, leaf, node, max_level, _ptr(out))
    if rc != 0:
        raise ValueError("bad tree parameters")
    return out.tobytes()


def skein_iv(state_words: int, hash_bits: int) -> np.ndarray:
    iv = np.zeros(state_words, dtype=np.uint64)
    lib().cw_oracle_skein_iv(state_words, hash_bits, _ptr(iv))
    return iv


def threefish_trace(state_words: int, key, tweak, block: bytes):
    """(states [1 + 72 + 18, state_words] uint64 in execution order, chaining output [state_words])."""
    k = np.array(key, dtype=np.uint64).copy()
    t = np.array(tweak, dtype=np.uint64)
    b = _buf(block)
    tr = np.zeros((91, state_words), dtype=np.uint64)
    lib().cw_oracle_threefish_trace(state_words, _ptr(k), _ptr(t), _ptr(b), _ptr(tr))
    return tr, k


def sha256(data) -> bytes:
    a = _buf(data)
    out = np.zeros(32, dtype=np.uint8)
    lib().cw_oracle_sha256(_ptr(a), a.size, _ptr(out))
    return out.tobytes()


def lz4_bound(n: int) -> int:
    return int(lib().cw_oracle_lz4_bound(n))


def lz4_compress(data) -> bytes:
    """LZ4_compress_default(src, dst, n, 2n) -- 2n raised to the bound for tiny n."""
    a = _buf(data)
    cap = max(2 * a.size, lz4_bound(a.size))
    out = np.zeros(cap, dtype=np.uint8)
    c = lib().cw_oracle_lz4_compress(_ptr(a), a.size, _ptr(out), cap)
    return out[:c].tobytes()


def lz4_decompress(data, cap: int) -> bytes | None:
    a = _buf(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    r = lib().cw_oracle_lz4_decompress(_ptr(a), a.size, _ptr(out), cap)
    return None if r < 0 else out[:r].tobytes()


def lzf_compress(data, cap: int | None = None) -> bytes:
    """lzf_compress(in, n, out, n-1); b'' means "did not fit" (returns 0)."""
    a = _buf(data)
    cap = a.size - 1 if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    c = lib().cw_oracle_lzf_compress(_ptr(a), a.size, _ptr(out), max(cap, 0))
    return out[:c].tobytes()


def lzf_decompress(data, cap: int) -> bytes | None:
    a = _buf(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    r = lib().cw_oracle_lzf_decompress(_ptr(a), a.size, _ptr(out), cap)
    return None if r < 0 else out[:r].tobytes()


def gen_random_blocks(seed: int, first_block: int, nblocks: int, block_bytes: int) -> np.ndarray:
    out = np.zeros(nblocks * block_bytes, dtype=np.uint8)
    lib().cw_oracle_gen_random_blocks(seed, first_block, nblocks, block_bytes, _ptr(out))
    return out


def gen_mixed_blocks(seed: int, first_block: int, nblocks: int, block_bytes: int) -> np.ndarray:
    out = np.zeros(nblocks * block_bytes, dtype=np.uint8)
    lib().cw_oracle_gen_mixed_blocks(seed, first_block, nblocks, block_bytes, _ptr(out))
    return out


def hash_and_compress(src: np.ndarray, block_bytes: int, hash_alg: int, comp_alg: int, threads: int = 1,
                      want_payload: bool = True):
    """Returns (seconds, digests[n,db] | None, sizes[n] | None, payload[n,stride] | None)."""
    src = _buf(src)
    n = src.size // block_bytes
    db = DIGEST_BYTES[hash_alg]
    digests = np.zeros((n, db), dtype=np.uint8) if db else None
    sizes = payload = None
    stride = 0
    if comp_alg != COMP_NONE:
        sizes = np.zeros(n, dtype=np.uint32)
        if want_payload:
            stride = max(2 * block_bytes, lz4_bound(block_bytes))
            payload = np.zeros((n, stride), dtype=np.uint8)
    secs = lib().cw_oracle_hash_and_compress(
        _ptr(src), n, block_bytes, hash_alg, comp_alg, threads,
        _ptr(digests) if digests is not None else None,
        _ptr(payload) if payload is not None else None, stride,
        _ptr(sizes) if sizes is not None else None)
    return secs, digests, sizes, payload


# ---- oracle/_ref: the reference's own Skein C, compiled where it lies ---------------------------
_ref = None


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def _ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_PATH)
    return _ref


def _ref_skein(prefix: str, data, hash_bits: int) -> bytes:
    """Init/Update/Final of the reference build (Optimized_64bit/skein.c:29-68,131-210 / :226-408)."""
    L = _ref_lib()
    a = _buf(data)
    ctx = C.create_string_buffer(512)  # larger than any Skein_*_Ctxt_t
    out = np.zeros(max((hash_bits + 7) // 8, 128), dtype=np.uint8)
    init, upd, fin = (getattr(L, f"{prefix}_{s}") for s in ("Init", "Update", "Final"))
    init.argtypes = [C.c_void_p, C.c_size_t]
    upd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fin.argtypes = [C.c_void_p, C.c_void_p]
    init(ctx, hash_bits)
    upd(ctx, _ptr(a), a.size)
    fin(ctx, _ptr(out))
    return out[: (hash_bits + 7) // 8].tobytes()


def ref_skein512(data, hash_bits: int = 512) -> bytes:
    return _ref_skein("Skein_512", data, hash_bits)


def ref_skein256(data, hash_bits: int = 128) -> bytes:
    return _ref_skein("Skein_256", data, hash_bits)
