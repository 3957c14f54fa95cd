"""compute-war_amd: MI355X-native back end for compute-war's per-block hash + compress hot path.

The product is ``libcwhc.so`` (hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/cw_hashcompress.h``).  This package is the thin host-side mirror of the reference's operator
interface (``doHashing`` / ``doCompression`` slots, ``HashOffload``) over that ABI via ctypes.
There is no CPU fallback: without the built library and a gfx950 device every compute call raises.
"""
from ._lib import (COMP_LZ4, COMP_LZF, COMP_NONE, HASH_NONE, HASH_SHA256, HASH_SKEIN256_128, HASH_SKEIN512,
                   CwError, lib, lib_path)
from .ops import (HashOffload, compress_blocks, compress_bound, decompress_blocks, do_decompression, dev_compress, dev_decompress, dev_gen_mixed, dev_gen_random, dev_hash, dev_hash_tree, dev_pack, hash_tree_blocks,
                  dev_hash_and_compress, dev_sum_sizes, digest_bytes, do_compression, do_hashing,
                  hash_and_compress_blocks, hash_and_compress_packed, hash_blocks, init, profile_enable, profile_kernels, profile_read, tune_reset, tune_set, tuned,
                  set_block_size, set_device, get_device, device_count,
                  shutdown)

__all__ = [n for n in dir() if not n.startswith("_")]
