"""ctypes binding of include/cw_hashcompress.h.  Fails loudly when the HIP library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

HASH_SKEIN512, HASH_SKEIN256_128, HASH_SHA256, HASH_NONE = 0, 1, 2, 3
COMP_LZ4, COMP_LZF, COMP_NONE = 0, 1, 2

# every symbol include/cw_hashcompress.h declares (tests/test_abi.py checks the header against this)
ABI_SYMBOLS = [
    "cw_init", "cw_shutdown", "cw_device_count", "cw_set_device", "cw_get_device", "cw_last_error", "cw_version",
    "cw_digest_bytes", "cw_compress_bound",
    "cw_set_block_size", "cw_get_block_size",
    "cw_hash_skein", "cw_hash_skein512", "cw_hash_sha256mb", "cw_compress_lz4", "cw_compress_lzf",
    "cw_decompress_lz4", "cw_decompress_lzf",
    "cw_hash_tree_blocks", "cw_dev_hash_tree",
    "cw_hash_blocks", "cw_compress_blocks", "cw_hash_and_compress_blocks", "cw_decompress_blocks",
    "cw_hash_and_compress_packed", "cw_prepare", "cw_host_alloc", "cw_host_free", "cw_host_register", "cw_host_unregister",
    "cw_dev_hash", "cw_dev_compress", "cw_dev_hash_and_compress", "cw_dev_gen_random", "cw_dev_gen_mixed", "cw_dev_sum_sizes",
    "cw_dev_decompress", "cw_dev_pack", "cw_dev_alloc", "cw_dev_free", "cw_dev_upload", "cw_dev_download", "cw_dev_synchronize", "cw_profile_enable", "cw_profile_read", "cw_profile_kernels",
    "cw_tune_set", "cw_tune_reset",
    "cw_offload_create", "cw_offload_destroy", "cw_offload_reset", "cw_offload_enqueue", "cw_offload_start",
    "cw_offload_complete", "cw_offload_completed", "cw_offload_state", "cw_offload_error", "cw_offload_do",
    "cw_offload_thread_start", "cw_offload_submit", "cw_offload_thread_stop",
    "cw_shard_range", "cw_mgpu_create", "cw_mgpu_destroy", "cw_mgpu_ndev", "cw_mgpu_device", "cw_mgpu_last_error", "cw_mgpu_gather",
]


class CwError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcwhc error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    # CW_LIB lets kernel experiments load an alternative build of the same ABI
    return os.environ.get("CW_LIB") or os.path.join(_HERE, "libcwhc.so")


_lib = None
ON_COMPLETE = C.CFUNCTYPE(None, C.c_void_p)


def _share_hip_runtime_with_torch() -> None:
    """A PyTorch-ROCm wheel bundles its own libamdhip64/libhsa-runtime64.  If libcwhc.so pulled in
    /opt/rocm's copies first, a later ``import torch`` would load a SECOND HSA runtime into the process and
    see no GPU.  When such a wheel is installed (and torch is not imported yet), load its runtime first so
    both sides share one; without torch the system ROCm runtime is used as linked."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib() -> C.CDLL:
    """Load libcwhc.so (built by __graft_entry__.build() / make -C compute_war_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CwError(-1, f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                          "g.build()'); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.c_void_p
    sigs = {
        "cw_init": ([C.c_int], C.c_int), "cw_shutdown": ([], None), "cw_device_count": ([], C.c_int),
        "cw_set_device": ([C.c_int], C.c_int), "cw_get_device": ([], C.c_int),
        "cw_last_error": ([], C.c_char_p), "cw_version": ([], C.c_char_p),
        "cw_digest_bytes": ([C.c_int], sz), "cw_compress_bound": ([C.c_int, sz], sz),
        "cw_set_block_size": ([sz], None), "cw_get_block_size": ([], sz),
        "cw_hash_skein": ([vp, vp, C.c_int], None), "cw_hash_skein512": ([vp, vp, C.c_int], None),
        "cw_hash_sha256mb": ([vp, vp, C.c_int], None),
        "cw_compress_lz4": ([vp, vp, sz], sz), "cw_compress_lzf": ([vp, vp, sz], sz),
        "cw_decompress_lz4": ([vp, vp, C.c_int, C.c_int], C.c_int),
        "cw_decompress_lzf": ([vp, C.c_uint, vp, C.c_uint], C.c_uint),
        "cw_decompress_blocks": ([C.c_int, vp, sz, u32p, sz, vp, sz, u32p], C.c_int),
        "cw_hash_tree_blocks": ([C.c_int, vp, sz, sz, C.c_uint, C.c_uint, C.c_uint, vp], C.c_int),
        "cw_dev_hash_tree": ([C.c_int, vp, sz, sz, sz, C.c_uint, C.c_uint, C.c_uint, vp, vp], C.c_int),
        "cw_hash_blocks": ([C.c_int, vp, sz, sz, vp], C.c_int),
        "cw_compress_blocks": ([C.c_int, vp, sz, sz, vp, sz, u32p], C.c_int),
        "cw_hash_and_compress_blocks": ([C.c_int, C.c_int, vp, sz, sz, vp, vp, sz, u32p], C.c_int),
        "cw_hash_and_compress_packed": ([C.c_int, C.c_int, vp, sz, sz, vp, vp, sz, vp, u32p], C.c_int),
        "cw_prepare": ([C.c_int, C.c_int, sz, sz, C.c_int], C.c_int),
        "cw_host_alloc": ([sz], vp), "cw_host_free": ([vp], None),
        "cw_host_register": ([vp, sz], C.c_int), "cw_host_unregister": ([vp], C.c_int),
        "cw_dev_hash": ([C.c_int, vp, sz, sz, sz, vp, vp], C.c_int),
        "cw_dev_compress": ([C.c_int, vp, sz, sz, sz, vp, sz, u32p, vp], C.c_int),
        "cw_dev_hash_and_compress": ([C.c_int, C.c_int, vp, sz, sz, sz, vp, vp, sz, u32p, vp], C.c_int),
        "cw_dev_gen_random": ([C.c_uint64, C.c_uint64, sz, sz, vp, vp], C.c_int),
        "cw_dev_gen_mixed": ([C.c_uint64, C.c_uint64, sz, sz, vp, vp], C.c_int),
        "cw_dev_sum_sizes": ([u32p, sz, C.c_uint32, vp, vp], C.c_int),
        "cw_dev_decompress": ([C.c_int, vp, sz, u32p, sz, vp, sz, u32p, vp], C.c_int),
        "cw_dev_pack": ([vp, sz, u32p, sz, vp, vp, vp], C.c_int),
        "cw_dev_alloc": ([sz], vp), "cw_dev_free": ([vp], None), "cw_dev_upload": ([vp, vp, sz], C.c_int),
        "cw_dev_download": ([vp, vp, sz], C.c_int), "cw_dev_synchronize": ([], C.c_int),
        "cw_profile_enable": ([C.c_int], None), "cw_profile_read": ([vp, vp, C.c_int], C.c_int),
        "cw_profile_kernels": ([C.c_int, vp, sz], C.c_int),
        "cw_tune_set": ([C.c_char_p, C.c_char_p], C.c_int), "cw_tune_reset": ([], None),
        "cw_offload_create": ([C.c_int, C.c_int, sz], vp), "cw_offload_destroy": ([vp], None),
        "cw_offload_reset": ([vp, vp, vp, ON_COMPLETE, vp], C.c_int),
        "cw_offload_enqueue": ([vp], C.c_int), "cw_offload_start": ([vp], C.c_int),
        "cw_offload_complete": ([vp], C.c_int), "cw_offload_completed": ([vp], C.c_int),
        "cw_offload_state": ([vp], C.c_int), "cw_offload_error": ([vp], C.c_int), "cw_offload_do": ([vp], C.c_int),
        "cw_offload_thread_start": ([], C.c_int), "cw_offload_submit": ([vp], C.c_int),
        "cw_offload_thread_stop": ([], None),
        "cw_shard_range": ([sz, C.c_int, C.c_int, vp, vp], None),
        "cw_mgpu_create": ([vp, C.c_int], vp), "cw_mgpu_destroy": ([vp], None), "cw_mgpu_ndev": ([vp], C.c_int),
        "cw_mgpu_device": ([vp, C.c_int], C.c_int), "cw_mgpu_last_error": ([], C.c_char_p),
        "cw_mgpu_gather": ([vp, vp, sz, vp, vp, sz], C.c_int),
    }
    assert set(sigs) == set(ABI_SYMBOLS)
    for name, (args, res) in sigs.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export the ABI
        fn.argtypes, fn.restype = args, res
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise CwError(rc, lib().cw_last_error().decode(errors="replace"))
