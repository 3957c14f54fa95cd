#!/usr/bin/env python3
"""Headline benchmark: GB/s ingested by the hash+compress hot path (Skein-512 + LZ4, 64 KiB blocks).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path (compress every block, then hash every block -- ProcessBlock,
src/hashandcompress/HashAndCompress.cpp:231-261) over this rank's blocks, which are generated on the device
beforehand and stay resident in HBM (the reference also keeps file reading outside its timed window,
:391-397).  Every rank owns a contiguous range of the global block index space (compute_war_amd/shard.py);
there is no collective on the data path, only the result gather (digests + byte totals) over RCCL.
Weak scaling: blocks per GPU are fixed (default 1 Mi x 64 KiB = 64 GiB per GPU).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel's algorithmic bytes / its HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline  the oracle's (CPU restatement of the reference path) throughput on this host, bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (imported before libcwhc.so so both share one HIP runtime)

import compute_war_amd as cw  # noqa: E402
from compute_war_amd.shard import gather_results, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
SEED = 0xC0FFEE


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--block-bytes", type=int, default=65536)
    ap.add_argument("--hash", default="skein512", choices=["skein512", "skein", "sha256mb"])
    ap.add_argument("--comp", default="lz4", choices=["lz4", "lzf"])
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--standalone", action="store_true",
                    help="after the timed region also launch each kernel alone (3x) and report its own roofline; off by default so "
                         "that a rocprofv3 summary of the default command holds fused launches only")
    return ap.parse_args()


def host_cpu_share() -> int:
    """CPUs this process may actually use: affinity mask, cgroup quota, and the pool's stated share of a
    1-GPU box (16) -- os.cpu_count() reports every core of the host."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, target_s: float):
    """Oracle (CPU restatement of the reference's worker loop) on a bounded sample of the same workload."""
    import oracle as O
    O.build()
    threads = host_cpu_share()
    bs = args.block_bytes
    h = {"skein512": O.HASH_SKEIN512, "skein": O.HASH_SKEIN256_128, "sha256mb": O.HASH_SHA256}[args.hash]
    c = {"lz4": O.COMP_LZ4, "lzf": O.COMP_LZF}[args.comp]
    # bounded sample: a fixed 2 GiB (or smaller) synthetic stream, passed over repeatedly until ~target_s seconds
    nb = max(threads, min((2 << 30) // bs, 1 << 20))
    data = O.gen_random_blocks(SEED, 0, nb, bs)
    secs, passes = 0.0, 0
    while secs < target_s and passes < 1000:
        t, *_ = O.hash_and_compress(data, bs, h, c, threads, want_payload=False)
        secs += t
        passes += 1
    nb *= passes
    return {
        "value": round(nb * bs / secs / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"{passes} passes over {nb // passes} x {bs} B synthetic random blocks ({nb // passes * bs / 2**20:.0f} MiB), "
                  f"{args.hash}+{args.comp}, oracle worker loop with {threads} threads, {secs:.1f} s",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
    cw.init(local_rank)

    bs, nb = args.block_bytes, args.blocks_per_gpu
    first, last = shard_range(nb * world, rank, world)
    assert last - first == nb
    db = cw.digest_bytes(args.hash)
    stride = (cw.compress_bound(args.comp, bs) + 15) // 16 * 16
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream

    src = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    # two digest / totals buffers: step i's result gather (RCCL, async) overlaps step i+1's kernels
    dig_bufs = [torch.zeros((nb, db), dtype=torch.uint8, device="cuda") for _ in range(2 if world > 1 else 1)]
    tot_bufs = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(len(dig_bufs))]
    digests = dig_bufs[0]
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    pending = [[] for _ in dig_bufs]  # outstanding RCCL work per buffer
    cw.dev_gen_random(SEED, first, nb, bs, src.data_ptr(), s)  # this rank's blocks of the global stream
    torch.cuda.synchronize()

    state = {"i": 0, "gathered": None}

    def step(timed: bool):
        cw.profile_enable(timed)  # the library brackets its own kernel launches with HIP events on their streams
        b = state["i"] % len(dig_bufs)
        state["i"] += 1
        for h in pending[b]:      # the gather that last read this buffer must be done before it is overwritten
            h.wait()
        dig, totals = dig_bufs[b], tot_bufs[b]
        totals.zero_()
        # codec + hash side by side (ProcessBlock, :243-257); see cw_dev_hash_and_compress for the stream layout
        cw.dev_hash_and_compress(args.hash, args.comp, src.data_ptr(), bs, nb, dig.data_ptr(), dst.data_ptr(),
                                 stride, sizes.data_ptr(), s)
        cw.dev_sum_sizes(sizes.data_ptr(), nb, bs, totals.data_ptr(), s)
        # the only exchange: gather digests + byte totals (no-op at N=1)
        all_d, all_t, pending[b] = gather_results(dig, totals, world, async_op=world > 1)
        state["gathered"] = (all_d, all_t, dig)

    def fence():
        for hs in pending:
            for h in hs:
                h.wait()
        if world > 1:
            import torch.distributed as dist
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    all_digests, all_totals, digests = state["gathered"]
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = cw.profile_read(reset=True)
    k_ms = {"comp": prof["codec"][0] / max(prof["codec"][1], 1), "hash": prof["hash"][0] / max(prof["hash"][1], 1)}

    # outside the timed region: each kernel on its own (3 launches), for the per-kernel rooflines
    solo_ms = {}
    cw.profile_enable(args.standalone)
    for _ in range(3 if args.standalone else 0):
        cw.dev_compress(args.comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
    torch.cuda.synchronize()
    p2 = cw.profile_read(reset=True)
    solo_ms["comp"] = p2["codec"][0] / max(p2["codec"][1], 1)
    for _ in range(3 if args.standalone else 0):
        cw.dev_hash(args.hash, src.data_ptr(), bs, nb, dig_bufs[0].data_ptr(), s)
    torch.cuda.synchronize()
    p2 = cw.profile_read(reset=True)
    solo_ms["hash"] = p2["hash"][0] / max(p2["hash"][1], 1)
    cw.profile_enable(False)
    if not args.standalone:
        solo_ms = {}
    total_blocks = nb * world
    bytes_out = int(all_totals[0].item())
    value = total_blocks * bs * args.steps / elapsed / 1e9

    # parity spot check (outside the timed region): sampled blocks against the CPU oracle
    spot = "skipped"
    if rank == 0:
        try:
            import oracle as O
            O.build()
            hs, hz = digests.cpu().numpy(), sizes.cpu().numpy()
            for i in (0, 1, nb // 2, nb - 1):
                blk = O.gen_random_blocks(SEED, first + i, 1, bs)
                want_d = {"skein512": lambda b: O.skein512(b, 512), "skein": lambda b: O.skein256(b, 128),
                          "sha256mb": O.sha256}[args.hash](blk)
                want_c = O.lz4_compress(blk) if args.comp == "lz4" else O.lzf_compress(blk)
                got_c = dst[i * stride:i * stride + int(hz[i])].cpu().numpy().tobytes()
                assert hs[i].tobytes() == want_d and got_c == want_c, f"block {i}"
            spot = "ok (4 sampled blocks bit-exact vs oracle)"
        except AssertionError as e:  # a wrong result must not pass silently as a benchmark number
            raise SystemExit(f"PARITY FAILURE in bench: {e}")

    if rank != 0:
        return
    # algorithmic bytes per block (DESIGN.md "Rooflines"): hash reads the block and writes its digest;
    # the codec reads the block and writes csize + 4
    csize = bytes_out / total_blocks
    alg_bytes = {"hash": bs + db, "comp": bs + csize + 4}
    # long Skein messages are hashed in 8 launches of cw::skein_slice_kernel (csrc/skein_kernels.hip);
    # k_ms["hash"] spans all of them, so "launches" says how to compare it with a per-launch average from rocprofv3
    nw = {"skein512": 8, "skein": 4}.get(args.hash, 0)
    sliced = nw and nb >= 4096 and bs % (nw * 8) == 0 and bs // (nw * 8) + 1 >= 256 and os.environ.get("CW_SKEIN_SLICED", "1")[0] != "0"
    launches = {"hash": 8 if sliced else 1, "comp": 1}
    names = {"hash": (f"cw::skein_slice_kernel<{nw},true>" if sliced else
                      {"skein512": "cw::skein_lines_kernel<8,true>", "skein": "cw::skein_lines_kernel<4,true>",
                       "sha256mb": "cw::sha256_blocks_kernel<true,false>"}[args.hash]),
             "comp": ("cw::lz4_scan_span_kernel (+ cw::lz4_parse_kernel on queued blocks)" if bs in (4096, 8192, 16384, 32768, 65536)
                      else "cw::lz4_scan_stream_kernel (+ cw::lz4_parse_kernel on queued blocks)") if args.comp == "lz4"
                     else "cw::lzf_links_kernel + cw::lzf_chain_kernel"}
    dom = max(k_ms, key=k_ms.get)
    kernels = {k: {"ms_per_step": round(k_ms[k], 3), "launches_per_step": launches[k], "ms_per_launch": round(k_ms[k] / launches[k], 3),
                   "alg_GBps": round(alg_bytes[k] * nb / (k_ms[k] / 1e3) / 1e9, 1),
                   "ingest_GBps": round(bs * nb / (k_ms[k] / 1e3) / 1e9, 1)} for k in k_ms}
    achieved = alg_bytes[dom] * nb / (k_ms[dom] / 1e3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes per launch, if measured
    if os.path.exists(tf):
        traffic = json.load(open(tf)).get(f"{dom}:{args.hash if dom == 'hash' else args.comp}:{bs}:{nb}")
    out = {
        "metric": "GB/s ingested (Skein-512 + LZ4, 64 KiB blocks)" if (args.hash, args.comp, bs) == ("skein512", "lz4", 65536)
                  else f"GB/s ingested ({args.hash} + {args.comp}, {bs} B blocks)",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64" if args.hash.startswith("skein") else "u32", "data": "synthetic",
        "config": {"workload": f"{args.hash}+{args.comp} over {nb} x {bs} B uniform-random blocks per GPU "
                               f"(splitmix64 stream, seed 0xC0FFEE), inputs resident in HBM",
                   "blocks_per_gpu": nb, "block_bytes": bs, "parallelism": f"block-sharded x{world}, gather-only RCCL"},
        "compression_ratio": round(total_blocks * bs / bytes_out, 4),
        "roofline": {"bound": "hbm", "kernel": names[dom],
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "launches_per_step": launches[dom], "ms_per_launch": round(k_ms[dom] / launches[dom], 3),
                     "alg_bytes_per_block": round(alg_bytes[dom], 1)},
        "kernels": dict(kernels, note="codec and hash run concurrently on two streams; durations overlap"),
        # the binding roof of the hash kernel is integer VALU issue, not HBM: instruction mix of one
        # Threefish-512 call x measured per-instruction cost = 2.65 us per wavefront-call per SIMD
        "valu_roofline": ({"kernel": names["hash"], "achieved": round(bs * nb / (k_ms["hash"] / 1e3) / 1e9, 1), "peak": 1580.0,
                           "unit": "GB/s", "frac": round(bs * nb / (k_ms["hash"] / 1e3) / 1e9 / 1580.0, 4),
                           "note": "peak = 1024 SIMDs x 4096 B per wavefront-call / 2.65 us; shared with the codec's VALU work "
                                   "when both kernels run"} if args.hash == "skein512" else None),
        # each kernel launched alone (not part of `value`): algorithmic bytes / duration against the HBM peak
        "standalone": {k: {"ms_per_step": round(solo_ms[k], 3), "launches_per_step": launches[k], "ingest_GBps": round(bs * nb / (solo_ms[k] / 1e3) / 1e9, 1),
                           "alg_GBps": round(alg_bytes[k] * nb / (solo_ms[k] / 1e3) / 1e9, 1),
                           "hbm_frac": round(alg_bytes[k] * nb / (solo_ms[k] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)} for k in solo_ms},
        "parity_spot_check": spot,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_baseline_seconds)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
