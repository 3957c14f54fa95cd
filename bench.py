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

Rank 0's LAST stdout line is the contract's JSON line, kept under ~6 KB so that it survives an 8 KB tail: the headline leg
(uniform-random blocks) in the top-level fields plus one compact record per further leg under "legs" (value, ms, ratio, roofline
fraction, traffic / algorithmic bytes, cpu baseline, parity counts).  The full record of every leg is printed BEFORE it, one
{"leg_detail": ...} line per leg, and written to gpurun_out/bench_detail.json.  The headline carries
  roofline      the dominant kernel's algorithmic bytes / its HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline  the oracle's (CPU restatement of the reference path) throughput on this host, bounded sample
and, at N = 1, further timed legs under "legs" -- outside `value`, each with its own ratio, kernels, roofline and
cpu_baseline -- that time what uniform noise cannot: the codecs' match/emit path:
  mixed                 Skein-512 + LZ4, 64 KiB, SURVEY 8(d)'s compressible synthetic mix (cw_dev_gen_mixed)
  corpus_skein512_lz4   BASELINE configs[2]: the in-tree corpora (canterbury + canterbury-large) tiled in HBM, 64 KiB
  corpus_skein256_lz4_4k  the reference's own default pair and block size (hc_sklz4: Skein-256-128 + LZ4 at 4 KiB, run_tests:19)
  corpus_sha256_lzf_4k / _64k   BASELINE configs[3], the reference's hc_shlzf pair (run_tests:20), 4 KiB and 64 KiB
  corpus_skein512_lz4_3233 / _51728   configs[2] at Silesia's literal size (3,233 x 64 KiB = 51,728 x 4 KiB blocks)
  corpus_skein512_lz4_16g / corpus_sha256_lzf_64k_16g   the two 64 KiB corpus legs at four times the batch (16 GiB = 256 Ki blocks in one call): the
                        codecs are sets of serial chains, and what a call reaches depends on how many blocks it brings
  corpus_skein256_lzf_4k / corpus_sha256_lz4_4k   the reference's other two pairs (run_tests:14,23)
plus "host_path" / "host_path_corpus": the drop-in host-buffer entry point over noise and over the corpus (PCIe-inclusive, never `value`).
After its timed region every leg is checked against the CPU oracle at the leg's own scale ("parity": the corpus legs are periodic, so
EVERY block's size and digest is compared with the oracle's for block i mod tile and the payload bytes of the first two and the last
tile; random / mixed legs regenerate 4,096 sampled blocks with the oracle's generator and compare input, digest, size and payload) and
decodes ALL of its slots on the device, comparing them with their input blocks ("roundtrip"); a mismatch exits non-zero instead of
printing a number.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0    # PCIe Gen5 x16, one direction (same guide)
SEED = 0xC0FFEE
# Rehearsal of the N > 1 path on a box with ONE GPU (tests/test_gpu_bench_world2.py): every rank uses cuda:0 and the gather runs over
# gloo, because RCCL refuses two ranks on one device.  Everything else -- shards, kernels, double-buffered gather, fences, timing,
# checks -- is the code an 8-GPU run executes.  The line is marked "rehearsal" and is never a benchmark number.
REHEARSE = os.environ.get("CW_BENCH_REHEARSE") == "gloo-one-gpu"
HASH_IDS = {"skein512": 0, "skein": 1, "sha256mb": 2}
COMP_IDS = {"lz4": 0, "lzf": 1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--block-bytes", type=int, default=65536)
    ap.add_argument("--hash", default="skein512", choices=list(HASH_IDS))
    ap.add_argument("--comp", default="lz4", choices=list(COMP_IDS))
    ap.add_argument("--data", default="random", choices=["random", "mixed", "corpus"], help="input of the headline leg")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline leg only")
    ap.add_argument("--no-roundtrip", action="store_true", help="skip the full-size decode-and-compare after each leg")
    ap.add_argument("--leg-bytes", type=int, default=4 << 30, help="input bytes per extra leg")
    ap.add_argument("--standalone", action="store_true",
                    help="after the timed region also launch each kernel alone (3x) and report its own roofline; off by default so "
                         "that a rocprofv3 summary of the default command holds fused launches only")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="rehearsal of the N > 1 control flow without a GPU: gloo, CPU tensors, the oracle standing in for the device "
                         "on a few small blocks (tests/test_bench_dryrun.py); prints a line marked dry_run, never a benchmark number")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = this repo's C restatement of the reference's worker loop; kind "port")
# ---------------------------------------------------------------------------------------------------------------
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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_rate(O, data, bs, hash_name, comp_name, threads, target_s, min_passes=1):
    """GB/s of the oracle's worker loop over `data`: passes until target_s, median over passes."""
    h, c = HASH_IDS[hash_name], COMP_IDS[comp_name]
    rates, secs = [], 0.0
    while (secs < target_s or len(rates) < min_passes) and len(rates) < 1000:
        t, *_ = O.hash_and_compress(data, bs, h, c, threads, want_payload=False)
        rates.append(len(data) / t / 1e9)
        secs += t
    return statistics.median(rates), len(rates), secs


def cpu_baseline(hash_name, comp_name, bs, sample, what, target_s):
    """Oracle on a bounded sample of the leg's own input (numpy uint8, whole blocks)."""
    import oracle as O
    O.build()
    threads = host_cpu_share()
    rate, passes, secs = _oracle_rate(O, sample, bs, hash_name, comp_name, threads, target_s, min_passes=3 if target_s >= 6 else 1)
    return {"value": round(rate, 4), "unit": "GB/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "sample": f"median of {passes} passes over {len(sample) // bs} x {bs} B blocks of {what} ({len(sample) / 2**20:.0f} MiB), "
                      f"{hash_name}+{comp_name}, oracle worker loop with {threads} threads, {secs:.1f} s"}


def cpu_baseline_reference_style(sample64k, target_s):
    """14 threads pinned as the reference's run_tests:17 does with taskset, and the reference's native
    pair Skein-256-128 + LZ4 on 4 KiB blocks (results/hc_sklz4.*), median of 3."""
    import oracle as O
    out = {}
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) >= 14:
        try:
            os.sched_setaffinity(0, set(cpus[-14:]))   # the reference pins cores 17-30 (run_tests:17): the LAST 14 here
            rate, passes, secs = _oracle_rate(O, sample64k, 65536, "skein512", "lz4", 14, target_s / 2, min_passes=3)
            out["pinned14_skein512_lz4_64k"] = {"value": round(rate, 4), "unit": "GB/s", "cores": 14, "pinned": True,
                                                "sample": f"median of {passes} passes, {secs:.1f} s, same sample as cpu_baseline"}
            rate, passes, secs = _oracle_rate(O, sample64k, 4096, "skein", "lz4", 14, target_s / 2, min_passes=3)
            out["pinned14_skein256_128_lz4_4k"] = {"value": round(rate, 4), "unit": "GB/s", "cores": 14, "pinned": True,
                                                   "sample": f"the reference's native pair on 4 KiB blocks (run_tests:17), median of {passes} passes, {secs:.1f} s"}
        finally:
            os.sched_setaffinity(0, set(cpus))
    else:
        out["note"] = f"only {len(cpus)} CPUs in this process's affinity mask: no 14-thread pinned run"
    return out


def cpu_baseline_pinned14(sample, bs, hash_name, comp_name, target_s):
    """The leg's own pair as the reference's run_tests pins it: 14 threads on the last 14 CPUs of the affinity mask."""
    import oracle as O
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 14:
        return {"note": f"only {len(cpus)} CPUs in the affinity mask"}
    try:
        os.sched_setaffinity(0, set(cpus[-14:]))
        rate, passes, secs = _oracle_rate(O, sample, bs, hash_name, comp_name, 14, target_s, min_passes=3)
    finally:
        os.sched_setaffinity(0, set(cpus))
    return {"value": round(rate, 4), "unit": "GB/s", "cores": 14, "pinned": True, "sample": f"median of {passes} passes, {secs:.1f} s"}


# ---------------------------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------------------------
def corpus_tile():
    """(bytes of the whole 64 KiB blocks of canterbury, then canterbury-large; #64K blocks of each) -- SURVEY 8(d):
    38 + 98 blocks; the files are the reference's dataset/ (tests/golden/corpus, inputs only)."""
    import lzma
    base = os.path.join(ROOT, "tests", "golden", "corpus")
    parts, counts = [], []
    for sub in ("canterbury", "canterbury-large"):
        n = 0
        for f in sorted(os.listdir(os.path.join(base, sub))):
            path = os.path.join(base, sub, f)
            data = lzma.open(path, "rb").read() if f.endswith(".xz") else open(path, "rb").read()
            whole = len(data) // 65536 * 65536
            parts.append(data[:whole])
            n += whole // 65536
        counts.append(n)
    return b"".join(parts), counts


def fill_input(cw, torch, kind, src, first, nb, bs, stream):
    """Generate the leg's blocks in HBM.  Returns a function sample(n) -> numpy bytes of the first n blocks (host)."""
    if kind in ("random", "mixed"):
        (cw.dev_gen_random if kind == "random" else cw.dev_gen_mixed)(SEED, first, nb, bs, src.data_ptr(), stream)
        return None
    tile, counts = corpus_tile()
    t = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda()
    tb = len(tile)
    reps = (nb * bs) // tb
    if reps:
        src[: reps * tb].view(reps, tb).copy_(t.unsqueeze(0).expand(reps, tb))
    rest = nb * bs - reps * tb
    if rest:
        src[reps * tb:].copy_(t[:rest])
    return counts


# ---------------------------------------------------------------------------------------------------------------
# parity at benchmark scale: the oracle as the checker, never the thing measured
# ---------------------------------------------------------------------------------------------------------------
def full_parity(torch, name, hash_name, comp_name, bs, nb, kind, first_block, src, dst, stride, sizes, digests):
    """Compare what the timed steps left in HBM with the CPU oracle.  Returns a dict; any difference ends the run (exit != 0).

    corpus   the input repeats every T blocks (the tile of corpus_tile()), so the oracle runs over one tile and EVERY block's
             size and digest is compared with the oracle's for block i mod T (on the device, no sampling); the payload bytes of
             the first two tiles and of the last whole tile are compared byte for byte.
    random / mixed   64 runs of 64 consecutive blocks spread over the leg (4,096 blocks) are regenerated with the
             oracle's generator and compared: input bytes, digest, size, payload."""
    import numpy as np
    import oracle as O
    O.build()
    h, c = HASH_IDS[hash_name], COMP_IDS[comp_name]
    threads = host_cpu_share()

    def fail(what):
        raise SystemExit(f"PARITY FAILURE in bench leg {name}: {what}")  # a wrong result must not pass as a number

    def payload_equal(dev_slots, osz, opay, where):
        for i in range(len(osz)):
            z = int(osz[i])
            if z and not np.array_equal(dev_slots[i, :z], opay[i, :z]):
                fail(f"payload of block {where + i}")

    if kind == "corpus":
        tile, _ = corpus_tile()
        T = len(tile) // bs
        if nb < T:
            tile, T = tile[: nb * bs], nb
        _, odig, osz, opay = O.hash_and_compress(np.frombuffer(tile, dtype=np.uint8), bs, h, c, threads, want_payload=True)
        reps = (nb + T - 1) // T
        want_sz = torch.from_numpy(osz.astype(np.int32)).cuda().repeat(reps)[:nb]
        bad = int((sizes != want_sz).sum().item())
        if bad:
            i = int(torch.nonzero(sizes != want_sz)[0].item())
            fail(f"{bad} of {nb} sizes differ from the oracle's, first at block {i}: {int(sizes[i])} != {int(want_sz[i])}")
        want_dig = torch.from_numpy(odig).cuda().repeat(reps, 1)[:nb]
        badd = int((digests != want_dig).any(dim=1).sum().item())
        if badd:
            fail(f"{badd} of {nb} digests differ from the oracle's")
        del want_sz, want_dig
        tiles = sorted({0, 1, nb // T - 1} & set(range(nb // T))) if nb >= T else []
        for r in tiles:
            slots = dst[r * T * stride:(r + 1) * T * stride].view(T, stride).cpu().numpy()
            payload_equal(slots, osz, opay, r * T)
        return {"blocks": nb, "sizes_equal": nb, "digests_equal": nb, "payload_blocks_equal": len(tiles) * T,
                "how": f"oracle over the {T}-block tile; every block vs block i mod {T}; payload bytes of tiles {tiles}", "ok": True}

    gen = O.gen_random_blocks if kind == "random" else O.gen_mixed_blocks
    runs, run_len = min(64, nb), min(64, max(1, nb // 64))
    starts = sorted({min(nb - run_len, r * (nb // runs) + (r % 7)) for r in range(runs)} | {0, nb - run_len})
    checked = 0
    for a in starts:
        blk = gen(SEED, first_block + a, run_len, bs)
        if not np.array_equal(src[a * bs:(a + run_len) * bs].cpu().numpy(), blk):
            fail(f"generator, blocks {a}..{a + run_len - 1}")
        _, odig, osz, opay = O.hash_and_compress(blk, bs, h, c, threads, want_payload=True)
        if not np.array_equal(sizes[a:a + run_len].cpu().numpy().astype(np.uint32), osz):
            fail(f"sizes of blocks {a}..{a + run_len - 1}")
        if not np.array_equal(digests[a:a + run_len].cpu().numpy(), odig):
            fail(f"digests of blocks {a}..{a + run_len - 1}")
        payload_equal(dst[a * stride:(a + run_len) * stride].view(run_len, stride).cpu().numpy(), osz, opay, a)
        checked += run_len
    return {"blocks": nb, "sampled_blocks_equal": checked, "how": f"{len(starts)} runs of {run_len} consecutive blocks regenerated by the oracle: "
            "input, digest, size, payload", "ok": True}


# ---------------------------------------------------------------------------------------------------------------
# one timed leg
# ---------------------------------------------------------------------------------------------------------------
def run_leg(cw, torch, args, name, hash_name, comp_name, bs, nb, kind, steps, warmup, world, rank, local_rank, first_block,
            baseline_s, standalone=False):
    from compute_war_amd.shard import gather_results
    db = cw.digest_bytes(hash_name)
    stride = (cw.compress_bound(comp_name, bs) + 15) // 16 * 16
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    src = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    # two digest / totals buffers: step i's result gather (RCCL, async) overlaps step i+1's kernels
    dig_bufs = [torch.zeros((nb, db), dtype=torch.uint8, device="cuda") for _ in range(2 if world > 1 else 1)]
    tot_bufs = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(len(dig_bufs))]
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    pending = [[] for _ in dig_bufs]
    counts = fill_input(cw, torch, kind, src, first_block, nb, bs, s)
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
        cw.dev_hash_and_compress(hash_name, comp_name, src.data_ptr(), bs, nb, dig.data_ptr(), dst.data_ptr(), stride, sizes.data_ptr(), s)
        cw.dev_sum_sizes(sizes.data_ptr(), nb, bs, totals.data_ptr(), s)
        all_d, all_t, pending[b] = gather_results(dig, totals, world, async_op=world > 1)  # the only exchange (no-op at N=1)
        state["gathered"] = (all_d, all_t, dig)

    def fence():
        for hs in pending:
            for h in hs:
                h.wait()
        if world > 1:
            import torch.distributed as dist
            dist.barrier() if REHEARSE else dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    all_digests, all_totals, digests = state["gathered"]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = cw.profile_read(reset=True)
    names = cw.profile_kernels()   # what the library launched, from the library
    k_ms = {"comp": prof["codec"][0] / max(prof["codec"][1], 1), "hash": prof["hash"][0] / max(prof["hash"][1], 1)}

    solo_ms = {}
    if standalone:  # outside the timed region: each kernel on its own (3 launches), for the per-kernel rooflines
        cw.profile_enable(True)
        for _ in range(3):
            cw.dev_compress(comp_name, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
        torch.cuda.synchronize()
        p2 = cw.profile_read(reset=True)
        solo_ms["comp"] = p2["codec"][0] / max(p2["codec"][1], 1)
        for _ in range(3):
            cw.dev_hash(hash_name, src.data_ptr(), bs, nb, dig_bufs[0].data_ptr(), s)
        torch.cuda.synchronize()
        p2 = cw.profile_read(reset=True)
        solo_ms["hash"] = p2["hash"][0] / max(p2["hash"][1], 1)
    cw.profile_enable(False)

    total_blocks = nb * world
    bytes_out = int(all_totals[0].item())
    stored_raw = int(all_totals[1].item())
    value = total_blocks * bs * steps / elapsed / 1e9

    # parity at the leg's own scale (outside the timed region): every block against the CPU oracle where the input is periodic,
    # 4,096 regenerated blocks otherwise
    spot, sample, per_corpus = "skipped", None, None
    if rank == 0:
        import numpy as np
        spot = full_parity(torch, name, hash_name, comp_name, bs, nb, kind, first_block, src, dst, stride, sizes, digests)
        hz = sizes.cpu().numpy()
        if counts:  # corpus: the ratio of each corpus over the first tile, against the survey's anchors
            k = 65536 // bs
            z = hz[: (counts[0] + counts[1]) * k].astype(np.int64)
            z = np.where(z == 0, bs, z)
            anchors = {(a["corpus"], a["block"]): a[comp_name] for a in json.load(open(os.path.join(ROOT, "tests", "golden", "survey_anchors.json")))["corpus_ratios"]}
            per_corpus = {}
            for cname, a, b in (("canterbury", 0, counts[0] * k), ("canterbury-large", counts[0] * k, (counts[0] + counts[1]) * k)):
                if b > nb:
                    continue
                r = (b - a) * bs / float(z[a:b].sum())
                exp = anchors.get((cname, bs))
                per_corpus[cname] = {"ratio": round(r, 4), "reference_ratio": exp, "equal": exp is not None and round(r, 4) == exp}
        if baseline_s > 0:
            nsample = max(host_cpu_share(), min(nb, (256 << 20) // bs))
            sample = src[: nsample * bs].cpu().numpy()

    # full-size round trip (outside the timed region): every block this rank compressed is decoded on the device and compared
    # with its input -- a check that does not depend on the oracle and covers all of the leg's blocks, not a sample
    roundtrip = None
    if not args.no_roundtrip:
        t_rt = time.perf_counter()
        chunk = max(1, min(nb, (2 << 30) // bs))
        back = torch.empty(chunk * bs, dtype=torch.uint8, device="cuda")
        status = torch.empty(chunk, dtype=torch.int32, device="cuda")
        wrong = raw = 0
        for first in range(0, nb, chunk):
            k = min(chunk, nb - first)
            cw.dev_decompress(comp_name, dst.data_ptr() + first * stride, stride, sizes.data_ptr() + first * 4, k, back.data_ptr(), bs,
                              status.data_ptr(), s)
            fits = sizes[first:first + k] != 0          # LZF: 0 = did not fit l - 1, the caller keeps the block raw
            same = (back[:k * bs].view(k, bs) == src[first * bs:(first + k) * bs].view(k, bs)).all(dim=1) & (status[:k] == 0)
            wrong += int((fits & ~same).sum().item())
            raw += int((~fits).sum().item())
            del same
        del back, status
        if wrong:
            raise SystemExit(f"ROUND-TRIP FAILURE in bench leg {name}: {wrong} of {nb} blocks do not decode to their input")
        roundtrip = {"blocks_decoded_and_compared": nb - raw, "stored_raw": raw, "ok": True, "seconds": round(time.perf_counter() - t_rt, 2),
                     "how": "cw_dev_decompress of every slot on the device, compared with the input block"}

    # the lane-per-block LZ4 parser is bound by the random memory lines of its probes: its probe rate against what
    # tools/random_line.hip measured the chip to retire for such a mix.  Probes and sequences per block are counted on the host
    # (tools/lz_probe_count.py, a pure-Python walk of the parser) over a few sampled blocks.
    line_roof = None
    rl = os.path.join(ROOT, "profiles", "random_line.json")
    if rank == 0 and comp_name == "lz4" and bs > 4096 and kind != "random" and "lanes" in names["codec"] and os.path.exists(rl):
        import importlib.util
        spec = importlib.util.spec_from_file_location("lz_probe_count", os.path.join(ROOT, "tools", "lz_probe_count.py"))
        lpc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(lpc)
        picks = [(i * nb // 24 + i) % nb for i in range(24)]   # (+ i: both parities of the synthetic mix)
        stats = [lpc.lz4_counts(src[i * bs:(i + 1) * bs].cpu().numpy().tobytes()) for i in picks]
        parsed = [(p, q) for p, q, _ in stats if q > 0]          # blocks without a match never reach the parser (the scan writes them)
        if parsed:
            probes = sum(p for p, _ in parsed) / len(parsed)
            seqs = sum(q for _, q in parsed) / len(parsed)
            lines_per_probe = (2 * probes + 2 * seqs) / probes     # entry load + store per probe; per sequence the ip-2 store and the match's candidate
            blocks_per_s = nb * len(parsed) / len(stats) / (k_ms["comp"] / 1e3)
            pts = sorted(json.load(open(rl))["points"], key=lambda q: q["lines_per_probe"])
            x0, x1 = pts[0]["lines_per_probe"], pts[-1]["lines_per_probe"]
            y0, y1 = x0 * pts[0]["probes_per_s"], x1 * pts[-1]["probes_per_s"]   # lines per second at both mixes
            t = min(1.0, max(0.0, (lines_per_probe - x0) / (x1 - x0)))
            peak_lines = y0 + t * (y1 - y0)
            got_lines = blocks_per_s * probes * lines_per_probe
            line_roof = {"bound": "random memory lines (lane-per-block parser)", "achieved": round(got_lines / 1e9, 2), "peak": round(peak_lines / 1e9, 2),
                         "unit": "G lines/s", "frac": round(got_lines / peak_lines, 4), "probes_per_block": round(probes),
                         "sequences_per_block": round(seqs), "blocks_sampled": len(stats), "blocks_parsed_share": round(len(parsed) / len(stats), 3),
                         "source": "profiles/random_line.json (tools/random_line.hip); counts: tools/lz_probe_count.py; time: the codec's whole call"}

    if rank != 0:
        return None, None
    csize = bytes_out / total_blocks
    alg_bytes = {"hash": bs + db, "comp": bs + csize + 4}   # the hash reads the block, writes its digest; the codec reads it, writes csize + 4
    launches = {"hash": 8 if "slice" in names["hash"] else 1, "comp": 1}
    # the kernel the step is bound by.  The two event spans say so only while the hash is the longer job (incompressible input); beside busy parsers
    # the hash's launches trickle in and its span is the whole call whatever its work, so on compressible input it is the codec
    dom = max(k_ms, key=k_ms.get) if kind == "random" or not k_ms.get("comp") else "comp"
    kernels = {k: {"name": names["codec" if k == "comp" else "hash"], "ms_per_step": round(k_ms[k], 3), "launches_per_step": launches[k],
                   "ms_per_launch": round(k_ms[k] / launches[k], 3),
                   "alg_GBps": round(alg_bytes[k] * nb / (k_ms[k] / 1e3) / 1e9, 1) if k_ms[k] else None,
                   "ingest_GBps": round(bs * nb / (k_ms[k] / 1e3) / 1e9, 1) if k_ms[k] else None} for k in k_ms}
    achieved = alg_bytes[dom] * nb / (k_ms[dom] / 1e3) / 1e9
    traffic, tsrc = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes per launch (separate rocprofv3 --pmc passes)
    if os.path.exists(tf):
        tj = json.load(open(tf))
        key = f"{dom}:{hash_name if dom == 'hash' else comp_name}:{bs}:{nb}:{kind}"
        if key in tj:
            traffic, tsrc = tj[key] / launches[dom], f"profiles/traffic.json[{key}] / launches (FETCH_SIZE x2 + WRITE_SIZE, measured at this size in separate --pmc passes)"
    leg = {
        "leg": name,
        "workload": f"{hash_name}+{comp_name} over {nb} x {bs} B blocks per GPU, {kind} input resident in HBM",
        "value": round(value, 2), "unit": "GB/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
        "compression_ratio": round(total_blocks * bs / bytes_out, 4), "blocks_stored_raw": stored_raw,
        "roofline": {"bound": "hbm", "kernel": kernels[dom]["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                     # HBM bytes the counters saw per algorithmic byte (1.0 = nothing re-read): the first thing to fix when it is large
                     "amplification": round(traffic / (alg_bytes[dom] * nb / launches[dom]), 2) if traffic else None,
                     "launches_per_step": launches[dom], "ms_per_launch": round(k_ms[dom] / launches[dom], 3),
                     "alg_bytes_per_block": round(alg_bytes[dom], 1)},
        "kernels": dict(kernels, note="codec and hash run concurrently on two streams; durations overlap"),
        "parity": spot,
        # what the result gather delivered on rank 0 (all ranks' digests in block order): comparable between an N-rank run and a
        # one-rank run over the same global block range
        "gathered": {"digests": int(all_digests.shape[0]), "sha256": hashlib.sha256(all_digests.cpu().numpy().tobytes()).hexdigest()},
    }
    if roundtrip:
        leg["roundtrip"] = roundtrip
    if line_roof:
        leg["line_roofline"] = line_roof
    if per_corpus:
        leg["corpus_ratios"] = per_corpus
    if solo_ms:
        leg["standalone"] = {k: {"ms_per_step": round(solo_ms[k], 3), "ingest_GBps": round(bs * nb / (solo_ms[k] / 1e3) / 1e9, 1),
                                 "alg_GBps": round(alg_bytes[k] * nb / (solo_ms[k] / 1e3) / 1e9, 1),
                                 "hbm_frac": round(alg_bytes[k] * nb / (solo_ms[k] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)} for k in solo_ms}
    if sample is not None:
        what = {"random": "the same uniform-random stream", "mixed": "the same synthetic mix", "corpus": "the same tiled corpus"}[kind]
        leg["cpu_baseline"] = cpu_baseline(hash_name, comp_name, bs, sample, what, baseline_s)
    # release HBM before the next leg
    del src, dst, dig_bufs, tot_bufs, sizes
    torch.cuda.empty_cache()
    return leg, sample


def host_path_leg(cw, torch, hash_name, comp_name, bs, nbytes, kind="random"):
    """The drop-in host-buffer entry point (cw_hash_and_compress_packed over page-locked buffers): PCIe-inclusive, never `value`.
    kind "corpus": the tiled in-tree corpus instead of noise -- the codec, not the link, bounds that call (chunks grow to 2 GiB)."""
    import ctypes as C
    import numpy as np
    L = cw.lib()
    nb = nbytes // bs
    cap = nb * cw.compress_bound(comp_name, bs)
    hs, hp = L.cw_host_alloc(nb * bs), L.cw_host_alloc(cap)
    if not hs or not hp:
        return {"error": L.cw_last_error().decode()}
    try:
        dev = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
        fill_input(cw, torch, kind, dev, 0, nb, bs, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cw.ops.check(L.cw_dev_download(hs, dev.data_ptr(), nb * bs))
        del dev
        torch.cuda.empty_cache()
        db = cw.digest_bytes(hash_name)
        dig = np.zeros((nb, db), dtype=np.uint8)
        sizes = np.zeros(nb, dtype=np.uint32)
        offs = np.zeros(nb + 1, dtype=np.uint64)
        cw.ops.check(L.cw_prepare(HASH_IDS[hash_name], COMP_IDS[comp_name], bs, nb, 1))
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            cw.ops.check(L.cw_hash_and_compress_packed(HASH_IDS[hash_name], COMP_IDS[comp_name], hs, bs, nb, dig.ctypes.data, hp, cap,
                                                       offs.ctypes.data, sizes.ctypes.data))
            times.append(time.perf_counter() - t0)
        t = statistics.median(times[1:])   # steady state: buffers the device has mapped before (a service re-uses its staging)
        out_bytes = int(offs[nb])
        return {"entry_point": "cw_hash_and_compress_packed (pinned input and output, three-stage pipeline, one calling thread)",
                "workload": f"{hash_name}+{comp_name} over {nb} x {bs} B {'uniform-random' if kind == 'random' else kind} blocks in host memory",
                "value": round(nb * bs / t / 1e9, 2), "unit": "GB/s", "seconds": round(t, 4), "bytes_in": nb * bs, "bytes_out": out_bytes,
                "first_pass_GBps": round(nb * bs / times[0] / 1e9, 2),   # the first pass after idle is slower (clocks, link; not page mapping: DESIGN.md 5)
                "roofline": {"bound": "pcie", "achieved": round(max(nb * bs, out_bytes) / t / 1e9, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                             "frac": round(max(nb * bs, out_bytes) / t / 1e9 / PCIE_PEAK_GBS, 4),
                             "note": "the busier direction's bytes / time against one direction of PCIe Gen5 x16"}}
    finally:
        L.cw_host_free(hs)
        L.cw_host_free(hp)


# ---------------------------------------------------------------------------------------------------------------
def dry_run_cpu(args):
    """The N > 1 control flow of main() on CPU tensors over gloo, the oracle standing in for the device: init, shard,
    double-buffered async gather, barrier, max-over-ranks timing, rank 0's line.  Not a measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle as O
    from compute_war_amd.shard import gather_results, shard_range
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if world > 1:
        dist.init_process_group("gloo")
    bs, nb = args.block_bytes, args.blocks_per_gpu
    first, last = shard_range(nb * world, rank, world)
    assert last - first == nb
    data = O.gen_random_blocks(SEED, first, nb, bs)
    dig_bufs = [torch.zeros((nb, 64), dtype=torch.uint8) for _ in range(2 if world > 1 else 1)]
    tot_bufs = [torch.zeros(2, dtype=torch.int64) for _ in dig_bufs]
    pending = [[] for _ in dig_bufs]
    state = {"i": 0}

    def step():
        b = state["i"] % len(dig_bufs)
        state["i"] += 1
        for h in pending[b]:
            h.wait()
        _, dig, sizes, _ = O.hash_and_compress(data, bs, O.HASH_SKEIN512, O.COMP_LZ4, 1, want_payload=False)
        dig_bufs[b].copy_(torch.from_numpy(dig))
        tot_bufs[b][0] = int(sizes.sum())
        tot_bufs[b][1] = 0
        all_d, all_t, pending[b] = gather_results(dig_bufs[b], tot_bufs[b], world, async_op=world > 1)
        state["gathered"] = (all_d, all_t)

    def fence():
        for hs in pending:
            for h in hs:
                h.wait()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    all_d, all_t = state["gathered"]
    if rank == 0:
        import hashlib
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "blocks": nb * world,
                          "bytes_out": int(all_t[0]), "digests_sha256": hashlib.sha256(all_d.numpy().tobytes()).hexdigest(),
                          "seconds": round(elapsed, 4)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.dry_run_cpu:
        return dry_run_cpu(args)
    import torch  # imported before libcwhc.so so both share one HIP runtime

    import compute_war_amd as cw
    from compute_war_amd.shard import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if REHEARSE:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if REHEARSE:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
    cw.init(local_rank)

    bs, nb = args.block_bytes, args.blocks_per_gpu
    first, last = shard_range(nb * world, rank, world)
    assert last - first == nb
    head, sample = run_leg(cw, torch, args, "headline", args.hash, args.comp, bs, nb, args.data, args.steps, args.warmup, world, rank, local_rank,
                           first, 0.0 if (args.no_cpu_baseline or world > 1) else args.cpu_baseline_seconds, standalone=args.standalone)
    if rank != 0:
        return
    detail_path = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    detail = {"headline": head}

    def emit_detail(leg):
        # the full record of a leg: its own stdout line (ahead of the final line) and the detail file
        print(json.dumps({"leg_detail": leg}), flush=True)
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            json.dump(detail, open(detail_path, "w"), indent=1)
        except OSError:
            pass

    emit_detail(head)
    rf = head["roofline"]
    out = {
        "metric": "GB/s ingested (Skein-512 + LZ4, 64 KiB blocks)" if (args.hash, args.comp, bs) == ("skein512", "lz4", 65536)
                  else f"GB/s ingested ({args.hash} + {args.comp}, {bs} B blocks)",
        "value": head["value"], "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,  # BASELINE.md holds no published number for this metric (BASELINE.json "published": {})
        "dtype": "u64" if args.hash.startswith("skein") else "u32", "data": "synthetic",
        "config": {"workload": f"{args.hash}+{args.comp}, {nb} x {bs} B {'uniform-random' if args.data == 'random' else args.data} blocks/GPU in HBM",
                   "blocks_per_gpu": nb, "block_bytes": bs, "parallelism": f"block-sharded x{world}, gather-only RCCL"},
        "compression_ratio": head["compression_ratio"],
        "roofline": {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "amplification",
                                        "launches_per_step", "ms_per_launch", "alg_bytes_per_block")},
        "kernel_ms": {k: v["ms_per_step"] for k, v in head["kernels"].items() if isinstance(v, dict)},
        "parity": {k: v for k, v in head["parity"].items() if k != "how"} if isinstance(head["parity"], dict) else head["parity"],
        "gathered": head["gathered"],
    }
    if REHEARSE:
        out["rehearsal"] = "CW_BENCH_REHEARSE=gloo-one-gpu: all ranks on cuda:0, gather over gloo; not a benchmark number"
    if "roundtrip" in head:
        out["roundtrip"] = {k: v for k, v in head["roundtrip"].items() if k != "how"}
    if "standalone" in head:
        out["standalone"] = head["standalone"]
    # the binding roof of the Skein-512 kernel is integer VALU issue, not HBM; the peak is derived in a tracked
    # file from the measured per-instruction issue costs (profiles/r02_ubench_valu_issue.txt) and the kernel's instruction mix
    vr = os.path.join(ROOT, "profiles", "valu_roofline.json")
    if args.hash == "skein512" and os.path.exists(vr):
        v = json.load(open(vr))
        hk = head["kernels"]["hash"]
        out["valu_roofline"] = {"achieved": hk["ingest_GBps"], "peak": v["peak_GBps"], "unit": "GB/s",
                                "frac": round(hk["ingest_GBps"] / v["peak_GBps"], 4), "source": "profiles/valu_roofline.json"}
    if "cpu_baseline" in head:
        cb = dict(head["cpu_baseline"])
        ref = cpu_baseline_reference_style(sample, args.cpu_baseline_seconds)
        detail["cpu_baseline_reference_style"] = ref
        cb["pinned14"] = {k: v["value"] for k, v in ref.items() if isinstance(v, dict)}
        out["cpu_baseline"] = cb
    if world == 1 and not args.no_legs:
        lb = args.leg_bytes
        bsec = 0.0 if args.no_cpu_baseline else 4.0
        # (leg, hash, codec, block bytes, input, blocks, steps, the reference's published MB/s for the pair at 14 threads -- its own host,
        #  4 KiB blocks, results/hc_*; context, not a baseline)
        plan = (("mixed", "skein512", "lz4", 65536, "mixed", lb // 65536, 3, None),
                ("corpus_skein512_lz4", "skein512", "lz4", 65536, "corpus", lb // 65536, 3, None),
                ("corpus_skein512_lz4_16g", "skein512", "lz4", 65536, "corpus", 4 * lb // 65536, 3, None),
                ("corpus_skein512_lz4_3233", "skein512", "lz4", 65536, "corpus", 3233, 10, None),
                ("corpus_skein512_lz4_51728", "skein512", "lz4", 4096, "corpus", 51728, 10, None),
                ("corpus_skein256_lz4_4k", "skein", "lz4", 4096, "corpus", lb // 4096, 3, 808.3),
                ("corpus_skein256_lzf_4k", "skein", "lzf", 4096, "corpus", lb // 4096, 3, 704.2),
                ("corpus_sha256_lz4_4k", "sha256mb", "lz4", 4096, "corpus", lb // 4096, 3, 4899.9),
                ("corpus_sha256_lzf_4k", "sha256mb", "lzf", 4096, "corpus", lb // 4096, 3, 3127.2),
                ("corpus_sha256_lzf_64k", "sha256mb", "lzf", 65536, "corpus", lb // 65536, 3, None),
                ("corpus_sha256_lzf_64k_16g", "sha256mb", "lzf", 65536, "corpus", 4 * lb // 65536, 3, None))
        legs = []
        for name, h, c, b, kind, blocks, lsteps, ref_mbps in plan:
            leg, lsample = run_leg(cw, torch, args, name, h, c, b, blocks, kind, lsteps, 1, 1, 0, local_rank, 0, bsec)
            if ref_mbps is not None:
                leg["reference_published_MBps"] = {"value": ref_mbps, "what": "the reference's own host, 14 pinned threads, 4 KiB blocks (results/hc_*, BASELINE.md section 1)"}
                if lsample is not None and bsec > 0:
                    leg["cpu_baseline"]["pinned14"] = cpu_baseline_pinned14(lsample, b, h, c, bsec / 2)
            detail[name] = leg
            emit_detail(leg)
            r, cbl, par = leg["roofline"], leg.get("cpu_baseline", {}), leg["parity"]
            legs.append({"leg": name, "blocks": blocks, "block_bytes": b, "value": leg["value"], "ms": leg["ms_per_step"],
                         "ratio": leg["compression_ratio"], "frac": r["frac"], "kernel_ms": r["ms_per_launch"], "traffic": r["traffic"],
                         "amp": r["amplification"], "cpu": cbl.get("value"), "cores": cbl.get("cores"),
                         "cpu14": (cbl.get("pinned14") or {}).get("value"), "ref_MBps": ref_mbps,
                         "parity": par.get("sizes_equal", par.get("sampled_blocks_equal")) if isinstance(par, dict) else par,
                         "rt": (leg.get("roundtrip") or {}).get("blocks_decoded_and_compared")})
        out["legs"] = legs
        out["legs_key"] = ("value GB/s; ms per step; frac = dominant codec/hash kernel's algorithmic bytes / its time / 8 TB/s; traffic = HBM bytes per launch "
                           "(PMC, profiles/traffic.json); amp = traffic / algorithmic; cpu = oracle port GB/s on `cores` threads, cpu14 = 14 pinned; "
                           "parity = blocks whose size+digest equal the oracle's; rt = blocks decoded on the device and compared")
        # `value` is measured on incompressible noise (the north star's input): the same pair on the compressible corpus, beside it
        comp_leg = next((l for l in out["legs"] if l["leg"] == "corpus_skein512_lz4"), None)
        if comp_leg:
            out["value_on_compressible_input"] = {"leg": comp_leg["leg"], "value": comp_leg["value"], "unit": "GB/s", "ratio": comp_leg["ratio"]}
        for key, kind in (("host_path", "random"), ("host_path_corpus", "corpus")):
            hp = host_path_leg(cw, torch, "skein512", "lz4", 65536, 8 << 30, kind=kind)
            detail[key] = hp
            emit_detail(dict(hp, leg=key))
            out[key] = {k: hp[k] for k in ("value", "unit", "seconds", "bytes_in", "bytes_out", "first_pass_GBps", "error") if k in hp}
            if "roofline" in hp:
                out[key]["pcie_frac"] = hp["roofline"]["frac"]
    line = json.dumps(out)
    if len(line) > 6000:   # the driver keeps an 8 KB tail: never let the contract line outgrow it
        out.pop("legs_key", None)
        line = json.dumps(out)
    print(line, flush=True)


if __name__ == "__main__":
    main()
