"""N>1 path on CPU: two gloo ranks shard the block index space, process their shards (with the oracle
standing in for the device here -- this tests the sharding + gather logic, not the kernels) and gather
digests/totals exactly as bench.py does over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, nblocks, bs, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from compute_war_amd.shard import gather_results, shard_range
    first, last = shard_range(nblocks, rank, world)
    data = O.gen_random_blocks(0xC0FFEE, first, last - first, bs)
    _, dig, sizes, _ = O.hash_and_compress(data, bs, O.HASH_SKEIN512, O.COMP_LZ4, threads=1, want_payload=False)
    totals = torch.tensor([int(sizes.sum()), 0], dtype=torch.int64)
    all_dig, all_tot, handles = gather_results(torch.from_numpy(dig), totals, world, async_op=True)
    for h in handles:
        h.wait()
    if rank == 0:
        np.save(os.path.join(out_dir, "dig.npy"), all_dig.numpy())
        np.save(os.path.join(out_dir, "tot.npy"), all_tot.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle):
    world, nblocks, bs = 2, 12, 4096
    mp.spawn(_rank_main, args=(world, _free_port(), nblocks, bs, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "dig.npy")
    tot = np.load(tmp_path / "tot.npy")
    data = oracle.gen_random_blocks(0xC0FFEE, 0, nblocks, bs)
    _, dig, sizes, _ = oracle.hash_and_compress(data, bs, oracle.HASH_SKEIN512, oracle.COMP_LZ4, threads=2, want_payload=False)
    assert got.shape == (nblocks, 64) and np.array_equal(got, dig)   # rank order == block order
    assert int(tot[0]) == int(sizes.sum())
