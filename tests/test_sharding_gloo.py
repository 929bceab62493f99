"""CPU suite: the N>1 host path (partition, frame scatter, result gather) with world_size-2 gloo.
The per-rank compute leg is replaced by the oracle HERE IN THE TEST ONLY (the product leg,
sharding.stream_flow, needs a GPU); what is checked is that sharded == unsharded, pair by pair."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties(lkflow):
    for n_frames in (0, 1, 2, 3, 9, 245, 256):
        for world in (1, 2, 3, 4, 8):
            parts = [lkflow.shard_pairs(n_frames, world, r) for r in range(world)]
            pairs = [max(n - 1, 0) for _, n in parts]
            assert sum(pairs) == max(n_frames - 1, 0)
            assert max(pairs) - min(pairs) <= 1
            nxt = 0
            for (f0, n), c in zip(parts, pairs):
                if c:
                    assert f0 == nxt and n == c + 1       # contiguous, 1-frame overlap
                    nxt = f0 + c
    # 256 frames / 8 GPUs -> 32 pairs on 7 ranks, 31 on the last; pacman 244 pairs -> 31/30
    assert [n - 1 for _, n in (lkflow.shard_pairs(256, 8, r) for r in range(8))] == [32] * 7 + [31]
    assert sorted({n - 1 for _, n in (lkflow.shard_pairs(245, 8, r) for r in range(8))}) == [30, 31]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sharding
    import synth
    from oracle import oracle as O
    H, W = 40, 64
    frames = None
    if rank == 0:
        v = synth.video(n_frames, H, W, seed=5)
        frames = torch.from_numpy(np.clip(np.rint(v * 255), 0, 255).astype(np.uint8))
    local = sharding.scatter_frames(frames, n_frames, H, W, src=0, device="cpu")
    f0, n_local = sharding.partition(n_frames, world)[rank]
    assert local.shape[0] == n_local
    rows = []
    for p in range(max(n_local - 1, 0)):                     # oracle stands in for the GPU leg (test only)
        a, b = O.u8_to_f64(local[p].numpy()), O.u8_to_f64(local[p + 1].numpy())
        u, vv = O.coarse_to_fine(a, b, 2, 5)
        rows.append([float(u.sum()), float(vv.sum()), float(f0 + p)])
    rows = torch.tensor(rows, dtype=torch.float64).reshape(-1, 3)
    allrows = sharding.gather_rows(rows, n_frames, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), allrows.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [6, 2])
def test_scatter_compute_gather_world2_equals_single(tmp_path, oracle, n_frames):
    import synth
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "rows.npy")
    v = synth.video(n_frames, 40, 64, seed=5)
    fr = np.clip(np.rint(v * 255), 0, 255).astype(np.uint8)
    want = []
    for p in range(n_frames - 1):
        u, vv = oracle.coarse_to_fine(oracle.u8_to_f64(fr[p]), oracle.u8_to_f64(fr[p + 1]), 2, 5)
        want.append([u.sum(), vv.sum(), float(p)])
    assert np.array_equal(got, np.array(want))               # bit-for-bit, in stream order
