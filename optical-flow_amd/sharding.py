"""Frame-pair sharding of a video stream over the GPUs of one node.

The hot path shards into independent units (frame pairs) once inputs are not mutated, so there is
NO collective on the data path.  RCCL (torch.distributed backend "nccl") is used only to move the
frame queue: scatter of uint8 frames from the ingest rank (8x smaller than fp64; converted with
lkflow_u8_to_f64_dev on arrival) and, optionally, a gather of per-pair results.  Gathering full
U/V planes to one GPU costs far more than computing them (2 fp64 planes per pair), so the default
result path is the fused consumer (3 scalars per pair, lkflow_flow_centroid_f64_dev); U/V stay
sharded unless explicitly gathered.

Works with any torch.distributed backend: "nccl" on GPUs, "gloo" in the CPU test-suite.
"""
import torch
import torch.distributed as dist

from lkflow_amd import shard_pairs


def partition(n_frames, world_size):
    """[(first_frame, n_frames_local)] for every rank (1-frame overlap between neighbours)."""
    return [shard_pairs(n_frames, world_size, r) for r in range(world_size)]


def _wire():
    """The device messages travel on, fixed by the backend on EVERY rank: RCCL ("nccl") moves device memory only,
    gloo (the CPU test-suite, the one-GPU rehearsal) moves host memory."""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def scatter_frames(frames_root, n_frames, height, width, src=0, device=None, dtype=torch.uint8):
    """Root `src` holds frames_root [F,H,W] (dtype); every rank returns its block
    [n_local, H, W] on `device`.  Point-to-point sends of exactly the needed frames (xGMI is
    point-to-point: the root's egress, 7 links, is the bound; nothing is broadcast)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    parts = partition(n_frames, world)
    first, n_local = parts[rank]
    # (Defaulting to the root tensor's device broke non-root ranks, which have no root tensor: they received into a
    # CPU tensor under nccl.)
    wire = _wire()
    device = torch.device(device) if device is not None else wire
    if wire.type == "cuda":
        assert device.type == "cuda", "the nccl (RCCL) backend scatters into device memory"
    if rank == src:
        assert frames_root is not None and tuple(frames_root.shape) == (n_frames, height, width)
        reqs, keep = [], []
        for r, (f0, n) in enumerate(parts):
            if r == src or n == 0:
                continue
            piece = frames_root[f0:f0 + n].to(wire).contiguous()   # host-side ingest rank under nccl: staged to the GPU
            keep.append(piece)
            reqs.append(dist.isend(piece, dst=r))
        local = frames_root[first:first + n_local].to(device).clone() if n_local else \
            torch.empty((0, height, width), dtype=dtype, device=device)
        for q in reqs:
            q.wait()
        return local
    local = torch.empty((n_local, height, width), dtype=dtype, device=wire)
    if n_local:
        dist.recv(local, src=src)
    return local.to(device)


def gather_rows(local_rows, n_frames, dst=0):
    """Gather per-pair result rows ([n_pairs_local, K], e.g. the centroid triples) on rank dst in stream
    order; returns [n_frames-1, K] on dst, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    parts = partition(n_frames, world)
    k = local_rows.shape[1]
    wire = _wire()
    if rank == dst:
        out = torch.empty((max(n_frames - 1, 0), k), dtype=local_rows.dtype, device=local_rows.device)
        for r, (f0, n) in enumerate(parts):
            cnt = max(n - 1, 0)
            if cnt == 0:
                continue
            if r == dst:
                out[f0:f0 + cnt] = local_rows
            else:
                buf = torch.empty((cnt, k), dtype=local_rows.dtype, device=wire)
                dist.recv(buf, src=r)
                out[f0:f0 + cnt] = buf.to(out.device)
        return out
    if local_rows.shape[0]:
        dist.send(local_rows.to(wire).contiguous(), dst=dst)
    return None


def stream_flow(frames_u8_local, ctx, levels, win, thresh=20.0, keep_flow=True):
    """GPU leg of one rank: uint8 frames [n,H,W] on the GPU -> (U, V, centroid rows) of its n-1 pairs.
    Needs liblkflow.so and a GPU (no fallback)."""
    n = frames_u8_local.shape[0]
    if n < 2:
        z = torch.empty((0,) + tuple(frames_u8_local.shape[1:]), dtype=torch.float64, device=frames_u8_local.device)
        return z, z.clone(), torch.empty((0, 3), dtype=torch.float64, device=frames_u8_local.device)
    f64 = ctx.u8_to_f64(frames_u8_local.contiguous())           # convertTo(CV_64F, 1/255) on device
    U, V = ctx.compute_video(f64, levels, win)
    rows = ctx.flow_centroid(U, V, thresh)
    return (U, V, rows) if keep_flow else (None, None, rows)
