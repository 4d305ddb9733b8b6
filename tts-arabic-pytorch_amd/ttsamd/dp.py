"""Data-parallel sharding of batched utterances over the GPUs of one node (one process per
GPU, torch.distributed: backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference is single-device; utterances are independent, so the only
communication is  (C1) a one-shot broadcast of the weights from rank 0 and
(C2) per call: all_gather of the int64 lengths + gather of the padded audio to rank 0 —
a fan-in over 7 independent xGMI links, no ring, no all-reduce.  No collective sits
inside the model."""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` (first n%world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_state_dict(sd, device, src=0):
    """C1: rank `src` holds {name: np.ndarray float32}; every rank returns the same dict.
    Metadata travels as a python object, the payload as ONE flat fp32 tensor (one large
    broadcast instead of hundreds of small ones — xGMI links are per-peer)."""
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape)) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    total = int(sum(int(np.prod(s)) for _, s in meta))
    if rank == src:
        flat = torch.from_numpy(np.concatenate([np.asarray(sd[k], np.float32).ravel() for k, _ in meta])).to(device)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out, off = {}, 0
    for k, s in meta:
        n = int(np.prod(s))
        out[k] = host[off:off + n].reshape(s).copy()
        off += n
    return out


def gather_audio(wave, lens, dst=0):
    """C2: wave [b_local, n_max_local] (device), lens int64 [b_local] (samples per utterance).
    Returns on `dst` a list (rank order, then local order) of 1-D device tensors; None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = wave.device
    meta = torch.tensor([wave.shape[0], wave.shape[1]], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    b_max = int(max(int(m[0]) for m in metas))
    n_max = int(max(int(m[1]) for m in metas))
    lens_pad = torch.zeros(b_max, dtype=torch.int64, device=dev)
    lens_pad[:lens.numel()] = lens
    all_lens = [torch.empty_like(lens_pad) for _ in range(world)]
    dist.all_gather(all_lens, lens_pad)
    pad = torch.zeros(b_max, n_max, dtype=wave.dtype, device=dev)
    pad[:wave.shape[0], :wave.shape[1]] = wave
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        for i in range(int(metas[r][0])):
            out.append(bufs[r][i, :int(all_lens[r][i])])
    return out


def tts_sharded(model, texts, batch_size=32, dst=0, **tts_kwargs):
    """Data-parallel `FastPitch2Wave.tts(list)`: every rank calls this with the SAME list.
    Utterances are ordered by length (so each rank's padded sub-batches are tight), dealt out in
    contiguous shards (`shard_bounds`), synthesised locally with `model.tts(shard, batch_size=...)`
    and gathered to rank `dst`, which returns the waves in the original order (other ranks: None).
    NB padded-batch FastPitch results depend on batch composition: an utterance's
    wave equals the single-GPU result for the same sub-batch, not for a different batching."""
    world, rank = dist.get_world_size(), dist.get_rank()
    order = sorted(range(len(texts)), key=lambda i: -len(texts[i]))
    lo, hi = shard_bounds(len(order), world, rank)
    mine = [texts[i] for i in order[lo:hi]]
    waves = model.tts(mine, batch_size=batch_size, **tts_kwargs) if mine else []
    dev = getattr(model, 'device', torch.device('cpu'))
    if dist.get_backend() == 'gloo':
        dev = torch.device('cpu')
    n_max = max([w.numel() for w in waves], default=1)
    pad = torch.zeros(len(waves), n_max, dtype=torch.float32, device=dev)
    lens = torch.zeros(len(waves), dtype=torch.int64, device=dev)
    for i, w in enumerate(waves):
        pad[i, :w.numel()] = w.to(dev)
        lens[i] = w.numel()
    gathered = gather_audio(pad, lens, dst=dst)
    if rank != dst:
        return None
    out = [None] * len(texts)
    for pos, w in zip(order, gathered):          # gathered is in rank order = sorted order
        out[pos] = w.cpu()
    return out
