"""Data-parallel sharding of batched utterances over the GPUs of one node: one process per GPU.

The reference is single-device; utterances are independent, so there is no collective
inside the model, only
  C1 once:     the weights from rank 0 — `Dp.broadcast_weights(engine)` (device blobs of a C-ABI handle:
               only rank 0 reads and packs the checkpoint) or `Dp.broadcast_state_dict` (one flat tensor);
  C2 per call: all-gather of the per-utterance lengths (`Dp.exchange_lens`), then a fan-in of the PACKED
               ragged audio (valid samples only) to rank 0 (`Dp.gather_audio`): world-1 independent
               point-to-point transfers over the xGMI links, no ring, no reduction.  The two exchanges run on
               two channels (one communicator / process group each), so the two-stream schedule of
               ttsamd.pipeline may issue them from its two streams.
Transports:
  'rccl'  — libttsamd's `ttsamd_dp_*` entry points (include/ttsamd.h; RCCL bound inside the library, device
            pointers in, nothing torch-specific), bootstrapped with a 128-byte id that rank 0 publishes
            through the already-initialised torch.distributed store.  The multi-GPU product path.
  'torch' — the same exchanges as torch.distributed calls on the default process group: backend "nccl"
            (= RCCL) or, for the world_size-2 CPU tests and the 2-ranks-on-one-GPU debug mode, "gloo"
            (device tensors are then staged through persistent pinned host buffers).
No buffer is allocated on the hot path: the length, send and receive buffers are persistent and grow
geometrically on the (rare) call that needs more."""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` (first n%world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_indices(order, world, rank):
    """Round-robin shard of a (length-sorted) index list: positions rank, rank + world, ...  Every rank's c-th chunk
    then covers the same stretch of the sorted list, so the per-chunk collectives meet ranks with equal work."""
    return list(order[rank::world])


class _Grow:
    """Persistent buffer that only ever grows (x1.5), so steady-state calls allocate nothing."""

    def __init__(self, dtype, device, pin=False):
        self.dtype, self.device, self.pin, self.buf = dtype, device, pin, None

    def get(self, n):
        n = int(n)
        if self.buf is None or self.buf.numel() < n:
            cap = max(n, int(1.5 * (self.buf.numel() if self.buf is not None else 0)), 1)
            self.buf = torch.empty(cap, dtype=self.dtype, device=self.device,
                                   pin_memory=bool(self.pin and torch.cuda.is_available()))
        return self.buf[:n]


class Dp:
    """One per process.  `device`: this rank's GPU (or cpu in the gloo tests).

    Ordering contract of the two channels (lengths / audio): every rank must issue the operations of ONE channel in the same HOST
    order -- the length all-gather of batch i + 1 and the audio fan-in of batch i may interleave freely across the two channels (they
    may run from two streams, and both RCCL kernels can be resident at once), but two calls on the same channel from different host
    threads of a rank, or in different orders on different ranks, deadlock the blocking collectives.  One thread per rank drives a Dp
    (ttsamd.pipeline.submit does); `close()` releases both communicators / the second process group."""

    def __init__(self, device, transport=None):
        assert dist.is_initialized(), 'init torch.distributed first (it carries the rendezvous)'
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device)
        self.backend = dist.get_backend()
        if transport is None:
            transport = os.environ.get('TTSAMD_DP_TRANSPORT')
        if transport is None:
            transport = 'rccl' if (self.backend == 'nccl' and self.device.type == 'cuda') else 'torch'
        assert transport in ('rccl', 'torch')
        self.transport = transport
        # TWO channels, so that the two exchanges of a call may be issued from two streams (ttsamd.pipeline: the length
        # all-gather of batch i + 1 on the acoustic stream while the audio fan-in of batch i is still queued behind its
        # vocoder on the other one).  Operations on ONE communicator must reach the device in the same order on every
        # rank, which two streams do not promise; each channel by itself is used from one stream at a time, in host order.
        #   comm / default group       : C1 weight broadcasts and the C2a length all-gather
        #   comm_audio / group_audio   : the C2b packed audio fan-in
        self.comm = self.comm_audio = None
        self.group_audio = None
        self.host_staged = self.backend == 'gloo' and self.device.type == 'cuda'
        if transport == 'rccl':
            from . import lib as L
            self._L, self._lib = L, L.load()
            idents = [(C.c_char * 128)(), (C.c_char * 128)()]
            if self.rank == 0:
                for ident in idents:
                    L.check(self._lib.ttsamd_dp_unique_id(ident), 'dp_unique_id')
            box = [bytes(ident) for ident in idents]
            dist.broadcast_object_list(box, src=0)            # 2 x 128 bytes through the rendezvous store
            comms = []
            with torch.cuda.device(self.device):
                for raw in box:
                    comm = C.c_void_p()
                    L.check(self._lib.ttsamd_dp_init(self.rank, self.world, (C.c_char * 128).from_buffer_copy(raw), C.byref(comm)),
                            'dp_init')
                    comms.append(comm)
            self.comm, self.comm_audio = comms
        else:
            self.group_audio = dist.new_group(backend=self.backend)      # every rank calls this (collective)
        st = self.device if not self.host_staged else torch.device('cpu')
        self._lens_send = _Grow(torch.int64, self.device)
        self._lens_recv = _Grow(torch.int64, self.device)
        self._lens_host = _Grow(torch.int64, 'cpu', pin=self.device.type == 'cuda')
        self._pack = _Grow(torch.float32, self.device)
        self._recv = _Grow(torch.float32, self.device)
        self._stage_send = _Grow(torch.float32, st, pin=True) if self.host_staged else None
        self._stage_recv = _Grow(torch.float32, st, pin=True) if self.host_staged else None

    def close(self):
        for name in ('comm_audio', 'comm'):
            comm = getattr(self, name, None)
            if comm is not None:
                self._lib.ttsamd_dp_destroy(comm)
                setattr(self, name, None)
        group = getattr(self, 'group_audio', None)
        if group is not None:                                 # torch transport: the second channel is a process group of its own
            self.group_audio = None
            if dist.is_initialized():
                dist.destroy_process_group(group)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ---------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr())

    # ---- C1 --------------------------------------------------------------------------
    def broadcast_weights(self, engine, src=0):
        """In place: this rank's HifiGanEngine / FastPitchEngine weight blobs <- rank `src`'s."""
        from .engine import FastPitchEngine, HifiGanEngine
        assert self.transport == 'rccl', 'handle-level broadcast is a C-ABI entry point (transport "rccl")'
        kind = {HifiGanEngine: 0, FastPitchEngine: 1}[type(engine)]
        with torch.cuda.device(self.device):
            self._L.check(self._lib.ttsamd_dp_broadcast_weights(self.comm, kind, engine.handle, src, self._stream()),
                          'dp_broadcast_weights')

    def broadcast_shapes(self, sd, src=0):
        """{name: array} on `src` -> {name: zeros of the same shape} elsewhere (what a non-root rank builds its
        handle from before `broadcast_weights`); `src` gets its own dict back."""
        box = [[(k, tuple(np.shape(v))) for k, v in sd.items()] if self.rank == src else None]
        dist.broadcast_object_list(box, src=src)
        if self.rank == src:
            return sd
        return {k: np.ones(s, np.float32) for k, s in box[0]}

    def broadcast_state_dict(self, sd, src=0):
        """Rank `src` holds {name: np.ndarray float32}; every rank returns the same dict.  Metadata travels as
        a python object, the payload as ONE flat fp32 tensor (one large transfer instead of hundreds)."""
        meta = [[(k, tuple(v.shape)) for k, v in sd.items()]] if self.rank == src else [None]
        dist.broadcast_object_list(meta, src=src)
        meta = meta[0]
        total = int(sum(int(np.prod(s)) for _, s in meta))
        dev = self.device if not self.host_staged else torch.device('cpu')
        if self.rank == src:
            flat = torch.from_numpy(np.concatenate([np.asarray(sd[k], np.float32).ravel() for k, _ in meta])).to(dev)
        else:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
        if self.transport == 'rccl':
            with torch.cuda.device(self.device):
                self._L.check(self._lib.ttsamd_dp_broadcast(self.comm, self._ptr(flat), total * 4, src, self._stream()),
                              'dp_broadcast')
        else:
            dist.broadcast(flat, src=src)
        host = flat.cpu().numpy()
        out, off = {}, 0
        for k, s in meta:
            n = int(np.prod(s))
            out[k] = host[off:off + n].reshape(s).copy()
            off += n
        return out

    # ---- C2a: lengths ----------------------------------------------------------------
    def exchange_lens(self, lens, b_cap):
        """lens int64 [b_local] on this rank's device (frames or samples), b_cap >= every rank's b_local.
        One small all-gather + ONE device->host copy; returns a host int64 array [world, 1 + b_cap] whose
        row r is (b_local_r, lens_r..., zero padding).  This is the only synchronisation point of a step:
        callers that need their own lengths on the host read them from row `rank` of the result."""
        b = int(lens.numel())
        assert b <= b_cap
        w = 1 + int(b_cap)
        send = self._lens_send.get(w)
        send.zero_()
        send[0] = b
        send[1:1 + b] = lens
        recv = self._lens_recv.get(self.world * w)
        host = self._lens_host.get(self.world * w)
        if self.transport == 'rccl':
            with torch.cuda.device(self.device):
                self._L.check(self._lib.ttsamd_dp_allgather(self.comm, self._ptr(send), self._ptr(recv), w * 8,
                                                            self._stream()), 'dp_allgather')
            host.copy_(recv, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        elif self.host_staged:
            mine = send.cpu()
            dist.all_gather_into_tensor(host, mine)
        else:
            dist.all_gather_into_tensor(recv, send)
            host.copy_(recv)
            if self.device.type == 'cuda':
                torch.cuda.current_stream(self.device).synchronize()
        return host.numpy().reshape(self.world, w).copy()

    # ---- C2b/c: audio ----------------------------------------------------------------
    def pack_audio(self, wave, nsamples, total):
        """wave [b, stride] + nsamples int64 [b] (device) -> persistent flat buffer holding the `total` valid
        samples back to back (one launch of ttsamd_dp_pack_audio; torch ops in the CPU tests)."""
        out = self._pack.get(max(int(total), 1))
        b, stride = wave.shape
        if total == 0 or b == 0:
            return out[:0]
        # the kernel takes raw pointers: a sliced wave or an int32 length tensor would be packed silently wrong
        assert wave.dtype == torch.float32 and wave.stride(1) == 1 and wave.stride(0) >= wave.shape[1], 'wave: fp32 rows, unit stride'
        assert nsamples.dtype == torch.int64 and nsamples.is_contiguous() and nsamples.device == wave.device and \
            nsamples.numel() == b, 'nsamples: contiguous int64 [b] on the wave\'s device'
        stride = wave.stride(0)
        if wave.device.type == 'cuda':
            from . import lib as L
            lib = L.load()
            with torch.cuda.device(wave.device):
                L.check(lib.ttsamd_dp_pack_audio(self._ptr(wave), stride, self._ptr(nsamples), b, wave.shape[1], self._ptr(out),
                                                 C.c_void_p(torch.cuda.current_stream(wave.device).cuda_stream)),
                        'dp_pack_audio')
        else:
            off = 0
            for i, n in enumerate(nsamples.tolist()):
                out[off:off + n] = wave[i, :n]
                off += n
        return out[:int(total)]

    def gather_flat(self, wave, nsamples, all_lens=None, dst=0, b_cap=None):
        """C2: wave [b_local, stride] (device), nsamples int64 [b_local] = valid samples per utterance.
        `all_lens` = result of `exchange_lens(nsamples, b_cap)` if the caller already has it (bench.py folds
        that exchange into FastPitch's own dec_lens read-back, so a step has ONE host sync); otherwise it is
        done here.  Returns (flat, all_lens): on `dst`, `flat` is a view of the persistent receive buffer
        (valid until the next call) holding every utterance's valid samples back to back, rank order then
        local order; None on the other ranks."""
        if all_lens is None:
            if b_cap is None:
                cap = torch.tensor([wave.shape[0]], dtype=torch.int64, device=wave.device if not self.host_staged else 'cpu')
                dist.all_reduce(cap, op=dist.ReduceOp.MAX)
                b_cap = int(cap.item())
            all_lens = self.exchange_lens(nsamples, b_cap)
        counts = np.array([int(all_lens[r, 1:1 + all_lens[r, 0]].sum()) for r in range(self.world)], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        mine, total = int(counts[self.rank]), int(counts.sum())
        packed = self.pack_audio(wave, nsamples, mine)
        recv = self._recv.get(max(total, 1)) if self.rank == dst else None
        if self.transport == 'rccl':
            cnt = (C.c_int64 * self.world)(*counts.tolist())
            off = (C.c_int64 * self.world)(*offsets.tolist())
            with torch.cuda.device(self.device):
                self._L.check(self._lib.ttsamd_dp_gather_audio(
                    self.comm_audio, self._ptr(packed) if mine else C.c_void_p(0),
                    self._ptr(recv) if recv is not None else C.c_void_p(0), cnt, off, dst, self._stream()), 'dp_gather_audio')
        else:
            self._gather_torch(packed, recv, counts, offsets, dst)
        return (recv[:total] if self.rank == dst else None), all_lens

    def gather_audio(self, wave, nsamples, all_lens=None, dst=0, b_cap=None):
        """`gather_flat` split into one 1-D view per utterance (rank order, then local order); None off `dst`."""
        flat, all_lens = self.gather_flat(wave, nsamples, all_lens=all_lens, dst=dst, b_cap=b_cap)
        if flat is None:
            return None
        sizes = [int(n) for r in range(self.world) for n in all_lens[r, 1:1 + all_lens[r, 0]]]
        return list(torch.split(flat, sizes)) if sizes else []

    def _gather_torch(self, packed, recv, counts, offsets, dst):
        """The fan-in through torch.distributed.  nccl: ONE `dist.gather` of equal-size slots (the longest rank's count;
        a persistent padded send buffer) — the most travelled primitive of the backend, then a device-side compaction on
        the root; gloo (CPU tests, one-device debug mode): point-to-point isend / irecv of the exact counts, staged
        through pinned host buffers when the tensors live on the GPU."""
        mine, total = int(counts[self.rank]), int(counts.sum())
        if self.backend == 'nccl':
            cap = max(int(counts.max()), 1)
            if not hasattr(self, '_slot_send'):
                self._slot_send = _Grow(torch.float32, self.device)
                self._slot_recv = _Grow(torch.float32, self.device)
            send = self._slot_send.get(cap)
            send[:mine].copy_(packed[:mine])
            slots = None
            if self.rank == dst:
                flat = self._slot_recv.get(cap * self.world)
                slots = [flat[r * cap:(r + 1) * cap] for r in range(self.world)]
            dist.gather(send, slots, dst=dst, group=self.group_audio)
            if self.rank == dst:
                for r in range(self.world):
                    c = int(counts[r])
                    if c:
                        recv[int(offsets[r]):int(offsets[r]) + c].copy_(slots[r][:c])
            return
        if self.host_staged:
            send = self._stage_send.get(max(mine, 1))[:mine]
            send.copy_(packed)
            rbuf = self._stage_recv.get(max(total, 1)) if self.rank == dst else None
        else:
            send, rbuf = packed, recv
        if self.rank == dst:
            rbuf[int(offsets[dst]):int(offsets[dst]) + mine] = send
            reqs = [dist.irecv(rbuf[int(offsets[r]):int(offsets[r] + counts[r])], src=r, group=self.group_audio)
                    for r in range(self.world) if r != dst and counts[r] > 0]
            for q in reqs:
                q.wait()
            if self.host_staged:
                recv[:total].copy_(rbuf[:total])
        elif mine > 0:
            dist.send(send, dst=dst, group=self.group_audio)


# ---- module-level conveniences (default Dp per process) --------------------------------
_default = None


def default(device=None):
    global _default
    if _default is None:
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if (
                torch.cuda.is_available() and dist.get_backend() != 'gloo') else torch.device('cpu')
        _default = Dp(device)
    return _default


def reset_default():
    global _default
    if _default is not None:
        _default.close()
    _default = None


def broadcast_state_dict(sd, device, src=0):
    return default(device).broadcast_state_dict(sd, src=src)


def gather_audio(wave, lens, dst=0):
    return default(wave.device).gather_audio(wave, lens, dst=dst)


def tts_sharded(model, texts, batch_size=32, dst=0, dp=None, **tts_kwargs):
    """Data-parallel `FastPitch2Wave.tts(list)`: every rank calls this with the SAME list.
    Utterances are ordered by length (so each rank's padded sub-batches are tight), dealt out ROUND-ROBIN over the
    ranks (`shard_indices`: rank r takes the sorted positions r, r + world, ... -- chunk c of every rank then holds
    texts of similar length, and since every chunk ends in a collective no rank waits for one that got all the long
    ones) and synthesised locally in chunks of `batch_size` with
    `model.tts_batch_device` — waves stay in HBM — then each chunk's valid samples are packed and fanned in to
    rank `dst`, which does ONE device->host copy per chunk and returns the waves in the original order
    (other ranks: None).  Models without `tts_batch_device` (test stubs) go through `model.tts` + a host pad.
    NB padded-batch FastPitch results depend on batch composition: an utterance's wave equals
    the single-GPU result for the same sub-batch, not for a different batching."""
    dpx = dp if dp is not None else default(getattr(model, 'device', None))
    world, rank = dpx.world, dpx.rank
    order = sorted(range(len(texts)), key=lambda i: -len(texts[i]))
    shards = [shard_indices(order, world, r) for r in range(world)]
    mine_idx = shards[rank]
    n_chunks = max((len(sh) + batch_size - 1) // batch_size for sh in shards) if texts else 0
    gathered = [[] for _ in range(world)]          # on dst: per source rank, waves in local order
    for c in range(n_chunks):
        mine = [texts[i] for i in mine_idx[c * batch_size:(c + 1) * batch_size]]
        if hasattr(model, 'tts_batch_device'):
            if mine:
                wave, nsamp = model.tts_batch_device(mine, **tts_kwargs)       # original order within the chunk
            else:
                wave = torch.zeros(0, 1, dtype=torch.float32, device=dpx.device)
                nsamp = torch.zeros(0, dtype=torch.int64, device=dpx.device)
        else:
            ws = model.tts(mine, batch_size=batch_size, **tts_kwargs) if mine else []
            dev = dpx.device
            wave = torch.zeros(len(ws), max([w.numel() for w in ws], default=1), dtype=torch.float32, device=dev)
            nsamp = torch.tensor([w.numel() for w in ws], dtype=torch.int64, device=dev)
            for i, w in enumerate(ws):
                wave[i, :w.numel()] = w.to(dev)
        flat, all_lens = dpx.gather_flat(wave, nsamp, dst=dst, b_cap=batch_size)
        if rank == dst:
            host = flat.cpu()                                  # one D2H per chunk for all ranks' audio
            o = 0
            for r in range(world):
                for n in all_lens[r, 1:1 + all_lens[r, 0]].tolist():
                    gathered[r].append(host[o:o + n].clone())
                    o += n
    if rank != dst:
        return None
    out = [None] * len(texts)
    for r in range(world):
        for i, w in zip(shards[r], gathered[r]):
            out[i] = w
    return out
