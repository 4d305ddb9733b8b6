"""CPU, world_size 2 over gloo: the data-parallel plumbing of ttsamd.dp (shard bounds,
weight broadcast C1, audio gather C2).  On the GPU box the same code runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'tts-arabic-pytorch_amd'))
    from ttsamd import dp
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cpu')
        # C1: weights exist only on rank 0
        rng = np.random.default_rng(0)
        sd = {'a.weight': rng.standard_normal((3, 5, 2)).astype(np.float32),
              'b.bias': rng.standard_normal((7,)).astype(np.float32)} if rank == 0 else None
        got = dp.broadcast_state_dict(sd, dev)
        ref = np.random.default_rng(0)
        assert np.array_equal(got['a.weight'], ref.standard_normal((3, 5, 2)).astype(np.float32))
        assert np.array_equal(got['b.bias'], ref.standard_normal((7,)).astype(np.float32))
        # sharding: 5 utterances over 2 ranks -> [0,3) and [3,5)
        lo, hi = dp.shard_bounds(5, world, rank)
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        # C2: ragged audio; utterance g has (g+1)*10 samples filled with g+1
        lens = torch.tensor([(g + 1) * 10 for g in range(lo, hi)], dtype=torch.int64)
        wave = torch.zeros(hi - lo, int(lens.max()))
        for i, g in enumerate(range(lo, hi)):
            wave[i, :lens[i]] = g + 1
        out = dp.gather_audio(wave, lens)
        if rank == 0:
            assert len(out) == 5
            for g, w in enumerate(out):
                assert w.shape == ((g + 1) * 10,) and bool((w == g + 1).all())
        else:
            assert out is None
        # end-to-end sharded tts with a stub model: utterance text 'x'*n -> wave of n samples valued n
        class Stub:
            device = torch.device('cpu')

            def tts(self, texts, batch_size=2, **kw):
                return [torch.full((len(t),), float(len(t))) for t in texts]
        texts = ['x' * n for n in (5, 9, 3, 7, 1)]
        res = dp.tts_sharded(Stub(), texts, batch_size=2)
        if rank == 0:
            assert [w.numel() for w in res] == [5, 9, 3, 7, 1]
            assert all(bool((w == w.numel()).all()) for w in res)
        else:
            assert res is None
        with open(os.path.join(tmpdir, f'ok{rank}'), 'w') as f:
            f.write('ok')
    finally:
        dist.destroy_process_group()


def test_dp_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def test_shard_bounds_cover():
    from ttsamd import dp
    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
