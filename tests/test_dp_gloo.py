"""CPU, world_size 2 over gloo: the data-parallel plumbing of ttsamd.dp (shard bounds, weight broadcast C1,
length exchange + packed audio fan-in C2, sharded tts).  On the GPU box the same class runs over RCCL through
libttsamd's ttsamd_dp_* entry points (tests/test_gpu_dp.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _Stub:
    """Deterministic stand-in for FastPitch2Wave: text 'x'*n -> n samples whose values depend only on the
    utterance (so any sharding must reproduce the 1-rank result bit for bit)."""
    device = torch.device('cpu')

    @staticmethod
    def _wave(t):
        g = torch.Generator().manual_seed(len(t))
        return torch.randn(len(t), generator=g)

    def tts(self, texts, batch_size=2, **kw):
        return [self._wave(t) for t in texts]


class _StubDevice(_Stub):
    """Same, through the device-resident entry point the real FastPitch2Wave offers."""

    def tts_batch_device(self, batch, **kw):
        ws = [self._wave(t) for t in batch]
        n = torch.tensor([w.numel() for w in ws], dtype=torch.int64)
        wave = torch.zeros(len(ws), int(n.max()))
        for i, w in enumerate(ws):
            wave[i, :w.numel()] = w
        return wave, n


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
        dpx = dp.Dp(dev)
        assert dpx.transport == 'torch' and not dpx.host_staged
        # C1: weights exist only on rank 0
        rng = np.random.default_rng(0)
        sd = {'a.weight': rng.standard_normal((3, 5, 2)).astype(np.float32),
              'b.bias': rng.standard_normal((7,)).astype(np.float32)} if rank == 0 else None
        got = dpx.broadcast_state_dict(sd)
        ref = np.random.default_rng(0)
        assert np.array_equal(got['a.weight'], ref.standard_normal((3, 5, 2)).astype(np.float32))
        assert np.array_equal(got['b.bias'], ref.standard_normal((7,)).astype(np.float32))
        shp = dpx.broadcast_shapes(sd)
        assert {k: v.shape for k, v in shp.items()} == {'a.weight': (3, 5, 2), 'b.bias': (7,)}
        # sharding: 5 utterances over 2 ranks -> [0,3) and [3,5)
        lo, hi = dp.shard_bounds(5, world, rank)
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        # C2: ragged audio; utterance g has (g+1)*10 samples filled with g+1
        lens = torch.tensor([(g + 1) * 10 for g in range(lo, hi)], dtype=torch.int64)
        wave = torch.zeros(hi - lo, int(lens.max()) + 3)              # stride > longest utterance
        for i, g in enumerate(range(lo, hi)):
            wave[i, :lens[i]] = g + 1
        all_lens = dpx.exchange_lens(lens, b_cap=4)
        assert all_lens.shape == (2, 5)
        assert all_lens[0].tolist() == [3, 10, 20, 30, 0] and all_lens[1].tolist() == [2, 40, 50, 0, 0]
        for trial in range(3):                                        # persistent buffers: repeated calls
            out = dpx.gather_audio(wave, lens, all_lens=all_lens if trial else None)
            if rank == 0:
                assert len(out) == 5
                for g, w in enumerate(out):
                    assert w.shape == ((g + 1) * 10,) and bool((w == g + 1).all())
            else:
                assert out is None
        # two channels: the fan-in travels on its own process group, so the length exchange of the NEXT batch may be issued
        # while a batch's audio is still on its way (the two-stream schedule of ttsamd.pipeline at world > 1)
        assert dpx.group_audio is not None and dpx.group_audio is not dist.group.WORLD
        nxt = dpx.exchange_lens(lens + 1, b_cap=4)
        out = dpx.gather_audio(wave, lens, all_lens=all_lens)
        assert nxt[0].tolist() == [3, 11, 21, 31, 0] and nxt[1].tolist() == [2, 41, 51, 0, 0]
        assert (out is None) == (rank != 0) and (rank != 0 or [w.numel() for w in out] == [10, 20, 30, 40, 50])
        ptr = dpx._recv.buf.data_ptr() if rank == 0 else dpx._pack.buf.data_ptr()
        dpx.gather_audio(wave, lens, all_lens=all_lens)
        assert ptr == (dpx._recv.buf.data_ptr() if rank == 0 else dpx._pack.buf.data_ptr()), 'steady state must not reallocate'
        # a rank with nothing to send
        e_lens = lens if rank == 0 else torch.zeros(0, dtype=torch.int64)
        e_wave = wave if rank == 0 else torch.zeros(0, 1)
        out = dpx.gather_audio(e_wave, e_lens, b_cap=4)
        if rank == 0:
            assert [w.numel() for w in out] == [10, 20, 30]
        # end-to-end sharded tts: rank-sharded result == 1-rank result, bit-identical per utterance
        texts = ['x' * n for n in (5, 9, 3, 7, 1, 12, 2)]
        for model in (_Stub(), _StubDevice()):
            for bs in (2, 3, 8):
                res = dp.tts_sharded(model, texts, batch_size=bs, dp=dpx)
                if rank == 0:
                    single = _Stub().tts(texts)
                    assert [w.numel() for w in res] == [len(t) for t in texts]
                    assert all(torch.equal(a, b) for a, b in zip(res, single))
                else:
                    assert res is None
        # module-level helpers keep working
        out = dp.gather_audio(wave, lens)
        assert (out is None) == (rank != 0)
        dp.reset_default()
        with open(os.path.join(tmpdir, f'ok{rank}'), 'w') as f:
            f.write('ok')
    finally:
        dist.destroy_process_group()


def test_dp_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def _worker_many(rank, world, port, tmpdir, n_utts):
    """world 4 / 8: uneven shards, ranks with ZERO utterances, the two-channel schedule (the next batch's length exchange issued
    between a batch's length exchange and its audio fan-in), sharded tts == 1-rank result bit for bit."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'tts-arabic-pytorch_amd'))
    from ttsamd import dp
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dpx = dp.Dp(torch.device('cpu'))
        spans = [dp.shard_bounds(n_utts, world, r) for r in range(world)]
        lo, hi = spans[rank]
        b_cap = max(h - l for l, h in spans)
        assert min(h - l for l, h in spans) == 0 or n_utts >= world          # the world-8 case has empty ranks
        for step in range(3):                                                 # three batches in flight order: lens(i+1) before audio(i)
            def batch(i):
                lens = torch.tensor([(g + 1) * 7 + i for g in range(lo, hi)], dtype=torch.int64)
                wave = torch.zeros(hi - lo, (int(lens.max()) if hi > lo else 0) + 2)
                for k, g in enumerate(range(lo, hi)):
                    wave[k, :lens[k]] = 100 * i + g + 1
                return lens, wave
            lens, wave = batch(step)
            all_lens = dpx.exchange_lens(lens, b_cap=b_cap)
            assert tuple(all_lens.shape) == (world, b_cap + 1)
            for r, (l, h) in enumerate(spans):
                assert int(all_lens[r, 0]) == h - l
                assert all_lens[r, 1:1 + h - l].tolist() == [(g + 1) * 7 + step for g in range(l, h)]
                assert all(int(v) == 0 for v in all_lens[r, 1 + h - l:])
            nxt_lens, _ = batch(step + 1)
            nxt = dpx.exchange_lens(nxt_lens, b_cap=b_cap)                    # channel 1 again, BEFORE this batch's fan-in on channel 2
            out = dpx.gather_audio(wave, lens, all_lens=all_lens)
            assert nxt[:, 0].tolist() == [h - l for l, h in spans]
            if rank == 0:
                assert len(out) == n_utts
                for g, w in enumerate(out):
                    assert w.shape == ((g + 1) * 7 + step,) and bool((w == 100 * step + g + 1).all())
            else:
                assert out is None
        texts = ['x' * n for n in (5, 9, 3, 7, 1, 12, 2, 4, 4, 11, 6, 8, 10)][:n_utts + 3]
        for model in (_Stub(), _StubDevice()):
            for bs in (2, 5):
                res = dp.tts_sharded(model, texts, batch_size=bs, dp=dpx)
                if rank == 0:
                    assert all(torch.equal(a, b) for a, b in zip(res, _Stub().tts(texts))) and len(res) == len(texts)
                else:
                    assert res is None
        dpx.close()
        assert dpx.group_audio is None                                        # the second process group is released
        with open(os.path.join(tmpdir, f'ok{rank}'), 'w') as f:
            f.write('ok')
    finally:
        dist.destroy_process_group()


def test_dp_world4_uneven_shards_gloo(tmp_path):
    """10 utterances over 4 ranks: shards of 3 / 3 / 2 / 2."""
    mp.spawn(_worker_many, args=(4, _free_port(), str(tmp_path), 10), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(4))


def test_dp_world8_ranks_without_utterances_gloo(tmp_path):
    """5 utterances over 8 ranks (the last batch of a list on an 8-GPU node): ranks 5..7 hold nothing and still take part in both
    exchanges of the two-channel schedule."""
    mp.spawn(_worker_many, args=(8, _free_port(), str(tmp_path), 5), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(8))


def test_shard_bounds_cover():
    from ttsamd import dp
    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_shard_indices_round_robin():
    """tts_sharded deals the length-sorted list round-robin: every index exactly once, shard sizes within one, and
    chunk c of every rank covers the same stretch of the sorted list (equal work per collective)."""
    from ttsamd import dp
    for n in (0, 1, 7, 32, 257):
        order = list(range(n))[::-1]
        for world in (1, 2, 3, 8):
            shards = [dp.shard_indices(order, world, r) for r in range(world)]
            assert sorted(i for sh in shards for i in sh) == sorted(order)
            assert max(map(len, shards)) - min(map(len, shards)) <= 1
            for sh in shards:
                assert all(order.index(a) + world == order.index(b) for a, b in zip(sh, sh[1:]))


def test_bench_self_launch_relays_failure():
    """`python bench.py --gpus 2` typed as is starts its own ranks; without a GPU every rank fails and the
    launcher must exit non-zero (never hang, never print a JSON line)."""
    import subprocess
    import sys
    from conftest import REPO
    if torch.cuda.is_available():
        import pytest
        pytest.skip('a GPU is present: covered by tests/test_gpu_dp.py')
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, timeout=300)
    assert p.returncode != 0 and b'{' not in p.stdout
