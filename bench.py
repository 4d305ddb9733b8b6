#!/usr/bin/env python3
"""bench.py — audio samples/s of the FastPitch -> HiFi-GAN hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--tokens 64]

Workload (BASELINE.json configs[1]): FastPitch+HiFi-GAN, synthetic 64-phoneme x batch 32
per GPU, fp32, synthetic weights (ttsamd.synth, seed 0), forced durations dur_tgt in [2,12]
(mean 7 frames/token, T_i ~ 448) so the work is deterministic.  A step = one
.tts_batch()-equivalent: ids already in HBM -> encoder+predictors -> (host reads dec_lens,
as the reference does) -> length regulator -> decoder -> ragged batched HiFi-GAN -> audio in
HBM.  N>1: one process per GPU — under torch.distributed.run, or started by this script itself when
`--gpus N` is typed without it — weights broadcast once from rank 0 over RCCL (ttsamd_dp_*), B utterances
PER RANK (weak scaling), every rank's lengths all-gathered in the step's one host sync and the packed
audio fanned in to rank 0 every step.  Prints ONE JSON line on rank 0; at N=1 the line also carries
`configs`: the batch-1 and batch-8 sub-results of the same step function.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
SAMPLE_RATE = 22050


def cpu_baseline(fp_sd, hg_sd, tokens, seconds_budget=25.0):
    """The CPU oracle (own restatement of the reference's torch-CPU path, pinned against the
    real reference by tests/test_oracle_golden.py) timed on this host's cores on a bounded
    sample of the same workload: batched FastPitch + per-utterance vocoder loop
    (models/fastpitch/networks.py:322-350)."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    fw = O.to_torch(fp_sd)
    hw = O.fold_weight_norm(hg_sd)
    b = 2
    ids = synth.synth_ids(b, tokens)
    dur = synth.synth_durations(b, tokens)
    n_default = torch.get_num_threads()
    sweep = {}
    with torch.inference_mode():
        O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids[:1, :8], dur_tgt=dur[:1, :8])      # warm-up
        # torch's intra-op pool does not scale to hundreds of host threads on these small per-utterance convs:
        # try a few pool sizes on one batch each and keep the best one for the timed sample
        for n_thr in sorted({n_default, min(n_default, 32), min(n_default, 8)}, reverse=True):
            torch.set_num_threads(n_thr)
            t0 = time.perf_counter()
            _, _, waves = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids, dur_tgt=dur)
            sweep[n_thr] = int(sum(w.numel() for w in waves)) / (time.perf_counter() - t0)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        n_samples, n_utts = 0, 0
        while True:
            _, dec_lens, waves = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids, dur_tgt=dur)
            n_samples += int(sum(w.numel() for w in waves))
            n_utts += b
            el = time.perf_counter() - t0
            if el > seconds_budget * 0.4:
                break
    out = {'value': n_samples / el, 'unit': 'audio samples/s', 'cores': torch.get_num_threads(),
           'kind': 'port', 'sample': f'{n_utts} utterances x {tokens} tokens (batch {b}, forced durations), '
                                     f'{el:.1f} s of torch-CPU fp32 on {os.cpu_count()} host cpus',
           'rtf': el / (n_samples / SAMPLE_RATE),
           'thread_sweep': {str(k): v for k, v in sweep.items()}}
    # single-thread figure on a quarter-length utterance so it stays within a few seconds
    try:
        torch.set_num_threads(1)
        q = max(8, tokens // 4)
        with torch.inference_mode():
            t0 = time.perf_counter()
            _, _, waves = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids[:1, :q], dur_tgt=dur[:1, :q])
            el1 = time.perf_counter() - t0
        out['one_thread'] = {'value': int(waves[0].numel()) / el1, 'unit': 'audio samples/s', 'cores': 1,
                             'sample': f'1 utterance x {q} tokens, {el1:.1f} s'}
    finally:
        torch.set_num_threads(n_default)
    return out


def _self_launch(args):
    """`python bench.py --gpus N` typed as is (no torch.distributed.run around it): start N ranks as child
    processes BEFORE anything in this process touches the GPU, relay rank 0's JSON line, fail if any child fails.
    With fewer than N GPUs on the box (the 1-GPU dev box) the ranks share device 0 and talk over gloo
    (TTSAMD_BENCH_ONE_DEVICE=1): a functional check of the N>1 path, labelled as such in the line."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()          # counts devices without initialising the HIP runtime
    env = dict(os.environ)
    if n_dev < args.gpus:
        env['TTSAMD_BENCH_ONE_DEVICE'] = '1'
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out0.decode().splitlines():                       # stdout carries the ONE json line; library chatter (gloo) goes to stderr
        print(ln, file=sys.stdout if ln.startswith('{') else sys.stderr)
    sys.stdout.flush()
    if any(rcs):
        print(f'bench.py: rank exit codes {rcs}', file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def _time_steps(step, steps, sync, barrier=None):
    """K steps bracketed by barrier + synchronize on both sides -> seconds."""
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    if barrier:
        barrier()
    sync()
    return time.perf_counter() - t0, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='utterances per GPU')
    ap.add_argument('--tokens', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-small', action='store_true', help='skip the batch 1 / batch 8 sub-results (N=1 only)')
    ap.add_argument('--pipeline', action='store_true',
                    help='two HIP streams (ttsamd.pipeline): FastPitch of step i+1 under HiFi-GAN of step i; measured +1.3 %% at '
                         'B=32 (the conv engine is busy either way), so the default stays the one-stream schedule')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'bf16x3'],
                    help='MFMA operand precision of the conv/linear GEMMs (f32 = BASELINE config 2; bf16 = config 3; '
                         'bf16x3 = split bf16, fp32-class accuracy)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        _self_launch(args)                      # never returns
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    # TTSAMD_BENCH_ONE_DEVICE=1 (set by _self_launch on a box with fewer GPUs than ranks): all ranks share GPU 0
    # and exchange over gloo with host staging — exercises the N>1 code path on a 1-GPU box; the real
    # multi-GPU run is one rank per GPU over RCCL.
    one_dev = os.environ.get('TTSAMD_BENCH_ONE_DEVICE') == '1' or world > torch.cuda.device_count()
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from ttsamd import synth, lib as L
    from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision
    import torch.distributed as dist

    set_precision(args.precision)
    dpx, transport = None, None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_dev:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        from ttsamd import dp
        transport = os.environ.get('TTSAMD_DP_TRANSPORT') or ('torch' if one_dev else 'rccl')
        if transport == 'rccl':
            # every rank probes the C-ABI RCCL binding (dlopen + id); if any rank cannot, ALL ranks use the same
            # exchanges through torch.distributed's RCCL instead (same wire, same buffers) and the line says so
            import ctypes
            ok = torch.tensor([1 if L.load().ttsamd_dp_unique_id((ctypes.c_char * 128)()) == 0 else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                print('bench.py: ttsamd_dp_* unavailable on some rank, using torch.distributed transport', file=sys.stderr)
                transport = 'torch'
        dpx = dp.Dp(dev, transport=transport)
        # C1: only rank 0 holds the checkpoint
        fp_sd = synth.fastpitch_state_dict() if rank == 0 else None
        hg_sd = synth.hifigan_state_dict() if rank == 0 else None
        if transport == 'rccl':
            fp = FastPitchEngine(dpx.broadcast_shapes(fp_sd), device=dev)
            hg = HifiGanEngine(dpx.broadcast_shapes(hg_sd), device=dev)
            dpx.broadcast_weights(fp)
            dpx.broadcast_weights(hg)
        else:
            fp_sd = dpx.broadcast_state_dict(fp_sd)
            hg_sd = dpx.broadcast_state_dict(hg_sd)
            fp = FastPitchEngine(fp_sd, device=dev)
            hg = HifiGanEngine(hg_sd, device=dev)
    else:
        fp_sd, hg_sd = synth.fastpitch_state_dict(), synth.hifigan_state_dict()
        fp = FastPitchEngine(fp_sd, device=dev)
        hg = HifiGanEngine(hg_sd, device=dev)
    B, Lt = args.batch, args.tokens
    # distinct synthetic utterances per rank (global batch = world * B)
    ids_all = synth.synth_ids(world * B, Lt)
    dur_all = synth.synth_durations(world * B, Lt)
    ids = torch.from_numpy(ids_all[rank * B:(rank + 1) * B]).to(dev)
    dur = torch.from_numpy(dur_all[rank * B:(rank + 1) * B]).to(dev)
    hop = hg.hop

    # --pipeline: two HIP streams (ttsamd.pipeline), the acoustic model of step i + 1 (150 short launches) under the vocoder of
    # step i.  Same work per step, same results; 79.4 -> 78.4 ms per step at B=32 (only FastPitch's non-conv kernels find idle CUs).
    from ttsamd.pipeline import FastPitchHifiGan
    pipe = FastPitchHifiGan(fp, hg, dev) if args.pipeline else None

    def make_step(ids_, dur_):
        if world == 1:
            if pipe is not None:
                def step():
                    _, dec_lens, wave = pipe.submit(ids_, dur_tgt=dur_)
                    return wave, dec_lens
                return step

            def step():
                mel, dec_lens, *_ = fp.infer(ids_, dur_tgt=dur_)
                return hg.forward(mel, dec_lens), dec_lens
            return step
        state = {}

        def hook(dec_lens):                      # ONE host sync per step: own + every rank's lengths
            state['all'] = dpx.exchange_lens(dec_lens, ids_.shape[0])
            return state['all'][rank, 1:1 + ids_.shape[0]]

        def vocode(mel, dec_lens):
            wave = hg.forward(mel, dec_lens)
            all_samples = state['all'].copy()
            all_samples[:, 1:] *= hop
            dpx.gather_flat(wave, dec_lens * hop, all_lens=all_samples)     # C2: audio fan-in to rank 0
            return wave

        def step():
            if pipe is not None:
                _, dec_lens, wave = pipe.submit(ids_, vocode=vocode, dur_tgt=dur_, lens_hook=hook)
                return wave, dec_lens
            mel, dec_lens, *_ = fp.infer(ids_, dur_tgt=dur_, lens_hook=hook)
            return vocode(mel, dec_lens), dec_lens
        return step

    step = make_step(ids, dur)
    sync = torch.cuda.synchronize
    barrier = dist.barrier if world > 1 else None
    if world > 1 and transport == 'rccl':
        # first DP step under guard: if the C-ABI RCCL exchange raises on ANY rank, every rank switches to the same
        # exchanges through torch.distributed (same wire) instead of failing the run; the line reports which one ran
        try:
            step()
            sync()
            ok = 1
        except Exception as e:                                   # noqa: BLE001
            print(f'bench.py rank {rank}: ttsamd_dp_* step failed ({e}); falling back to torch.distributed', file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            dpx.close()
            transport = 'torch'
            dpx = dp.Dp(dev, transport=transport)
            step = make_step(ids, dur)
    for _ in range(args.warmup):
        step()
    sync()

    lib = L.load()
    import ctypes
    lib.ttsamd_profile_enable(1)
    elapsed, (wave, dec_lens) = _time_steps(step, args.steps, sync, barrier)
    prof = (ctypes.c_double * 3)()
    L.check(lib.ttsamd_profile_read(prof), 'profile_read')
    lib.ttsamd_profile_enable(0)

    frames = int(dec_lens.sum().item())
    samples = frames * hop * args.steps
    tot = torch.tensor([elapsed, float(samples)], dtype=torch.float64, device=dev if not one_dev else 'cpu')
    if world > 1:
        mx = tot[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot[1:].clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, samples = float(mx[0]), float(sm[0])

    from ttsamd.config import NET_CONFIG as NC, HIFIGAN_CONFIG as HC
    hg_fpf = hifigan_flops_per_frame(HC)
    dec_fpt, enc_fpt = fastpitch_conv_flops_per_pos(NC)
    # peak of the MFMA instruction actually issued: fp32 157.3; bf16 2500 dense; split bf16 issues
    # 3 bf16 MFMAs per algorithmic product -> 2500/3 algorithmic TFLOP/s
    peak = {'f32': PEAK_F32_MFMA_TFLOPS, 'bf16': 2500.0, 'bf16x3': 2500.0 / 3}[args.precision]

    def small_config(b):
        """Batch-b sub-result (north star: batch 1 / 8 / 32): same step function, own warm-up, timed WITHOUT the
        per-launch events (they cost a B=1 call 17 %); its roofline figure is algorithmic FLOPs over the call's
        WALL time — launch gaps and the three-stream overlap included — so it can never exceed what ran."""
        ids_b, dur_b = ids[:b].contiguous(), dur[:b].contiguous()
        st = make_step(ids_b, dur_b)
        for _ in range(5):
            st()
        sync()
        n = max(args.steps, 20)
        el, (_, dl) = _time_steps(st, n, sync)
        fr, tm = int(dl.sum().item()), int(dl.max().item())
        flops = hg_fpf * fr + dec_fpt * b * tm + enc_fpt * b * Lt
        ach = flops / (el / n) / 1e12
        return {'batch': b, 'ms_per_step': el / n * 1e3, 'value': fr * hop * n / el, 'unit': 'audio samples/s',
                'rtf': el / (fr * hop * n / SAMPLE_RATE), 'frames': fr, 'steps': n,
                'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                             'basis': 'algorithmic conv FLOPs / wall time of the whole call (no per-launch events)'}}

    if rank == 0:
        conv_ms, n_launch = prof[0], prof[1]
        # algorithmic FLOPs (2*Cout*Cin*K per output position) of the bracketed MFMA conv launches on
        # this rank: HiFi-GAN convs process sum(frames)*upsampling positions (ragged, early exit),
        # FastPitch decoder convs B*T_max positions, encoder/predictor convs B*L positions.
        t_max = int(dec_lens.max().item())
        flops = args.steps * (hg_fpf * frames + dec_fpt * B * t_max + enc_fpt * B * Lt)
        achieved = flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # HBM bytes per conv launch: OFFLINE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (profiles/rN/traffic.json; gpurun forbids mixing PMC with the timed run), newest round first
        traffic, traffic_src = None, None
        if B == 32 and Lt == 64 and args.precision == 'f32' and world == 1:
            for rnd in ('r2', 'r1'):
                try:
                    with open(os.path.join(REPO, 'profiles', rnd, 'traffic.json')) as f:
                        traffic = json.load(f)['bytes_per_conv_launch_corrected']
                    traffic_src = f'offline PMC pass, profiles/{rnd}/traffic.json'
                    break
                except (OSError, KeyError):
                    pass
        prec_name = {'f32': 'fp32', 'bf16': 'bf16 MFMA', 'bf16x3': 'split-bf16 MFMA'}[args.precision]
        par = f'dp{world}' + (' (ranks share ONE device, gloo + host staging: functional check, not a scaling number)' if one_dev else '')
        out = {
            'metric': 'audio samples/sec (FastPitch+HiFi-GAN, synthetic 64-phoneme inputs)',
            'value': samples / elapsed, 'unit': 'audio samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': {'f32': 'f32', 'bf16': 'bf16', 'bf16x3': 'f32 via split-bf16 (3x bf16 MFMA, fp32 accumulate)'}[args.precision], 'data': 'synthetic (ids, forced durations, random-init weights)',
            'rtf': elapsed / (samples / SAMPLE_RATE),
            'config': {'workload': f'FastPitch+HiFi-GAN, synthetic {Lt}-phoneme x batch{B} per GPU, {prec_name}, '
                                   f'{world}xMI355X', 'batch_per_gpu': B, 'n_tokens': Lt,
                       'frames_per_step_rank0': frames, 'parallelism': par, 'dp_transport': transport,
                       'schedule': ('two HIP streams: FastPitch of step i+1 under HiFi-GAN of step i (ttsamd.pipeline)' if args.pipeline else
                                    'one stream per step')},
            'roofline': {'bound': 'mfma' if args.precision == 'f32' else 'hbm (fp32 activations; MFMA figures for reference)',
                         'kernel': ('MFMA conv engine: conv1d_mfma_f32 + resblock_pair + convt_mfma_f32' if args.precision == 'f32' else 'conv1d_mfma_bf16') + ' (all instantiations)',
                         'kernel_time_basis': 'HIP events on the launch stream: one pair per launch, one pair per fork..join section of the three-stream ResBlock schedule (wall time of the section); kernel time = length of the UNION of these intervals (the two pipeline streams overlap)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'B/launch', 'traffic_source': traffic_src,
                         'launches': int(n_launch), 'avg_launch_ms': conv_ms / max(1.0, n_launch),
                         'kernel_ms_per_step': conv_ms / args.steps},
        }
        if world == 1 and not args.no_small and B > 8:
            out['configs'] = [small_config(1), small_config(8)]
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(fp_sd, hg_sd, Lt)
        elif world > 1:
            out['cpu_baseline'] = None
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dpx.close()
        dist.destroy_process_group()


def hifigan_flops_per_frame(h):
    """Algorithmic FLOPs (2*MAC) of the MFMA conv launches per mel frame: conv_pre, the
    polyphase upsamplers and every ResBlock1 conv (614.1 MFLOP/frame incl. conv_post)."""
    c0 = h['upsample_initial_channel']
    f = 2.0 * c0 * h['num_mels'] * 7
    ch, mul = c0, 1
    for u, k in zip(h['upsample_rates'], h['upsample_kernel_sizes']):
        f += 2.0 * (ch // 2) * ch * k / u * (mul * u)        # k/u taps per output sample
        ch, mul = ch // 2, mul * u
        for kk, dil in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
            f += len(dil) * 2 * (2.0 * ch * ch * kk) * mul
    return f


def fastpitch_conv_flops_per_pos(c):
    """(decoder FLOP per frame incl. proj, encoder+predictor FLOP per token) of the MFMA conv launches."""
    d = c['symbols_embedding_dim']

    def layer(dh, nh, filt, k):
        return 2.0 * d * 3 * nh * dh + 2.0 * nh * dh * d + 2.0 * d * filt * k * 2
    dec = c['out_fft_n_layers'] * layer(c['out_fft_d_head'], c['out_fft_n_heads'], c['out_fft_conv1d_filter_size'],
                                        c['out_fft_conv1d_kernel_size']) + 2.0 * d * c['n_mel_channels']
    enc = c['in_fft_n_layers'] * layer(c['in_fft_d_head'], c['in_fft_n_heads'], c['in_fft_conv1d_filter_size'],
                                       c['in_fft_conv1d_kernel_size'])
    for p in ('dur', 'pitch', 'energy'):
        if p == 'energy' and not c['energy_conditioning']:
            continue
        f, k, n = c[f'{p}_predictor_filter_size'], c[f'{p}_predictor_kernel_size'], c[f'{p}_predictor_n_layers']
        enc += 2.0 * d * f * k + (n - 1) * 2.0 * f * f * k
    return dec, enc


if __name__ == '__main__':
    main()
