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
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # one hardware queue per stream the path uses (ttsamd/__init__.py), before any GPU call

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E, spec
# what v_mfma_f32_32x32x16_bf16 sustains on random data with nothing else running (register-only loop on every SIMD, power-managed
# clock 1.72-1.78 GHz): tools/mfma_peak_bench.hip, profiles/r3/mfma_sustained_peak.txt.  Reported NEXT to the guide's peak, never instead.
BF16_MFMA_SUSTAINED_TFLOPS = 1700.0
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
    out = {'value': n_samples / el, 'unit': 'audio samples/s', 'cores': torch.get_num_threads(), 'host_cpus': os.cpu_count(),
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


def cpu_baseline_c1(fp_sd, hg_sd, threads, seconds_budget=10.0):
    """BASELINE configs[0] on this host's cores: the reference's own CPU case (inference.py:45-58 -> FastPitch2Wave.tts on
    data/infer_text.txt, batch_size=1; models/fastpitch/networks.py:402-411) restated by the oracle -- the first N committed lines
    (token ids of the REAL reference's tokeniser, tests/golden/infer_text_ids.npz) one line at a time, PREDICTED durations, vocoder
    per utterance, denoise off -- for as many lines as fit the budget.  Beside `configs[]` C1's GPU figure (same lines, batch_size=1)."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import tts_oracle as O
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    g = dict(np.load(os.path.join(REPO, 'tests', 'golden', 'infer_text_ids.npz'), allow_pickle=False))
    fw, hw = O.to_torch(fp_sd), O.fold_weight_norm(hg_sd)
    n_default = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        with torch.inference_mode():
            n_lines = n_samples = n_tok = 0
            t0 = time.perf_counter()
            while n_lines < len(g['offsets']) - 1:
                ids = np.asarray(g['flat'][g['offsets'][n_lines]:g['offsets'][n_lines + 1]], np.int64)[None]
                _, _, waves = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids)
                n_samples += int(waves[0].numel())
                n_tok += ids.shape[1]
                n_lines += 1
                el = time.perf_counter() - t0
                if el > seconds_budget:
                    break
    finally:
        torch.set_num_threads(n_default)
    return {'value': n_samples / el, 'unit': 'audio samples/s', 'cores': threads, 'kind': 'port', 'ms_per_utterance': el * 1e3 / n_lines,
            'rtf': el / (n_samples / SAMPLE_RATE), 'lines': n_lines, 'tokens': n_tok, 'samples': n_samples,
            'sample': f'the first {n_lines} of the 100 infer_text.txt lines ({n_tok} tokens), batch_size 1, predicted durations, denoise 0, '
                      f'{el:.1f} s of torch-CPU fp32 on {threads} threads (tokenisation not timed: ids of the reference tokeniser)'}


def _cpu_worker(job):
    """One worker process of cpu_baseline_all_cores (spawned: never touches the GPU): its own copy of the synthetic weights,
    `threads` intra-op threads, distinct utterances (rows first, first + stride, ... of the synthetic batch), timed for
    `seconds` after one warm-up utterance.  Returns (samples, utterances, seconds)."""
    first, stride, threads, tokens, seconds = job
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    torch.set_num_threads(threads)
    fw = O.to_torch(synth.fastpitch_state_dict())
    hw = O.fold_weight_norm(synth.hifigan_state_dict())
    ids = synth.synth_ids(first + 64 * stride + 1, tokens)
    dur = synth.synth_durations(first + 64 * stride + 1, tokens)
    with torch.inference_mode():
        O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids[:1, :8], dur_tgt=dur[:1, :8])
        t0 = time.perf_counter()
        n_samples = n_utts = 0
        row = first
        while time.perf_counter() - t0 < seconds and n_utts < 64:
            _, _, waves = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids[row:row + 1], dur_tgt=dur[row:row + 1])
            n_samples += int(waves[0].numel())
            n_utts += 1
            row += stride
        return n_samples, n_utts, time.perf_counter() - t0


def _usable_cpus():
    """Cores this process may actually use: the affinity mask and the cgroup CPU quota (cpu.max / cfs_quota) cap what os.cpu_count()
    reports -- a 256-thread host behind a 32-core quota runs 256 threads SLOWER than 32 (the thread sweep of cpu_baseline shows it)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                     # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()
            if q != 'max':
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:   # cgroup v1
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline_all_cores(tokens, threads=8, seconds=12.0):
    """Utterance-parallel CPU baseline on the WHOLE host: N = host_cpus // threads worker processes x `threads` intra-op threads,
    each synthesising distinct utterances one at a time (utterances are independent, so this is how a CPU deployment of the
    reference would use a many-core host; torch's intra-op pool alone stops scaling at ~8 threads on these per-utterance convs).
    Aggregate = total samples / the slowest worker's time."""
    import multiprocessing as mp
    host = os.cpu_count() or 1
    usable, quota = _usable_cpus()
    eff = int(min(usable, quota)) if quota else usable
    n = max(1, min(eff // threads, 64))
    ctx = mp.get_context('spawn')
    t0 = time.perf_counter()
    with ctx.Pool(n) as pool:
        # bounded: worker start-up (a cold `import torch`, weight synthesis) has taken minutes on a loaded host; the leg is a context
        # figure, not worth more than ~1.5 min of the bench's wall time (TimeoutError -> the caller records the error string)
        res = pool.map_async(_cpu_worker, [(w, n, threads, tokens, seconds) for w in range(n)]).get(timeout=90.0 + seconds)
    wall = time.perf_counter() - t0
    samples, utts, slowest = sum(r[0] for r in res), sum(r[1] for r in res), max(r[2] for r in res)
    return {'value': samples / slowest, 'unit': 'audio samples/s', 'cores': n * threads, 'workers': n, 'threads_per_worker': threads,
            'host_cpus': host, 'usable_cpus': usable, 'cgroup_cpu_quota': quota, 'kind': 'port',
            'sample': f'{utts} distinct utterances x {tokens} tokens over {n} worker processes x {threads} threads, '
                      f'{slowest:.1f} s timed per worker ({wall:.1f} s incl. process start-up and weight synthesis)'}


def _count_gpus():
    """GPUs on this node WITHOUT initialising the HIP runtime in this process (the launcher / supervisor parents must never hold a
    GPU context: they kill and restart their children): the KFD topology in sysfs (a node with simd_count > 0 is a GPU), narrowed
    by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES; torch.cuda.device_count() only if sysfs is not there."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(',') if x.strip() != ''])
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(base):
            props = {}
            with open(os.path.join(base, node, 'properties')) as f:
                for ln in f:
                    kv = ln.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
            if int(props.get('simd_count', '0')) <= 0:
                continue                                        # a CPU node
            minor = props.get('drm_render_minor')
            # a container may see the whole host's topology but own only some of the render nodes
            if minor is None or os.access(f'/dev/dri/renderD{minor}', os.R_OK | os.W_OK):
                n += 1
        return n
    except (OSError, ValueError):
        return torch.cuda.device_count()


def _self_launch(args):
    """`python bench.py --gpus N` typed as is (no torch.distributed.run around it): start N ranks as child
    processes BEFORE anything in this process touches the GPU, relay rank 0's JSON line, fail if any child fails.
    With fewer than N GPUs on the box (the 1-GPU dev box) the ranks share device 0 and talk over gloo
    (TTSAMD_BENCH_ONE_DEVICE=1): a functional check of the N>1 path, labelled as such in the line.

    This process never touches the GPU, so it is the WATCHDOG of the ranks: if rank 0 has not printed its line
    within the budget (TTSAMD_BENCH_WATCHDOG_S, default 300 s + 2 s per step: a cold `import torch` alone can take
    two minutes on a fresh box), or any rank dies, every child is killed (exact PIDs) and FRESH children are started
    once more with TTSAMD_DP_TRANSPORT=torch (the exchanges through torch.distributed's own RCCL process group
    instead of the ttsamd_dp_* communicator); the line then says which transport produced it and why.  A hang of
    a collective between ranks (the one failure a try/except inside a rank cannot see) therefore costs the budget
    once, never the caller's whole time-out."""
    import socket
    import subprocess
    n_dev = _count_gpus()                      # sysfs: this process never initialises the HIP runtime
    base_env = dict(os.environ)
    if n_dev < args.gpus:
        base_env['TTSAMD_BENCH_ONE_DEVICE'] = '1'
    base_env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    budget = float(os.environ.get('TTSAMD_BENCH_WATCHDOG_S', 300.0 + 2.0 * (args.steps + args.warmup)))

    def attempt(n, extra):
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(base_env, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus),
                   TTSAMD_BENCH_ATTEMPT=str(n), TTSAMD_BENCH_WORKER='1', **extra)
        procs = []
        for r in range(args.gpus):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        deadline, out0, why = time.monotonic() + budget, None, None
        while True:
            try:
                out0, _ = procs[0].communicate(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                dead = [r for r, p in enumerate(procs[1:], 1) if p.poll() not in (None, 0)]
                if dead:
                    why = f'rank(s) {dead} exited with {[procs[r].returncode for r in dead]} while rank 0 was still running'
                    break
                if time.monotonic() > deadline:
                    why = f'no result within the watchdog budget of {budget:.0f} s'
                    break
        if why is not None:                                        # stalled or broken: end every child we started
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            return None, why
        rcs = [procs[0].returncode]
        for p in procs[1:]:
            try:
                rcs.append(p.wait(timeout=60))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append(p.wait())
        if any(rcs):
            return None, f'rank exit codes {rcs}'
        return out0.decode(), None

    out, why = attempt(0, {})
    if out is None:
        print(f'bench.py watchdog: {why}; restarting all ranks with TTSAMD_DP_TRANSPORT=torch', file=sys.stderr)
        out, why2 = attempt(1, {'TTSAMD_DP_TRANSPORT': 'torch', 'TTSAMD_BENCH_FALLBACK_REASON': why})
        why = why2
    if out is None:
        print(f'bench.py: {why}', file=sys.stderr)
        sys.exit(1)
    for ln in out.splitlines():                                   # stdout carries the ONE json line; library chatter (gloo) goes to stderr
        print(ln, file=sys.stdout if ln.startswith('{') else sys.stderr)
    sys.stdout.flush()
    sys.exit(0)


def _supervise_rank(args, rank, world):
    """A rank started by an EXTERNAL launcher (`python -m torch.distributed.run ... bench.py --gpus N`, what the driver uses)
    gets the same protection as the self-launched case: this process never touches the GPU; it runs the real rank as a child
    process and watches it.  If the child stalls past the budget, dies, or ANY rank's supervisor on the node reports a failure,
    every supervisor kills its child and starts a FRESH one with TTSAMD_DP_TRANSPORT=torch on a rendezvous of its own.  The
    launcher itself only ever sees the supervisors, which exit non-zero only when the second attempt failed too.

    The supervisors talk through files in the temp directory keyed by a per-RUN nonce (the launcher's TORCHELASTIC_RUN_ID, its
    port and its pid: an earlier run on the same port can leave nothing behind that this one would read).  Each rank writes only
    ITS OWN failure flag `<key>_a<attempt>_failed_r<rank>`, reads everybody's, and removes its own files when it exits: no
    rank ever deletes a file another rank may just have written.  The retry's rendezvous port is probed free by rank 0's
    supervisor (bind to port 0) and published in `<key>_retry_port`; the launcher's own store still holds the first attempt's keys."""
    import glob
    import socket
    import subprocess
    import tempfile
    import threading
    port0 = int(os.environ.get('MASTER_PORT', '29500'))
    run_id = os.environ.get('TORCHELASTIC_RUN_ID', 'none')
    key = os.path.join(tempfile.gettempdir(), f'ttsamd_bench_{run_id}_{port0}_{os.getppid()}')
    budget = float(os.environ.get('TTSAMD_BENCH_WATCHDOG_S', 300.0 + 2.0 * (args.steps + args.warmup)))
    mine = []                                                    # files this supervisor wrote

    def put(path, text=''):
        tmp = f'{path}.tmp{os.getpid()}'
        try:
            with open(tmp, 'w') as f:
                f.write(text)
            os.replace(tmp, path)                                # atomic: a reader never sees a half-written port
            mine.append(path)
        except OSError:
            pass

    def cleanup():
        for path in mine:
            try:
                os.remove(path)
            except OSError:
                pass

    why = None
    try:
        for attempt in (0, 1):
            env = dict(os.environ, TTSAMD_BENCH_WORKER='1', TTSAMD_BENCH_ATTEMPT=str(attempt),
                       TTSAMD_BENCH_INIT_TIMEOUT_S=str(int(budget) + 120))
            if attempt == 1:
                port_file = f'{key}_retry_port'
                if rank == 0:
                    with socket.socket() as sk:
                        sk.bind(('127.0.0.1', 0))
                        put(port_file, str(sk.getsockname()[1]))
                t_wait = time.monotonic() + 60
                while not os.path.exists(port_file) and time.monotonic() < t_wait:
                    time.sleep(0.2)
                try:
                    with open(port_file) as f:
                        retry_port = int(f.read())
                except (OSError, ValueError):
                    print(f'bench.py supervisor of rank {rank}: no retry port published by rank 0', file=sys.stderr)
                    break
                env.update(TTSAMD_DP_TRANSPORT='torch', MASTER_PORT=str(retry_port), TORCHELASTIC_USE_AGENT_STORE='False',
                           TTSAMD_BENCH_FALLBACK_REASON=why or 'first attempt failed')
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                     stdout=subprocess.PIPE if rank == 0 else None)
            lines = []

            def pump(stream=child.stdout):
                for ln in iter(stream.readline, b''):
                    lines.append(ln.decode())
            th = None
            if rank == 0:
                th = threading.Thread(target=pump, daemon=True)
                th.start()
            deadline, why, t_line = time.monotonic() + budget, None, None
            while child.poll() is None:
                time.sleep(0.5)
                if rank == 0 and t_line is None and any(ln.startswith('{') for ln in lines):
                    t_line = time.monotonic()                    # the line is out: only teardown is left
                if t_line is not None and time.monotonic() - t_line > 60:
                    break                                        # hung in teardown after a complete result: keep the result
                if glob.glob(f'{key}_a{attempt}_failed_r*'):
                    why = 'another rank reported a failure'
                    break
                if time.monotonic() > deadline:
                    why = f'no result within the watchdog budget of {budget:.0f} s'
                    break
            if child.poll() is None:
                child.kill()
            rc = child.wait()
            if th is not None:
                th.join(timeout=5)
            got = [ln for ln in lines if ln.startswith('{')]
            ok = (why is None and rc == 0) or (rank == 0 and got and why is None)
            if why is None and not ok:
                why = f'rank {rank} exited with {rc}'
            if not ok:
                put(f'{key}_a{attempt}_failed_r{rank}')          # tell the other supervisors
                print(f'bench.py supervisor of rank {rank}: {why}' + ('; restarting on TTSAMD_DP_TRANSPORT=torch' if attempt == 0 else ''),
                      file=sys.stderr)
                continue
            for ln in lines:
                print(ln, end='', file=sys.stdout if ln.startswith('{') else sys.stderr)
            sys.stdout.flush()
            # peers may still be polling for this attempt's flags: leave ours (there are none on success) and go
            cleanup()
            sys.exit(0)
    finally:
        if why is not None:
            time.sleep(2.0)                                      # let the peers' 0.5 s polls see the flags before they go
        cleanup()
    sys.exit(1)


def _time_steps(step, steps, sync, barrier=None, tick=None):
    """K steps bracketed by barrier + synchronize on both sides -> (seconds, last output, per-step ms list or None).
    `tick()` (optional) records a HIP event on the stream the step's LAST launch went to and returns it: the differences between
    consecutive events are the per-step times the median is taken over (no host synchronisation inside the timed region)."""
    if barrier:
        barrier()
    sync()
    evs = [tick()] if tick else None
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
        if tick:
            evs.append(tick())
    sync()
    if barrier:
        barrier()
    sync()
    el = time.perf_counter() - t0
    per_step = [a.elapsed_time(b) for a, b in zip(evs, evs[1:])] if tick else None
    return el, out, per_step


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return None if n == 0 else (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2]))


BF16_RIDGE_FLOP_PER_BYTE = 2500.0e12 / 8000.0e9      # 312.5: above it a bf16 launch sequence is bound by the matrix pipe, below by HBM


def bf16_roofline(flops, byts, sec, extra=None):
    """Roofline entry of a bf16 launch sequence: the bound is whichever roof the ALGORITHMIC intensity (FLOP / HBM byte) puts first.
    With the fused pairs / chained ResBlocks the step sits at ~560 FLOP/B, above the 312.5 FLOP/B ridge: MFMA-bound, so `frac` is the
    fraction of the 2.5 PFLOP/s dense bf16 peak; the HBM side rides along (and the other way round below the ridge)."""
    tf = flops / sec / 1e12
    gbs = byts / sec / 1e9
    inten = flops / max(byts, 1.0)
    hbm = {'hbm_achieved_gbs': gbs, 'hbm_peak_gbs': PEAK_HBM_GBS, 'hbm_frac': gbs / PEAK_HBM_GBS}
    mf = {'mfma_achieved_tflops': tf, 'mfma_peak_tflops': 2500.0, 'mfma_frac': tf / 2500.0,
          'mfma_sustained_tflops': BF16_MFMA_SUSTAINED_TFLOPS, 'mfma_frac_of_sustained': tf / BF16_MFMA_SUSTAINED_TFLOPS}
    if inten >= BF16_RIDGE_FLOP_PER_BYTE:
        r = {'bound': 'mfma', 'achieved': tf, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': tf / 2500.0}
    else:
        r = {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS}
    r.update({'algorithmic_flop_per_byte': inten, 'ridge_flop_per_byte': BF16_RIDGE_FLOP_PER_BYTE})
    r.update(hbm)
    r.update(mf)
    if extra:
        r.update(extra)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='utterances per GPU')
    ap.add_argument('--tokens', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-small', action='store_true', help='skip the batch 1 / batch 8 sub-results (N=1 only)')
    ap.add_argument('--pipeline', action='store_true', default=None,
                    help='two HIP streams (ttsamd.pipeline): FastPitch of step i+1 under HiFi-GAN of step i, same work and same '
                         'results per step (bit-identical).  The default at every precision and every N: fp32 +2 %% at B=32 (the conv '
                         'engine is busy either way), bf16 11.2 -> 9.8 ms per step (FastPitch is 18 %% of that step and mostly '
                         'launch- and latency-bound)')
    ap.add_argument('--no-pipeline', dest='pipeline', action='store_false', help='force the one-stream schedule')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='N > 1: weak = --batch utterances PER RANK (default); strong = --batch utterances in total, B/N per rank '
                         '(north star: B=256 on 1/2/4/8 GPUs)')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the extra driver-visible sub-results (config 3 bf16 share, config 4 Tacotron2, config 5 Vocos, D2H)')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'bf16x3'],
                    help='MFMA operand precision of the conv/linear GEMMs (f32 = BASELINE config 2; bf16 = config 3; '
                         'bf16x3 = split bf16, fp32-class accuracy)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        _self_launch(args)                      # never returns
    if world > 1 and os.environ.get('TTSAMD_BENCH_WORKER') != '1':
        _supervise_rank(args, rank, world)      # never returns: ranks of an external launcher watch their own worker
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
    red_dev = dev                                   # where the small host-side reductions (timings, flags) live
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from ttsamd import dp
        transport = os.environ.get('TTSAMD_DP_TRANSPORT') or ('torch' if one_dev else 'rccl')
        if os.environ.get('TTSAMD_BENCH_ATTEMPT', '0') == '0':      # tests/test_gpu_dp.py: fault injection, first attempt only
            if os.environ.get('TTSAMD_BENCH_TEST_STALL') == f'{rank}':
                time.sleep(3600)                    # a rank that never arrives
            if os.environ.get('TTSAMD_BENCH_TEST_DIE') == f'{rank}':
                os._exit(5)                         # a rank that dies before the rendezvous
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get('TTSAMD_BENCH_INIT_TIMEOUT_S', '600')))
        if one_dev or transport == 'rccl':
            # rendezvous, barriers and the few host-side reductions over gloo: with the ttsamd_dp_* transport a rank then
            # holds exactly ONE RCCL communicator (the library's), created right below, not a second one next to torch's
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=pg_timeout)
            red_dev = torch.device('cpu')
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        if transport == 'rccl':
            # every rank probes the C-ABI RCCL binding (dlopen + id) BEFORE the first collective; if any rank cannot,
            # ALL ranks leave (exit code 3) and the parent's watchdog restarts them on the torch.distributed transport
            import ctypes
            ok = torch.tensor([1 if L.load().ttsamd_dp_unique_id((ctypes.c_char * 128)()) == 0 else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                print(f'bench.py rank {rank}: ttsamd_dp_* unavailable on some rank', file=sys.stderr)
                dist.destroy_process_group()
                sys.exit(3)
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
    Lt = args.tokens
    # distinct synthetic utterances per rank: weak scaling = --batch per rank (global batch world * B), strong scaling =
    # --batch in total, contiguous balanced shards (ttsamd.dp.shard_bounds)
    if args.scaling == 'strong' and world > 1:
        from ttsamd.dp import shard_bounds
        g_batch = args.batch
        lo, hi = shard_bounds(g_batch, world, rank)
        assert g_batch >= world, '--scaling strong needs at least one utterance per rank'
    else:
        g_batch = world * args.batch
        lo, hi = rank * args.batch, (rank + 1) * args.batch
    B = hi - lo
    b_cap = -(-g_batch // world)                    # the largest shard (length exchange slot count)
    ids_all = synth.synth_ids(g_batch, Lt)
    dur_all = synth.synth_durations(g_batch, Lt)
    ids = torch.from_numpy(ids_all[lo:hi]).to(dev)
    dur = torch.from_numpy(dur_all[lo:hi]).to(dev)
    hop = hg.hop
    xch_ev = []                                     # (start, end) event pairs around the per-step audio fan-in

    # --pipeline: two HIP streams (ttsamd.pipeline), the acoustic model of step i + 1 (150 short launches) under the vocoder of
    # step i.  Same work per step, same results; 79.4 -> 78.4 ms per step at B=32 (only FastPitch's non-conv kernels find idle CUs).
    from ttsamd.pipeline import FastPitchHifiGan
    if args.pipeline is None:
        # default: every precision, every N (fp32 B=32: 78.1 -> 76.7 ms per step, bf16: 11.2 -> 9.8), so that `--gpus N` runs the schedule
        # the N = 1 line advertises; --no-pipeline gives the one-stream schedule.  At N > 1 the acoustic stream issues
        # the length all-gather and the vocoder stream the audio fan-in, each on its OWN communicator (ttsamd.dp.Dp.comm / comm_audio):
        # operations on one communicator would have to reach the device in the same order on every rank, which two streams do not promise
        args.pipeline = True
    pipe_obj = []

    def get_pipe():
        if not pipe_obj:
            pipe_obj.append(FastPitchHifiGan(fp, hg, dev))
        return pipe_obj[0]

    def make_tick(pipe):
        """event recorder on the stream a step's last launch goes to (the vocoder stream of the pipeline, else the caller's)"""
        def tick():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(pipe.s_hg if pipe is not None else torch.cuda.current_stream(dev))
            return ev
        return tick

    def make_step(ids_, dur_, pipelined=None):
        pipe = get_pipe() if (args.pipeline if pipelined is None else pipelined) else None
        step = _make_step(ids_, dur_, pipe)
        step.tick = make_tick(pipe)
        return step

    def _make_step(ids_, dur_, pipe):
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
            state['all'] = dpx.exchange_lens(dec_lens, max(b_cap, ids_.shape[0]))
            return state['all'][rank, 1:1 + ids_.shape[0]]

        def vocode(mel, dec_lens):
            wave = hg.forward(mel, dec_lens)
            all_samples = state['all'].copy()
            all_samples[:, 1:] *= hop
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dpx.gather_flat(wave, dec_lens * hop, all_lens=all_samples)     # C2: audio fan-in to rank 0
            e1.record()
            xch_ev.append((e0, e1))
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
        # first DP step under guard.  A failure on one rank can leave the others inside an unmatched Send / Recv, which no
        # in-process fallback can recover: a rank that raises exits with code 3, the parent's watchdog sees it (or, when
        # the run was started by torch.distributed.run, the launcher does), ends every rank and -- in the self-launched
        # case -- restarts fresh ranks on the torch.distributed transport
        try:
            step()
            sync()
        except Exception as e:                                   # noqa: BLE001
            print(f'bench.py rank {rank}: ttsamd_dp_* step failed ({e})', file=sys.stderr)
            os._exit(3)
    for _ in range(args.warmup):
        step()
    sync()
    xch_ev.clear()

    lib = L.load()
    import ctypes
    lib.ttsamd_profile_enable(0 if os.environ.get("TTSAMD_BENCH_NO_EVENTS") else 1)
    elapsed, (wave, dec_lens), per_step = _time_steps(step, args.steps, sync, barrier, tick=step.tick)
    prof = (ctypes.c_double * 3)()
    L.check(lib.ttsamd_profile_read(prof), 'profile_read')
    lib.ttsamd_profile_enable(0)

    frames = int(dec_lens.sum().item())
    samples = frames * hop * args.steps
    xch_ms = sum(a.elapsed_time(b) for a, b in xch_ev) / max(1, args.steps) if xch_ev else 0.0
    tot = torch.tensor([elapsed, float(samples), xch_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        mx = tot[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot[1:2].clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        xm = tot[2:].clone()
        dist.all_reduce(xm, op=dist.ReduceOp.MAX)
        elapsed, samples, xch_ms = float(mx[0]), float(sm[0]), float(xm[0])

    from ttsamd.config import NET_CONFIG as NC, HIFIGAN_CONFIG as HC
    hg_fpf = hifigan_flops_per_frame(HC)
    dec_fpt, enc_fpt = fastpitch_conv_flops_per_pos(NC)
    hg_bpf = hifigan_octet_bytes_per_frame(HC)
    dec_bpt, enc_bpt = fastpitch_conv_bytes_per_pos(NC)
    # peak of the MFMA instruction actually issued: fp32 157.3; bf16 2500 dense; split bf16 issues
    # 3 bf16 MFMAs per algorithmic product -> 2500/3 algorithmic TFLOP/s
    PEAKS = {'f32': PEAK_F32_MFMA_TFLOPS, 'bf16': 2500.0, 'bf16x3': 2500.0 / 3}
    peak = PEAKS[args.precision]

    def step_flops(b, fr):
        """algorithmic FLOPs (2*Cout*Cin*K per VALID output position) of one step's MFMA conv launches: HiFi-GAN and the
        FastPitch decoder on sum(frames) (+ one position per utterance for the ragged tile edges), encoder / predictors on
        B * L tokens (padded positions past an utterance's end are skipped by the kernels and are not counted)"""
        return hg_fpf * fr + dec_fpt * (fr + b) + enc_fpt * b * Lt

    def step_bytes_bf16(b, fr):
        """algorithmic HBM bytes of the same launches in the bf16 mode: HiFi-GAN on the octet engine (bf16 activations, fused
        pairs = one read + one write per pair), FastPitch on the bf16 MFMA engine with fp32 activations"""
        return hg_bpf * fr + dec_bpt * (fr + b) + enc_bpt * b * Lt

    def wall_roofline(prec, flops, byts, sec):
        """roofline entry over the WALL time of a whole call (launch gaps, small kernels and stream overlap included)"""
        ach = flops / sec / 1e12
        r = {'bound': 'mfma', 'achieved': ach, 'peak': PEAKS[prec], 'unit': 'TFLOP/s', 'frac': ach / PEAKS[prec],
             'basis': 'algorithmic conv FLOPs / wall time of the whole call (no per-launch events)'}
        if prec == 'bf16':
            r = bf16_roofline(flops, byts, sec, {'basis': 'algorithmic conv FLOPs (and HBM bytes) / wall time of the whole call '
                                                          '(no per-launch events)'})
        return r

    def small_config(b, prec=None, name=None, pipelined=False, inputs=None, predicted=False):
        """Batch-b sub-result (north star: batch 1 / 8 / 32): same step function, own warm-up, timed WITHOUT the
        per-launch events (they cost a B=1 call 17 %); its roofline figure is algorithmic work over the call's
        WALL time — launch gaps and the three-stream overlap included — so it can never exceed what ran."""
        prec = prec or args.precision
        set_precision(prec)
        try:
            ids_src, dur_src = inputs if inputs is not None else (ids, dur)
            ids_b, dur_b = ids_src[:b].contiguous(), (None if predicted else dur_src[:b].contiguous())
            st = make_step(ids_b, dur_b, pipelined)
            for _ in range(5):
                st()
            sync()
            n = max(args.steps, 20)
            el, (_, dl), ps = _time_steps(st, n, sync, tick=st.tick)
        finally:
            set_precision(args.precision)
        fr = int(dl.sum().item())
        out_c = {'batch': b, 'ms_per_step': el / n * 1e3, 'ms_per_step_median': _median(ps), 'value': fr * hop * n / el, 'unit': 'audio samples/s',
                 'rtf': el / (fr * hop * n / SAMPLE_RATE), 'frames': fr, 'steps': n, 'dtype': prec,
                 'roofline': wall_roofline(prec, step_flops(b, fr), step_bytes_bf16(b, fr), el / n)}
        out_c['schedule'] = ('two HIP streams: FastPitch of step i+1 under HiFi-GAN of step i (ttsamd.pipeline)' if pipelined
                             else 'one call after the other on one stream')
        if name:
            out_c['config'] = name
        if predicted:
            out_c['durations'] = ('PREDICTED by the duration predictor (clamp(exp(log_dur) - 1, 0, 75), model.py:366-368) on the calibrated synthetic '
                                  'weights of ttsamd.synth (~7 frames per token); every other entry forces dur_tgt so that the work is fixed')
        return out_c

    if rank == 0:
        conv_ms, n_launch, n_sections = prof[0], prof[1], prof[2]
        flops = args.steps * step_flops(B, frames)
        achieved = flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # HBM bytes per conv launch: OFFLINE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (profiles/rN/traffic.json; gpurun forbids mixing PMC with the timed run), newest round first
        traffic, traffic_src, traffic_alg = None, None, None
        if B == 32 and Lt == 64 and world == 1:
            for rnd in ('r6', 'r5', 'r4', 'r3', 'r2', 'r1'):
                try:
                    with open(os.path.join(REPO, 'profiles', rnd, 'traffic.json' if args.precision == 'f32' else f'traffic_{args.precision}.json')) as f:
                        tj = json.load(f)
                    traffic = tj['bytes_per_conv_launch_corrected']
                    traffic_alg = tj.get('algorithmic_bytes_per_conv_launch')
                    traffic_src = f'offline PMC pass, profiles/{rnd}/' + ('traffic.json' if args.precision == 'f32' else f'traffic_{args.precision}.json')
                    break
                except (OSError, KeyError):
                    pass
        prec_name = {'f32': 'fp32', 'bf16': 'bf16 MFMA', 'bf16x3': 'split-bf16 MFMA'}[args.precision]
        par = f'dp{world}' + (' (ranks share ONE device, gloo + host staging: functional check, not a scaling number)' if one_dev else '')
        time_basis = ('HIP events on the launch stream: one pair per launch, one pair per fork..join section of the three-stream ResBlock '
                      'schedule (wall time of the section); kernel time = length of the UNION of these intervals (the two pipeline '
                      'streams overlap)')
        if args.precision == 'bf16':
            byts = args.steps * step_bytes_bf16(B, frames)
            roof = bf16_roofline(flops, byts, max(conv_ms, 1e-9) * 1e-3, {
                'kernel': 'bf16 octet engine (bfo_resblock_pair / _chain + bfo_conv1d + bfo_convt; HiFi-GAN, FastPitch FFT blocks and predictors) + conv1d_mfma_bf16 for the remaining FastPitch convs',
                'kernel_time_basis': time_basis,
                'mfma_sustained_note': 'register-only v_mfma_f32_32x32x16_bf16 loop on random data, power-managed clock 1.72-1.78 GHz '
                                       '(tools/mfma_peak_bench.hip, profiles/r3/mfma_sustained_peak.txt); the 2.5 PFLOP/s peak needs 2.4 GHz',
                'algorithmic_bytes_per_step': byts / args.steps})
        else:
            roof = {'bound': 'mfma',
                    'kernel': ('MFMA conv engine: conv1d_wino4_f32 (Winograd F(4,3) decomposition, k = 3 / 7 / 11, C >= 64) + resblock_pair2 (fused C = 32 / 64 pairs, both phases on F(2,3)) + conv1d_mfma_f32 + convt_mfma_f32' if args.precision == 'f32' else 'split-bf16 octet engine: bfo3_resblock_pair + bfo3_conv1d + bfo3_convt (HiFi-GAN, FastPitch FFT blocks and predictors)') + ' (all instantiations)',
                    'kernel_time_basis': time_basis, 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak}
            if args.precision == 'f32':
                # products the Winograd launches do NOT issue (conv_wino2.hip: k = 3 / 7 / 11 as F(2,3) sub-filters + single taps: 4/6,
                # 10/14, 16/22 of the direct conv's): every ResBlock conv of HiFi-GAN's C >= 128 stages, FastPitch's decoder conv-FF
                saved = args.steps * (hifigan_wino_saved_flops_per_frame(HC) * frames +
                                      NC['out_fft_n_layers'] * 2.0 * NC['symbols_embedding_dim'] * NC['out_fft_conv1d_filter_size'] *
                                      NC['out_fft_conv1d_kernel_size'] * 2 * (1.0 - wino_product_ratio(NC['symbols_embedding_dim'], NC['out_fft_conv1d_kernel_size'], 1, fused=False)) * (frames + B)) if wino_on() else 0.0
                issued = (flops - saved) / (max(conv_ms, 1e-9) * 1e-3) / 1e12
                roof['issued'] = issued
                roof['frac_issued'] = issued / peak
                roof['flops_basis'] = ('`achieved` / `frac`: UN-REDUCED algorithmic FLOPs (2 Cout Cin K per valid output position) over kernel time.  '
                                       '`issued` / `frac_issued`: the same minus the products the Winograd launches do not issue, from the IDEAL product '
                                       'counts of the routed kernels -- F(4,3) decomposition (conv_wino4.hip: 6/12, 16/28, 23/44 of the direct products at k = 3 / 7 / 11; '
                                       'every un-fused ResBlock conv of HiFi-GAN (C >= 64) and FastPitch\'s decoder conv-FF), F(2,3) (4/6, 10/14, 16/22: both phases of '
                                       'the fused pairs -- C = 32 every k, C = 64 k = 3).  Idle tuple slots at dilation 3 / 5, the fused pairs\' halo recompute and small launches '
                                       'routed back to the direct kernel are NOT counted: a LOWER bound on what the MFMA pipe executed.  '
                                       'TTSAMD_WINO=0 runs the direct kernels, TTSAMD_WINO4=0 the round-5 F(2,3) routing')
        roof.update({'traffic': traffic, 'traffic_unit': 'B/launch', 'traffic_source': traffic_src,
                     'traffic_algorithmic': traffic_alg, 'traffic_ratio': (traffic / traffic_alg) if (traffic and traffic_alg) else None,
                     'launches': int(n_launch), 'sections': int(n_sections),
                     'avg_section_ms': conv_ms / max(1.0, n_sections), 'avg_launch_ms': conv_ms / max(1.0, n_launch),
                     'launch_note': 'a section = one event pair = a single launch or a fork..join group of up to 18 concurrent '
                                    'ResBlock launches on three streams; avg_launch_ms = kernel time / launches (overlapped launches share time)',
                     'kernel_ms_per_step': conv_ms / args.steps})
        out = {
            'metric': 'audio samples/sec (FastPitch+HiFi-GAN, synthetic 64-phoneme inputs)',
            'value': samples / elapsed, 'unit': 'audio samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'ms_per_step_median': _median(per_step),
            'ms_per_step_note': 'ms_per_step (and value) = wall time of the K timed steps / K, as the contract prescribes; the median is over '
                                'the K per-step intervals between HIP events recorded after each step\'s last launch (rank 0).  On the two-stream '
                                'schedule the host runs the acoustic model many steps ahead, so the first intervals absorb later steps\' FastPitch '
                                'and the last ones are vocoder-only: there the MEAN is the rate, the median is not',
            'higher_is_better': True, 'scaling': args.scaling if world > 1 else 'weak',
            'vs_baseline': None, 'dtype': {'f32': 'f32', 'bf16': 'bf16', 'bf16x3': 'f32 via split-bf16 (3x bf16 MFMA, fp32 accumulate)'}[args.precision], 'data': 'synthetic (ids, forced durations, random-init weights)',
            'rtf': elapsed / (samples / SAMPLE_RATE),
            'config': {'workload': f'FastPitch+HiFi-GAN, synthetic {Lt}-phoneme x batch{B} per GPU, {prec_name}, '
                                   f'{world}xMI355X', 'batch_per_gpu': B, 'global_batch': g_batch, 'n_tokens': Lt,
                       'frames_per_step_rank0': frames, 'parallelism': par, 'dp_transport': transport,
                       'dp_fallback_reason': os.environ.get('TTSAMD_BENCH_FALLBACK_REASON'),
                       'schedule': ('two HIP streams: FastPitch of step i+1 under HiFi-GAN of step i (ttsamd.pipeline)' if args.pipeline else
                                    'one stream per step')},
            'roofline': roof,
        }
        if world > 1:
            out['exchange_ms_per_step'] = xch_ms          # C2 audio fan-in (pack + transfer), max over ranks; the length exchange
            out['exchange_note'] = ('HIP events around the packed audio fan-in on each rank, max over ranks; the all-gather of the '
                                    'lengths rides in the step\'s one host synchronisation')
        t_sec = time.perf_counter()
        sections = {}

        def lap(name):
            nonlocal t_sec
            now = time.perf_counter()
            sections[name] = round(now - t_sec, 1)
            t_sec = now

        if world == 1 and not args.no_small and B > 8:
            out['configs'] = [small_config(1), small_config(8)]
            lap('batch 1 / 8')
            if args.precision == 'f32' and args.pipeline and B == 32:
                out['configs'].append(small_config(B, pipelined=False, name='C2 (this line\'s workload) on the ONE-stream schedule: every '
                                                                           'launch of a step on the caller\'s stream, same work and results'))
            if args.precision == 'f32' and B == 32:
                # SURVEY §8(d) primary synthetic run: the same workload on PREDICTED durations (no dur_tgt), both schedules
                c2p = small_config(B, pipelined=bool(args.pipeline), predicted=True,
                                   name='C2 on PREDICTED durations (dur_tgt=None: duration predictor -> regulate_len), calibrated synthetic weights')
                c2p['ms_per_step_one_stream'] = small_config(B, pipelined=False, predicted=True)['ms_per_step']
                out['configs'].append(c2p)
                lap('C2 predicted durations')
            if not args.no_extra:
                out['configs'] += extra_configs(args, dev, fp, hg, ids, dur, hop, small_config, wall_roofline, sync, lap)
        if world == 1 and not args.no_extra:
            out['d2h'] = d2h_probe(wave, dec_lens, hop)
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(fp_sd, hg_sd, Lt)
            lap('cpu_baseline')
            try:
                out['cpu_baseline']['c1'] = cpu_baseline_c1(fp_sd, hg_sd, out['cpu_baseline']['cores'])
            except Exception as e:                               # noqa: BLE001
                out['cpu_baseline']['c1'] = {'error': str(e)[:300]}
            lap('cpu_baseline.c1')
            try:
                out['cpu_baseline']['all_cores'] = cpu_baseline_all_cores(Lt)
            except Exception as e:                               # noqa: BLE001
                out['cpu_baseline']['all_cores'] = {'error': str(e)[:300]}
            lap('cpu_baseline.all_cores')
        elif world > 1:
            out['cpu_baseline'] = None
        if sections:
            out['bench_sections_s'] = sections        # wall seconds of the sub-results behind this line (the timed K steps are not in it)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dpx.close()
        dist.destroy_process_group()


def d2h_probe(wave, dec_lens, hop):
    """the GPU number excludes the device -> host copy of the audio; reported here.  One copy of the padded
    [B, n_max] wave into a pinned host buffer (what FastPitch2Wave.tts_batch does once per batch; the reference copies per
    utterance, models/fastpitch/networks.py:345), median of 5."""
    host = torch.empty(wave.shape, dtype=wave.dtype, pin_memory=True)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host.copy_(wave, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    nbytes = wave.numel() * 4
    return {'d2h_ms': ts[len(ts) // 2] * 1e3, 'bytes': nbytes, 'valid_bytes': int(dec_lens.sum().item()) * hop * 4,
            'gb_per_s': nbytes / ts[len(ts) // 2] / 1e9, 'what': 'padded [B, n_max] fp32 wave -> pinned host buffer, one copy per batch'}


def extra_configs(args, dev, fp, hg, ids, dur, hop, small_config, wall_roofline, sync, lap=lambda name: None):
    """The other BASELINE configs as driver-visible sub-results (each: ms_per_step, audio samples/s, a wall-time roofline):
    C3  the per-GPU share (B = 256 / 8 = 32) of the bf16 configuration;
    C4  Tacotron2 autoregressive decode + HiFi-GAN, batch 8, 448 forced decoder steps (the gate is biased shut so the work is
        fixed; prenet dropout on);
    C5  FastPitch 4-speaker + MelVocos('22k'), batch 32, one speaker id per call cycling 0..3."""
    from ttsamd import engine as E, synth
    from ttsamd.config import NET_CONFIG, TACOTRON2_CONFIG, HIFIGAN_CONFIG
    res = []
    B = ids.shape[0]
    # the throughput configuration runs the two-stream schedule (FastPitch of batch i+1 under HiFi-GAN of batch i: FastPitch is 18 % of
    # a bf16 step and mostly launch- and latency-bound); the one-stream time of the same work rides along.  Batch 8 / 1 are latency
    # numbers: one call after the other on one stream.
    c3 = small_config(B, prec='bf16', pipelined=True,
                      name=f'C3 per-GPU share: FastPitch+HiFi-GAN, synthetic 64-phoneme x batch{B}, bf16 MFMA '
                           '(HiFi-GAN and the FastPitch FFT blocks on the bf16 octet engine)')
    c3['ms_per_step_one_stream'] = small_config(B, prec='bf16')['ms_per_step']
    res.append(c3)
    for b_small in (8, 1):                          # north star: batch 1 / 8 / 32 -- the bf16 configuration at the small batches too
        res.append(small_config(b_small, prec='bf16', name=f'C3 at batch {b_small}: FastPitch+HiFi-GAN, synthetic 64-phoneme, bf16 MFMA'))
    # C3 at its REAL size on one GPU: all 256 utterances of the 8-GPU configuration = the N = 1 point of its strong-scaling curve
    # (`bench.py --gpus N --batch 256 --scaling strong --precision bf16` gives the other points)
    try:
        b_full = 256
        ids256 = torch.from_numpy(synth.synth_ids(b_full, ids.shape[1])).to(dev)
        dur256 = torch.from_numpy(synth.synth_durations(b_full, ids.shape[1])).to(dev)
        c3f = small_config(b_full, prec='bf16', pipelined=True, inputs=(ids256, dur256),
                           name='C3 at its full size on ONE GPU (N = 1 point of the strong-scaling curve): FastPitch+HiFi-GAN, synthetic '
                                '64-phoneme x batch256, bf16 MFMA')
        c3f['ms_per_step_one_stream'] = small_config(b_full, prec='bf16', inputs=(ids256, dur256))['ms_per_step']
        res.append(c3f)
        del ids256, dur256
    except Exception as e:                                       # noqa: BLE001
        res.append({'config': 'C3 at batch 256 on one GPU', 'error': str(e)[:300]})
    # C3 INSIDE north_star's tolerance (mel 1e-3 / wave 1e-4): the same engine in its split-bf16 mode -- every operand hi + lo, three
    # v_mfma_f32_32x32x16_bf16 per product, x3 tensors (4 bytes per element) in HBM; tests: test_full_batch_every_utterance[bf16x3],
    # test_config3_full_size_256_utterances_bf16x3.  `peak` of its roofline entry = 2.5 PFLOP/s / 3 algorithmic.
    try:
        c3x = small_config(B, prec='bf16x3', pipelined=True,
                           name=f'C3 per-GPU share inside the 1e-3 / 1e-4 tolerance: FastPitch+HiFi-GAN, synthetic 64-phoneme x batch{B}, '
                                'split-bf16 MFMA on the octet engine (bfo3_*)')
        c3x['ms_per_step_one_stream'] = small_config(B, prec='bf16x3')['ms_per_step']
        res.append(c3x)
        for b_small in (8, 1):                      # north star: batch 1 / 8 / 32 -- the in-tolerance mode at the small batches too (latency, one stream)
            res.append(small_config(b_small, prec='bf16x3', name=f'C3 at batch {b_small} inside the tolerance: FastPitch+HiFi-GAN, synthetic 64-phoneme, split-bf16 MFMA'
                                                                 + (' (FastPitch of a batch <= 2 call on the fp32 kernels)' if b_small <= 2 else '')))
        b_full = 256
        ids256 = torch.from_numpy(synth.synth_ids(b_full, ids.shape[1])).to(dev)
        dur256 = torch.from_numpy(synth.synth_durations(b_full, ids.shape[1])).to(dev)
        res.append(small_config(b_full, prec='bf16x3', pipelined=True, inputs=(ids256, dur256),
                                name='C3 at its full size on ONE GPU inside the 1e-3 / 1e-4 tolerance: synthetic 64-phoneme x batch256, split-bf16 MFMA'))
        del ids256, dur256
    except Exception as e:                                       # noqa: BLE001
        res.append({'config': 'C3 split-bf16', 'error': str(e)[:300]})
    torch.cuda.empty_cache()
    lap('C3 (bf16 32 / 8 / 1 / 256, bf16x3 32 / 8 / 1 / 256)')
    n = max(args.steps, 10)
    hgf = hifigan_flops_per_frame(HIFIGAN_CONFIG)
    # ---- C4
    try:
        frames_t, bt = 448, 8
        taco = E.Tacotron2Engine(synth.tacotron2_state_dict(TACOTRON2_CONFIG, seed=0, gate_bias=-30.0), TACOTRON2_CONFIG, device=dev)
        tids = torch.from_numpy(synth.synth_ids(bt, 64)).to(dev)
        tlens = torch.full((bt,), 64, dtype=torch.int64, device=dev)
        sids = torch.zeros(bt, dtype=torch.int64, device=dev)

        def c4(seed):
            mel, mel_lens, _ = taco.infer(tids, sids, tlens, max_step=frames_t, dropout_seed=seed)
            return hg.forward(mel.contiguous(), mel_lens.to(torch.int64)), mel_lens
        for i in range(2):
            c4(i)
        sync()
        t0 = time.perf_counter()
        for i in range(n):
            _, ml = c4(100 + i)
        sync()
        el = (time.perf_counter() - t0) / n
        t1 = time.perf_counter()
        for i in range(n):
            taco.infer(tids, sids, tlens, max_step=frames_t, dropout_seed=200 + i)
        sync()
        el_t = (time.perf_counter() - t1) / n
        fr = int(ml.sum().item())
        # decoder GEMVs per step and utterance: two LSTM cells (4 x 1024 rows over 1792 + 2560 inputs... counted from the config)
        c = TACOTRON2_CONFIG
        mem = c['encoder_embedding_dim'] + (c['speaker_embedding_dim'] if c.get('num_speakers', 1) > 1 else 0)
        lstm = 2.0 * 4 * c['attention_rnn_dim'] * (c['prenet_dim'] + mem + c['attention_rnn_dim']) + \
            2.0 * 4 * c['decoder_rnn_dim'] * (c['attention_rnn_dim'] + mem + c['decoder_rnn_dim'])
        flops = hgf * fr + lstm * fr
        res.append({'config': 'C4 Tacotron2 (autoregressive LSTM-attention decoder, persistent kernel) + HiFi-GAN, batch 8 x 64 tokens, '
                              '448 forced decoder steps, fp32', 'batch': bt, 'ms_per_step': el * 1e3, 'value': fr * hop / el,
                    'unit': 'audio samples/s', 'rtf': el / (fr * hop / SAMPLE_RATE), 'frames': fr, 'steps': n, 'dtype': 'f32',
                    'ms_tacotron2': el_t * 1e3, 'us_per_decoder_step': el_t / frames_t * 1e6,
                    'parity': 'unpinned (torchaudio Tacotron2 is not in the reference tree; SURVEY §8c)',
                    # two parts with two different bounds: the decoder is a 448-step recurrence whose step is ONE dependent chain (six
                    # cross-XCD hand-offs, the cells' serial MFMAs, operand round trips through L2), the vocoder is the MFMA conv engine
                    'roofline': dict(wall_roofline('f32', hgf * fr, 0.0, max(el - el_t, 1e-9)),
                                     what='HiFi-GAN part only: conv FLOPs / (whole call - Tacotron2 alone)'),
                    'decoder': {'bound': 'latency (one dependent chain per step inside one persistent launch)',
                                'us_per_step': el_t / frames_t * 1e6, 'handoffs_per_step': TACO_HANDOFFS_PER_STEP,
                                'handoff_us': TACO_HANDOFF_US, 'mfma_us_per_step': TACO_MFMA_US_PER_STEP,
                                'floor_us_per_step': TACO_HANDOFFS_PER_STEP * TACO_HANDOFF_US + TACO_MFMA_US_PER_STEP,
                                'frac_of_floor': (TACO_HANDOFFS_PER_STEP * TACO_HANDOFF_US + TACO_MFMA_US_PER_STEP) / (el_t / frames_t * 1e6),
                                'lstm_tflops': lstm * fr / el_t / 1e12,
                                'note': 'ms_tacotron2 also holds the encoder, the postnet and the host stop test (once per call)'}})
        del taco
    except Exception as e:                                       # noqa: BLE001  (a sub-result must not take the headline line down)
        res.append({'config': 'C4 Tacotron2 + HiFi-GAN', 'error': str(e)[:300]})
    lap('C4')
    # ---- C5
    try:
        cfg4 = dict(NET_CONFIG, n_speakers=4, speaker_emb_weight=1.0)
        fp4 = E.FastPitchEngine(synth.fastpitch_state_dict(cfg4), cfg4, device=dev)
        voc = E.VocosEngine(synth.vocos_state_dict(), device=dev)
        spk = [0]

        def c5():
            spk[0] = (spk[0] + 1) % 4
            mel, dl, *_ = fp4.infer(ids, dur_tgt=dur, speaker=spk[0])
            return voc.forward(mel, dl), dl
        from ttsamd.pipeline import FastPitchHifiGan
        pipe5 = FastPitchHifiGan(fp4, voc, dev)                  # the same two-stream schedule with Vocos as the second stage

        def c5p():
            spk[0] = (spk[0] + 1) % 4
            _, dl, wave = pipe5.submit(ids, vocode=voc.forward, dur_tgt=dur, speaker=spk[0])
            return wave, dl

        def timed(f):
            for _ in range(3):
                f()
            sync()
            t0 = time.perf_counter()
            for _ in range(n):
                _, dl_ = f()
            sync()
            return (time.perf_counter() - t0) / n, dl_
        el1, dl = timed(c5)
        el, dl = timed(c5p)
        fr = int(dl.sum().item())
        dec_fpt, enc_fpt = fastpitch_conv_flops_per_pos(cfg4)
        flops = VOCOS_FLOPS_PER_FRAME * fr + dec_fpt * (fr + B) + enc_fpt * B * ids.shape[1]
        res.append({'config': f'C5 FastPitch 4-speaker + MelVocos(22k) (ISTFT head), batch {B}, fp32', 'batch': B, 'ms_per_step': el * 1e3,
                    'ms_per_step_one_stream': el1 * 1e3,
                    'schedule': 'two HIP streams: FastPitch of step i+1 under Vocos of step i (ttsamd.pipeline)',
                    'value': fr * hop / el, 'unit': 'audio samples/s', 'rtf': el / (fr * hop / SAMPLE_RATE), 'frames': fr, 'steps': n,
                    'dtype': 'f32', 'roofline': wall_roofline('f32', flops, 0.0, el)})
    except Exception as e:                                       # noqa: BLE001
        res.append({'config': 'C5 FastPitch 4-speaker + MelVocos', 'error': str(e)[:300]})
    # ---- C2 with the denoiser on (SURVEY §8d secondary: denoise = 0.005, the default of FastPitch2Wave.tts)
    try:
        from vocoder.hifigan.denoiser import Denoiser

        class _Voc:                                              # what Denoiser needs of a vocoder: a device and a call
            device = dev

            def __call__(self, mel):
                return hg.forward(mel)

            def to(self, d):
                return self
        den = Denoiser(_Voc())

        def c2d():
            mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
            wave = hg.forward(mel, dl)
            return den.forward_batch(wave, dl * hop, 0.005, nsamples_min=513), dl
        from ttsamd.pipeline import FastPitchHifiGan
        pipe_d = FastPitchHifiGan(fp, hg, dev)

        def c2dp():                                              # the headline's two-stream schedule with the denoiser behind the vocoder
            _, dl, wave = pipe_d.submit(ids, vocode=lambda mel, dl_: den.forward_batch(hg.forward(mel, dl_), dl_ * hop, 0.005, nsamples_min=513),
                                        dur_tgt=dur)
            return wave, dl

        def c2p():                                               # ... and without it, timed right beside (same box, same minute)
            _, dl, wave = pipe_d.submit(ids, dur_tgt=dur)
            return wave, dl

        def timed_d(f):
            for _ in range(3):
                f()
            sync()
            t0 = time.perf_counter()
            for _ in range(n):
                _, dl_ = f()
            sync()
            return (time.perf_counter() - t0) / n, dl_
        el1, dl = timed_d(c2d)
        el, dl = timed_d(c2dp)
        el_off, _ = timed_d(c2p)
        fr = int(dl.sum().item())
        res.append({'config': f'C2 with denoise=0.005 (Denoiser: STFT -> spectral subtraction -> ISTFT after the vocoder), batch {B}, fp32',
                    'batch': B, 'ms_per_step': el * 1e3, 'ms_per_step_one_stream': el1 * 1e3, 'ms_per_step_denoise_off': el_off * 1e3,
                    'denoise_cost_ms': (el - el_off) * 1e3,
                    'schedule': 'two HIP streams as the headline (FastPitch of step i+1 under HiFi-GAN + denoiser of step i); '
                                'ms_per_step_denoise_off = the same loop without the denoiser, timed right beside it',
                    'value': fr * hop / el, 'unit': 'audio samples/s',
                    'rtf': el / (fr * hop / SAMPLE_RATE), 'frames': fr, 'steps': n, 'dtype': 'f32',
                    'parity': 'denoiser unpinned at the torchaudio boundary (golden made with a torch.stft stand-in; SURVEY §8c)'})
    except Exception as e:                                       # noqa: BLE001
        res.append({'config': 'C2 with denoise=0.005', 'error': str(e)[:300]})
    lap('C5 + C2 denoise')
    # ---- C1: the reference's own case (inference.py:55-58) -- the 100 lines of data/infer_text.txt through FastPitch2Wave.tts
    try:
        res += c1_configs(dev)
    except Exception as e:                                       # noqa: BLE001
        res.append({'config': 'C1 FastPitch2Wave.tts on the 100 infer_text lines', 'error': str(e)[:300]})
    return res


TACO_HANDOFFS_PER_STEP = 6           # csrc/tacotron2.hip: dependent cross-CU hand-offs of one decoder step in the dataflow schedule
TACO_HANDOFF_US = 0.5                # one word across XCDs, sc1 store -> sc1 poll (tools/handoff_bench.hip, profiles/r4/handoff_bench.txt)
TACO_MFMA_US_PER_STEP = 5.8          # 108 super-steps x 4 v_mfma_f32_16x16x4_f32 (32 cycles each) per wave and step at 2.4 GHz: the serial matrix work


def c1_configs(dev):
    """BASELINE config 1 on the GPU: the 100 committed lines of the reference's data/infer_text.txt (tests/golden/infer_text_lines.json,
    35-268 tokens each) through the drop-in `FastPitch2Wave.tts(list, batch_size=...)` -- Arabic text in, CPU waves out, i.e.
    tokenisation, FastPitch with PREDICTED durations (synthetic weights: ~7 frames per token), HiFi-GAN, denoiser as asked and the
    device -> host copies are all inside the timed call (the reference's plumbing, models/fastpitch/networks.py:352-435)."""
    import tempfile
    import text
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    from models.fastpitch import FastPitch2Wave
    with open(os.path.join(REPO, 'tests', 'golden', 'infer_text_lines.json'), encoding='utf-8') as f:
        lines = json.load(f)
    res = []
    with tempfile.TemporaryDirectory() as d:
        fp_sd = {k: torch.from_numpy(v.copy()) for k, v in synth.fastpitch_state_dict().items()}
        torch.save({'model': fp_sd, 'config': dict(NET_CONFIG), 'symbols': list(text.symbols)}, os.path.join(d, 'fp.pth'))
        torch.save({'generator': {k: torch.from_numpy(v.copy()) for k, v in synth.hifigan_state_dict().items()}}, os.path.join(d, 'hg.pth'))
        with open(os.path.join(d, 'config.json'), 'w') as f:
            json.dump(HIFIGAN_CONFIG, f)
        model = FastPitch2Wave(os.path.join(d, 'fp.pth'), vocoder_sd=os.path.join(d, 'hg.pth'),
                               vocoder_config=os.path.join(d, 'config.json')).to(dev)
    for bs, denoise in ((1, 0.0), (32, 0.0), (1, 0.005)):
        model.tts(lines[:4], batch_size=bs, denoise=denoise)                      # warm-up (engines, workspaces)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            waves = model.tts(lines, batch_size=bs, denoise=denoise)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        el = sorted(ts)[1]
        ns = int(sum(w.numel() for w in waves))
        res.append({'config': f'C1 FastPitch2Wave.tts(100 lines of infer_text.txt, batch_size={bs}, denoise={denoise}), text in -> CPU waves out, fp32',
                    'batch': bs, 'ms_total': el * 1e3, 'ms_per_utterance': el * 1e3 / len(lines), 'value': ns / el, 'unit': 'audio samples/s',
                    'rtf': el / (ns / SAMPLE_RATE), 'samples': ns, 'utterances': len(lines), 'dtype': 'f32',
                    'timing': 'median of 3 whole calls, host wall time incl. tokenisation and the device -> host copies',
                    'schedule': 'three HIP streams over the chunks of the list (FastPitch2Wave._tts_list_pipelined): tokenise + FastPitch of the next '
                                'chunks (each chunk = the reference\'s padded batch of `batch_size` lines) under vocoder + denoiser of the previous '
                                'ones, D2H on a third stream; the batch-independent vocoder takes the mels of up to 16 utterances per ragged call; '
                                'batch_size 1: the lines sorted by length through FastPitch AND the vocoder in balanced ragged groups of <= 32 whose '
                                'FastPitch rows are computed as if alone (ttsamd_fastpitch_set_batch_mode 1), waves back in input order'})
    return res


VOCOS_FLOPS_PER_FRAME = 26.85e6      # MelVocos('22k'), convs and linears per mel frame (embed 0.57 + 8 x (dwconv 0.007 + pwconv 3.146) + head 1.05 MFLOP; the ISTFT is not in it)


def hifigan_octet_bytes_per_frame(h):
    """Algorithmic HBM bytes per mel frame of the HiFi-GAN conv launches on the bf16 octet engine: every
    tensor crosses HBM as bf16; a fused c1 -> c2 pair (C <= 128) reads its input once and writes once, a whole k = 3 ResBlock
    (three pairs) likewise; at C = 256 c1 and c2 are two launches (read + write, read + residual + write); the two accumulating last pairs of a stage re-read the sum."""
    c0 = h['upsample_initial_channel']
    by = 2.0 * (h['num_mels'] + c0)
    ch, mul = c0, 1
    nk = len(h['resblock_kernel_sizes'])
    for u in h['upsample_rates']:
        by += 2.0 * (ch * mul + (ch // 2) * mul * u)
        ch, mul = ch // 2, mul * u
        per_pair = 2.0 * (2 if ch <= 128 else 5) * ch * mul
        for k, dil in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
            # a k = 3 ResBlock at C <= 128 is ONE launch (bfo_chain.hip): one read + one write for its three pairs
            by += per_pair if (k == 3 and ch <= 128 and len(dil) == 3) else len(dil) * per_pair
        by += 2.0 * (nk - 1) * ch * mul
    return by


def fastpitch_conv_bytes_per_pos(c):
    """(decoder, encoder + predictors) algorithmic HBM bytes per position of FastPitch's conv launches in the bf16 mode:
    4 * (Cin + Cout * (1 + residual)) per conv with fp32 activations, 2 bytes per element where a tensor is bf16."""
    d = c['symbols_embedding_dim']

    def layer(dh, nh, filt):
        # qkv and o_net on the bf16 MFMA engine with fp32 activations; the conv-FF pair on the octet engine: bf16 input copy,
        # bf16 1536-channel intermediate, fp32 residual in and fp32 stream out
        return 4.0 * ((d + 3 * nh * dh) + (nh * dh + 2 * d)) + 2.0 * (d + filt) + (2.0 * filt + 4.0 * 2 * d)
    dec = c['out_fft_n_layers'] * layer(c['out_fft_d_head'], c['out_fft_n_heads'], c['out_fft_conv1d_filter_size']) + \
        4.0 * (d + c['n_mel_channels'])
    enc = c['in_fft_n_layers'] * layer(c['in_fft_d_head'], c['in_fft_n_heads'], c['in_fft_conv1d_filter_size'])
    for p in ('dur', 'pitch', 'energy'):
        if p == 'energy' and not c['energy_conditioning']:
            continue
        f, n = c[f'{p}_predictor_filter_size'], c[f'{p}_predictor_n_layers']
        enc += 4.0 * ((d + f) + (n - 1) * 2 * f)
    return dec, enc


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


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def wino_product_ratio(ch, kk, dil=1, fused=None):
    """Products the routed kernel of a ResBlock / conv-FF conv issues, as a fraction of the direct conv's 2 k per output pair -- from the
    routing masks the library reads (TTSAMD_WINO, TTSAMD_WINO4, TTSAMD_WINO2, TTSAMD_FUSED2_WB; csrc/conv_wino.hip: wino_route, csrc/hifigan.hip):
      F(4,3) decomposition (conv_wino4.hip): 6 / 16 / 23 products per output quad at k = 3 / 7 / 11  -> 0.500 / 0.571 / 0.523
      F(2,3) decomposition (conv_wino2.hip, both phases of the fused pairs): 4 / 10 / 16 per pair     -> 0.667 / 0.714 / 0.727
    IDEAL counts of the routed kernels: idle tuple slots at dilation 3 / 5 (120 of 128 columns), the halo the fused pairs recompute and
    launches that the >= 192-block rule sends back to the direct kernel (small batches) are not in it, so `issued` is a LOWER bound on what
    the matrix pipe executed and `frac_issued` a lower bound on its utilisation."""
    if kk not in (3, 7, 11) or os.environ.get('TTSAMD_WINO', '1') == '0':
        return 1.0
    kbit = {3: 1, 7: 2, 11: 4}[kk]
    f23 = (4 * (kk // 3) + 2 * (kk % 3)) / (2.0 * kk)
    f43 = {3: 6, 7: 16, 11: 23}[kk] / (4.0 * kk)
    if fused is None:
        fused = ch <= 32 or (ch == 64 and kk == 3)             # hifigan.hip: kFused2Mask 00f
    if fused:
        return f23 if os.environ.get('TTSAMD_FUSED2_WB', '1') != '0' else 1.0
    m4, m2 = _env_int('TTSAMD_WINO4', 31), _env_int('TTSAMD_WINO2', 31)
    if (m4 & kbit) and (dil == 1 or (m4 & 8)):
        return f43
    if (m2 & kbit) and (dil == 1 or (m2 & 8)):
        return f23
    return 1.0


def hifigan_wino_saved_flops_per_frame(h):
    """Products per mel frame that the Winograd kernels do not issue (wino_product_ratio per ResBlock conv): the stages with >= 128 channels on
    csrc/conv_wino4.hip (F(4,3): every un-fused conv, C >= 64), both phases of the fused pairs (C = 32 every k, C = 64 k = 3) on F(2,3)
    (csrc/resblock_fused2.hip)."""
    ch, mul, f = h['upsample_initial_channel'], 1, 0.0
    for u in h['upsample_rates']:
        ch, mul = ch // 2, mul * u
        for kk, dil in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
            if ch < 32:
                continue
            for d in dil:
                f += (2.0 * ch * ch * kk) * mul * ((1.0 - wino_product_ratio(ch, kk, d)) + (1.0 - wino_product_ratio(ch, kk, 1)))   # c1 (dilated) + c2
    return f


def wino_on():
    return os.environ.get('TTSAMD_WINO', '1') != '0'


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
