import hashlib, os, sys, torch
sys.path.insert(0, '/root/repo/tts-arabic-pytorch_amd')
from ttsamd import synth, engine as E
dev = torch.device('cuda:0')
for prec in ('f32', 'bf16x3'):
    E.set_precision(prec)
    fp = E.FastPitchEngine(synth.fastpitch_state_dict(), device=dev)
    ids = torch.from_numpy(synth.synth_ids(32, 64)).to(dev)
    dur = torch.from_numpy(synth.synth_durations(32, 64)).to(dev)
    mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
    h = hashlib.sha256()
    for b in range(32):
        h.update(mel[b, :, :int(dl[b])].contiguous().cpu().numpy().tobytes())
    mel2, dl2, *_ = fp.infer(ids)        # predicted durations
    for b in range(32):
        h.update(mel2[b, :, :int(dl2[b])].contiguous().cpu().numpy().tobytes())
    print(prec, os.environ.get('TTSAMD_LIB', 'default'), h.hexdigest()[:16], int(dl.sum()), int(dl2.sum()))
    if prec == 'f32' and len(sys.argv) > 1:
        torch.save([mel.cpu(), mel2.cpu(), dl.cpu(), dl2.cpu()], sys.argv[1])
