import torch
a = torch.load('gpurun_out/fp_new.pt'); b = torch.load('gpurun_out/fp_old.pt')
for k, (m0, m1, dl) in enumerate(((a[0], b[0], a[2]), (a[1], b[1], a[3]))):
    w = 0.0; nd = 0
    for r in range(32):
        t = int(dl[r]); d = (m0[r, :, :t] - m1[r, :, :t]).abs()
        w = max(w, float(d.max())); nd += int((d.amax(0) > 0).sum())
    print('run', k, 'max-abs diff on valid frames %.3e' % w, 'frames that differ', nd, 'of', int(dl.sum()))
