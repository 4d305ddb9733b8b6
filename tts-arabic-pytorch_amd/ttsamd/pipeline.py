"""Two-stage pipeline over the FastPitch and HiFi-GAN engines: the acoustic model of batch i + 1 runs on its own HIP stream
UNDER the vocoder of batch i.

FastPitch at B = 32 is ~150 launches of 10-500 us, most of which do not fill the 256 CUs (8.5 ms of a 79 ms step at 0.75 of
the matrix peak for its convs, far less for the rest); HiFi-GAN is 70 ms of chip-filling launches.  Issued on one stream the
two follow each other; on two streams the dispatcher fills FastPitch's idle CUs with vocoder workgroups.  Nothing is shared
between the stages but the mel / length tensors of a batch (fresh tensors per call, handed over with an event), each engine
keeps its own workspace and sees its own calls in order, so results are bit-identical to the one-stream schedule.
Used by bench.py and by the drop-in `FastPitch2Wave.tts` for lists that span several batches."""
import torch


class FastPitchHifiGan:
    def __init__(self, fp, hg, device=None):
        self.fp, self.hg = fp, hg
        self.device = torch.device(device) if device is not None else fp.device
        self.s_fp = torch.cuda.Stream(self.device)
        # the vocoder is the stage that fills the chip: its stream (and the engine's two branch streams, csrc/hifigan.hip) get the higher
        # priority, the acoustic model of the next batch takes what is left (fp32 B = 32: 52.86 -> 52.59 ms per step on one box)
        self.s_hg = torch.cuda.Stream(self.device, priority=-1)

    def submit(self, ids, vocode=None, **infer_kw):
        """Queue one batch: FastPitch.infer(ids, **infer_kw) on the acoustic stream, then `vocode(mel, dec_lens)` (default:
        HiFi-GAN forward) on the vocoder stream.  Returns (mel, dec_lens, result of vocode); the tensors are valid on the vocoder
        stream: call `join()` (or `torch.cuda.synchronize`) before reading them from another stream or the host."""
        cur = torch.cuda.current_stream(self.device)
        self.s_fp.wait_stream(cur)                                 # the inputs were produced on the caller's stream
        with torch.cuda.stream(self.s_fp):
            mel, dec_lens, *_ = self.fp.infer(ids, **infer_kw)
        self.s_hg.wait_stream(self.s_fp)
        with torch.cuda.stream(self.s_hg):
            mel.record_stream(self.s_hg)                           # allocated on the acoustic stream, consumed here
            dec_lens.record_stream(self.s_hg)
            out = (self.hg.forward(mel, dec_lens) if vocode is None else vocode(mel, dec_lens))
        return mel, dec_lens, out

    def join(self):
        """Make the caller's stream wait for everything queued so far."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.s_hg)
        cur.wait_stream(self.s_fp)
