"""utils.audio: the wav writer read back with scipy, peak normalisation."""
import numpy as np
import torch


def test_save_wav_roundtrip(tmp_path):
    from scipy.io import wavfile
    from utils.audio import save_wav
    w = torch.sin(torch.arange(3000) * 0.05) * 0.7
    save_wav(tmp_path / 'a.wav', w)
    sr, data = wavfile.read(tmp_path / 'a.wav')
    assert sr == 22050 and data.dtype == np.int16 and data.shape == (3000,)
    assert np.max(np.abs(data / 32767.0 - w.numpy())) <= 0.5 / 32767 + 1e-7
    save_wav(tmp_path / 'b.wav', w[None], encoding='PCM_F')
    sr, data = wavfile.read(tmp_path / 'b.wav')
    assert data.dtype == np.float32 and np.array_equal(data, w.numpy())
    save_wav(tmp_path / 'c.wav', np.array([2.0, -2.0], np.float32))          # clipped, not wrapped
    assert wavfile.read(tmp_path / 'c.wav')[1].tolist() == [32767, -32768]
    save_wav(tmp_path / 'd.wav', torch.zeros(0))
    assert wavfile.read(tmp_path / 'd.wav')[1].shape == (0,)


def test_peak_normalise():
    from utils.audio import peak_normalise
    w = torch.tensor([0.1, -0.5, 0.25])
    assert torch.allclose(peak_normalise(w), torch.tensor([0.198, -0.99, 0.495]))
    assert torch.equal(peak_normalise(torch.zeros(4)), torch.zeros(4))
