"""Output side of the path: wav files and the peak normalisation of the web manager.
The reference calls torchaudio.save(path, wave[None], 22050) (inference.py:61-63, utils/app_utils.py:76-77);
torchaudio is not a dependency here, so the RIFF container is written directly."""
import struct

import numpy as np


def save_wav(path, wave, sample_rate=22_050, encoding='PCM_S', bits_per_sample=16):
    """wave: 1-D (or [1, n]) float tensor/array in [-1, 1].  'PCM_S' 16-bit (torchaudio.save's default for
    .wav from float32 is 32-bit float: pass encoding='PCM_F') -> little-endian RIFF/WAVE, mono."""
    a = wave.detach().cpu().numpy() if hasattr(wave, 'detach') else np.asarray(wave)
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    if encoding == 'PCM_F':
        fmt, bits, data = 3, 32, a.astype('<f4').tobytes()
    elif encoding == 'PCM_S' and bits_per_sample == 16:
        fmt, bits = 1, 16
        data = np.clip(np.round(a * 32767.0), -32768, 32767).astype('<i2').tobytes()
    else:
        raise ValueError(f'unsupported wav encoding {encoding}/{bits_per_sample}')
    block = bits // 8
    hdr = b'RIFF' + struct.pack('<I', 36 + len(data)) + b'WAVE' + b'fmt ' + struct.pack(
        '<IHHIIHH', 16, fmt, 1, sample_rate, sample_rate * block, block, bits) + b'data' + struct.pack('<I', len(data))
    with open(path, 'wb') as f:
        f.write(hdr)
        f.write(data)


def peak_normalise(wave, peak=0.99):
    """wave / max|wave| * 0.99 (utils/app_utils.py:73-74); returns a new tensor/array."""
    m = abs(wave).max()
    return wave / m * peak if float(m) > 0 else wave
