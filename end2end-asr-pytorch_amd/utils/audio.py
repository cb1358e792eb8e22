"""Waveform loading and the spectrogram front end (reference: utils/audio.py:7-15 and SpectrogramParser.parse_audio,
utils/data_loader.py:60-91).  Own implementation on numpy only: the reference's torchaudio / librosa / sox dependencies
are not part of this build.  sox-based tempo/gain augmentation and noise injection (audio.py:17-61) are host-side data
preparation outside the accelerated path and are not provided.
"""
import wave

import numpy as np


def load_audio(path):
    """16-bit / 32-bit PCM wav -> float32 in [-1, 1], channels averaged (reference: audio.py:7-15)."""
    with wave.open(path, "rb") as f:
        nch, width, n = f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError("unsupported sample width %d in %s" % (width, path))
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    return x


def hamming_window(n):
    """Symmetric Hamming window: the reference passes scipy.signal.hamming as a CALLABLE to librosa, which evaluates
    it as window(n) i.e. sym=True (data_loader.py:20-21,77-79)."""
    k = np.arange(n, dtype=np.float64)
    return (0.54 - 0.46 * np.cos(2.0 * np.pi * k / (n - 1))).astype(np.float32)


def log_spectrogram(y, sample_rate=16000, window_size=0.02, window_stride=0.01, normalize=True):
    """float waveform -> (n_fft/2+1, frames) log1p(|STFT|), optionally (x-mean)/std over the whole utterance with the
    unbiased std torch uses (data_loader.py:72-89).  STFT convention: n_fft = win_length = sr*window_size (320),
    hop = sr*window_stride (160), centred frames with reflect padding (librosa's default of that era)."""
    n_fft = int(sample_rate * window_size)
    hop = int(sample_rate * window_stride)
    y = np.asarray(y, dtype=np.float32)
    if y.size < 2:
        y = np.pad(y, (0, 2 - y.size))
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="reflect") if y.size > pad else np.pad(y, (pad, pad), mode="constant")
    n_frames = 1 + (yp.size - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = yp[idx] * hamming_window(n_fft)[None, :]
    spec = np.abs(np.fft.rfft(frames, n=n_fft, axis=1)).T.astype(np.float32)        # (bins, frames)
    spec = np.log1p(spec)
    if normalize:
        mean = spec.mean()
        std = spec.std(ddof=1)
        spec = (spec - mean) / std
    return spec


def gpu_front_end(inputs, input_sizes, sample_rate=16000, window_size=0.02, window_stride=0.01, src_max_len=None):
    """--gpu-frontend: `inputs` (B,1,1,Lmax) are the loader's zero padded WAVEFORMS and `input_sizes` (B) their sample
    counts (the collate function is unchanged: a waveform is a 1-bin "spectrogram").  Returns what the host path would have
    put in the batch: log-spectrograms (B,1,F,T) normalised per utterance, cut to --src-max-len frames AFTER the
    normalisation (data_loader.py:49-53), and the frame counts."""
    import torch
    from asr_hip import ops
    n_fft, hop = int(sample_rate * window_size), int(sample_rate * window_stride)
    wav = inputs.reshape(inputs.shape[0], inputs.shape[-1]).float().contiguous()
    lens = torch.as_tensor(input_sizes).to(device=wav.device, dtype=torch.int32)
    spect, n_frames = ops.log_spectrogram(wav, lens, n_fft=n_fft, hop=hop, normalize=True)
    if src_max_len is not None and spect.shape[-1] > src_max_len:
        spect = spect[..., :src_max_len].contiguous()
        n_frames = torch.clamp(n_frames, max=src_max_len)
    return spect, n_frames.cpu()
