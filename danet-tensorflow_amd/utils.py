'''
Audio front-end / back-end helpers (reference app/utils.py:53-135) on the GPU:
STFT and iSTFT are HIP kernels (ops.stft / ops.istft); wav I/O and FFT
resampling stay on the host like the reference (scipy), they are not on the
hot path.
'''
from random import randint
from math import ceil

import numpy as np
import torch

from .hparams import hparams
from . import ops


def _window(device):
    return torch.as_tensor(np.asarray(hparams.FFT_WND, dtype=np.float32)).to(device)


def stft(data, device=None):
    '''the reference's STFT call (app/utils.py:117-122) for a float waveform
    (numpy or torch, 1-D or [n_sig, Ls]); returns complex64 torch tensor
    [.., time, FEATURE_SIZE] on the GPU.'''
    if not torch.is_tensor(data):
        data = torch.as_tensor(np.asarray(data, dtype=np.float32))
    device = device or (data.device if data.is_cuda else 'cuda')
    data = data.to(device=device, dtype=torch.float32)
    return ops.stft(data, _window(data.device), hparams.FFT_SIZE, hparams.FFT_STRIDE)


def istft(X, stride, window):
    '''app/utils.py:53-75.  X complex [length, 1+fft_size//2] (numpy or torch)
    -> float64 waveform (numpy if numpy came in)'''
    was_np = not torch.is_tensor(X)
    Xt = torch.as_tensor(np.asarray(X)) if was_np else X
    Xt = Xt.to(device='cuda' if not Xt.is_cuda else Xt.device, dtype=torch.complex64)
    w = torch.as_tensor(np.asarray(window, dtype=np.float32)).to(Xt.device) \
        if not torch.is_tensor(window) else window.to(Xt.device, torch.float32)
    y = ops.istft(Xt, stride, w)
    return y.cpu().numpy() if was_np else y


def random_zeropad(X, padlen, axis=-1):
    '''randomly zero-pads both sides of `axis`, total `padlen`
    (app/utils.py:78-92); host-side numpy like the reference (dataset batching)'''
    if padlen == 0:
        return X
    l = randint(0, padlen)
    r = padlen - l
    ndim = X.ndim
    assert -ndim <= axis < ndim
    axis %= X.ndim
    pad = [(0, 0)] * axis + [(l, r)] + [(0, 0)] * (ndim - axis - 1)
    return np.pad(X, pad, mode='constant')


def load_wavfile(filename):
    '''WAV -> resample to hparams.SMPRATE -> STFT (app/utils.py:95-122);
    returns numpy complex [time, FEATURE_SIZE]'''
    import scipy.io.wavfile
    import scipy.signal
    if filename is None:
        raise IOError('WAV file not specified, please specify via --input-file argument.')
    smprate, data = scipy.io.wavfile.read(filename)
    if smprate != hparams.SMPRATE:
        data = scipy.signal.resample(
            data, int(ceil(len(data) * hparams.SMPRATE / smprate)))
    return stft(np.asarray(data, dtype=np.float32)).cpu().numpy().astype(hparams.COMPLEXX)


def save_wavfile(filename, feature):
    '''[time, FEATURE_SIZE] complex -> WAV (app/utils.py:125-135)'''
    import scipy.io.wavfile
    data = istft(np.asarray(feature), stride=hparams.FFT_STRIDE, window=hparams.FFT_WND)
    scipy.io.wavfile.write(filename, hparams.SMPRATE, data)
