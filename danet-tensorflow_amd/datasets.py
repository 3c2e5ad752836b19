'''
Datasets.  Only the reference's data-free `toy` generator
(app/datasets/dataset.py:43-63) is re-stated; `timit` / `wsj0` need licensed
corpora and are out of scope.  `synth` is the speech-shaped 8 kHz
2-speaker generator the benchmarks use.
'''
import numpy as np

from .hparams import hparams


class Dataset(object):
    '''base class (app/datasets/dataset.py:12-35)'''
    def __init__(self):
        self.is_loaded = False

    def epoch(self, subset, batch_size, shuffle=False):
        raise NotImplementedError()

    def install_and_load(self):
        raise NotImplementedError()


@hparams.register_dataset('toy')
class WhiteNoiseData(Dataset):
    '''always generates uniform noise rand(batch, 128, FEATURE_SIZE), 10 batches
    per epoch (app/datasets/dataset.py:43-63)'''
    def epoch(self, subset, batch_size, shuffle=False):
        if not self.is_loaded:
            raise RuntimeError('Dataset is not loaded.')
        for _ in range(10):
            signal = np.random.rand(
                batch_size, 128, hparams.FEATURE_SIZE).astype(hparams.FLOATX)
            yield (signal,)

    def install_and_load(self):
        self.is_loaded = True
        return


def speech_shaped_wave(rng, n_samples, smprate=8000, rms=1000.0, phase=0.0):
    '''white N(0,1) -> one-pole low-pass (0.95) -> 3 Hz amplitude envelope ->
    int16-like RMS; the reference feeds un-normalised int16-scale waveforms
    (app/datasets/TIMIT/process.py:44-46)'''
    import scipy.signal
    x = rng.randn(n_samples).astype(np.float32)
    y = scipy.signal.lfilter([1.0], [1.0, -0.95], x)
    t = np.arange(n_samples) / float(smprate)
    y = y * (0.5 * (1.0 + np.sin(2 * np.pi * 3.0 * t + phase)))
    y = y * (rms / (np.sqrt(np.mean(y ** 2)) + 1e-12))
    return y.astype(np.float32)


def synth_waves(seed, n_utt, n_frames, smprate=None):
    '''[n_utt, Ls] float32 waveforms whose STFT has exactly `n_frames` frames
    (T = 1 + ceil(Ls/S))'''
    rng = np.random.RandomState(seed)
    S = hparams.FFT_STRIDE
    Ls = (n_frames - 1) * S
    smprate = smprate or hparams.SMPRATE
    return np.stack([
        speech_shaped_wave(rng, Ls, smprate, phase=rng.uniform(0, 2 * np.pi))
        for _ in range(n_utt)])


@hparams.register_dataset('synth')
class SynthSpeechData(Dataset):
    '''speech-shaped synthetic 2..C-speaker data (SURVEY 8d cfg 2/3): yields
    complex64 spectra [batch, T, FEATURE_SIZE] of independent single-speaker
    utterances, like the reference's TIMIT / WSJ0 iterators
    (app/datasets/timit.py, wsj0.py); mixing happens in the model (main.py:233).
    The STFT runs on the GPU (danet_stft).'''
    N_BATCH = {'train': 10, 'valid': 2, 'test': 2}
    N_FRAMES = 160

    def epoch(self, subset, batch_size, shuffle=False):
        if not self.is_loaded:
            raise RuntimeError('Dataset is not loaded.')
        import torch
        from . import utils
        base = {'train': 0, 'valid': 10 ** 6, 'test': 2 * 10 ** 6}[subset]
        for i in range(self.N_BATCH[subset]):
            seed = base + i if not shuffle else base + int(np.random.randint(0, 10 ** 5))
            waves = synth_waves(seed, batch_size, self.N_FRAMES)
            yield (utils.stft(torch.as_tensor(waves)).cpu().numpy(),)

    def install_and_load(self):
        self.is_loaded = True


@hparams.register_dataset('synth-varlen')
class SynthVarLenSpeechData(SynthSpeechData):
    '''the same generator with utterances of DIFFERENT lengths, batched the way the
    reference's WSJ0 / TIMIT iterators batch variable-length utterances
    (app/datasets/wsj0.py:51-55, timit.py): every spectrogram is zero-padded on both sides
    of the time axis to the longest of the batch by `utils.random_zeropad` (python
    `random`, app/utils.py:78-92).  Padded frames are exact zeros, so a mixture whose
    sources are all padded at a frame has |mix| = 0 there: the argmax / threshold /
    weighted-mean tie cases of the estimators reach the full model.'''
    MIN_FRAMES = 96

    def epoch(self, subset, batch_size, shuffle=False):
        if not self.is_loaded:
            raise RuntimeError('Dataset is not loaded.')
        import torch
        from . import utils
        base = {'train': 0, 'valid': 10 ** 6, 'test': 2 * 10 ** 6}[subset]
        for i in range(self.N_BATCH[subset]):
            seed = base + i if not shuffle else base + int(np.random.randint(0, 10 ** 5))
            rng = np.random.RandomState(seed + 7)
            lens = rng.randint(self.MIN_FRAMES, self.N_FRAMES + 1, size=batch_size)
            waves = synth_waves(seed, batch_size, self.N_FRAMES)
            S = hparams.FFT_STRIDE
            data = [utils.stft(torch.as_tensor(w[:(n - 1) * S])).cpu().numpy()
                    for w, n in zip(waves, lens)]
            max_len = max(map(len, data))
            spectra_li = [utils.random_zeropad(x, max_len - len(x), axis=-2) for x in data]
            yield (np.stack(spectra_li),)
