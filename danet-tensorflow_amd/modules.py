'''
Encoder / Estimator / Separator plugins -- the drop-in boundary.

Same class hierarchy, constructor signature `(model, name)`, `__call__`
signatures, `USE_TRUTH` flags, `debug_fetches` behaviour and
`@hparams.register_*` names as the reference's `app/modules.py:11-603`, so
`hparams.get_encoder()` / `get_estimator(name)` / `get_separator(name)` resolve
to MI355X-native implementations unchanged.  Tensors are eager torch CUDA
tensors instead of TF symbolic tensors; every `__call__` runs hand-written HIP
kernels through `ops` (no CPU fallback).

Extension (defaults = the reference's hard-coded values, app/modules.py:153,
212,223-242): `hparams.NUM_LSTM_LAYERS`, `hparams.LSTM_HDIM`.
'''
from math import sqrt

import numpy as np
import torch

from .hparams import hparams
from . import ops


class ModelModule(object):
    '''abstract sub-module of model (app/modules.py:11-25)'''
    def __init__(self, model, name):
        if hparams.DEBUG:
            self.debug_fetches = {}
        self.name = name
        self.model = model

    def __call__(self, s_dropout_keep=1.):
        raise NotImplementedError()


class Encoder(ModelModule):
    '''maps log-magnitude-spectra to embedding (app/modules.py:28-50)'''
    def __init__(self, model, name):
        super(Encoder, self).__init__(model, name)

    def __call__(self, s_mixture, s_dropout_keep=1.):
        '''[batch_size, length, feature_size] ->
        [batch_size, length, feature_size, embedding_size]'''
        raise NotImplementedError()


class Estimator(ModelModule):
    '''Estimates attractor location (app/modules.py:53-70)'''
    USE_TRUTH = True

    def __init__(self, model, name):
        super(Estimator, self).__init__(model, name)

    def __call__(self, s_embed, **kwargs):
        '''[B, T, F, E] -> [B, num_signals, E]'''
        raise NotImplementedError()


class Separator(ModelModule):
    '''mixture magnitudes + attractors + embedding -> separated magnitudes
    (app/modules.py:73-93)'''
    def __init__(self, model, name):
        super(Separator, self).__init__(model, name)

    def __call__(self, s_mixed_signals_pwr, s_attractors, s_embed_flat):
        raise NotImplementedError()


def _uniform_init(r):
    def init(shape, gen):
        return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(r).float()
    return init


def _const_init(arr):
    def init(shape, gen):
        return torch.as_tensor(np.asarray(arr, dtype=np.float32)).reshape(shape).clone()
    return init


def _lstm_bias(hdim, i_bias=1.5):
    b = np.zeros([hdim * 4], dtype=np.float32)
    b[hdim * 1:hdim * 2] = i_bias   # input gate   (modules.py:218)
    b[hdim * 2:hdim * 3] = -1.      # forget gate  (modules.py:219)
    b[hdim * 3:hdim * 4] = 1.       # output gate  (modules.py:220)
    return b


@hparams.register_encoder('toy')
class ToyEncoder(Encoder):
    '''2-layer MLP for debugging (app/modules.py:96-116)'''
    def __init__(self, model, name):
        super(ToyEncoder, self).__init__(model, name)

    def __call__(self, s_signals, s_dropout_keep=1.):
        F, E = hparams.FEATURE_SIZE, hparams.EMBED_SIZE
        m = self.model
        hid = hparams.FFT_SIZE * 2
        # TF default initialiser for get_variable is glorot_uniform
        w0 = m.get_variable(self.name + '/linear0/W', [F, hid],
                            _uniform_init(sqrt(6. / (F + hid))))
        b0 = m.get_variable(self.name + '/linear0/B', [hid], _const_init(np.zeros(hid)))
        w1 = m.get_variable(self.name + '/linear1/W', [hid, F * E],
                            _uniform_init(sqrt(6. / (hid + F * E))))
        b1 = m.get_variable(self.name + '/linear1/B', [F * E], _const_init(np.zeros(F * E)))
        s_mid = ops.lyr_linear(s_signals, w0, b0)
        s_mid = ops.relu(s_mid, hparams.RELU_LEAKAGE)
        s_out = ops.lyr_linear(s_mid, w1, b1)
        return s_out.reshape(hparams.BATCH_SIZE, -1, F, E)


def _lyr_bilstm(name_, model_, s_input_, hdim_, t_axis_, axis_, w_init_, b_init_,
                s_dropout_keep_):
    '''one bidirectional layer on a batch-major tensor (app/modules.py:120-137);
    fwd and reversed-bwd scans run concurrently in one persistent launch.
    Dropout is the identity: the reference never wires keep_prob in
    (main.py:243) and its default is 1.'''
    assert t_axis_ in (-2, 1) and axis_ in (-1, 2)
    D = s_input_.shape[-1]
    params = []
    for d in ('_fwd', '_bwd'):
        W = model_.get_variable('%s%s/LSTM/linear/W' % (name_, d), [D + hdim_, 4 * hdim_], w_init_)
        b = model_.get_variable('%s%s/LSTM/linear/B' % (name_, d), [4 * hdim_], b_init_)
        params += [W, b]
    return ops.LstmLayerFn.apply(s_input_, hdim_, *params)


class _RnnEncoderBase(Encoder):
    NDIR = 1
    INIT_SCALE = 1.15

    def _dims(self):
        raise NotImplementedError()

    def __call__(self, s_signals, s_dropout_keep=1.):
        hdim, nlayer = self._dims()
        F, E = hparams.FEATURE_SIZE, hparams.EMBED_SIZE
        m = self.model
        init_range = self.INIT_SCALE / sqrt(hdim)           # modules.py:154 / :213
        w_initer = _uniform_init(init_range)
        b_initer = _const_init(_lstm_bias(hdim))            # modules.py:158-162 / :217-221
        params = []
        D = F
        for l in range(nlayer):
            for d in (('_fwd', '_bwd') if self.NDIR == 2 else ('',)):
                base = '%s/lstm%d%s/LSTM/linear/' % (self.name, l, d)
                params.append(m.get_variable(base + 'W', [D + hdim, 4 * hdim], w_initer))
                params.append(m.get_variable(base + 'B', [4 * hdim], b_initer))
            D = self.NDIR * hdim
        params.append(m.get_variable(self.name + '/output/W', [D, F * E],
                                     _uniform_init(1.85)))  # modules.py:184-191 / :248-255
        s_out = ops.RnnEncoderFn.apply(s_signals, hdim, nlayer, self.NDIR, *params)
        return s_out.reshape(hparams.BATCH_SIZE, -1, F, E)  # modules.py:192-195 / :256-259


@hparams.register_encoder('lstm-orig')
class LstmEncoder(_RnnEncoderBase):
    '''unidirectional LSTM stack as in the original paper (app/modules.py:140-196);
    reference shape 4 x 600'''
    NDIR = 1
    INIT_SCALE = 1.15

    def __init__(self, model, name):
        super(LstmEncoder, self).__init__(model, name)

    def _dims(self):
        # the reference hard-codes 600 units for the unidirectional stack
        # (modules.py:153); LSTM_HDIM's default (300) is the BiLSTM width, so
        # only a non-default value overrides it.
        h = hparams.LSTM_HDIM
        return (600 if h == 300 else h), hparams.NUM_LSTM_LAYERS


@hparams.register_encoder('bilstm-orig')
class BiLstmEncoder(_RnnEncoderBase):
    '''Bi-LSTM stack as in the original paper (app/modules.py:199-260);
    reference shape 4 x 300 per direction'''
    NDIR = 2
    INIT_SCALE = .75

    def __init__(self, model, name):
        super(BiLstmEncoder, self).__init__(model, name)

    def _dims(self):
        return hparams.LSTM_HDIM, hparams.NUM_LSTM_LAYERS


class _TruthEstimatorBase(Estimator):
    USE_TRUTH = True
    MODE = None

    def __call__(self, s_embed, s_src_pwr, s_mix_pwr, s_embed_flat=None):
        s_attractors = ops.TruthAttractorFn.apply(
            s_embed, s_src_pwr, s_mix_pwr, ops.TRUTH_MODES[self.MODE], float(hparams.EPS))
        if hparams.DEBUG:
            self.debug_fetches = dict()
        return s_attractors                                  # float[B, C, E]


@hparams.register_estimator('truth')
class AverageEstimator(_TruthEstimatorBase):
    '''simple average of true assignment (app/modules.py:382-412)'''
    MODE = 'truth'

    def __init__(self, model, name):
        super(AverageEstimator, self).__init__(model, name)


@hparams.register_estimator('truth-threshold')
class ThreshouldedAverageEstimator(_TruthEstimatorBase):
    '''true assignment, bins with |mix| <= 5 cut (app/modules.py:415-450)'''
    MODE = 'truth-threshold'

    def __init__(self, model, name):
        super(ThreshouldedAverageEstimator, self).__init__(model, name)


@hparams.register_estimator('truth-weighted')
class WeightedAverageEstimator(_TruthEstimatorBase):
    '''|mix|-weighted average of true assignment (app/modules.py:453-487)'''
    MODE = 'truth-weighted'

    def __init__(self, model, name):
        super(WeightedAverageEstimator, self).__init__(model, name)


@hparams.register_estimator('anchor')
class AnchoredEstimator(Estimator):
    '''best anchor subset + one EM step (app/modules.py:490-545)'''
    USE_TRUTH = False

    def __init__(self, model, name):
        super(AnchoredEstimator, self).__init__(model, name)
        self.name = name

    def __call__(self, s_embed, s_src_pwr=None, s_mix_pwr=None, s_embed_flat=None):
        v_anchors = self.model.get_variable(
            self.name + '/anchors', [hparams.NUM_ANCHOR, hparams.EMBED_SIZE],
            lambda shape, gen: torch.randn(shape, generator=gen, dtype=torch.float64).float())
        s_attractors, s_attractor_sets, s_subset_choice = ops.AnchorAttractorFn.apply(
            s_embed, v_anchors, hparams.MAX_N_SIGNAL)
        if hparams.DEBUG:
            self.debug_fetches = dict(
                asets=s_attractor_sets, anchors=v_anchors, subset_choice=s_subset_choice)
        return s_attractors


@hparams.register_estimator('kmeans')
class KMeansEstimator(Estimator):
    '''k-means-style attractor estimation for inference (BASELINE cfg 5).  NOT in
    the reference ("Only the favorable anchor method is implemented",
    README.md:216) -- an extension with no reference behaviour to match.
    Initialised from the anchor estimator's attractors (so it shares the
    `anchors` variable name and shape, app/modules.py:503-506), then
    hparams.KMEANS_ITERS Lloyd iterations: assign every time-frequency bin to its
    nearest attractor by dot product (the separator's own similarity), recompute
    each attractor as the |mix|-weighted mean of its bins (the reference's
    truth-weighted formula, app/modules.py:476-482, with estimated instead of
    ideal assignments).  Built entirely from the existing HIP kernels:
    danet_separate_fwd -> argmax inside danet_attractor_truth_fwd.'''
    USE_TRUTH = False

    def __init__(self, model, name):
        super(KMeansEstimator, self).__init__(model, name)
        self.name = name

    def __call__(self, s_embed, s_src_pwr=None, s_mix_pwr=None, s_embed_flat=None):
        B, T, F, E = s_embed.shape
        C = hparams.MAX_N_SIGNAL
        v_anchors = self.model.get_variable(
            self.name + '/anchors', [hparams.NUM_ANCHOR, hparams.EMBED_SIZE],
            lambda shape, gen: torch.randn(shape, generator=gen, dtype=torch.float64).float())
        with torch.no_grad():
            s_embed_d = s_embed.detach()
            if s_mix_pwr is None:
                s_mix_pwr = torch.ones(B, T, F, device=s_embed.device)
            s_attr, _, _ = ops.AnchorAttractorFn.apply(s_embed_d, v_anchors.detach(), C)
            s_flat = s_embed_d.reshape(B, -1, E)
            for _ in range(int(hparams.KMEANS_ITERS)):
                # score[b,c,n] = |mix| * softmax_c(embed . attr): argmax_c = nearest attractor
                s_score, _ = ops.SeparateFn.apply(s_mix_pwr, s_attr, s_flat, 0, False)
                s_attr = ops.TruthAttractorFn.apply(
                    s_embed_d, s_score, s_mix_pwr, ops.TRUTH_MODES['truth-weighted'],
                    float(hparams.EPS))
        if hparams.DEBUG:
            self.debug_fetches = dict(anchors=v_anchors)
        return s_attr


class _DotSeparatorBase(Separator):
    ACT = None

    def __call__(self, s_mixed_signals_pwr, s_attractors, s_embed_flat):
        s_sep, s_masks = ops.SeparateFn.apply(
            s_mixed_signals_pwr, s_attractors, s_embed_flat, self.ACT, bool(hparams.DEBUG))
        if hparams.DEBUG:
            self.debug_fetches['masks'] = s_masks
        return s_sep                                         # [B, C, T, F]


@hparams.register_separator('dot-sigmoid-orig')
class DotSeparatorSigmoid(_DotSeparatorBase):
    '''sigmoid(embed . attractor) masks (app/modules.py:548-574)'''
    ACT = 1

    def __init__(self, model, name):
        super(DotSeparatorSigmoid, self).__init__(model, name)


@hparams.register_separator('dot-softmax-orig')
class DotSeparatorSoftmax(_DotSeparatorBase):
    '''softmax_c(embed . attractor) masks (app/modules.py:577-603)'''
    ACT = 0

    def __init__(self, model, name):
        super(DotSeparatorSoftmax, self).__init__(model, name)
