'''Size-independent properties at the FULL size of BASELINE cfg 2 (B = 32, T = 128, 129 bins,
3 x 300 BiLSTM, E = 20, 6 anchors) and cfg 4 as written, where the CPU oracle is too slow to be the
checker: batch-permutation equivariance (bit-exact: a row's arithmetic must not depend on which
workgroup / cluster / MFMA row it lands in), time-reversal symmetry of the two scan directions
(bit-exact), masks on the simplex, permutation invariance of the PIT loss, linearity of backward.'''
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok(), 'persistent LSTM kernel reported a hand-off timeout'


def _cfg2_model(hp, **kw):
    from danet_amd.model import Model
    base = dict(BATCH_SIZE=32, MAX_N_SIGNAL=2, FFT_SIZE=256, FFT_STRIDE=64, EMBED_SIZE=20,
                NUM_LSTM_LAYERS=3, LSTM_HDIM=300, NUM_ANCHOR=6, ENCODER_TYPE='bilstm-orig',
                TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
                SEPARATOR_TYPE='dot-softmax-orig', MAX_TRAIN_LEN=128)
    base.update(kw)
    hp.load(base)
    hp.digest()
    return Model('prop', device='cuda', seed=11).build()


def _src(hp, T, seed=0, scale=300.0):
    rng = np.random.RandomState(seed)
    B, C, F = hp.BATCH_SIZE, hp.MAX_N_SIGNAL, hp.FEATURE_SIZE
    x = (rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * scale
    x *= rng.rand(B, C, T, 1) ** 2          # frames of very different level (as speech has)
    return torch.as_tensor(x.astype(np.complex64)).cuda()


def test_cfg2_full_size_batch_permutation_is_bit_exact(hp):
    '''utterances are independent all the way to the masks: permuting the batch permutes every
    per-utterance output BIT FOR BIT (32 rows = 2 clusters of 16 in the recurrent kernels, 4 of 8 in
    BPTT, different MFMA rows, different workgroups of the estimator -- none of it may matter)'''
    model = _cfg2_model(hp)
    src = _src(hp, 128)
    perm = torch.as_tensor(np.random.RandomState(3).permutation(hp.BATCH_SIZE)).cuda()
    a = model.debug_fetch(src)
    b = model.debug_fetch(src[perm].contiguous())
    for k in ('embed', 'attrs', 'output'):
        assert torch.equal(a[k][perm], b[k]), k
    # masks on the simplex (dot-softmax separator), finite everywhere
    o = model.forward(src)
    sep, mix = o['sep_pwr'], src.sum(dim=1).abs()
    assert torch.isfinite(sep).all() and (sep >= 0).all()
    assert float(((sep.sum(dim=1) - mix).abs() / (mix + 1e-3)).max().detach()) < 1e-4


def test_cfg2_full_size_pit_loss_and_gradients_ignore_speaker_order(hp):
    '''permutation-invariant training: swapping the two sources of every mixture changes neither
    the loss nor any parameter gradient (beyond summation-order rounding), and backward is linear
    in the incoming gradient'''
    model = _cfg2_model(hp)
    src = _src(hp, 128, seed=1)
    grads = []
    for s, scale in ((src, 1.0), (src.flip(1).contiguous(), 1.0), (src, 3.0)):
        model.zero_grad()
        out = model.forward(s)
        (out['loss'] * scale).backward()
        g = torch.cat([v.grad.reshape(-1) for v in model.vars.values() if v.grad is not None]).clone()
        grads.append((float(out['loss'].detach()), g))
    (l0, g0), (l1, g1), (l2, g2) = grads
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    gmax = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-4 * gmax
    assert float((g2 - 3.0 * g0).abs().max()) <= 2e-5 * 3.0 * gmax


@pytest.mark.parametrize('B,T,D,H', [(32, 128, 600, 300), (32, 128, 129, 300), (32, 24, 1200, 600),
                                     (1, 1251, 600, 300)])     # cfg 5: the B = 1 GEMV kernel, full length
def test_bilstm_time_reversal_symmetry(B, T, D, H):
    '''a BiLSTM whose two directions swap their weights, fed the time-reversed input, returns the
    time-reversed output with the two halves swapped -- bit for bit in forward, to summation-order
    rounding in backward: the forward and the reversed scan are the same arithmetic on mirrored indices (fused
    forward at cfg-2 widths, hoisted 12-unit forward and U = 32 BPTT at the cfg-4 width)'''
    from danet_amd import ops
    rng = np.random.RandomState(B + T + H)
    r = 0.75 / np.sqrt(H)
    x = torch.as_tensor((rng.randn(B, T, D) * 0.7).astype(np.float32)).cuda()
    Wf, Wb = [torch.as_tensor((rng.uniform(-r, r, size=(D + H, 4 * H)) * 2).astype(np.float32)).cuda()
              for _ in range(2)]
    bf, bb = [torch.as_tensor((rng.randn(4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    dy = torch.as_tensor(rng.randn(B, T, 2 * H).astype(np.float32)).cuda()

    def run(xin, params, dyin):
        xin = xin.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in params]
        y = ops.LstmLayerFn.apply(xin, H, *ps)
        y.backward(dyin)
        return y.detach(), xin.grad, [p.grad for p in ps]

    y1, dx1, g1 = run(x, [Wf, bf, Wb, bb], dy)
    swap = lambda t: torch.cat([t[..., H:], t[..., :H]], dim=-1)
    y2, dx2, g2 = run(x.flip(1).contiguous(), [Wb, bb, Wf, bf], swap(dy).flip(1).contiguous())
    from danet_amd import _lib
    # (LstmLayerFn hands the kernel rows of length D: the fused forward needs D % 4 == 0)
    if D % 4 == 0 and _lib.load().danet_lstm_fwd_fused_supported(T, B, H, 2, D) == 1:
        assert torch.equal(swap(y2).flip(1), y1)           # whole cell in one kernel: same arithmetic
    else:
        # hoisted input projection: both directions' products are ONE grouped stream-K launch whose
        # tiles are cut at different k positions -> the two directions differ by rounding
        assert float((swap(y2).flip(1) - y1).abs().max()) <= 2e-6 * float(y1.abs().max())
    # dX = da_f Wx_f^T + da_b Wx_b^T is one K-concatenated product: swapping the directions swaps
    # the order of its two halves -> rounding only (da itself is bit-exact, or y's test above and
    # the weight gradients below would not hold)
    # (2400-term float32 sums in two different orders, the stream-K launch cuts them at other k)
    assert float((dx2.flip(1) - dx1).abs().max()) <= 5e-6 * float(dx1.abs().max())
    # weight gradients: same products, the K (time) order of the GEMM reversed -> rounding only
    for a, b in ((g1[0], g2[2]), (g1[1], g2[3]), (g1[2], g2[0]), (g1[3], g2[1])):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
