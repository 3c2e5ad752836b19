'''
Where does the mask error at TRAINED parameters come from?  (diagnostic; GPU)

Trains cfg 2 like tests/test_gpu_trained_parity.py, then splits the HIP-vs-float64 mask
error into (a) the encoder's contribution (HIP embedding -> float64 estimator + separator),
(b) the heads' contribution (float64 embedding rounded to float32 -> HIP estimator +
separator), and prints the same split for the float32 oracle, plus the logit scale at the
worst bin.  python tools/parity_decompose.py [steps]
'''
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from danet_amd.hparams import hparams as hp  # noqa: E402
from danet_amd import ops  # noqa: E402
from oracle import danet_oracle as O  # noqa: E402
import test_gpu_trained_parity as TP  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    hp.reset()
    model = TP._setup(hp, 32)
    batches = TP._batches(hp, 4)
    for i in range(steps):
        model.train_step(batches[i % 4])
    torch.cuda.synchronize()
    n = 4
    src = batches[0][:n].cpu().numpy()
    p = model.param_dict()
    cfg = dict(H=300, L=3, E=20, C=2, A=6, train_est='anchor', infer_est='anchor',
               separator='dot-softmax-orig')
    got = TP.product_outputs(model, hp, batches[0], n)
    r64 = O.model_forward(src.astype(np.complex128), p, cfg)
    r32 = O.model_forward(src.astype(np.complex64), p, cfg, dtype=np.float32)
    anchors = p['global/train_estimator/anchors'].astype(np.float64)

    def heads64(embed, mix_pwr):
        e = np.asarray(embed, dtype=np.float64)
        attr = O.est_anchor(e, anchors, 2)
        _, m = O.sep_dot(mix_pwr, attr, e.reshape(n, -1, 20), 'softmax', return_masks=True)
        return m

    def mx(a, b):
        return float(np.abs(np.asarray(a, np.float64) - b).max())

    m64 = r64['masks']
    print('masks  HIP vs f64      : %.3e' % mx(got['masks'], m64))
    print('masks  f32 vs f64      : %.3e' % mx(r32['masks'], m64))
    print('(a) f64 heads on HIP embed : %.3e' % mx(heads64(got['embed'], r64['mix_pwr']), m64))
    print('(a) f64 heads on f32 embed : %.3e' % mx(heads64(r32['embed'], r64['mix_pwr']), m64))
    print('    f64 heads on f32-ROUNDED f64 embed: %.3e' %
          mx(heads64(r64['embed'].astype(np.float32), r64['mix_pwr']), m64))
    # (b) HIP heads on the rounded float64 embedding
    e32 = torch.as_tensor(r64['embed'].astype(np.float32)).cuda()
    B = hp.BATCH_SIZE
    ebig = torch.zeros(B, *e32.shape[1:], device='cuda'); ebig[:n] = e32
    mixp = torch.zeros(B, *e32.shape[1:3], device='cuda')
    mixp[:n] = torch.as_tensor(r64['mix_pwr'].astype(np.float32)).cuda()
    with torch.no_grad():
        attr, _, _ = ops.AnchorAttractorFn.apply(ebig, model.vars['global/train_estimator/anchors'], 2)
        _, mk = ops.SeparateFn.apply(mixp, attr, ebig.reshape(B, -1, 20), 0, True)
    print('(b) HIP heads on rounded f64 embed: %.3e' % mx(mk[:n].cpu().numpy(), m64))
    print('embed  HIP vs f64 max abs %.3e (max |embed| %.1f);  f32: %.3e' %
          (mx(got['embed'], r64['embed']), np.abs(r64['embed']).max(), mx(r32['embed'], r64['embed'])))
    # per-stage encoder error: stack output before the projection
    x = r64['mix_log']
    _, acts64, y64 = O.bilstm_encoder(x, {k: v.astype(np.float64) for k, v in p.items()}, 300, 3, 20, return_all=True)
    _, acts32, y32 = O.bilstm_encoder(r32['mix_log'], {k: v.astype(np.float32) for k, v in p.items()}, 300, 3, 20, return_all=True)
    print('stack output (centred) f32 vs f64 max abs %.3e' % mx(y32, y64))
    W = p['global/encoder/output/W']
    print('projection of the f64 stack output in f32 (numpy sgemm) vs f64: %.3e' %
          mx((y64.astype(np.float32) @ W).reshape(r64['embed'].shape), r64['embed']))
    yc = torch.as_tensor(y64.astype(np.float32)).cuda().contiguous()
    out = torch.empty(n, 128, W.shape[1], device='cuda')
    ops.gemm(yc, model.vars['global/encoder/output/W'], out, n * 128, W.shape[1], 600, 600, W.shape[1], W.shape[1])
    print('projection of the f64 stack output by the HIP GEMM vs f64     : %.3e' %
          mx(out.cpu().numpy().reshape(r64['embed'].shape), r64['embed']))
    # the HIP stack, layer by layer (activations kept by the encoder's autograd node)
    with torch.enable_grad():
        fe = ops.frontend(batches[0])
        emb = model.encoder(fe['mix_log'])
    node = emb.grad_fn
    while node is not None and not hasattr(node, 'ctxs'):
        node = node.next_functions[0][0] if node.next_functions else None
    T = 128
    for l, c in enumerate(node.ctxs):
        yh = c.ypad[1:T + 1].transpose(0, 1)[:n].cpu().numpy()          # [n, T, 2H]
        print('layer %d output: HIP vs f64 max abs %.3e rms %.3e | f32 oracle %.3e rms %.3e' % (
            l, mx(yh, acts64[l]), np.sqrt(((yh - acts64[l]) ** 2).mean()),
            mx(acts32[l], acts64[l]), np.sqrt(((acts32[l].astype(np.float64) - acts64[l]) ** 2).mean())))
    # the SAME parameters through the hoisted forward (input GEMM + recurrent kernel)
    from danet_amd import _lib
    _lib.set_option('lstm_fwd_fused', 0)
    with torch.enable_grad():
        emb2 = model.encoder(fe['mix_log'])
    node2 = emb2.grad_fn
    while node2 is not None and not hasattr(node2, 'ctxs'):
        node2 = node2.next_functions[0][0] if node2.next_functions else None
    for l, c in enumerate(node2.ctxs):
        yh = c.ypad[1:T + 1].transpose(0, 1)[:n].cpu().numpy()
        print('layer %d output, HOISTED forward at the same parameters: HIP vs f64 max abs %.3e rms %.3e' % (
            l, mx(yh, acts64[l]), np.sqrt(((yh - acts64[l]) ** 2).mean())))
    print('embed, hoisted forward: HIP vs f64 max abs %.3e' % mx(emb2[:n].detach().cpu().numpy(), r64['embed']))
    _lib.apply_env_options()
    # input centring: the per-utterance mean (a COMMON-MODE error of every input of layer 0)
    ml64 = r64['mix_log']
    mean64 = ml64.mean(axis=(1, 2))
    mean32 = r32['mix_log'].mean(axis=(1, 2), dtype=np.float32)
    xin = torch.zeros(B, T, 129, device='cuda'); xin[:n] = torch.as_tensor(ml64.astype(np.float32)).cuda()
    xc = torch.empty(T, B, 132, device='cuda')
    mean_hip = ops.center(xin, B, T, 129, 0, 129, xc, 1, 132)[:n].cpu().numpy()
    print('input mean: f64 %s' % mean64)
    print('input mean error: HIP %s | numpy f32 %s' % (mean_hip - mean64, mean32 - mean64))
    xc64 = (ml64 - mean64[:, None, None])
    print('centred input: HIP vs f64 max %.3e rms %.3e | numpy f32 max %.3e rms %.3e' % (
        mx(xc.transpose(0, 1)[:n, :, :129].cpu().numpy(), xc64),
        np.sqrt(((xc.transpose(0, 1)[:n, :, :129].cpu().numpy() - xc64) ** 2).mean()),
        mx((r32['mix_log'] - mean32[:, None, None]).astype(np.float32), xc64),
        np.sqrt((((r32['mix_log'] - mean32[:, None, None]).astype(np.float32) - xc64) ** 2).mean())))
    # layer 0 in isolation on the exactly centred input (rounded to float32)
    x0 = xc64.astype(np.float32)
    names0 = ['global/encoder/lstm0_%s/LSTM/linear/%s' % (d, w) for d in ('fwd', 'bwd') for w in ('W', 'B')]
    W0f, b0f, W0b, b0b = [p[k] for k in names0]
    ref0 = O.lyr_bilstm(x0.astype(np.float64), W0f.astype(np.float64), b0f.astype(np.float64),
                        W0b.astype(np.float64), b0b.astype(np.float64), 300)
    np0 = O.lyr_bilstm(x0, W0f, b0f, W0b, b0b, 300)
    xb0 = torch.zeros(B, T, 129, device='cuda'); xb0[:n] = torch.as_tensor(x0).cuda()
    with torch.no_grad():
        y0 = ops.LstmLayerFn.apply(xb0, 300, *[model.vars[k] for k in names0])[:n].cpu().numpy()
    print('layer 0 alone (exact centred input): HIP max %.3e rms %.3e | numpy f32 max %.3e rms %.3e' % (
        mx(y0, ref0), np.sqrt(((y0 - ref0) ** 2).mean()), mx(np0, ref0), np.sqrt(((np0 - ref0) ** 2).mean())))
    # layer 1 in ISOLATION: the float64 oracle's layer-0 output (rounded to float32) through
    # layer 1 only -- intrinsic noise of one layer: HIP fused / HIP hoisted / numpy float32
    x1 = acts64[0].astype(np.float32)                                  # [n, T, 2H]
    names = ['global/encoder/lstm1_%s/LSTM/linear/%s' % (d, w) for d in ('fwd', 'bwd') for w in ('W', 'B')]
    Wf, bf, Wb, bb = [p[k] for k in names]
    ref1 = O.lyr_bilstm(x1.astype(np.float64), Wf.astype(np.float64), bf.astype(np.float64),
                        Wb.astype(np.float64), bb.astype(np.float64), 300)
    np32 = O.lyr_bilstm(x1, Wf, bf, Wb, bb, 300)
    print('layer 1 alone: numpy f32 vs f64 max %.3e rms %.3e' % (mx(np32, ref1), np.sqrt(((np32 - ref1) ** 2).mean())))
    xb = torch.zeros(B, T, 600, device='cuda'); xb[:n] = torch.as_tensor(x1).cuda()
    prm = [model.vars[k] for k in names]
    for fused in (1, 0):
        _lib.set_option('lstm_fwd_fused', fused)
        with torch.no_grad():
            yh = ops.LstmLayerFn.apply(xb, 300, *prm)[:n].cpu().numpy()
        print('layer 1 alone: HIP (fused=%d) vs f64 max %.3e rms %.3e' % (
            fused, mx(yh, ref1), np.sqrt(((yh - ref1) ** 2).mean())))
        et = np.abs(yh - ref1).max(axis=(0, 2))
        print('   max error by time step (every 16th): ' + ' '.join('%.1e' % v for v in et[::16]))
    et = np.abs(np32 - ref1).max(axis=(0, 2))
    print('   numpy f32, by time step (every 16th):   ' + ' '.join('%.1e' % v for v in et[::16]))
    _lib.apply_env_options()
    ych = node.yc[:n].cpu().numpy()
    print('centred stack output: HIP vs f64 max abs %.3e | f32 oracle %.3e' % (mx(ych, y64), mx(y32, y64)))
    print('f64 projection of the HIP stack output vs f64 embed: %.3e' %
          mx((ych.astype(np.float64) @ W.astype(np.float64)).reshape(r64['embed'].shape), r64['embed']))
    xin = (r64['mix_log'] - r64['mix_log'].mean(axis=(1, 2), keepdims=True))
    print('front-end: mix_log HIP vs f64 %.3e' % mx(fe['mix_log'][:n].cpu().numpy(), r64['mix_log']))
    d = np.abs(got['masks'].astype(np.float64) - m64)
    idx = np.unravel_index(d.argmax(), d.shape)
    b, t, f, c = idx
    lg = r64['embed'][b, t, f] @ r64['attrs'][b].T
    print('worst bin', idx, 'mask', m64[idx], 'logits', lg, '|embed|', np.abs(r64['embed'][b, t, f]).max(),
          '|attr|', np.abs(r64['attrs'][b]).max())
    lg_all = np.einsum('btfe,bce->btfc', r64['embed'], r64['attrs'])
    print('logit |max| %.1f, rms %.1f' % (np.abs(lg_all).max(), np.sqrt((lg_all ** 2).mean())))


if __name__ == '__main__':
    main()
