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
