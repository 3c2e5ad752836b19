#!/usr/bin/env python
'''Estimator / separator / PIT-loss kernels of one cfg-2 (or --cfg4) train step, each timed in a
loop with HIP events (GPU box):  python tools/bench_heads.py [--cfg4]'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import ops, _lib  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device('cuda')
    cfg4 = '--cfg4' in sys.argv
    B, T, F = 32, 128, 129
    E, C, A = (40, 3, 6) if cfg4 else (20, 2, 6)
    torch.manual_seed(0)
    embed = (torch.randn(B, T, F, E, device=dev) * 0.5).requires_grad_(True)
    anchors = (torch.randn(A, E, device=dev) * 0.5).requires_grad_(True)
    mix = torch.rand(B, T, F, device=dev)
    src = torch.randn(B, C, T, F, device=dev, dtype=torch.complex64)
    phasor = torch.randn(B, T, F, 2, device=dev)
    mb = embed.numel() * 4 / 1e6
    L = _lib.load()
    for _ in range(50):      # clocks
        ops.AnchorAttractorFn.apply(embed.detach(), anchors.detach(), C)
    res = []
    attr, asets, choice = ops.AnchorAttractorFn.apply(embed, anchors, C)
    res.append(('anchor fwd (+final)', timeit(lambda: ops.AnchorAttractorFn.apply(embed.detach(), anchors.detach(), C)), mb))
    sep, _ = ops.SeparateFn.apply(mix, attr, embed.view(B, T * F, E), 0, False)
    res.append(('separate fwd', timeit(lambda: ops.SeparateFn.apply(mix, attr.detach(), embed.detach().view(B, T * F, E), 0, False)), mb))
    loss, snr, perm = ops.PitMseFn.apply(src, sep, phasor, 0, 1e-7)
    res.append(('pit fwd (+final)', timeit(lambda: ops.PitMseFn.apply(src, sep.detach(), phasor, 0, 1e-7)), 0))

    def bwd():
        embed.grad = None
        anchors.grad = None
        loss.backward(retain_graph=True)
    res.append(('whole backward of the heads (pit, separate, anchor + torch glue)', timeit(bwd), 3 * mb))

    def chain():
        a, _, _ = ops.AnchorAttractorFn.apply(embed, anchors, C)
        s, _ = ops.SeparateFn.apply(mix, a, embed.view(B, T * F, E), 0, False)
        l, _, _ = ops.PitMseFn.apply(src, s, phasor, 0, 1e-7)
        embed.grad = None
        anchors.grad = None
        l.backward()
    res.append(('forward + backward chain', timeit(chain), 0))
    for name, us, mbs in res:
        print('%-70s %7.1f us%s' % (name, us, '   (%.1f TB/s on %.0f MB)' % (mbs / us, mbs) if mbs else ''))


if __name__ == '__main__':
    main()
