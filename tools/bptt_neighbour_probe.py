#!/usr/bin/env python
'''GPU box: what a neighbour on a second stream costs the BPTT kernel (cfg 2: B 32, T 128, H 300).

The weight-gradient products of a layer (17.7 GFLOP) run beside the NEXT layer's latency-bound
BPTT kernel.  This probe launches danet_lstm_bwd on one stream and, right behind it on another,
one of: nothing; the layer's four TN products as the capped stream-K group of the step (exact
fp32 matrix instructions); the same number of flops as packed-weight products on the bf16 matrix
cores (csrc/gemm_x6.hip) in 1, 2 or 4 launches.  Prints the BPTT kernel's time and the
neighbour's.   python tools/bptt_neighbour_probe.py
'''
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import _lib, ops  # noqa: E402

L = _lib.load()
ptr = _lib.ptr


def main():
    B, T, H, D = 32, 128, 300, 600
    dev = torch.device('cuda')
    ops.prepare_streams(dev)
    torch.manual_seed(0)
    gx = [torch.randn(T * B, 4 * H, device=dev) * 0.5 for _ in range(2)]
    Wh = [torch.randn(H, 4 * H, device=dev) * (0.75 / H ** 0.5) for _ in range(2)]
    dy = torch.randn(T, B, 2 * H, device=dev)
    n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, 2)
    ws = torch.zeros(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ypad = torch.empty(T + 2, B, 2 * H, device=dev)
    gates = [x.clone() for x in gx]
    cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
    _lib.check(L.danet_lstm_fwd(st, T, B, H, 2, ptr(gates[0]), ptr(gates[1]), ptr(Wh[0]), ptr(Wh[1]),
                                4 * H, ptr(ypad), 2 * H, ptr(gates[0]), ptr(gates[1]),
                                ptr(cells[0]), ptr(cells[1]), ptr(ws), n, None, 0))
    das = [torch.empty(T * B, 4 * H, device=dev) for _ in range(2)]
    side = torch.cuda.Stream(dev)

    # the neighbours
    x = torch.randn(T * B, D, device=dev)
    hprev = torch.randn(T * B, 2 * H, device=dev)
    dWx = [torch.zeros(D, 4 * H, device=dev) for _ in range(2)]
    dWh = [torch.zeros(H, 4 * H, device=dev) for _ in range(2)]
    da = [torch.randn(T * B, 4 * H, device=dev) for _ in range(2)]

    def group_fp32(wgs):
        def run():
            ops.gemm_group([(x, D, da[0], 4 * H, dWx[0], 4 * H, D, 4 * H, 0.0),
                            (x, D, da[1], 4 * H, dWx[1], 4 * H, D, 4 * H, 0.0),
                            (hprev, 2 * H, da[0], 4 * H, dWh[0], 4 * H, H, 4 * H, 0.0),
                            (hprev[:, H:], 2 * H, da[1], 4 * H, dWh[1], 4 * H, H, 4 * H, 0.0)],
                           T * B, transA=True, max_workgroups=wgs)
        return run

    # the same 17.7 GFLOP as NT products with a packed weight: M 4096, N 600, K 3600 / pieces
    Kx = 3600
    Ax = torch.randn(T * B, Kx, device=dev)
    Wx = torch.randn(600, Kx, device=dev) * 0.05
    Cx = torch.empty(T * B, 600, device=dev)

    def x6(pieces):
        k = Kx // pieces
        def run():
            for i in range(pieces):
                ops.gemm_w(Ax[:, i * k:], Kx, Wx[:, i * k:], Kx, 1, Cx, T * B, 600, k, 600)
        return run

    def measure(nb, reps=12):
        tb, tn = [], []
        for it in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.danet_lstm_bwd(st, T, B, H, 2, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
                                        ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
                                        ptr(das[0]), ptr(das[1]), None, None, 0.0, ptr(ws), n, None, 0))
            e1.record()
            if nb is not None:
                with torch.cuda.stream(side):
                    f0.record()
                    nb()
                    f1.record()
            torch.cuda.synchronize()
            assert int(ws[:4].view(torch.int32)[0]) == 0, 'status'
            tb.append(e0.elapsed_time(e1) * 1e3)
            tn.append(f0.elapsed_time(f1) * 1e3 if nb is not None else 0.0)
        tb, tn = sorted(tb[2:]), sorted(tn[2:])
        return tb[len(tb) // 2], tn[len(tn) // 2]

    def group_x6():
        ops.GEMM_X6 = 3
        try:
            group_fp32(0)()
        finally:
            ops.GEMM_X6 = 1

    ops.GEMM_X6 = 1
    for name, nb in (('alone', None), ('fp32 group, 256 workgroups (the step)', group_fp32(256)),
                     ('x6 TN group (the same four products)', group_x6),
                     ('fp32 group, 128 workgroups', group_fp32(128)),
                     ('x6 NT, 1 launch (same flops)', x6(1)), ('x6 NT, 3 launches', x6(3)),
                     ('x6 NT, 6 launches', x6(6))):
        if nb is not None:
            nb()
        b, t = measure(nb)
        print('%-42s BPTT %6.1f us   neighbour %6.1f us' % (name, b, t), flush=True)


if __name__ == '__main__':
    main()
