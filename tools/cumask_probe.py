#!/usr/bin/env python
'''CU-masked HIP streams (hipExtStreamCreateWithCUMask): (1) which CUs does mask bit i enable,
(2) does confining the weight-gradient GEMM group to a CU subset remove the slow-down it causes
the BPTT kernel beside it?  (GPU box; build tools/csrc/libprobe.so first, see contention_probe.py.)

Findings (round 2): bit b = CU (b / 8) of XCD (b % 8), consecutive indices of an XCD rotate over
its 4 shader engines (unequal shares per engine dispatch badly: a group on 10 CUs per XCD took
875 us, on 12 CUs 472 us).  A CU-masked stream is a BLOCKING stream: it serialises with the NULL
stream (torch's default), so per-stream event timings look perfect (BPTT 338 us "beside" a 327 us
group) while the wall time is the sum.  Against a non-blocking stream it does overlap, but the
pair is slower than today's schedule (one group workgroup on every CU): wall 604 / 581 / 838 us
with the group on 96 / 128 / 160 CUs vs 501 us -- the BPTT kernel needs all its 152 workgroups
resident, and CUs that carry two group workgroups have no room for one.  Not adopted.'''
import collections
import ctypes
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
P = ctypes.CDLL(os.path.join(ROOT, 'tools', 'csrc', 'libprobe.so'))
P.probe_stream_create_cumask.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
P.probe_whereami.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    out = ctypes.c_void_p()
    rc = P.probe_stream_create_cumask(words, 8, ctypes.byref(out))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(out.value)


def where(stream, wgs=256):
    out = torch.zeros(wgs, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    P.probe_whereami(stream.cuda_stream, wgs, ptr(out), 200000)
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype('uint32')
    # HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    return collections.Counter(((int(x) >> 16) & 0xf, (int(x) >> 13) & 7, (int(x) >> 12) & 1, (int(x) >> 8) & 0xf)
                               for x in v)


def contention(dev):
    B, T, H, D = 32, 128, 300, 600
    torch.manual_seed(0)
    Wh = [torch.randn(H, 4 * H, device=dev) * (0.75 / H ** 0.5) for _ in range(2)]
    dy = torch.randn(T, B, 2 * H, device=dev)
    n = L.danet_lstm_workspace_bytes(T, B, H, 2)
    ws = torch.zeros(n, dtype=torch.uint8, device=dev)
    ypad = torch.empty(T + 2, B, 2 * H, device=dev)
    cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
    das = [torch.randn(T * B, 4 * H, device=dev) for _ in range(2)]
    gates = [torch.randn(T * B, 4 * H, device=dev) * 0.5 for _ in range(2)]
    main_s = torch.cuda.current_stream()
    _lib.check(L.danet_lstm_fwd(
        main_s.cuda_stream, T, B, H, 2, ptr(gates[0]), ptr(gates[1]), ptr(Wh[0]), ptr(Wh[1]),
        4 * H, ptr(ypad), 2 * H, ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
        ptr(ws), n, None, 0))
    torch.cuda.synchronize()
    x = torch.randn(T * B, D, device=dev)
    h = torch.randn(T * B, H, device=dev)
    dW = [torch.empty(D, 4 * H, device=dev) for _ in range(2)] + [torch.empty(H, 4 * H, device=dev) for _ in range(2)]
    da2 = [torch.randn(T * B, 4 * H, device=dev) for _ in range(2)]

    def bptt(stream):
        with torch.cuda.stream(stream):
            _lib.check(L.danet_lstm_bwd(
                stream.cuda_stream, T, B, H, 2, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
                ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]), ptr(das[0]), ptr(das[1]),
                ptr(ws), n, None))

    def group(stream, wgs):
        with torch.cuda.stream(stream):
            ops.gemm_group([(x, D, da2[0], 4 * H, dW[0], 4 * H, D, 4 * H, 0.0),
                            (x, D, da2[1], 4 * H, dW[1], 4 * H, D, 4 * H, 0.0),
                            (h, H, da2[0], 4 * H, dW[2], 4 * H, H, 4 * H, 0.0),
                            (h, H, da2[1], 4 * H, dW[3], 4 * H, H, 4 * H, 0.0)], T * B,
                           transA=True, max_workgroups=wgs)

    nA = int(os.environ.get('CUS_A', '160'))
    sA = masked_stream(range(0, nA))
    sB = masked_stream(range(nA, 256))
    side = torch.cuda.Stream()
    print('mask A (bits 0..%d): %d CUs;  mask B: %d CUs' % (nA - 1, len(where(sA)), len(where(sB))))
    # warm the clocks
    for _ in range(30):
        group(main_s, 512)
    torch.cuda.synchronize()
    Wx = torch.randn(D, 4 * H, device=dev)
    dxo = torch.empty(T * B, D, device=dev)

    def dx(stream):
        with torch.cuda.stream(stream):
            ops.gemm_kcat(das[0], 4 * H, Wx, 4 * H, 4 * H, das[1], 4 * H, Wx, 4 * H, 4 * H, dxo,
                          T * B, D, D, transB=True)

    def timed(fn_b, sb, fn_g, sg):
        tb, tg = [], []
        for it in range(7):
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            if sg is not None:
                ev[2].record(sg); fn_g(); ev[3].record(sg)
            if sb is not None:
                ev[0].record(sb); fn_b(); ev[1].record(sb)
            torch.cuda.synchronize()
            if sb is not None:
                tb.append(ev[0].elapsed_time(ev[1]) * 1e3)
            if sg is not None:
                tg.append(ev[2].elapsed_time(ev[3]) * 1e3)
        tb.sort(); tg.sort()
        return (tb[len(tb) // 2] if tb else 0.0), (tg[len(tg) // 2] if tg else 0.0)

    print('BPTT alone %.0f us; dX alone %.0f us; group alone (512 wgs) %.0f us' % (
        timed(lambda: bptt(main_s), main_s, None, None)[0], timed(lambda: dx(main_s), main_s, None, None)[0],
        timed(None, None, lambda: group(main_s, 512), main_s)[1]))
    print('today: BPTT + group(unmasked, 256 wgs): BPTT %.0f us, group %.0f us' % timed(
        lambda: bptt(main_s), main_s, lambda: group(side, 256), side))
    def wall(fn_b, fn_g, sg):
        ts = []
        for it in range(7):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            sg.wait_stream(main_s)
            fn_g()
            fn_b()
            main_s.wait_stream(sg)
            e1.record(main_s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    def small_kernels():
        t = torch.zeros(1024, device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(200):
            t.add_(1.0)
        e1.record(main_s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 200

    def wall2(sb, fn_b, sg, fn_g):
        ts = []
        for it in range(7):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            sg.wait_stream(main_s)
            if sb is not main_s:
                sb.wait_stream(main_s)
            fn_g()
            fn_b()
            main_s.wait_stream(sg)
            if sb is not main_s:
                main_s.wait_stream(sb)
            e1.record(main_s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    def tiny(stream):
        t = torch.zeros(1024, device=dev)
        with torch.cuda.stream(stream):
            for _ in range(20):
                t.add_(1.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(200):
                t.add_(1.0)
            e1.record(stream)
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 200

    nb_main = torch.cuda.Stream()      # non-blocking, like every torch side stream
    print('tiny kernels back to back: %.2f us each on the NULL stream, %.2f us on a non-blocking stream '
          '(CU-masked = blocking streams exist)' % (tiny(main_s), tiny(nb_main)))
    print('WALL {BPTT null stream || group unmasked side 256 wgs}: %.0f us' % wall2(main_s, lambda: bptt(main_s), side, lambda: group(side, 256)))
    for nB in (96, 128, 160, 192):
        sB = masked_stream(range(256 - nB, 256))
        print('group on %d masked CUs (%d wgs): WALL beside BPTT on the null stream %.0f us, on a non-blocking stream %.0f us; '
              'beside dX on a non-blocking stream %.0f us' % (
                  nB, 2 * nB, wall2(main_s, lambda: bptt(main_s), sB, lambda: group(sB, 2 * nB)),
                  wall2(nb_main, lambda: bptt(nb_main), sB, lambda: group(sB, 2 * nB)),
                  wall2(nb_main, lambda: dx(nb_main), sB, lambda: group(sB, 2 * nB))), flush=True)


def main():
    if '--contention' in sys.argv:
        return contention(torch.device('cuda'))

    dev = torch.device('cuda')
    torch.zeros(1, device=dev)
    print('unmasked: %d distinct (xcc, se, sh, cu)' % len(where(torch.cuda.current_stream())))
    for bits in ([0], [1], [8], [32], range(0, 8), range(0, 32), range(0, 256, 2)):
        c = where(masked_stream(list(bits)), 64)
        ks = sorted(c)
        print('mask bits %s -> %d CUs: %s' % (list(bits)[:10], len(c), ks[:12]))


if __name__ == '__main__':
    main()
