'''Micro-benchmark of danet_center: one-launch vs two-launch form at the step's shapes.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.load_package()
from danet_amd import ops, _lib

def run(B, T, D, one, inl, outl):
    _lib.set_option('center_one', one)
    x = torch.randn(B, T, D, device='cuda') if inl == 0 else torch.randn(T, B, D, device='cuda')
    ldo = (D + 3) // 4 * 4
    out = torch.empty(T * B * ldo, device='cuda')
    for _ in range(5):
        ops.center(x, B, T, D, inl, D, out, outl, ldo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.center(x, B, T, D, inl, D, out, outl, ldo)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3

for (B, T, D, inl, outl) in [(32, 128, 129, 0, 1), (32, 128, 600, 1, 0), (32, 128, 600, 0, 1), (16, 128, 129, 0, 1)]:
    print(B, T, D, inl, outl, 'two-launch %.1f us' % run(B, T, D, 0, inl, outl), 'one-launch %.1f us' % run(B, T, D, 1, inl, outl))
