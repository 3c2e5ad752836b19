'''
Optimizers (reference app/ozers.py:9-18): factories registered under the same
names ('sgd', 'adam') and called with the same keyword arguments.  They return
objects bound to the model's flat parameter / gradient buffers; the update is
one fused HIP kernel (value clip + TF1 Adam, main.py:359-363) or, for SGD, one
torch axpy.
'''
import math

import torch

from .hparams import hparams
from . import ops


class _FlatOptimizer(object):
    def __init__(self, learn_rate):
        self.learn_rate = learn_rate
        self.theta = self.grad = None

    def bind(self, theta, grad):
        self.theta, self.grad = theta, grad


class TfAdam(_FlatOptimizer):
    '''tf.train.AdamOptimizer(learning_rate) with TF defaults beta1=0.9,
    beta2=0.999, epsilon=1e-8: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    theta -= lr_t * m / (sqrt(v) + eps)  -- eps OUTSIDE the bias-corrected root,
    unlike torch.optim.Adam.'''
    def __init__(self, learn_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        super(TfAdam, self).__init__(learn_rate)
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.m = self.v = None

    def bind(self, theta, grad):
        super(TfAdam, self).bind(theta, grad)
        self.m = torch.zeros_like(theta)
        self.v = torch.zeros_like(theta)

    def step(self, t, lr, clip=None, grad_scale=1.0, zero_grad=False, ranges=None):
        '''ranges: [lo, hi) element ranges of the flat buffers to update (default: all) --
        the update is elementwise, so a step may be issued in pieces as gradients become final'''
        lr_t = lr * math.sqrt(1. - self.beta2 ** t) / (1. - self.beta1 ** t)
        for lo, hi in (ranges if ranges is not None else [(0, self.theta.numel())]):
            ops.adam_clip_step(self.theta[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi],
                               lr_t, self.beta1, self.beta2, self.epsilon, clip or 0.0,
                               grad_scale, zero_grad)


class TfSgd(_FlatOptimizer):
    '''tf.train.GradientDescentOptimizer'''
    def step(self, t, lr, clip=None, grad_scale=1.0, zero_grad=False, ranges=None):
        for lo, hi in (ranges if ranges is not None else [(0, self.theta.numel())]):
            g = self.grad[lo:hi] * grad_scale
            if clip:
                g = g.clamp_(-clip, clip)
            self.theta[lo:hi].add_(g, alpha=-lr)
            if zero_grad:
                self.grad[lo:hi].zero_()


@hparams.register_optimizer('sgd')
def sgd_ozer(learn_rate, lr_decay=None, lr_decay_epoch=2, **kwargs):
    return TfSgd(learn_rate)


@hparams.register_optimizer('adam')
def adam_ozer(learn_rate, lr_decay=None, lr_decay_epoch=2, **kwargs):
    return TfAdam(learn_rate, **kwargs)
