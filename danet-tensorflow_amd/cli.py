'''
Command-line driver: the reference's `main()` and train / valid / test / demo
loops (main.py:402-532, :551-750) over the MI355X-native Model.

Same flags (main.py:553-582): -n/--name -m/--mode -i/--input-pfile
-o/--output-pfile -c/--config-file -ne/--num-epoch -if/--input-file
-ds/--dataset -lr/--learn-rate -tl/--train-length -bs/--batch-size
--no-save-on-epoch --no-valid-on-epoch; modes train | valid | test | demo | debug.
Host-side Python like the reference (north_star: "host code stays Python").
Not carried over: TensorBoard summaries, matplotlib plots, the interactive
prompt mode (observability / UI, out of scope).  Checkpoints are .npz
files keyed by the reference's TF variable names.
'''
from __future__ import print_function
import argparse
import os
import sys
from collections import OrderedDict
from math import isnan

import numpy as np
import torch

from .hparams import hparams
from . import datasets  # noqa: F401  (registers datasets)
from . import dist
from . import feed
from . import utils
from .model import Model


# DANET_FEED_MODE=sync (or --sync-feed): the reference's literal loop (blocking upload, a host read of every metric
# every step) instead of the one-batch-ahead feed
SYNC_FEED = os.environ.get('DANET_FEED_MODE', 'ahead') == 'sync'      # (also: --sync-feed)


def _dict_format(di):
    return ' '.join('='.join((k, str(v))) for k, v in di.items())


def train_epoch(model, batches, out=sys.stdout, sync_feed=None):
    '''the batch loop of Model.train (main.py:413-436) over host batches [B*C, T, F]: crop to
    MAX_TRAIN_LEN (main.py:422-426), one train step per batch, a ':' per step, epoch means of the
    fetched metrics.  The host never waits for the device inside the loop: batches are staged,
    cropped and uploaded one step ahead (feed.BatchFeed), the per-step metrics stay on the device
    until the epoch mean is read (feed.StepReport).  sync_feed=True is the reference's literal
    form -- blocking upload, `float()` of every metric every step -- kept for the equality test
    and `bench.py --e2e --sync-feed`.  Returns (OrderedDict of epoch means, number of batches).
    LIFETIME of a batch tensor in the 'ahead' mode: it is a view of one of three reused device
    buffers and is overwritten by the upload of the batch after the next one -- anything kept
    beyond the step that consumes it (e.g. `model.debug_fetches['input']` under hparams.DEBUG) must
    be cloned by the caller; Model.train_step itself keeps nothing.'''
    if sync_feed is None:
        sync_feed = SYNC_FEED
    src = feed.BatchFeed(batches, model.device, hparams.MAX_TRAIN_LEN,
                         mode='sync' if sync_feed else None)
    report = feed.StepReport(flush_every=1 if sync_feed else 1024)
    for spectra in src:
        step_fetch = model.train_step(spectra)
        model.reset_state()
        out.write(':')
        out.flush()
        report.add(step_fetch)
    return report.mean(), report.n


def train(model, n_epoch, dataset, args, out=sys.stdout):
    '''Model.train (main.py:402-510)'''
    best_loss, best_loss_time = float('+inf'), 0
    model.set_learn_rate(hparams.LR)
    out.write('Set learning rate to %f\n' % hparams.LR)
    i_epoch = 0
    while i_epoch < n_epoch:
        cli_report, _n = train_epoch(model, dataset.epoch(
            'train', hparams.BATCH_SIZE * hparams.MAX_N_SIGNAL, shuffle=True), out,
            sync_feed=getattr(args, 'sync_feed', None))
        model.check_status()       # hand-off timeouts of the persistent kernels surface here at the latest
        # data parallel: every rank must take the SAME learning-rate and NaN decisions, so
        # they are taken on the rank-mean of the epoch metrics (NaN anywhere -> NaN everywhere)
        keys = list(cli_report.keys())
        for k, v in zip(keys, dist.allreduce_mean_scalars([cli_report[k] for k in keys],
                                                          model.device)):
            cli_report[k] = v
        best_loss, best_loss_time = lr_decay_update(model, cli_report['loss'], best_loss,
                                                    best_loss_time, out)
        if not args.no_save_on_epoch:
            if any(map(isnan, cli_report.values())):                             # main.py:462-476
                if i_epoch:
                    out.write('\nEpoch %d/%d got NAN values, restoring last checkpoint ... '
                              % (i_epoch + 1, n_epoch))
                    restore_checkpoint(model, 'saves/' + model.name + ('_e%d' % i_epoch))
                    out.write('done')
                    continue                  # redo this epoch from the restored parameters
                out.write('\nRun into NAN during 1st epoch, exiting ...')
                sys.exit(-1)
            if dist.rank() == 0:              # only the file write is rank 0's
                model.save_params('saves/' + model.name + ('_e%d' % (i_epoch + 1)))
            out.write('S')
        out.write('\nEpoch %d/%d %s\n' % (i_epoch + 1, n_epoch, _dict_format(cli_report)))
        out.flush()
        i_epoch += 1
        if args.no_valid_on_epoch:
            continue
        rep = evaluate(model, dataset, 'valid', out, sync_feed=getattr(args, 'sync_feed', None))
        out.write('\nValid  %d/%d %s\n' % (i_epoch, n_epoch, _dict_format(rep)))
        out.flush()


def lr_decay_update(model, epoch_loss, best_loss, best_loss_time, out=sys.stdout):
    '''learning-rate schedule of Model.train (main.py:439-459): 'adaptive' counts epochs
    without a new best loss, 'fixed' counts every epoch, None never decays; after
    NUM_EPOCH_PER_LR_DECAY counted epochs LR *= LR_DECAY.  Returns the updated
    (best_loss, best_loss_time).'''
    if hparams.LR_DECAY_TYPE == 'adaptive':                                      # main.py:439-451
        if epoch_loss < best_loss:
            best_loss, best_loss_time = epoch_loss, 0
        else:
            best_loss_time += 1
    elif hparams.LR_DECAY_TYPE == 'fixed':
        best_loss_time += 1
    elif hparams.LR_DECAY_TYPE is not None:
        raise ValueError('Unknown LR_DECAY_TYPE "%s"' % hparams.LR_DECAY_TYPE)
    if best_loss_time == hparams.NUM_EPOCH_PER_LR_DECAY:                         # main.py:453-459
        best_loss_time = 0
        old_lr = model.get_learn_rate()
        new_lr = old_lr * hparams.LR_DECAY
        model.set_learn_rate(new_lr)
        out.write('[LR %f -> %f]' % (old_lr, new_lr))
    return best_loss, best_loss_time


def restore_checkpoint(model, filename):
    '''NaN-restore (main.py:462-476) under data parallelism: rank 0 reads the file, every
    rank receives the parameters by broadcast, so the replicas stay identical.'''
    if dist.rank() == 0:
        model.load_params(filename)
    dist.broadcast_params_(model._flat)


def evaluate(model, dataset, subset, out=sys.stdout, sync_feed=None):
    '''validation / test sweep with valid_fetches (main.py:486-510, :512-532); fed like
    train_epoch (no crop: main.py:497-498)'''
    if sync_feed is None:
        sync_feed = SYNC_FEED
    src = feed.BatchFeed(dataset.epoch(subset, hparams.BATCH_SIZE * hparams.MAX_N_SIGNAL,
                                       shuffle=False), model.device, None,
                         mode='sync' if sync_feed else None)
    report = feed.StepReport(flush_every=1 if sync_feed else 1024)
    for spectra in src:
        report.add(model.valid_step(spectra))
        model.reset_state()
        out.write('.')
        out.flush()
    rep = report.mean()
    # a hand-off timeout in any step of the sweep invalidates its metrics: surface it here
    # (blocking; collective under data parallelism -- every rank reaches this line)
    model.check_status()
    return rep


def _draw_aligned_sources(dataset, shuffle=False):
    '''one data point of hparams.MAX_N_SIGNAL single-speaker spectra from the test subset,
    each zero-padded at a random position of the time axis (utils.random_zeropad) to the
    longest of them rounded up to a multiple of hparams.LENGTH_ALIGN -- main.py:662-672
    (demo) and :719-727 (debug).  Returns complex [C, T, F].'''
    src_signals = []
    for src_signals in dataset.epoch('test', hparams.MAX_N_SIGNAL, shuffle=shuffle):
        break
    max_len = max(map(len, src_signals[0]))
    max_len += (-max_len) % hparams.LENGTH_ALIGN                     # main.py:668 / :724
    return np.stack([utils.random_zeropad(x, max_len - len(x), axis=-2) for x in src_signals[0]])


def demo(model, args, dataset=None, out=sys.stdout):
    '''wav -> separated wavs (main.py:655-696).  Without `-if` the mixture is made from
    MAX_N_SIGNAL utterances of the dataset's test subset and written to demo.wav first
    (main.py:662-672).'''
    if args.input_file is None:
        if dataset is None:
            raise ValueError('demo mode without an input file needs a dataset')
        filename = 'demo.wav'
        src_signals = _draw_aligned_sources(dataset)
        feat = np.sum(src_signals, axis=0)                           # raw_mixture, main.py:673
        utils.save_wavfile(filename, feat)
        out.write('wrote %s\n' % filename)
    else:
        filename = args.input_file
        feat = utils.load_wavfile(args.input_file)                   # [T, F] complex
    x = torch.as_tensor(feat.astype(np.complex64)).to(model.device)[None]
    sep = model.infer(x)[0].cpu().numpy()                            # [C, T, F]
    model.check_status()          # never write wavs computed from a timed-out launch
    base, ext = os.path.splitext(filename)
    for i, s in enumerate(sep):
        fn = '%s_separated_%d%s' % (base, i + 1, ext)                # main.py:692-694
        utils.save_wavfile(fn, s)
        out.write('wrote %s\n' % fn)


def debug(model, dataset, out=sys.stdout):
    '''dump the debug fetches to debug/debug_data.npz (main.py:717-737 writes a .mat):
    ONE data point (BATCH_SIZE is forced to 1, main.py:623-629) of MAX_N_SIGNAL test
    utterances, drawn shuffled, aligned and randomly zero-padded (main.py:719-727)'''
    input_ = _draw_aligned_sources(dataset, shuffle=True)[None]      # [1, C, T, F]
    fetch = model.debug_fetch(torch.as_tensor(input_.astype(np.complex64)).to(model.device))
    model.check_status()
    os.makedirs('debug', exist_ok=True)
    np.savez('debug/debug_data.npz',
             **{k: v.detach().cpu().numpy() for k, v in fetch.items() if torch.is_tensor(v)})
    out.write('wrote debug/debug_data.npz: %s\n' % ', '.join(sorted(fetch)))


def build_parser():
    p = argparse.ArgumentParser(description='MI355X-native Deep Attractor Network')
    p.add_argument('-n', '--name', default='UnnamedExperiment')
    p.add_argument('-m', '--mode', default='train',
                   help='train | valid | test | demo | debug')
    p.add_argument('-i', '--input-pfile', help='path to input model parameter file')
    p.add_argument('-o', '--output-pfile', help='path to output model parameters file')
    p.add_argument('-c', '--config-file', help='JSON hyperparameter file (keys of default.json)')
    p.add_argument('-ne', '--num-epoch', type=int, default=10)
    p.add_argument('-if', '--input-file', help='input WAV file for "demo" mode')
    p.add_argument('-ds', '--dataset', help='choose dataset to use, overrides hparams.DATASET_TYPE')
    p.add_argument('-lr', '--learn-rate', type=float, help='overrides hparams.LR')
    p.add_argument('-tl', '--train-length', type=int, help='overrides hparams.MAX_TRAIN_LEN')
    p.add_argument('-bs', '--batch-size', type=int, help='overrides hparams.BATCH_SIZE')
    p.add_argument('--no-save-on-epoch', action='store_true')
    p.add_argument('--no-valid-on-epoch', action='store_true')
    p.add_argument('--sync-feed', action='store_true', default=None,
                   help='(not in the reference) blocking uploads + per-step metric reads')
    return p


def main(argv=None, out=sys.stdout):
    args = build_parser().parse_args(argv)
    if args.config_file is not None:                                  # main.py:587-592
        hparams.load_json(args.config_file)
    if args.learn_rate is not None:                                   # main.py:594-604
        hparams.LR = args.learn_rate
    if args.train_length is not None:
        hparams.MAX_TRAIN_LEN = args.train_length
    if args.dataset is not None:
        hparams.DATASET_TYPE = args.dataset
    if args.batch_size is not None:
        hparams.BATCH_SIZE = args.batch_size
    if args.mode in ('demo', 'debug'):
        hparams.BATCH_SIZE = 1                                        # main.py:623-627
        if args.mode == 'debug':
            hparams.DEBUG = True                                      # main.py:628-629
    hparams.digest()

    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist.init_from_env('nccl', device)
    np.random.seed(dist.shard_seed(1337))

    out.write('Preparing dataset "%s" ... ' % hparams.DATASET_TYPE)      # main.py:607-612
    dataset = hparams.get_dataset()()
    dataset.install_and_load()
    out.write('done\n')
    out.write('Building model ... ')
    model = Model(name=args.name, device=device).build()
    out.write('done (%d parameters)\n' % model.parameter_count())
    if args.input_pfile is not None:
        model.load_params(args.input_pfile)

    if args.mode == 'train':
        train(model, args.num_epoch, dataset, args, out)
        if args.output_pfile is not None and dist.rank() == 0:
            model.save_params(args.output_pfile)
    elif args.mode in ('valid', 'test'):
        rep = evaluate(model, dataset, args.mode, out)
        out.write('\n%s: %s\n' % (args.mode.capitalize(), _dict_format(rep)))
    elif args.mode == 'demo':
        demo(model, args, dataset, out)
    elif args.mode == 'debug':
        debug(model, dataset, out)
    else:
        raise ValueError('Unknown mode "%s"' % args.mode)
    return model
