#!/usr/bin/env python
'''
Benchmark of the DANet hot path on MI355X -- contract in the task brief.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg4|cfg4h600|cfg5|cfg5-kmeans]

`--gpus N` with N > 1 needs no prepared environment: when WORLD_SIZE is unset the script
re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per
GPU, RCCL over xGMI, rendezvous on 127.0.0.1); launched by torch.distributed.run directly it
reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  DANET_FORCE_DIST=1
takes the same spawn path with N = 1 (a 1-rank RCCL group).

Default workload (the driver's line) = BASELINE.json configs[1] ("cfg2"): a "step" is one full
train step (front-end -> BiLSTM encoder -> attractor estimator -> separator -> PIT loss ->
backward -> gradient all-reduce -> value clip + Adam) on one synthetic batch per GPU:
TIMIT-shaped synthetic 8 kHz 2-speaker mixtures, FFT 256 / stride 64 (129 bins), 128 frames,
3 x 300 BiLSTM, E=20, BATCH_SIZE=32 per GPU, anchor estimator (A=6), dot-softmax separator.
Inputs (complex spectra) are resident in HBM before the timed region.  Metric:
mixture-seconds/s, with mixture-seconds per step per GPU = B*T*FFT_STRIDE/SMPRATE = 32.768.

Other configs (BASELINE.json configs[3], [4]; not the driver's line):
  cfg4      3-speaker, E=40, 4 x 300 BiLSTM, truth-weighted training estimator, train step
  cfg4h600  the same at 600 units per direction ("4x600")
  cfg5      16 kHz, FFT 512 / stride 128 (257 bins), one 10 s utterance (T=1251), B=1
            INFERENCE: danet_stft -> front-end -> 4 x 300 BiLSTM -> anchor estimator ->
            separator -> phase re-attach -> danet_istft; a step = one utterance
  cfg5-kmeans  the same with the k-means attractor estimator (extension)

Prints ONE JSON line on rank 0 (the last line of stdout).
'''
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
# fp32 products as six bf16 piece products (csrc/gemm_x6.hip): the dense bf16 peak / 6
PEAK_X6_TFLOPS = 2516.6 / 6
PEAK_HBM_GBS = 8000.0
DTYPE_NOTE = ('fp32 tensors, fp32 accumulation, fp32-accurate results everywhere.  The large products '
              '(projection, dYc, dX, weight gradients, hoisted LSTM input halves) are formed on the bf16 '
              'matrix cores from an EXACT three-way split of every fp32 operand (hi + mid + lo bf16 pieces; '
              'six of the nine piece products, the dropped ones < 2^-25 |a||b|): error against float64 at or '
              'below the exact-fp32 matrix instructions\' (tests/test_gpu_gemm_x6.py, parity block of this '
              'line; DESIGN.md 3.2).  DANET_GEMM_X6=0 runs the same step on v_mfma_f32_* only.')

CONFIGS = {
    'cfg2': dict(kind='train', batch=32, frames=128, layers=3, hdim=300,
                 hp=dict(MAX_N_SIGNAL=2, FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000, EMBED_SIZE=20,
                         NUM_ANCHOR=6, TRAIN_ESTIMATOR_METHOD='anchor',
                         INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig')),
    'cfg4': dict(kind='train', batch=32, frames=128, layers=4, hdim=300,
                 hp=dict(MAX_N_SIGNAL=3, FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000, EMBED_SIZE=40,
                         NUM_ANCHOR=6, TRAIN_ESTIMATOR_METHOD='truth-weighted',
                         INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig')),
    'cfg4h600': dict(kind='train', batch=32, frames=128, layers=4, hdim=600,
                     hp=dict(MAX_N_SIGNAL=3, FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000,
                             EMBED_SIZE=40, NUM_ANCHOR=6, TRAIN_ESTIMATOR_METHOD='truth-weighted',
                             INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig')),
    'cfg5': dict(kind='infer', batch=1, frames=1251, layers=4, hdim=300,
                 hp=dict(MAX_N_SIGNAL=2, FFT_SIZE=512, FFT_STRIDE=128, SMPRATE=16000,
                         EMBED_SIZE=20, NUM_ANCHOR=6, TRAIN_ESTIMATOR_METHOD='truth-weighted',
                         INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig')),
    'cfg5-kmeans': dict(kind='infer', batch=1, frames=1251, layers=4, hdim=300,
                        hp=dict(MAX_N_SIGNAL=2, FFT_SIZE=512, FFT_STRIDE=128, SMPRATE=16000,
                                EMBED_SIZE=20, NUM_ANCHOR=6,
                                TRAIN_ESTIMATOR_METHOD='truth-weighted',
                                INFER_ESTIMATOR_METHOD='kmeans',
                                SEPARATOR_TYPE='dot-softmax-orig')),
}


def log(*a):
    print('[bench r%s]' % os.environ.get('RANK', '0'), *a, file=sys.stderr, flush=True)


def setup_hparams(args, cfg):
    from danet_amd.hparams import hparams
    hparams.reset()
    kw = dict(cfg['hp'])
    kw.update(BATCH_SIZE=args.batch, NUM_LSTM_LAYERS=args.layers, LSTM_HDIM=args.hdim,
              MAX_TRAIN_LEN=args.frames, ENCODER_TYPE='bilstm-orig', OPTIMIZER_TYPE='adam')
    hparams.load(kw)
    hparams.digest()
    return hparams


def oracle_cfg(hp):
    return dict(H=hp.LSTM_HDIM, L=hp.NUM_LSTM_LAYERS, E=hp.EMBED_SIZE, C=hp.MAX_N_SIGNAL,
                A=hp.NUM_ANCHOR, train_est=hp.TRAIN_ESTIMATOR_METHOD,
                infer_est='anchor', separator=hp.SEPARATOR_TYPE)


def make_batches(hp, rank, n_batches, device):
    '''speech-shaped synthetic waveforms -> HIP STFT -> complex64 [B,C,T,F] in HBM'''
    from danet_amd import datasets, utils
    B, C, T = hp.BATCH_SIZE, hp.MAX_N_SIGNAL, hp.MAX_TRAIN_LEN
    out = []
    for i in range(n_batches):
        waves = datasets.synth_waves(1337 + rank + 1000 * i, B * C, T)
        spec = utils.stft(torch.as_tensor(waves).to(device))          # [B*C, T, F]
        assert spec.shape[1] == T, spec.shape
        out.append(spec.reshape(B, C, T, hp.FEATURE_SIZE).contiguous())
    return out


def make_host_batches(hp, rank, n_batches, device, extra_frames=32):
    '''what a dataset iterator yields (app/datasets/*.py): host numpy complex64 [B*C, T+extra, F]
    single-speaker spectra, longer than MAX_TRAIN_LEN so that the loop's random crop
    (main.py:422-426) is exercised'''
    from danet_amd import datasets, utils
    B, C, T = hp.BATCH_SIZE, hp.MAX_N_SIGNAL, hp.MAX_TRAIN_LEN + extra_frames
    out = []
    for i in range(n_batches):
        waves = datasets.synth_waves(7331 + rank + 1000 * i, B * C, T)
        out.append(utils.stft(torch.as_tensor(waves).to(device)).cpu().numpy())
    return out


def run_e2e(args, hp, model, device, rank, use_dist, barrier, sync_feed=False):
    '''the drop-in train loop (cli.train_epoch = main.py:413-436) over HOST-resident batches:
    staging, crop, upload and the metric reads are inside the timed region'''
    import io
    import random
    from danet_amd import cli
    host = make_host_batches(hp, rank, 4, device)
    random.seed(1337 + rank)

    def epoch(n):
        for i in range(n):
            yield (host[i % len(host)],)
    # untimed: the same settle + warmup steps as the HBM-resident region (the runtime's one-off
    # 30-70 ms reaction shows up a few steps after the first synchronisation of a NEW kind of loop
    # -- tools/feed_trace.py: always inside the first ten steps, never again -- and must not land
    # in a 30-step timed region; an epoch of real training has thousands of steps)
    n_untimed = int(os.environ.get('DANET_BENCH_SETTLE_STEPS', '8')) + max(args.warmup, 4)
    cli.train_epoch(model, epoch(n_untimed), io.StringIO(), sync_feed=sync_feed)
    barrier()
    # one "epoch" of at least 2K steps (its end-of-epoch metric read is inside); >= 150 so that one of the
    # occasional 5-8 ms hiccups of a fresh loop moves the mean by < 2 %
    n_timed = max(2 * args.steps, 150)
    t0 = time.perf_counter()
    rep, n = cli.train_epoch(model, epoch(n_timed), io.StringIO(), sync_feed=sync_feed)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, device, use_dist) / n_timed * args.steps
    assert n == n_timed and np.isfinite(rep['loss']), (n, rep)
    return dt, host[0].nbytes, rep, n_untimed, n_timed


def host_info():
    return dict(os_cpu_count=os.cpu_count(),
                sched_affinity=len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None)


def _cpu_baseline_rows(hpd, cfg, params_np, sample_b, rows, q=None):
    '''rows: [(threads, n_timed_steps, budget_s)] -> [(threads, s/step, n)].  Self-contained
    (runs in a child process for the all-cores row): hpd = plain dict of hparams values.'''
    import types
    graft.load_package()
    from oracle import torch_ref as R
    from oracle import danet_oracle as O
    from danet_amd import datasets
    from danet_amd.hparams import hparams
    hparams.load({k: v for k, v in hpd.items() if k in ('FFT_SIZE', 'FFT_STRIDE', 'SMPRATE')})
    hparams.digest()
    hp = types.SimpleNamespace(**hpd)
    C, T = hp.MAX_N_SIGNAL, hp.MAX_TRAIN_LEN
    waves = datasets.synth_waves(4242, sample_b * C, T)
    w = O.fft_window(hp.FFT_SIZE)
    spec = np.stack([O.stft(x, w, hp.FFT_SIZE, hp.FFT_STRIDE) for x in waves])
    src = torch.tensor(spec.reshape(sample_b, C, T, hp.FFT_SIZE // 2 + 1))
    tp = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in params_np.items()}
    m = {k: torch.zeros_like(v) for k, v in tp.items()}
    v = {k: torch.zeros_like(x) for k, x in tp.items()}
    tstep = [0]

    def one_step():
        tstep[0] += 1
        t0 = time.time()
        for k in tp:
            tp[k].grad = None
        R.model_forward(src, tp, cfg)['loss'].backward()
        R.tf_adam_step_(tp, {k: tp[k].grad for k in tp}, m, v, tstep[0], hp.LR,
                        clip=hp.GRAD_CLIP_THRES)
        return time.time() - t0

    out = []
    for threads, n, budget_s in rows:
        torch.set_num_threads(threads)
        one_step()                               # untimed first step at this thread count
        times, t_start = [], time.time()
        for _ in range(n):
            times.append(one_step())
            if time.time() - t_start > budget_s:
                break
        log('cpu_baseline %d threads: %s s/step' % (threads, ['%.2f' % t for t in times]))
        out.append((threads, float(np.mean(times)), len(times)))
    if q is not None:
        q.put(out)
    return out


def x6_vs_exact(hp, model, src, mask_f32_floor, n_check=4):
    '''the SAME (trained) parameters evaluated with every product on the bf16 matrix cores (the
    default: six bf16 piece products per fp32 product, csrc/gemm_x6.hip and the recurrent half of the
    fused forward kernel) and with the exact-fp32 matrix instructions only (DANET_GEMM_X6=0 +
    DANET_LSTM_FWD_FUSED=0): the two must agree far inside the parity bar'''
    from danet_amd import ops, _lib
    B, E = hp.BATCH_SIZE, hp.EMBED_SIZE
    act = 0 if hp.SEPARATOR_TYPE == 'dot-softmax-orig' else 1

    def fetch():
        with torch.no_grad():
            out = model.forward(src)
            _, masks = ops.SeparateFn.apply(out['mix_pwr'], out['attrs'],
                                            out['embed'].reshape(B, -1, E), act, True)
        return dict(embed=out['embed'][:n_check].double(), attrs=out['attrs'][:n_check].double(),
                    masks=masks[:n_check].double())
    a = fetch()
    x6, fused = ops.GEMM_X6, _lib.get_option('lstm_fwd_fused')
    ops.GEMM_X6 = 0
    _lib.set_option('lstm_fwd_fused', 0)
    try:
        b = fetch()
    finally:
        ops.GEMM_X6 = x6
        _lib.set_option('lstm_fwd_fused', fused)
    rep = {k: float((a[k] - b[k]).abs().max() / max(float(b[k].abs().max()), 1e-30)) for k in a}
    floor = max(1e-4, 2.0 * float(mask_f32_floor))
    rep['ok'] = bool(rep['embed'] <= 5e-6 and rep['attrs'] <= 5e-6 and rep['masks'] <= floor)
    rep['rule'] = ('max|x6 - exact| / max|exact| at the final parameters: embed, attrs <= 5e-6 (what the products '
                   'themselves differ by); masks <= max(1e-4, 2 * err(f32 oracle, f64)) = %.2e -- two fp32 '
                   'evaluations of the same sharp softmax differ by about the sum of their own distances to '
                   'float64 (tests/test_gpu_trained_parity.py: 9.3e-5 exact, 9.5e-5 x6, 1.1e-4 between them at '
                   '200 steps), so the masks cannot be held to the embedding\'s bar' % floor)
    return rep


def cpu_baseline(hp, params_np, sample_b, n_steps=3, brief=False):
    '''the oracle's torch-CPU float32 restatement of the same train step (per-timestep loop
    like tf.scan), timed on the host cores at 16, 32 and 64 threads plus a single-thread row
    (each row a few timed steps under its own time budget).  `value` / `cores` = the best row.
    A row on ALL cores of a 256-thread host is not attempted any more: rounds 2-3 measured that
    torch's intra-op pool needs more than 60 s per step there (the per-timestep products are
    [32 x 900] x [900 x 1200]: they stop scaling long before that).'''
    ncpu = os.cpu_count() or 1
    hpd = dict(MAX_N_SIGNAL=hp.MAX_N_SIGNAL, MAX_TRAIN_LEN=hp.MAX_TRAIN_LEN, FFT_SIZE=hp.FFT_SIZE,
               FFT_STRIDE=hp.FFT_STRIDE, SMPRATE=hp.SMPRATE, LR=hp.LR,
               GRAD_CLIP_THRES=hp.GRAD_CLIP_THRES)
    cfg = oracle_cfg(hp)
    mix_s = sample_b * hp.MAX_TRAIN_LEN * hp.FFT_STRIDE / hp.SMPRATE
    want = [(t, n_steps if t == 16 else 2, 10.0 if t == 16 else 6.0) for t in (16, 32, 64) if t <= ncpu]
    if brief:                       # a sub-record of the default run: one row
        want = [(min(16, ncpu), 1, 8.0)]
    if not want:
        want = [(ncpu, n_steps, 10.0)]
    rows = _cpu_baseline_rows(hpd, cfg, params_np, sample_b, want + [(1, 1, 1.0)])
    multi, (_, dt1, _) = rows[:-1], rows[-1]
    cores, dt, n = min(multi, key=lambda r: r[1])
    torch.set_num_threads(min(ncpu, 16))
    return dict(value=mix_s / dt, unit='mixture-seconds/s', cores=cores, kind='port',
                host_cpu_count=ncpu, single_thread_value=mix_s / dt1,
                rows=[dict(threads=t, value=mix_s / d, s_per_step=round(d, 3), timed_steps=k)
                      for t, d, k in multi],
                all_cores_note='not attempted: > 60 s per step at %d threads in rounds 2-3' % ncpu
                if ncpu > 64 else None,
                sample='%d of %d mixtures/step, same T/F/L/H, %d timed train steps (%.2f s each) at '
                       '%d threads (best of the rows), torch-CPU fp32 restatement of the reference '
                       '(TF1 unavailable)' % (sample_b, hp.BATCH_SIZE, n, dt, cores))


def pmc_traffic(kernel):
    '''HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/run_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate runs, KB units).  Correction
    per MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports half of a 16-B/lane coalesced
    read stream, so reads = 2*FETCH_SIZE (calibrated on adam_clip_kernel); WRITE_SIZE as is.
    The record carries the summary's tag and git revision; a summary without this kernel
    symbol yields None (the number is not carried over to a different kernel).'''
    pdir = os.path.join(ROOT, 'profiles')
    cands = sorted(f for f in os.listdir(pdir) if f.endswith('pmc_summary.json')) if os.path.isdir(pdir) else []
    for name in reversed(cands):
        d = json.load(open(os.path.join(pdir, name)))
        meta = d.get('_meta', {})
        for k, v in d.items():
            if k.startswith('_') or not isinstance(v, dict):
                continue
            if k.replace('void ', '').startswith(kernel + '<') or k.replace('void ', '') == kernel:
                if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
                    b = int((2.0 * v['FETCH_SIZE']['per_launch'] + v['WRITE_SIZE']['per_launch']) * 1024)
                    return b, dict(file='profiles/' + name, git=meta.get('git'), symbol=k,
                                   workload=meta.get('workload', 'cfg2'))
    return None, None


def parity_vs_oracle(hp, model, src, n_check=4):
    '''SDR-proxy + parity gate at the parameters the benchmark ends with (TRAINED for as
    many steps as the run took): masks / embedding / attractors / separated magnitudes of
    the first `n_check` mixtures (mixtures are independent) against the float64 oracle,
    with the float32 oracle's own distance to it as the noise floor (oracle/parity.py):
    ok = err(HIP, f64) <= max(1e-4, 2 * err(f32 oracle, f64)) for every tensor.'''
    from oracle import parity as P
    from danet_amd import ops
    B, E = hp.BATCH_SIZE, hp.EMBED_SIZE
    act = 0 if hp.SEPARATOR_TYPE == 'dot-softmax-orig' else 1
    with torch.no_grad():
        out = model.forward(src)
        _, masks = ops.SeparateFn.apply(out['mix_pwr'], out['attrs'],
                                        out['embed'].reshape(B, -1, E), act, True)
    n = n_check
    got = dict(embed=out['embed'][:n].cpu().numpy(), attrs=out['attrs'][:n].cpu().numpy(),
               masks=masks[:n].cpu().numpy(), sep_pwr=out['sep_pwr'][:n].cpu().numpy(),
               perm_idx=out['perm_idx'][:n].cpu().numpy())
    rep = P.parity_report(got, src[:n].cpu().numpy(), model.param_dict(), oracle_cfg(hp))
    # the kernels train_step ACTUALLY runs for separator + loss are the fused pair
    # (ops.SeparatePitFn); the tensors above come from the unfused kernels (the fused ones keep
    # masks / magnitudes in registers).  Same arithmetic in the same order: loss, SNR and the
    # permutation of all B mixtures must agree to the bit, and the n checked permutations with
    # the float64 oracle's.
    if model.fuse_heads and getattr(model.separator, 'ACT', None) is not None:
        with torch.no_grad():
            fo = model.forward(src, fuse_heads=True)
        snr_f, snr_u = float(fo['SNR']), float(out['SNR'])
        l_f, l_u = float(fo['loss']), float(out['loss'])
        fused = dict(loss_bit_equal=bool(torch.equal(fo['loss'], out['loss'])),      # (informative:
                     loss_rel_diff=abs(l_f - l_u) / max(abs(l_u), 1e-30),            # bit-equal at C = 2,
                     snr_rel_diff=abs(snr_f - snr_u) / max(abs(snr_u), 1e-30),       # another order of
                     perm_idx_equal=bool(torch.equal(fo['perm_idx'], out['perm_idx'])),   # the chunk sums
                     loss=l_f, snr=snr_f)                                            # at C = 3)
        fused['ok'] = (fused['loss_rel_diff'] <= 1e-6 and fused['snr_rel_diff'] <= 1e-5 and
                       fused['perm_idx_equal'])
        rep['fused_heads'] = fused
        rep['ok'] = bool(rep['ok'] and fused['ok'])
    return rep, rep['masks']['hip_vs_f64']['mse']


def _fill_parity(res, rep, mse, model):
    log('parity at the final parameters (%d train steps): %s' % (model.step_count, json.dumps(rep)))
    res['train_steps_before_mask_check'] = model.step_count
    res['mask_mse_vs_oracle'] = mse
    res['mask_max_abs_err_vs_oracle'] = rep['masks']['hip_vs_f64']['max_rel']
    res['mask_err_f32_oracle'] = rep['masks']['f32_vs_f64']['max_rel']
    res['mask_rms_rel_err_vs_oracle'] = rep['masks']['hip_vs_f64']['rms_rel']
    res['parity_ok'] = rep['ok']
    res['parity'] = dict(
        rule='err(HIP,f64) <= max(1e-4, 2*err(f32 oracle,f64)); err = max|a-b|/max|b|; '
             '4 mixtures of batch 0 at the final (trained) parameters; logits = embed.attr^T formed in float64 from each '
             'path\'s own embedding and attractors, relative to max|logit| (bar 1e-5): what the softmax amplifies; fused separator+loss '
             'kernels (the ones train_step runs): permutation equal to the unfused ones, loss to 1e-6, SNR to 1e-5',
        **{k: dict(hip=rep[k]['hip_vs_f64']['max_rel'], f32=rep[k]['f32_vs_f64']['max_rel'],
                   hip_rms=rep[k]['hip_vs_f64']['rms_rel'], f32_rms=rep[k]['f32_vs_f64']['rms_rel'],
                   ok=rep[k]['ok']) for k in ('embed', 'attrs', 'logits', 'masks', 'sep_pwr') if k in rep},
        perm_idx_equal=rep['perm_idx_equal'], fused_heads=rep.get('fused_heads'))


# functional test of the N > 1 path on a 1-GPU box (tests/test_gpu_dist.py): every rank uses
# device 0 and the group is gloo (RCCL refuses two ranks on one device).  The line it prints is
# marked and is NOT a measurement.
SHARED_GPU_TEST = os.environ.get('DANET_BENCH_TEST_SHARED_GPU') == '1'


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args):
    '''plain `python bench.py --gpus N` (no torchrun environment): become the launcher'''
    if 'WORLD_SIZE' in os.environ:
        return
    if args.gpus == 1 and os.environ.get('DANET_FORCE_DIST') != '1':
        return
    n_vis = torch.cuda.device_count()
    if SHARED_GPU_TEST:
        n_vis = args.gpus
    if n_vis < args.gpus:
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible' % (args.gpus, n_vis))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log('spawning: %s' % ' '.join(cmd))
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg2', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int)
    ap.add_argument('--frames', type=int)
    ap.add_argument('--layers', type=int)
    ap.add_argument('--hdim', type=int)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-check', action='store_true',
                    help='diagnostic sweeps only: skip the oracle comparison (parity_ok = null)')
    ap.add_argument('--step-times', action='store_true',
                    help='diagnostic: one HIP event per timed step, per-step GPU times to stderr')
    ap.add_argument('--cpu-sample', type=int)
    ap.add_argument('--no-e2e', action='store_true',
                    help='skip the second timed region (the train loop over host-resident batches)')
    ap.add_argument('--allreduce-schedule', choices=['auto', '0', 'tail', '1'], default=None,
                    help="gradient reduction schedule under data parallelism (Model.grad_schedule; "
                         "default 'auto': starts as '0' = ONE all-reduce per step and decides between '0' and "
                         "'tail' from its own measured all-reduce / step times, dist.choose_schedule)")
    ap.add_argument('--no-also', action='store_true',
                    help='default cfg2 run on one GPU: skip the short cfg4h600 / cfg5 sub-records (`also`)')
    ap.add_argument('--no-schedules', action='store_true',
                    help='--gpus N > 1: skip the passes that time the other gradient-reduction schedules')
    args = ap.parse_args()
    maybe_spawn(args)
    cfg = CONFIGS[args.config]

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if SHARED_GPU_TEST else int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = 'WORLD_SIZE' in os.environ
    graft.load_package()
    from danet_amd import _lib, ops
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # the side streams must exist before the RCCL communicator (ops.prepare_streams)
        ops.prepare_streams(device)
        if SHARED_GPU_TEST:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=device)
        assert torch.distributed.get_world_size() == args.gpus, \
            'RCCL group has %d ranks, --gpus %d' % (torch.distributed.get_world_size(), args.gpus)
    explicit_shape = any(getattr(args, k) is not None for k in ('batch', 'frames', 'layers', 'hdim'))
    for k in ('batch', 'frames', 'layers', 'hdim'):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    hp = setup_hparams(args, cfg)
    if cfg['kind'] == 'infer':
        res = run_infer(args, cfg, hp, device, rank, world, use_dist)
    else:
        res = run_train(args, cfg, hp, device, rank, world, use_dist)
    # The driver runs ONE command (this default).  BASELINE.json's other single-GPU workloads --
    # configs[3] as written (4 x 600, 3-spk, truth-weighted: cfg4h600) and configs[4] (10 s inference:
    # cfg5) -- ride along as short sub-records so that they are driver-observed too; cfg 2 stays
    # the headline `value`.  Same code paths as `--config cfg4h600 / cfg5`, fewer steps, one CPU row.
    # (diagnostic invocations -- --no-cpu-baseline / --no-parity-check, as the profiling scripts use -- skip them)
    if (args.config == 'cfg2' and world == 1 and not use_dist and not args.no_also and not explicit_shape
            and not args.no_cpu_baseline and not args.no_parity_check and res is not None):
        res['also'] = {}
        for name in ('cfg4h600', 'cfg5'):
            t_a = time.perf_counter()
            res['also'][name] = run_also(name, args, device)
            res['also'][name]['wall_s'] = round(time.perf_counter() - t_a, 1)
        res['also_note'] = ('short driver-visible passes of BASELINE configs[3] (as written) and configs[4]; '
                            'full-length runs: python bench.py --config cfg4h600 | cfg5')
    if use_dist:
        torch.distributed.destroy_process_group()
    if rank == 0:
        if args.gpus > 1:
            assert res.get('rccl_ranks') == args.gpus, (res.get('rccl_ranks'), args.gpus)
        # RCCL writes a version banner to the C stdout of the process; flush it first so
        # that the JSON line is the LAST line of stdout
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        if SHARED_GPU_TEST:
            res['test_mode'] = 'ranks share ONE GPU over gloo: functional test of the N > 1 path, not a measurement'
        print(json.dumps(res), flush=True)
        if res.get('parity_ok') is False:
            # the line above is still the record; a parity failure fails the run
            log('PARITY FAILED: %s' % json.dumps(res.get('parity')))
            sys.exit(4)


def run_also(name, args, device):
    '''one short pass of another BASELINE workload inside the default run (single GPU): the same
    run_train / run_infer as `--config name`, K = 8 timed steps, parity gate on, CPU baseline =
    one 16-thread row'''
    import copy
    from danet_amd import ops
    cfg = CONFIGS[name]
    a = copy.copy(args)
    a.config, a.steps, a.warmup, a.no_e2e, a.brief = name, 8, 2, True, True
    a.allreduce_schedule, a.cpu_sample, a.step_times = None, None, False
    for k in ('batch', 'frames', 'layers', 'hdim'):
        setattr(a, k, cfg[k])
    torch.cuda.synchronize(device)
    ops.drop_packs(device)                   # the previous model's operand-layout weight copies
    torch.cuda.empty_cache()
    hp = setup_hparams(a, cfg)
    full = (run_infer if cfg['kind'] == 'infer' else run_train)(a, cfg, hp, device, 0, 1, False)
    keep = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config', 'roofline',
            'parity_ok', 'parity', 'mask_max_abs_err_vs_oracle', 'mask_err_f32_oracle', 'cpu_baseline',
            'untimed_steps_total', 'train_steps_before_mask_check')
    out = {k: full[k] for k in keep if k in full}
    if isinstance(out.get('roofline'), dict):
        out['roofline'] = {k: v for k, v in out['roofline'].items() if k not in ('note', 'fused')}
    return out


def make_barrier(use_dist):
    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    return barrier


def max_over_ranks(dt, device, use_dist):
    if not use_dist:
        return dt
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    return float(tt.item())


def kernels_table(prof_all, nb):
    kern = {}
    for label, (n, ms) in prof_all.items():
        kern[label] = dict(launches=n, total_ms=round(ms, 3), avg_us=round(1e3 * ms / n, 2))
    kern['_note'] = 'separate instrumented pass of %d steps after the timed region' % nb
    return kern


def run_train(args, cfg, hp, device, rank, world, use_dist):
    from danet_amd import _lib, ops
    from danet_amd.model import Model
    batches = make_batches(hp, rank, 4, device)
    model = Model('bench', device=device, seed=1337, grad_schedule=args.allreduce_schedule).build()
    log('built: %d params' % model.parameter_count())
    barrier = make_barrier(use_dist)

    # one initialisation step outside the W warmup steps: HIP code objects are loaded and the
    # allocator pools are grown on first use (a 0.15 s one-off that must not land in the
    # timed region when the caller asks for W = 0)
    model.train_step(batches[0])
    # ... and so must the runtime's one-off reaction to the host running far ahead of the GPU for
    # the first time.  The host enqueues a step in ~1.2 ms, the GPU runs it in 3.4 ms; the first
    # time the host gets 8-9 steps ahead (it then blocks for 6-16 ms in a launch), the recurrent
    # kernel that is running at that moment takes 13-19 ms instead of 0.4 (tools/catch_stall.py:
    # one `lstm_fwd` / `lstm_bwd` call, always 3-4 steps after a synchronisation, once per
    # process, never under rocprofv3 where the host is slower).  It hit one benchmark run in five
    # -- with K = 20 that is +1.2 ms per step.  A FIXED number (every rank must issue the same
    # collectives) of un-synchronised untimed steps takes it here: 0 of 90 runs afterwards.
    # Round 3: Model.train_step bounds the host's run-ahead (ops.MAX_STEPS_IN_FLIGHT steps),
    # so that situation no longer arises and the settle phase is 8 steps (was 128); the record
    # carries every untimed step (`init_steps`, `settle_steps`, `warmup`).
    settle = int(os.environ.get('DANET_BENCH_SETTLE_STEPS', '8'))
    _lib.prepare_timing()        # the timing events of the timed region exist before the untimed steps
    for i in range(settle):
        model.train_step(batches[i % len(batches)])
    torch.cuda.synchronize(device)
    for i in range(args.warmup):
        model.train_step(batches[i % len(batches)])
    barrier()
    log('warmup done')
    # timed region: HIP events only around the recurrent kernels (the roofline kernel) and
    # only on every 4th step -- a timing event costs ~10 us of stream time on this runtime
    # (60 pairs per step were 4 % of the step, 6 pairs per step still 3 %)
    _lib.profile_start(only=('lstm_fwd', 'lstm_bwd'))
    step_ev = []
    if getattr(args, 'step_times', False):
        step_ev.append(torch.cuda.Event(enable_timing=True)); step_ev[-1].record()
    t0 = time.perf_counter()
    host_ms = []
    for i in range(args.steps):
        _lib.profile_enable(i % 4 == 0 and os.environ.get('DANET_BENCH_EVENTS', '1') != '0')
        th = time.perf_counter()
        model.train_step(batches[i % len(batches)])
        if step_ev:
            step_ev.append(torch.cuda.Event(enable_timing=True)); step_ev[-1].record()
            host_ms.append(1e3 * (time.perf_counter() - th))
    _lib.profile_enable(True)
    barrier()
    dt = time.perf_counter() - t0
    if step_ev:
        log('per-step GPU ms: ' + ' '.join('%.2f' % step_ev[i].elapsed_time(step_ev[i + 1])
                                           for i in range(len(step_ev) - 1)))
        log('per-step host enqueue ms: ' + ' '.join('%.2f' % h for h in host_ms))
        log('allocator: %s' % {k: v for k, v in torch.cuda.memory_stats().items()
                               if k in ('num_alloc_retries', 'num_device_alloc', 'num_device_free',
                                        'reserved_bytes.all.peak')})
    prof = _lib.profile_stop()
    prof_in_region = bool(prof)
    # per-entry-point breakdown: a separate, untimed pass with every call instrumented
    nb = min(10, args.steps)
    _lib.profile_start()
    for i in range(nb):
        model.train_step(batches[i % len(batches)])
    barrier()
    prof_all = _lib.profile_stop()
    ok = ops.lstm_status_ok()
    log('timed region: %.3f s for %d steps; lstm status ok=%s' % (dt, args.steps, ok))
    dt = max_over_ranks(dt, device, use_dist)
    assert ok, 'persistent LSTM kernel reported a hand-off timeout'

    # the gradient all-reduce on its own (HIP events on the current stream, which the RCCL
    # stream is joined to): what an un-overlapped reduction adds to a step
    allreduce_ms = None
    if use_dist:
        g = model._flat_grad
        for _ in range(3):
            torch.distributed.all_reduce(g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.distributed.all_reduce(g)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = e0.elapsed_time(e1) / 10
        g.zero_()

    mix_s_per_step = hp.BATCH_SIZE * hp.MAX_TRAIN_LEN * hp.FFT_STRIDE / hp.SMPRATE
    value = world * mix_s_per_step * args.steps / dt
    # second timed region (not `value`): the same K steps through the drop-in train loop with
    # batches that start in HOST memory (every rank runs it: the steps contain the all-reduce)
    e2e = None
    if not args.no_e2e:
        dt_e, nbytes, rep_e, n_unt, n_tim = run_e2e(args, hp, model, device, rank, use_dist, barrier)
        dt_s, _, _, _, _ = run_e2e(args, hp, model, device, rank, use_dist, barrier, sync_feed=True)
        assert ops.lstm_status_ok()
        e2e = dict(loop='cli.train_epoch (main.py:413-436): host numpy batches [B*C, T+32, F] -> crop '
                        '-> pinned staging -> async upload one batch ahead (side stream) -> train_step; metrics '
                        'read once per epoch',
                   ms_per_step=round(1e3 * dt_e / args.steps, 3),
                   value=round(world * mix_s_per_step * args.steps / dt_e, 2),
                   frac_of_resident=round(dt / dt_e, 4), untimed_steps=n_unt, timed_steps=n_tim,
                   host_batch_bytes=int(nbytes),
                   uploaded_bytes_per_step=int(nbytes * hp.MAX_TRAIN_LEN // (hp.MAX_TRAIN_LEN + 32)),
                   sync_feed_ms_per_step=round(1e3 * dt_s / args.steps, 3),
                   sync_feed_note='the reference\'s literal loop: blocking upload from pageable '
                                  'memory + float() of loss / SNR every step',
                   epoch_mean_loss=rep_e['loss'])
        log('e2e train loop: %.3f ms/step (resident %.3f, sync feed %.3f)'
            % (e2e['ms_per_step'], 1e3 * dt / args.steps, e2e['sync_feed_ms_per_step']))
    # N > 1: ONE run settles the reduction schedule.  `value` above is the schedule the model was
    # built with (default 'auto': '0' = one all-reduce after backward, or 'tail' if the model's own measurement of the
    # collective against its step says so -- `grad_allreduce_decision`); the same K steps are now timed with
    # the other schedules and with NO reduction at all (what the step costs without communication),
    # every rank taking part in every pass.  exposed = schedule - no_reduction.
    sched_ms = None
    if use_dist and world > 1 and not args.no_schedules:
        from danet_amd import dist as D

        def time_schedule(sched, reduce=True):
            m = Model('bench_' + sched, device=device, seed=1337, grad_schedule=sched).build()
            orig = D.allreduce_grads_
            if not reduce:
                D.allreduce_grads_ = lambda g: 1.0 / D.world_size()
            try:
                for i in range(4 + max(args.warmup, 2)):
                    m.train_step(batches[i % len(batches)])
                barrier()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    m.train_step(batches[i % len(batches)])
                barrier()
                d = max_over_ranks(time.perf_counter() - t0, device, use_dist)
            finally:
                D.allreduce_grads_ = orig
            assert ops.lstm_status_ok()
            return round(1e3 * d / args.steps, 3)
        sched_ms = {model.grad_schedule: round(1e3 * dt / args.steps, 3)}
        for sc in ('0', 'tail', '1'):
            if sc not in sched_ms:
                sched_ms[sc] = time_schedule(sc)
        none = time_schedule('0', reduce=False)
        sched_ms = dict(ms_per_step=sched_ms, no_reduction_ms_per_step=none,
                        exposed_comm_ms={k: round(v - none, 3) for k, v in sched_ms.items()},
                        note="'0' one all-reduce after backward | 'tail' everything outside the bottom encoder "
                             "layer reduced under that layer's weight-gradient GEMMs | '1' per-layer buckets; "
                             "no_reduction = the same steps with the collective skipped (replicas drift: timing only)")
    if rank != 0:
        return None

    B, T, H, L = hp.BATCH_SIZE, hp.MAX_TRAIN_LEN, hp.LSTM_HDIM, hp.NUM_LSTM_LAYERS
    C, E, F = hp.MAX_N_SIGNAL, hp.EMBED_SIZE, hp.FFT_SIZE // 2 + 1
    # algorithmic flops per launch: the recurrent half of one BiLSTM layer, plus --
    # where the fused kernels run (csrc/lstm.hip) -- the input projection (forward) / the
    # weight gradients (BPTT) that the same launch computes; averaged over the L layers
    L_ = _lib.load()
    Ds = [F if l == 0 else 2 * H for l in range(L)]
    fwd_fused = [L_.danet_lstm_fwd_fused_supported(T, B, H, 2, D) == 1 for D in Ds]
    bwd_fused = [False] * L          # (weight gradients are never fused into BPTT: EXPERIMENTS.md)
    rec = 2.0 * 2 * B * T * H * 4 * H
    flops = dict(
        lstm_fwd=sum(rec + (2.0 * 2 * B * T * D * 4 * H if f else 0.0) for D, f in zip(Ds, fwd_fused)) / L,
        lstm_bwd=sum(rec + (2.0 * 2 * B * T * (D + H) * 4 * H if f else 0.0) for D, f in zip(Ds, bwd_fused)) / L)
    if not prof_in_region:     # DANET_BENCH_EVENTS=0 (diagnostic): fall back to the separate pass
        prof = prof_all
    dom = max(('lstm_fwd', 'lstm_bwd'), key=lambda k: prof.get(k, (1, 0.0))[1])
    n, ms = prof[dom]
    achieved = flops[dom] / (ms / n * 1e-3) / 1e12
    # kernel symbol behind the label (csrc/lstm.hip)
    ksym = {'lstm_fwd': 'lstm_fwd_fx_kernel' if any(fwd_fused) else 'lstm_fwd_kernel',
            'lstm_bwd': 'lstm_bwd_rs_kernel'}[dom]
    traffic, tsrc = pmc_traffic(ksym)
    if tsrc is not None and tsrc.get('workload', 'cfg2') != args.config:
        traffic, tsrc = None, None
    roofline = dict(kernel=ksym, bound='mfma', achieved=round(achieved, 3),
                    peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    traffic=traffic, traffic_source=tsrc,
                    us_per_timestep=round(1e3 * ms / n / T, 3),
                    flops_per_launch=flops[dom],
                    fused=dict(forward_input_projection=fwd_fused, bptt_weight_gradients=bwd_fused),
                    events_in_timed_region=prof_in_region,
                    note='latency-bound recurrence: T dependent steps per launch; see DESIGN.md '
                         'for the step-latency model.  HIP events bracket the two recurrent '
                         'entry points on every 4th step INSIDE the timed region')
    for other in ('lstm_fwd', 'lstm_bwd'):
        if other in prof:
            on, oms = prof[other]
            roofline[other + '_us'] = round(1e3 * oms / on, 1)
            roofline[other + '_tflops'] = round(flops[other] / (oms / on * 1e-3) / 1e12, 2)
    # second-largest consumer: the fp32 MFMA GEMMs (all launches of one step together)
    gemm_flops = 0.0
    for l in range(L):
        D = Ds[l]
        per = 2 * (2.0 * B * T * D * 4 * H)                  # one x-sized product, both directions
        gemm_flops += (0 if fwd_fused[l] else per)           # gx
        gemm_flops += (per if l > 0 else 0)                  # dX (layer 0 needs none)
        if not bwd_fused[l]:
            gemm_flops += per + 2 * (2.0 * B * T * H * 4 * H)    # dWx, dWh
    gemm_flops += 3 * 2.0 * B * T * 2 * H * F * E            # projection, dWout, dYc
    gtimers = [k for k in ('gemm_f32', 'gemm_f32_group', 'gemm_x6', 'gemm_x6_tn_group') if k in prof_all]
    if gtimers:
        gn = sum(prof_all[k][0] for k in gtimers)
        gms = sum(prof_all[k][1] for k in gtimers)
        x6_ms = sum(prof_all[k][1] for k in gtimers if k.startswith('gemm_x6'))
        gach = gemm_flops * nb / (gms * 1e-3) / 1e12
        # priced against the peak of the instruction mix that ran: time-weighted between the exact
        # fp32 matrix instructions and six bf16 piece products per fp32 product
        gpeak = (PEAK_X6_TFLOPS * x6_ms + PEAK_F32_MFMA_TFLOPS * (gms - x6_ms)) / gms
        roofline['gemm_f32'] = dict(kernel='gemm_x6_*_kernel + gemm_f32_*_kernel' if x6_ms else 'gemm_f32_kernel',
                                    bound='mfma', achieved=round(gach, 2), peak=round(gpeak, 1),
                                    unit='TFLOP/s (fp32-equivalent)', frac=round(gach / gpeak, 4),
                                    x6_time_share=round(x6_ms / gms, 3),
                                    note='sum over the %d launches of a step, timed in-step '
                                         '(instrumented pass; some run concurrently); fp32 operands '
                                         'and fp32-accurate results throughout: the x6 kernels form '
                                         'each product from six bf16 piece products '
                                         '(csrc/gemm_x6.hip)' % (gn // nb))
    est = hp.TRAIN_ESTIMATOR_METHOD
    res = dict(metric='mixture-seconds/s (train step)', value=round(value, 2),
               unit='mixture-seconds/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(1e3 * dt / args.steps, 3), higher_is_better=True,
               scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
               dtype_note=DTYPE_NOTE,
               config=dict(workload='%s: synthetic 8 kHz %d-spk, FFT %d/%d (%d bins), T=%d, '
                                    '%dx%d BiLSTM, E=%d, %s estimator%s, %s, B=%d/GPU'
                                    % (args.config, C, hp.FFT_SIZE, hp.FFT_STRIDE, F, T, L, H, E, est,
                                       ' (A=%d)' % hp.NUM_ANCHOR if est == 'anchor' else '',
                                       hp.SEPARATOR_TYPE.replace('-orig', ''), B),
                           global_batch=B * world, parallelism='dp%d' % world,
                           grad_allreduce_bytes=int(model._grad_store.numel() * 4),
                           grad_allreduce_schedule=model.grad_schedule,
                           grad_allreduce_decision=model.schedule_decision,
                           collectives_per_step=model.collectives_per_step()),
               init_steps=1, settle_steps=settle,
               untimed_steps_total=1 + settle + args.warmup,
               max_steps_in_flight=ops.MAX_STEPS_IN_FLIGHT,
               rccl_ranks=(torch.distributed.get_world_size() if use_dist else 0),
               allreduce_ms_standalone=(round(allreduce_ms, 4) if allreduce_ms is not None else None),
               roofline=roofline, kernels=kernels_table(prof_all, nb), host=host_info())
    if e2e is not None:
        res['e2e'] = e2e
    if world == 1:
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        rep, mse = (None, None) if args.no_parity_check else parity_vs_oracle(hp, model, batches[0], 4)
        if rep is None:
            res['parity_ok'] = None
        else:
            _fill_parity(res, rep, mse, model)
            if not getattr(args, 'brief', False):
                res['parity']['x6_vs_exact'] = x6_vs_exact(hp, model, batches[0],
                                                           rep['masks']['f32_vs_f64']['max_rel'])
                res['parity_ok'] = bool(res['parity_ok'] and res['parity']['x6_vs_exact']['ok'])
        if not args.no_cpu_baseline:
            sample = args.cpu_sample or (B if (H <= 300 and L <= 3) else max(2, B // 8))
            res['cpu_baseline'] = cpu_baseline(hp, model.param_dict(), sample,
                                               brief=getattr(args, 'brief', False))
    res['timed_region_note'] = (
        '`value` = %d steps = %.0f ms of GPU time between two barriers (the driver chooses K); the '
        'sturdier figure is e2e.ms_per_step, >= 150 steps through the drop-in train loop from host '
        'batches, which must agree with it' % (args.steps, 1e3 * dt))
    if sched_ms:
        res['schedules'] = sched_ms
    return res


def infer_parity(hp, model, X):
    '''cfg 5 parity gate: embedding / attractors / masks / separated magnitudes of the WHOLE
    utterance (B = 1, T = 1251) against the float64 torch-CPU restatement of the inference
    fetches (main.py:384-385, :685-690), with the float32 evaluation of the same restatement
    as the noise floor: ok = err(HIP, f64) <= max(1e-4, 2 * err(f32, f64)), err =
    max|a-b| / max|b|.  X: complex64 [T, F] on the device.'''
    from oracle import torch_ref as R
    from danet_amd import ops
    B, E = 1, hp.EMBED_SIZE
    act = 0 if hp.SEPARATOR_TYPE == 'dot-softmax-orig' else 1
    with torch.no_grad():
        sep = model.infer(X[None])                                  # the product call
        fe = ops.frontend(X[None, None].contiguous())               # ... and its intermediates
        emb = model.encoder(fe['mix_log'])
        attr = model.valid_estimator(emb, s_mix_pwr=fe['mix_pwr'])
        _, masks = ops.SeparateFn.apply(fe['mix_pwr'], attr, emb.reshape(B, -1, E), act, True)
    got = dict(embed=emb.cpu().numpy(), attrs=attr.cpu().numpy(), masks=masks.cpu().numpy(),
               sep_pwr=sep.abs().cpu().numpy())
    cfg = dict(oracle_cfg(hp), infer_est=hp.INFER_ESTIMATOR_METHOD,
               kmeans_iters=int(hp.KMEANS_ITERS), eps=float(hp.EPS))
    refs = {}
    for name, rt, ct in (('f64', torch.float64, torch.complex128), ('f32', torch.float32, torch.complex64)):
        tp = {k: torch.tensor(v, dtype=rt) for k, v in model.param_dict().items()}
        with torch.no_grad():
            r = R.infer_forward(X[None].cpu().to(ct), tp, cfg)
        refs[name] = {k: r[k].double().numpy() for k in got}

    def err(a, b):
        d = np.asarray(a, dtype=np.float64) - b
        return dict(max_rel=float(np.abs(d).max() / (np.abs(b).max() + 1e-300)),
                    rms_rel=float(np.sqrt((d * d).mean()) / (np.sqrt((b * b).mean()) + 1e-300)))
    rep, ok = {}, True
    for k in got:
        e_hip, e_f32 = err(got[k], refs['f64'][k]), err(refs['f32'][k], refs['f64'][k])
        bound = max(1e-4, 2.0 * e_f32['max_rel'])
        rep[k] = dict(hip=e_hip['max_rel'], f32=e_f32['max_rel'], hip_rms=e_hip['rms_rel'],
                      f32_rms=e_f32['rms_rel'], ok=bool(e_hip['max_rel'] <= bound))
        ok = ok and rep[k]['ok']
    d = got['masks'].astype(np.float64) - refs['f64']['masks']
    return ok, rep, float((d * d).mean())


def infer_cpu_baseline(hp, params_np, wave_np, n_steps=3):
    '''the same step (waveform -> STFT -> inference fetches -> iSTFT of every source) through the
    CPU restatements: numpy STFT / iSTFT (oracle/danet_oracle.py) + torch-CPU float32 model
    (oracle/torch_ref.py, per-timestep loop like tf.scan), on <= 16 host threads'''
    from oracle import torch_ref as R
    from oracle import danet_oracle as O
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 16)
    torch.set_num_threads(cores)
    w = O.fft_window(hp.FFT_SIZE)
    tp = {k: torch.tensor(v, dtype=torch.float32) for k, v in params_np.items()}
    cfg = dict(oracle_cfg(hp), infer_est=hp.INFER_ESTIMATOR_METHOD,
               kmeans_iters=int(hp.KMEANS_ITERS), eps=float(hp.EPS))

    def one():
        t0 = time.time()
        X = O.stft(wave_np, w, hp.FFT_SIZE, hp.FFT_STRIDE)
        with torch.no_grad():
            r = R.infer_forward(torch.tensor(X[None]), tp, cfg)
        for c in range(hp.MAX_N_SIGNAL):
            O.istft(r['sep'][0, c].numpy(), hp.FFT_STRIDE, w)
        return time.time() - t0
    one()
    times = [one() for _ in range(n_steps)]
    log('cpu_baseline (inference) %d threads: %s s/utterance' % (cores, ['%.2f' % t for t in times]))
    dt = float(np.mean(times))
    mix_s = hp.MAX_TRAIN_LEN * hp.FFT_STRIDE / hp.SMPRATE
    return dict(value=mix_s / dt, unit='mixture-seconds/s', cores=cores, kind='port',
                host_cpu_count=ncpu,
                sample='the whole %.1f s utterance, %d timed passes (%.2f s each) at %d threads: numpy '
                       'STFT/iSTFT + torch-CPU fp32 restatement of the reference\'s inference '
                       'fetches (TF1 unavailable)' % (mix_s, n_steps, dt, cores))


def run_infer(args, cfg, hp, device, rank, world, use_dist):
    '''cfg 5: the demo path (main.py:623-627, 655-696) on one long utterance per step:
    waveform in HBM -> danet_stft -> Model.infer -> danet_istft of every separated source.
    N > 1 = independent replicas (B = 1 does not shard).'''
    from danet_amd import _lib, ops, utils, datasets
    from danet_amd.model import Model
    assert hp.BATCH_SIZE == 1, 'the inference configs are B = 1 (main.py:623-624)'
    T, S, N = hp.MAX_TRAIN_LEN, hp.FFT_STRIDE, hp.FFT_SIZE
    Ls = (T - 1) * S
    rng = np.random.RandomState(1337 + rank)
    waves = []
    for i in range(4):
        w = sum(datasets.speech_shaped_wave(rng, Ls, hp.SMPRATE, phase=0.3 + 1.8 * c)
                for c in range(hp.MAX_N_SIGNAL))
        waves.append(torch.as_tensor(w.astype(np.float32)).to(device))
    model = Model('bench', device=device, seed=1337).build()
    wnd = torch.as_tensor(np.asarray(hp.FFT_WND)).to(device)
    log('built: %d params' % model.parameter_count())
    barrier = make_barrier(use_dist)

    def step(w):
        X = utils.stft(w)                                   # [T, F] complex64
        sep = model.infer(X[None])                          # [1, C, T, F]
        return ops.istft(sep[0].contiguous(), S, wnd)       # [C, T*S] float64

    y = step(waves[0])
    assert tuple(y.shape) == (hp.MAX_N_SIGNAL, T * S), y.shape
    settle = int(os.environ.get('DANET_BENCH_SETTLE_STEPS', '8'))                  # see run_train
    _lib.prepare_timing()
    for i in range(settle):
        step(waves[i % len(waves)])
    torch.cuda.synchronize(device)
    for i in range(args.warmup):
        step(waves[i % len(waves)])
    barrier()
    _lib.profile_start(only=('lstm_fwd',))
    t0 = time.perf_counter()
    for i in range(args.steps):
        _lib.profile_enable(i % 4 == 0)
        step(waves[i % len(waves)])
    _lib.profile_enable(True)
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_stop()
    nb = min(10, args.steps)
    _lib.profile_start()
    for i in range(nb):
        step(waves[i % len(waves)])
    barrier()
    prof_all = _lib.profile_stop()
    ok = ops.lstm_status_ok()
    dt = max_over_ranks(dt, device, use_dist)
    assert ok, 'persistent LSTM kernel reported a hand-off timeout'
    if rank != 0:
        return None
    H, L, F, E = hp.LSTM_HDIM, hp.NUM_LSTM_LAYERS, hp.FFT_SIZE // 2 + 1, hp.EMBED_SIZE
    mix_s = T * S / hp.SMPRATE
    n, ms = prof['lstm_fwd']
    per_launch_s = ms / n * 1e-3
    # B = 1 recurrence: the per-step product is a GEMV.  Algorithmic HBM bytes of one launch
    # (both directions): gx read + gates, cell, y written; Wh is read ONCE (it stays in LDS /
    # registers for all T steps).  A formulation that re-streams Wh every step would move
    # T * 2 * H*4H*4 bytes instead -- `wh_restream_equiv` prices our launch against that.
    alg_bytes = 2 * (T * (4 * H + 4 * H + H + H) * 4 + H * 4 * H * 4)
    restream = T * 2 * H * 4 * H * 4
    roofline = dict(kernel='lstm_fwd_kernel', bound='hbm',
                    achieved=round(alg_bytes / per_launch_s / 1e9, 2), peak=PEAK_HBM_GBS,
                    unit='GB/s', frac=round(alg_bytes / per_launch_s / 1e9 / PEAK_HBM_GBS, 5),
                    traffic=None, us_per_timestep=round(1e6 * per_launch_s / T, 3),
                    algorithmic_bytes_per_launch=alg_bytes,
                    wh_restream_equiv=dict(bytes=restream,
                                           GBps=round(restream / per_launch_s / 1e9, 1),
                                           frac_of_hbm_peak=round(restream / per_launch_s / 1e9 / PEAK_HBM_GBS, 4)),
                    mfma_frac=round(2.0 * 2 * T * H * 4 * H / per_launch_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 5),
                    events_in_timed_region=True,
                    note='B=1: T dependent GEMV steps per launch, latency-bound; weights stationary, '
                         'so the HBM fraction of the algorithmic bytes is tiny by design '
                         '(DESIGN.md 3.1); us_per_timestep is the meaningful figure')
    res = dict(metric='mixture-seconds/s (inference, demo path)', value=round(world * mix_s * args.steps / dt, 2),
                unit='mixture-seconds/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=round(1e3 * dt / args.steps, 3), higher_is_better=True,
                scaling='weak', vs_baseline=None, dtype='f32', data='synthetic', dtype_note=DTYPE_NOTE,
                config=dict(workload='%s: synthetic %d Hz %d-spk utterance of %.1f s, FFT %d/%d (%d bins), '
                                     'T=%d, %dx%d BiLSTM, E=%d, %s estimator, B=1: stft -> infer -> istft'
                                     % (args.config, hp.SMPRATE, hp.MAX_N_SIGNAL, mix_s, N, S, F, T, L, H,
                                        E, hp.INFER_ESTIMATOR_METHOD),
                            global_batch=world, parallelism='replicas%d' % world),
                init_steps=1, settle_steps=settle, untimed_steps_total=1 + settle + args.warmup,
                max_steps_in_flight=ops.MAX_STEPS_IN_FLIGHT,
                rccl_ranks=(torch.distributed.get_world_size() if use_dist else 0),
                roofline=roofline, kernels=kernels_table(prof_all, nb), host=host_info())
    if world == 1:
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        if args.no_parity_check:
            res['parity_ok'] = None
        else:
            ok, rep, mse = infer_parity(hp, model, utils.stft(waves[0]))
            log('parity (whole utterance): %s' % json.dumps(rep))
            res['parity_ok'] = ok
            res['mask_mse_vs_oracle'] = mse
            res['mask_max_abs_err_vs_oracle'] = rep['masks']['hip']
            res['mask_err_f32_oracle'] = rep['masks']['f32']
            res['parity'] = dict(
                rule='err(HIP,f64) <= max(1e-4, 2*err(f32 oracle,f64)); err = max|a-b|/max|b|; the '
                     'whole utterance (T=%d) at the initial parameters; oracle = float64 torch-CPU '
                     'restatement of main.py:384-385,685-690%s' % (
                         T, ' (k-means: restated extension, no reference behaviour)'
                         if hp.INFER_ESTIMATOR_METHOD == 'kmeans' else ''), **rep)
        if not args.no_cpu_baseline:
            res['cpu_baseline'] = infer_cpu_baseline(hp, model.param_dict(), waves[0].cpu().numpy(),
                                                     n_steps=1 if getattr(args, 'brief', False) else 3)
    return res


if __name__ == '__main__':
    main()
