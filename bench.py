#!/usr/bin/env python
'''
Benchmark of the DANet hot path on MI355X -- contract in the task brief.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`)

A "step" = one full train step (front-end -> BiLSTM encoder -> attractor
estimator -> separator -> PIT loss -> backward -> gradient all-reduce ->
value clip + Adam) on one synthetic batch per GPU.  Workload = BASELINE.json
configs[1]: TIMIT-shaped synthetic 8 kHz 2-speaker mixtures, FFT 256 / stride
64 (129 bins), 128 frames, 3 x 300 BiLSTM, E=20, BATCH_SIZE=32 per GPU, anchor
estimator (A=6), dot-softmax separator.  Inputs (complex spectra) are resident
in HBM before the timed region.  Metric: mixture-seconds/s, with
mixture-seconds per step per GPU = B*T*FFT_STRIDE/SMPRATE = 32.768.

Prints ONE JSON line on rank 0.
'''
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBS = 8000.0


def setup_hparams(args):
    from danet_amd.hparams import hparams
    hparams.reset()
    hparams.load(dict(
        BATCH_SIZE=args.batch, MAX_N_SIGNAL=2, FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000,
        EMBED_SIZE=20, NUM_LSTM_LAYERS=args.layers, LSTM_HDIM=args.hdim, NUM_ANCHOR=6,
        MAX_TRAIN_LEN=args.frames, ENCODER_TYPE='bilstm-orig',
        TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
        SEPARATOR_TYPE='dot-softmax-orig', OPTIMIZER_TYPE='adam'))
    hparams.digest()
    return hparams


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


def cpu_baseline(hp, params_np, sample_b, n_steps=3):
    '''the oracle's torch-CPU float32 restatement of the same train step
    (per-timestep loop like tf.scan), timed on the host cores.'''
    from oracle import torch_ref as R
    from danet_amd import datasets
    from oracle import danet_oracle as O
    # tiny per-timestep matmuls stop scaling (and thrash) beyond a few threads
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    C, T = hp.MAX_N_SIGNAL, hp.MAX_TRAIN_LEN
    waves = datasets.synth_waves(4242, sample_b * C, T)
    w = O.fft_window(hp.FFT_SIZE)
    spec = np.stack([O.stft(x, w, hp.FFT_SIZE, hp.FFT_STRIDE) for x in waves])
    src = torch.tensor(spec.reshape(sample_b, C, T, hp.FEATURE_SIZE))
    cfg = dict(H=hp.LSTM_HDIM, L=hp.NUM_LSTM_LAYERS, E=hp.EMBED_SIZE, C=C, A=hp.NUM_ANCHOR,
               train_est='anchor', infer_est='anchor', separator='dot-softmax-orig')
    tp = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in params_np.items()}
    m = {k: torch.zeros_like(v) for k, v in tp.items()}
    v = {k: torch.zeros_like(x) for k, x in tp.items()}
    times = []
    budget_s, t_start = 20.0, time.time()
    for t in range(1, n_steps + 2):
        t0 = time.time()
        for k in tp:
            tp[k].grad = None
        R.model_forward(src, tp, cfg)['loss'].backward()
        R.tf_adam_step_(tp, {k: tp[k].grad for k in tp}, m, v, t, hp.LR, clip=hp.GRAD_CLIP_THRES)
        times.append(time.time() - t0)
        print('[bench] cpu_baseline step %d: %.2f s' % (t, times[-1]), file=sys.stderr, flush=True)
        if len(times) >= 2 and time.time() - t_start > budget_s:
            break
    n_steps = len(times) - 1
    dt = float(np.mean(times[1:]))
    mix_s = sample_b * T * hp.FFT_STRIDE / hp.SMPRATE
    # one more step on a single thread (SURVEY 8d asks for a 1-thread row beside it)
    torch.set_num_threads(1)
    t0 = time.time()
    for k in tp:
        tp[k].grad = None
    R.model_forward(src, tp, cfg)['loss'].backward()
    R.tf_adam_step_(tp, {k: tp[k].grad for k in tp}, m, v, n_steps + 2, hp.LR, clip=hp.GRAD_CLIP_THRES)
    dt1 = time.time() - t0
    torch.set_num_threads(cores)
    print('[bench] cpu_baseline single-thread step: %.2f s' % dt1, file=sys.stderr, flush=True)
    return dict(value=mix_s / dt, unit='mixture-seconds/s', cores=cores, kind='port',
                single_thread_value=mix_s / dt1,
                sample='%d of %d mixtures/step, same T/F/L/H, %d timed train steps (%.2f s each), '
                       'torch-CPU fp32 restatement of the reference (TF1 unavailable)'
                       % (sample_b, hp.BATCH_SIZE, n_steps, dt))


def pmc_traffic_bytes(kernel):
    '''HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/run_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate runs, KB units).
    Correction per MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports half of
    a 16-B/lane coalesced read stream, so reads = 2*FETCH_SIZE (calibrated here on
    adam_clip_kernel: 2*52.7 MB vs 110.5 MB algorithmic); WRITE_SIZE as is.
    None if no PMC summary was collected for this kernel (cfg 2 only).'''
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    for k, v in d.items():
        if k.replace('void ', '').startswith(kernel) and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            return int((2.0 * v['FETCH_SIZE']['per_launch'] + v['WRITE_SIZE']['per_launch']) * 1024)
    return None


def mask_mse_vs_oracle(hp, model, src, n_check=2):
    '''SDR-proxy: MSE between the HIP path's masks and the float64 oracle's on
    the first `n_check` mixtures of the batch (mixtures are independent).'''
    from oracle import torch_ref as R
    with torch.no_grad():
        out = model.forward(src)
    masks = (out['sep_pwr'] / out['mix_pwr'][:, None].clamp_min(1e-30))[:n_check].double().cpu()
    cfg = dict(H=hp.LSTM_HDIM, L=hp.NUM_LSTM_LAYERS, E=hp.EMBED_SIZE, C=hp.MAX_N_SIGNAL,
               A=hp.NUM_ANCHOR, train_est='anchor', infer_est='anchor',
               separator='dot-softmax-orig')
    tp = {k: torch.tensor(v, dtype=torch.float64) for k, v in model.param_dict().items()}
    with torch.no_grad():
        r = R.model_forward(src[:n_check].cpu().to(torch.complex128), tp, cfg)
    ref = r['masks'].permute(0, 3, 1, 2)
    return float(((masks - ref) ** 2).mean()), float((masks - ref).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--hdim', type=int, default=300)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=32)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('DANET_FORCE_DIST') == '1'   # 1-rank RCCL smoke
    graft.load_package()
    from danet_amd import _lib, ops
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        # the side streams must exist before the RCCL communicator (ops.prepare_streams)
        ops.prepare_streams(device)
        torch.distributed.init_process_group('nccl', device_id=device)
    from danet_amd.model import Model
    hp = setup_hparams(args)
    batches = make_batches(hp, rank, 4, device)
    model = Model('bench', device=device, seed=1337).build()
    log = lambda *a: print('[bench r%d]' % rank, *a, file=sys.stderr, flush=True)
    log('built: %d params' % model.parameter_count())

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # one initialisation step outside the W warmup steps: HIP code objects are loaded and the
    # allocator pools are grown on first use (a 0.15 s one-off that must not land in the
    # timed region when the caller asks for W = 0)
    model.train_step(batches[0])
    for i in range(args.warmup):
        model.train_step(batches[i % len(batches)])
    barrier()
    log('warmup done')
    # timed region: HIP events only around the recurrent kernels (the roofline kernel) and
    # only on every 4th step -- a timing event costs ~10 us of stream time on this runtime
    # (60 pairs per step were 4 % of the step, 6 pairs per step still 3 %)
    _lib.profile_start(only=('lstm_fwd', 'lstm_bwd'))
    t0 = time.perf_counter()
    for i in range(args.steps):
        _lib.profile_enable(i % 4 == 0)
        model.train_step(batches[i % len(batches)])
    _lib.profile_enable(True)
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_stop()
    # per-entry-point breakdown: a separate, untimed pass with every call instrumented
    nb = min(10, args.steps)
    _lib.profile_start()
    for i in range(nb):
        model.train_step(batches[i % len(batches)])
    barrier()
    prof_all = _lib.profile_stop()
    ok = ops.lstm_status_ok()
    log('timed region: %.3f s for %d steps; lstm status ok=%s' % (dt, args.steps, ok))
    if use_dist:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert ok, 'persistent LSTM kernel reported a hand-off timeout'

    mix_s_per_step = hp.BATCH_SIZE * hp.MAX_TRAIN_LEN * hp.FFT_STRIDE / hp.SMPRATE
    value = world * mix_s_per_step * args.steps / dt

    if rank == 0:
        B, T, H, L = hp.BATCH_SIZE, hp.MAX_TRAIN_LEN, hp.LSTM_HDIM, hp.NUM_LSTM_LAYERS
        # algorithmic flops per launch: recurrent half of one BiLSTM layer
        lstm_flops = 2.0 * 2 * B * T * H * 4 * H
        kern = {}
        for label, (n, ms) in prof_all.items():
            kern[label] = dict(launches=n, total_ms=round(ms, 3), avg_us=round(1e3 * ms / n, 2))
        kern['_note'] = 'separate instrumented pass of %d steps after the timed region' % nb
        dom = max(('lstm_fwd', 'lstm_bwd'), key=lambda k: prof.get(k, (1, 0.0))[1])
        n, ms = prof[dom]
        achieved = lstm_flops / (ms / n * 1e-3) / 1e12
        # kernel symbol behind the label (csrc/lstm.hip): BPTT runs the reduce-scatter
        # kernel unless DANET_LSTM_BWD_RS=0 selects the all-gather one
        ksym = {'lstm_fwd': 'lstm_fwd_kernel',
                'lstm_bwd': 'lstm_bwd_kernel' if os.environ.get('DANET_LSTM_BWD_RS') == '0'
                else 'lstm_bwd_rs_kernel'}[dom]
        roofline = dict(kernel=ksym, bound='mfma', achieved=round(achieved, 3),
                        peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                        frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=pmc_traffic_bytes(ksym),
                        us_per_timestep=round(1e3 * ms / n / T, 3),
                        note='latency-bound recurrence: T dependent steps per launch; '
                             'see DESIGN.md for the step-latency model')
        # second-largest consumer: the fp32 MFMA GEMMs (all launches of one step together)
        F, E = hp.FFT_SIZE // 2 + 1, hp.EMBED_SIZE
        gemm_flops = 0.0
        for l in range(L):
            D = F if l == 0 else 2 * H
            gemm_flops += 2 * (2.0 * B * T * D * 4 * H) * 3      # gx, dWx, dX per direction
            gemm_flops += 2 * (2.0 * B * T * H * 4 * H)          # dWh per direction
        gemm_flops -= 2 * (2.0 * B * T * F * 4 * H)              # layer 0 needs no dX
        gemm_flops += 3 * 2.0 * B * T * 2 * H * F * E            # projection, dWout, dYc
        if 'gemm_f32' in prof_all:
            gn, gms = prof_all['gemm_f32']
            if 'gemm_f32_group' in prof_all:       # grouped launches
                gn, gms = gn + prof_all['gemm_f32_group'][0], gms + prof_all['gemm_f32_group'][1]
            gach = gemm_flops * nb / (gms * 1e-3) / 1e12
            roofline['gemm_f32'] = dict(kernel='gemm_f32_kernel', bound='mfma',
                                        achieved=round(gach, 2), peak=PEAK_F32_MFMA_TFLOPS,
                                        unit='TFLOP/s', frac=round(gach / PEAK_F32_MFMA_TFLOPS, 4),
                                        note='sum over the %d launches of a step, timed in-step '
                                             '(instrumented pass; some run concurrently)' % (gn // nb))
        res = dict(metric='mixture-seconds/s (train step)', value=round(value, 2),
                   unit='mixture-seconds/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(1e3 * dt / args.steps, 3), higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='cfg2: synthetic 8 kHz 2-spk, FFT 256/64 (129 bins), '
                                        'T=%d, %dx%d BiLSTM, E=20, anchor estimator (A=6), '
                                        'dot-softmax, B=%d/GPU' % (T, L, H, B),
                               global_batch=B * world, parallelism='dp%d' % world,
                               grad_allreduce_bytes=int(model._flat_grad.numel() * 4)),
                   roofline=roofline, kernels=kern)
        if world == 1:
            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            mse, mx = mask_mse_vs_oracle(hp, model, batches[0])
            log('mask mse vs oracle %.3e (max abs %.3e)' % (mse, mx))
            res['mask_mse_vs_oracle'] = mse
            res['mask_max_abs_err_vs_oracle'] = mx
            if not args.no_cpu_baseline:
                res['cpu_baseline'] = cpu_baseline(hp, model.param_dict(), args.cpu_sample)
    if use_dist:
        torch.distributed.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner to the C stdout of the process; flush it first so
        # that the JSON line is the LAST line of stdout
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
