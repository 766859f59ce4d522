#!/usr/bin/env python
"""Headline benchmark: images/sec of the part-detector + spatial-model forward (ending in
argmax coordinates on the device) on 480x720x3 synthetic images, K=9, on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one pass of the whole path over one batch per GPU (BASELINE.json configs[1]: batch 64,
fp32) with inputs already resident in HBM.  Ranks shard by batch (images are independent);
the only collective is the all-gather of [B,2,9] int32 coordinates (main.py:573-574 -> RCCL).
Rank 0 prints ONE compact JSON line (<= 4 KB: it must survive the driver's 8 KB stdout tail) as its LAST line of stdout: the driver's
contract keys, `roofline`, `cpu_baseline` and `configs` = {name: {value, ms_per_step, dtype, frac, bound}} for every other configuration
measured in the same run; the verbose record of every configuration goes to `bench_detail.json` (gpurun_out/ when that directory exists,
else the working directory) and to stderr.  `roofline` (dominant
kernel = the hand-written channel GEMM `cgemm_split_kernel` of the frequency-domain conv4_fullres /
conv5 layers -- or, with the direct kernels selected, their MFMA implicit-GEMM launch -- timed live
with HIP events on its stream) and `cpu_baseline` (the CPU restatement timed on the host cores;
TensorFlow is unavailable).  Beside the headline (configs[1], fp32, batch 64) the line carries
`bf16_config2` (configs[2], default route), `bf16_config2_mfma` (configs[2] on the direct bf16 MFMA
kernels only, `conv9_fft=0`) and `config3_gb2048` (configs[3]: a fixed global batch of 2048 images
sharded over the ranks, strong scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import joint_cnn_mrf_amd  # noqa: E402,F401
from joint_cnn_mrf_amd import dist as jdist  # noqa: E402
from joint_cnn_mrf_amd import synth  # noqa: E402
from joint_cnn_mrf_amd.engine import Engine  # noqa: E402

# Algorithmic FLOPs (2*MAC) per image, SURVEY.md 8d / BASELINE.md section 2.
FLOPS_PD_SM = 413_188_758_480
# The dominant kernel is one instantiation of conv_igemm (9x9 taps, 4x32 pixel patch, 128-channel
# N-tile); per step it is launched twice: conv4_fullres (256->512) and conv5 (512->512), both on
# 60x90 maps.  Algorithmic FLOPs per image of each launch = 60*90 * Cout * 81*Cin * 2:
FLOPS_DOMINANT = {'conv4_fullres': 114_661_785_600, 'conv5': 229_323_571_200}
PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                            # HBM3E spec (6.3 TB/s is what a streaming copy achieves)
# Frequency-domain route (conv_fft.hip): 60x90 maps -> 64 x 96 circular transforms, 64 * 49 frequencies; the channel GEMM
# (cgemm_split.hip) is one complex [B x Cin] x [Cin x Cout] product per frequency on the 16-bit matrix cores with operands split into
# two 16-bit parts, three real products per real multiply: bf16 parts on bf16 handles, FP16 parts of spectra scaled by powers of two on
# fp32 handles (22 significant bits: fp32-class).
FFT_FREQS = 64 * 49
FLOPS_SM = 4_856_014_800                 # the 81 pairwise convolutions as the reference computes them
SM_ALGO_BYTES_PER_IMAGE = 14e6           # algorithmic HBM traffic of an FFT spatial model per image
GEMM_LAYERS = {'conv4_fullres': (256, 512), 'conv5': (512, 512)}


def gemm_mtile(dtype, b):
    """cgemm_split_mtile() of cgemm_split.hip: rows of the M tile the activation spectra are laid out for."""
    if dtype == 'bf16':
        return 256 if b > 128 else 128 if b > 64 else 64
    return 128 if b > 64 else 64


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE) KB, FETCH doubled per the gfx950
    correction in MI355X_MICROARCH.md).  PMC cannot be read from inside the timed run, so the
    value is the last profiled one for this (dtype, batch), else null."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
            rec = json.load(fh).get(key, {})
            return rec.get('traffic_bytes_per_launch'), rec.get('source')
    except OSError:
        return None, None


def step_traffic(key):
    """HBM bytes of ONE step (every kernel) of a configuration, from the committed counter passes (profiles/pmc_traffic.json: step_bytes)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
            rec = json.load(fh).get(key, {})
            return rec.get('step_bytes'), rec.get('step_source')
    except OSError:
        return None, None


def sm_traffic(dtype):
    """Measured HBM bytes per image of the spatial model's kernels (profiles/pmc_traffic.json: "sm_<dtype>")."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
            rec = json.load(fh).get('sm_' + dtype, {})
            return rec.get('bytes_per_image'), rec.get('source')
    except OSError:
        return None, None


def agreement_reference(params, x, torso, local_rank, use_sm):
    """The fp32 default engine's heat maps and coordinates on the resident images `x` (walked 64 at a time inside jcm_forward)."""
    eng = Engine(device=local_rank, precision='fp32').load_params(params)
    r = eng.forward(x, torso if use_sm else None, use_sm=use_sm, want_prob=True)
    torch.cuda.synchronize()
    eng.close()
    return r


def smi_sample():
    """Power / clock of GPU 0 from rocm-smi (best effort: None when the tool is missing or prints something unexpected)."""
    import subprocess
    try:
        txt = subprocess.run(['rocm-smi', '-d', '0', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=20).stdout
        rec = json.loads(txt[txt.index('{'):])
        card = rec[sorted(rec)[0]]
        out = {}
        import re
        for k, v in card.items():
            kl = k.lower()
            if 'power' in kl and 'w' in kl:
                out['power_w'] = float(str(v).split()[0])
            elif kl.startswith('sclk clock speed'):      # "sclk clock speed:": "(2100Mhz)"  (the "sclk clock level" key holds the level index, not MHz)
                m = re.search(r'([0-9.]+)\s*mhz', str(v).lower())
                if m:
                    out['sclk_mhz'] = float(m.group(1))
        return out or None
    except Exception:
        return None


def sustained(args, dtype, B, params, local_rank, dev, use_sm, seconds=10.0):
    """>= `seconds` of back-to-back steps (no host sync inside batches of 10 steps): shows whether the short timed region of the headline rides a
    boost clock.  rocm-smi power / clock samples are taken while the loop runs (after a third and at the end)."""
    eng = Engine(device=local_rank, precision=dtype).load_params(params)
    x, torso = resident_inputs(B, 0, dev)
    for _ in range(3):
        eng.forward(x, torso if use_sm else None, use_sm=use_sm, want_prob=False)
    torch.cuda.synchronize()
    samples, n, t0 = [], 0, time.perf_counter()
    while True:
        for _ in range(10):
            r = eng.forward(x, torso if use_sm else None, use_sm=use_sm, want_prob=False)
        n += 10
        el = time.perf_counter() - t0
        if (len(samples) == 0 and el > seconds / 3) or (len(samples) == 1 and el > seconds * 0.9):
            samples.append(smi_sample())          # (the GPU keeps working on the queued steps while rocm-smi runs)
        if el >= seconds:
            break
        if n % 50 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del r
    eng.close()
    return {'seconds': dt, 'steps': n, 'value': n * B / dt, 'unit': 'images/sec', 'ms_per_step': dt / n * 1e3, 'dtype': dtype, 'batch': B, 'smi': samples}


def cpu_baseline(params, reps=5, batches=(1, 8), budget_s=150.0):
    """The oracle's torch-CPU formulation (fp32) on the host cores: the stand-in for the reference's TF-CPU path, which cannot run
    here (no TensorFlow).  BASELINE.md section 3: B = 1 and B = 8, median of `reps` runs after one warm-up; bounded by `budget_s`
    (a batch size stops repeating once the budget is spent; at least one run each).  `value` = the B = 8 median."""
    from oracle import jcm_oracle_torch as T
    nmax = max(batches)
    x, torso = synth.make_images(nmax, seed=99), synth.make_torso(nmax, seed=98)
    T.forward(x[:1], torso[:1], params, dtype=torch.float32)          # warm-up (thread pools, oneDNN primitives)
    # torch's default of one thread per physical core is not the fastest setting on a many-core host (measured on the GPU box, 256 logical CPUs:
    # B=8 0.33 images/s at 128 threads, 0.44 at 64, 0.47 at 32): one B=1 run per candidate, the fastest is used and reported as `cores`
    n0, best = torch.get_num_threads(), None
    for n in sorted({n0, min(n0, 64), min(n0, 32)}, reverse=True):
        torch.set_num_threads(n)
        t0 = time.time()
        T.forward(x[:1], torso[:1], params, dtype=torch.float32)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, n)
    torch.set_num_threads(best[1])
    t_start, per = time.time(), {}
    for b in batches:
        times = []
        for _ in range(reps):
            t0 = time.time()
            T.forward(x[:b], torso[:b], params, dtype=torch.float32)
            times.append(time.time() - t0)
            if time.time() - t_start > budget_s * (0.3 if b == batches[0] and len(batches) > 1 else 1.0):
                break
        per['b%d' % b] = {'images_per_s': b / float(np.median(times)), 'median_s': float(np.median(times)), 'runs': len(times)}
    last = per['b%d' % batches[-1]]
    threads = torch.get_num_threads()
    torch.set_num_threads(n0)
    return {'value': last['images_per_s'], 'unit': 'images/sec', 'cores': threads, 'kind': 'port', **per,
            'sample': 'B=%s synthetic 480x720 images, median of <=%d runs each after 1 warm-up, full-size network PD+SM, fp32 torch-CPU/oneDNN '
                      'restatement (TensorFlow unavailable); value = B=%d; %d torch threads (fastest of the candidates tried), host has %d logical CPUs'
                      % ('/'.join(str(b) for b in batches), reps, batches[-1], threads, os.cpu_count())}


def resident_inputs(B, rank, dev):
    """B synthetic images + torso maps in HBM.  Large batches (a rank's share of configs[3]'s 2048) are generated in
    slices of 256 so the host never holds more than 1 GB of them."""
    x = torch.empty((B, 480, 720, 3), dtype=torch.float32, device=dev)
    torso = torch.empty((B, 60, 90, 1), dtype=torch.float32, device=dev)
    for b0 in range(0, B, 256):
        n = min(256, B - b0)
        x[b0:b0 + n] = torch.as_tensor(synth.make_images(n, seed=1234 + rank + 1000 * (b0 // 256)))
        torso[b0:b0 + n] = torch.as_tensor(synth.make_torso(n, seed=4321 + rank + 1000 * (b0 // 256)))
    return x, torso


def dist_on():
    """Is there a process group?  Every barrier / collective of this script runs when there is one -- also with ONE rank (torch.distributed.run
    --nproc-per-node 1, JCM_BENCH_FORCE_DIST=1: the nccl branch executed on a one-GPU box, tests/test_gpu_bench_ranks.py)."""
    return dist.is_available() and dist.is_initialized()


def run_config(args, dtype, B, params, world, rank, local_rank, dev, use_sm, f32_conv=None, config_name=None, micro_batch=None, conv9_fft=None, fft_single=None, fft_t16=None, agree=False):
    """Time `args.steps` steps of one (dtype, batch) configuration; returns the result dict on
    rank 0 (None elsewhere).  Timed region: barrier + synchronize on both sides, max over ranks.
    f32_conv='split16': the fp32 path with its stride-1 layers on the direct fp16x3 split kernels (conv_split.hip).
    B is the rank's batch per step; jcm_forward walks it in micro-batches (256 bf16 / 64 fp32 unless `micro_batch`)."""
    eng = Engine(device=local_rank, precision=dtype, f32_conv=f32_conv, micro_batch=micro_batch, conv9_fft=conv9_fft, fft_single=fft_single, fft_t16=fft_t16,
                 fft_fuse=args.fft_fuse).load_params(params)
    x, torso = resident_inputs(B, rank, dev)                                          # resident in HBM
    agree_ref = agreement_reference(params, x, torso, local_rank, use_sm) if agree and rank == 0 else None      # (outside every timed region)

    def step():
        r = eng.forward(x, torso if use_sm else None, use_sm=use_sm, want_prob=False)
        return jdist.allgather_coords(r['sm_coords' if use_sm else 'pd_coords'])

    # The per-launch HIP events of the roofline object come from a pool inside the library: one profiled step
    # before the warm-up creates them, so that the profiled region below only records (2 hipEventRecord per conv
    # launch, no event creation or destruction).
    eng.set_profile(True)
    step()
    eng.set_profile(False)
    for _ in range(args.warmup):
        step()

    def timed_region(profile):
        """EXACTLY args.steps steps between barrier + synchronize on both sides; returns (wall seconds, per-step HIP-event ms, coords)."""
        torch.cuda.synchronize()
        if dist_on():
            dist.barrier()
        torch.cuda.synchronize()
        if profile:
            eng.set_profile(True)          # recycles the warm-up record; nothing is allocated here
        # one event per step boundary on the launch stream (the engine enqueues on torch's current stream): the median step
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        for i in range(args.steps):
            evs[i].record()
            c = step()
        evs[args.steps].record()
        torch.cuda.synchronize()
        if dist_on():
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if profile:
            eng.set_profile(False)
        ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        if dist_on():
            t = torch.tensor([d], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d, ms, c

    # Two timed regions of the same K steps (round 6): the first WITHOUT the library's per-launch events -- `value` / `ms_per_step`, what a
    # caller of jcm_forward gets --, the second WITH them (two hipEventRecord around every conv layer, its GEMM and the spatial model): the roofline object's
    # launch durations, and `ms_per_step_profiled` beside the headline so that the cost of the instrumentation is on the line.
    dt, step_ms, coords = timed_region(False)
    dt_prof, step_ms_prof, _ = timed_region(True)
    assert coords.shape == (world * B, 2, 9)
    out = None
    if rank == 0 and args.layer_times:
        for scope, _k, _s, cin, cout, _l in synth.conv_scopes(args.debug):
            if cin != 3:
                ms, n = eng.profile_read(scope)     # destructive read: the roofline object below then sees no launches
                print('layer %-18s %8.3f ms/launch  (%d launches)' % (scope, ms / max(n, 1), n), file=sys.stderr)
    if rank == 0:
        tot_ms, tot_n, tot_flops = 0.0, 0, 0.0
        for scope, fl in FLOPS_DOMINANT.items():
            ms, n = eng.profile_read(scope)
            tot_ms, tot_n = tot_ms + ms, tot_n + n
            tot_flops += fl / (16 if args.debug else 1) * B * args.steps if n else 0.0     # n = steps x micro-batches launches cover steps x B images
        kname = eng.conv_kernel_name('conv5', min(B, micro_batch or (256 if dtype == 'bf16' else 64)), 60, 90) if not args.debug else 'debug'
        layer_ms = tot_ms / max(tot_n, 1)                      # average duration of a whole conv4_fullres / conv5 layer
        freq_domain = kname.startswith('conv_fft')
        mb = micro_batch or (256 if dtype == 'bf16' else 64)
        gemm = None
        if freq_domain:
            # wide 9x9 layers in the frequency domain: the dominant kernel is the channel GEMM cgemm_split_kernel (one complex matrix
            # product per frequency); its own executed bf16-MFMA FLOPs and algorithmic HBM bytes, its own HIP events
            # 16-bit parts per operand and real products per multiply: fp32 handles two scaled fp16 parts / three products; bf16 handles ONE scaled fp16
            # part / one product (default, "fft_single") or two bf16 parts / three products
            single = dtype == 'bf16' and fft_single is not False
            np_parts, nprod = (1, 1) if single else (2, 3)
            mrows = min(B, mb)
            mt = gemm_mtile(dtype, mrows)
            rows_p = -(-mrows // mt) * mt
            tot_ms, tot_n, tot_flops, tot_bytes, tot_flops32 = 0.0, 0, 0.0, 0.0, 0.0
            for scope, (cin, cout) in GEMM_LAYERS.items():
                ms, n = eng.profile_read(scope + '/gemm')
                tot_ms, tot_n = tot_ms + ms, tot_n + n
                tot_flops32 += 8.0 * cin * cout * FFT_FREQS * mrows * n              # complex multiply-adds as real FLOPs
                tot_flops += nprod * 8.0 * cin * cout * FFT_FREQS * mrows * n        # executed on the bf16 matrix cores
                # activation spectra and filter spectra (2 B x np parts x re|im per complex number), product spectra (complex fp32; complex fp16 on the
                # default route of bf16 handles, whose intermediates are all 16-bit)
                ybytes = 4 if single and fft_t16 is not False else 8
                tot_bytes += FFT_FREQS * (rows_p * cin * 4 * np_parts + cin * cout * 4 * np_parts + mrows * cout * ybytes) * n
            gemm = {'np_parts': np_parts, 'products': nprod, 'flops32_per_launch': tot_flops32 / max(tot_n, 1), 'bytes_per_launch': tot_bytes / max(tot_n, 1)}
        launch_ms = tot_ms / max(tot_n, 1)                     # average launch duration (HIP events, launch stream)
        flops_launch = tot_flops / max(tot_n, 1)               # average algorithmic (direct kernels) / executed (split GEMM) FLOPs per launch
        achieved = flops_launch / (launch_ms * 1e-3) / 1e12 if tot_n else None
        peak = PEAK_TFLOPS['bf16' if freq_domain else dtype]     # the channel GEMM runs on the bf16 matrix cores for both handle types
        fp32_equiv = None
        if f32_conv == 'split16' and achieved:      # the roofline of these kernels is the 16-bit matrix-core peak
            fp32_equiv = achieved
            achieved = achieved * 3
            peak = PEAK_TFLOPS['bf16']
        value = world * B * args.steps / dt
        out = {
            'value': value, 'ms_per_step': dt / args.steps * 1e3, 'ms_per_step_profiled': dt_prof / args.steps * 1e3,
            # the arithmetic the path computes in: exact fp32 MFMA chain, or fp32 operands carried as 16-bit parts
            # 'f32' alone = the exact fp32 MFMA accumulation chain; the default fp32 route carries every fp32 spectrum as two scaled fp16 parts (22 bits)
            'dtype': ((('bf16(fp16 spectra + fp16 row-transformed tensors, fft)' if fft_single is not False else 'bf16(bf16x2 spectra, fft)') if freq_domain else 'bf16') if dtype == 'bf16' else
                      {'split16': 'f32(fp16x3)'}.get(f32_conv, 'f32(fp16x2 spectra, fft)' if freq_domain else 'f32')),
            'config': {'workload': '%s: batch=%d/GPU synthetic 480x720x3, part detector%s forward + argmax, %s%s%s'
                                   % (config_name or ('configs[1]' if dtype == 'fp32' else 'configs[2]'), B, ' + spatial model' if use_sm else '',
                                      dtype + (' operands, stride-1 layers as fp16x3 split MFMA' if f32_conv == 'split16' else ''),
                                      ', micro-batches of %d' % mb if B > mb else '',
                                      ', DEBUG filters/4' if args.debug else ''),
                       'batch_per_gpu': B, 'global_batch': world * B, 'micro_batch': min(mb, B), 'use_sm': use_sm,
                       'collective': 'all_gather coords int32 [B,2,9]'},
            'path_tflops': value * FLOPS_PD_SM / (16 if args.debug else 1) / 1e12,
            'scaling': 'weak',
            'roofline': {'bound': 'mfma', 'kernel': ('cgemm_split_kernel: channel GEMM of the frequency-domain 9x9 layers (conv4_fullres + conv5; %s)' if freq_domain else
                                                     'conv_igemm 9x9, 60x90 maps (conv4_fullres + conv5 launches; %s)')
                                   % ('hand-written, v_mfma_f32_32x32x16_%s, operands split into %d %s parts (%d real products per multiply), one complex [B x Cin] x [Cin x Cout] '
                                      'product per frequency of the 64 x 96 transform = 3136 per launch, LDS-DMA operand rings; achieved = executed bf16 MFMA FLOPs / GEMM time'
                                      % ('bf16' if dtype == 'bf16' and gemm['np_parts'] == 2 else 'f16', gemm['np_parts'], 'bf16' if dtype == 'bf16' and gemm['np_parts'] == 2 else 'fp16 (scaled spectra)', gemm['products'])
                                      if freq_domain else
                                      'fp32 operands as 2 fp16 parts, 3 x fp16 MFMA 32x32x16 per k16 step, 12x32 patch x 256 ch; achieved = executed fp16 MFMA FLOPs (3 x algorithmic) against the fp16 peak'
                                      if f32_conv == 'split16' else
                                      'fp32 MFMA 32x32x2, 128-pixel strip tiles x 128 ch' if dtype == 'fp32' else
                                      'bf16 MFMA 32x32x16, 384-pixel strips of the flattened batch x 256 ch, LDS-DMA halo + weight rings' if kname == 'conv_strip_bf16_kernel'
                                      else 'bf16 MFMA 32x32x16, 12x32 patch x 256 ch'),
                         'kernel_name': kname,
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': (achieved / peak) if achieved else None,
                         'launch_ms': launch_ms, 'launches': tot_n, 'flops_per_launch': flops_launch},
        }
        rf = out['roofline']
        tkey = '%s_b%d_fft%s' % (dtype, min(B, mb), '_bf16x2' if dtype == 'bf16' and fft_single is False else '') if freq_domain else '%s%s_b%d' % (dtype, '_' + f32_conv if f32_conv == 'split16' else '', min(B, mb))
        rf['traffic'], rf['traffic_source'] = pmc_traffic(tkey) if not args.debug else (None, None)
        # SURVEY 8d's yardstick beside the executed-work one: images/s x 413.19 GFLOP (the direct-convolution FLOPs of the path) / MFMA peak of the handle's type
        rf['algorithmic_frac'] = out['path_tflops'] / PEAK_TFLOPS[dtype]
        if freq_domain:
            # the GEMM against both ceilings; `bound` names the one it sits closer to
            gbs = gemm['bytes_per_launch'] / (launch_ms * 1e-3) / 1e9 if tot_n else None
            useful = gemm['flops32_per_launch'] / (launch_ms * 1e-3) / 1e12 if tot_n else None      # complex-GEMM FLOPs without the split multiplier
            rf['mfma'] = {'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': (achieved / peak) if achieved else None,
                          'fp32_equivalent_tflops': useful, 'useful_frac': (useful / peak) if useful else None,
                          'note': 'achieved counts the %d real products per multiply of the split operands as executed FLOPs; useful_frac does not' % gemm['products']}
            rf['hbm'] = {'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': (gbs / PEAK_HBM_GBS) if gbs else None,
                         'algorithmic_bytes_per_launch': gemm['bytes_per_launch']}
            if gbs and achieved and gbs / PEAK_HBM_GBS > achieved / peak:
                rf.update({'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS})
            rf['layer_ms'] = layer_ms  # the whole layer: four transform kernels + the GEMM
            # the whole step against the HBM roof: bytes of every kernel of one step (2 x FETCH_SIZE + WRITE_SIZE summed over the committed counter pass of
            # this configuration) / measured step time
            sb, ssrc = step_traffic(tkey) if not args.debug else (None, None)
            if sb:
                imgs = min(B, mb)
                gbs_step = sb * (B / imgs) / (dt / args.steps) / 1e9
                rf['step'] = {'bytes_per_step': sb * (B / imgs), 'bytes_per_image': sb / imgs, 'achieved': gbs_step, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                              'frac': gbs_step / PEAK_HBM_GBS, 'traffic_source': ssrc,
                              'note': 'SURVEY 8d prices the direct formulation at ~77 MB of activations per image; the five-pass frequency-domain route moves several times that'}
            out['path_tflops_note'] = 'images/s x direct-convolution FLOPs: the wide 9x9 layers execute 35x fewer in the frequency domain'
        free_b, total_b = torch.cuda.mem_get_info(dev)
        out['device_mem_used_gb'] = (total_b - free_b) / 1e9      # everything resident on this rank's GPU right after the timed steps (inputs, weights, filter spectra, workspace)
        med = float(np.median(step_ms))
        out['median_ms_per_step'] = med
        out['value_median'] = B / (med * 1e-3)      # this rank's images / median step (HIP events on the launch stream; no barrier inside)
        out['step_ms_min_max'] = [float(min(step_ms)), float(max(step_ms))]
        if use_sm:
            # an FFT spatial model reports HBM GB/s beside the MFMA yardstick: ~14 MB of algorithmic traffic per image (10 forward + 81
            # product + 81 inverse spectra / frames); the fused route (sm_fused.hip) keeps all but the 10 likelihood spectra on the CU -- measured
            # bytes per image (2 x FETCH_SIZE + WRITE_SIZE of its two kernels) in profiles/pmc_traffic.json
            ms, n = eng.profile_read('sm')
            if n:
                imgs = B * args.steps / n                      # images per sm launch group (micro-batch)
                gbs = SM_ALGO_BYTES_PER_IMAGE * imgs / (ms / n * 1e-3) / 1e9
                smb, smsrc = sm_traffic(dtype)
                rf['sm'] = {'ms_per_call': ms / n, 'images_per_call': imgs, 'algorithmic_bytes_per_image': SM_ALGO_BYTES_PER_IMAGE, 'achieved': gbs,
                            'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS, 'traffic_bytes_per_image': smb, 'traffic_source': smsrc,
                            'mfma_yardstick_tflops': FLOPS_SM * imgs / (ms / n * 1e-3) / 1e12}
        if agree_ref is not None:
            # north_star judges the bf16 configuration on arg-max agreement: this engine's coordinates against the fp32 default engine's on
            # the same resident images (evaluation.py:15-24, main.py:389-397)
            from joint_cnn_mrf_amd.evaluation import argmax_agreement
            r = eng.forward(x, torso if use_sm else None, use_sm=use_sm, want_prob=True)
            key = 'sm' if use_sm else 'pd'
            a = argmax_agreement(agree_ref[key + '_prob'], agree_ref[key + '_coords'], r[key + '_prob'], r[key + '_coords'])
            out['argmax_agreement'] = {'vs': 'fp32 default engine, same %d images, %s coordinates' % (B, 'spatial-model' if use_sm else 'part-detector'),
                                       'exact': a['exact'], 'within1': a['within1'], 'mean_dist': a['mean_dist'], 'n_joints': a['n_joints'],
                                       'safe': a['safe'], 'margin_mult': a['margin_mult'], 'rms_logprob_err': a['rms_logprob_err']}
            if use_sm:
                a = argmax_agreement(agree_ref['pd_prob'], agree_ref['pd_coords'], r['pd_prob'], r['pd_coords'])
                out['argmax_agreement']['pd'] = {'exact': a['exact'], 'within1': a['within1'], 'mean_dist': a['mean_dist'], 'safe': a['safe']}
            del r
        if fp32_equiv is not None:
            out['roofline']['fp32_equivalent_tflops'] = fp32_equiv
            out['roofline']['x_fp32_mfma_peak'] = fp32_equiv / PEAK_TFLOPS['fp32']
    eng.close()
    del x, torso
    torch.cuda.empty_cache()
    return out


# Training step (BASELINE.json configs[4]): algorithmic FLOPs per image = forward + data gradient + weight
# gradient of every conv (no data gradient for the three conv1 layers) + 3x the pairwise term.
FLOPS_CONV1 = 1_088_640_000 + 272_160_000 + 68_040_000          # conv1 full/half/quarter
FLOPS_TRAIN = 3 * FLOPS_PD_SM - FLOPS_CONV1


def run_train(args, B, params, world, rank, local_rank, dev, use_sm, prec=None, f32_conv='args'):
    """`--train`: time the joint training step (loss + gradients, gradient all-reduce, clip + Adam, table
    refresh) on B images per GPU; data-parallel, one RCCL all-reduce of the flat gradient buffer per step."""
    from joint_cnn_mrf_amd.train import Trainer
    prec = prec or args.dtype or 'fp32'
    f32_conv = args.f32_conv if f32_conv == 'args' else f32_conv
    eng = Engine(device=local_rank, precision=prec, f32_conv=f32_conv if prec == 'fp32' else None).load_params(params)
    tr = Trainer(eng, optimizer='adam', lr=0.001, lmbd=0.001, use_sm=use_sm, overlap_allreduce=args.overlap)
    x = torch.as_tensor(synth.make_images(B, seed=1234 + rank), device=dev)
    y = torch.as_tensor(synth.make_targets(B, seed=4321 + rank), device=dev)
    moving = Trainer.moving_statistics_of(params) if world > 1 else None
    for _ in range(args.warmup):
        tr.train_step(x, y, moving=moving)
    torch.cuda.synchronize()
    if dist_on():
        dist.barrier()
    torch.cuda.synchronize()
    eng.set_profile(True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i].record()
        losses, _ = tr.train_step(x, y, moving=moving)
    evs[args.steps].record()
    torch.cuda.synchronize()
    if dist_on():
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on():
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.set_profile(False)
    out = None
    if rank == 0:
        value = world * B * args.steps / dt
        scale = 16 if args.debug else 1
        # dominant training kernels: the three conv5 passes, each 229.3 GFLOP per image as a direct convolution (60x90 px, 81 taps, 512x512)
        kern = {}
        sp = f32_conv == 'split16' and prec == 'fp32'
        fd = prec == 'fp32' and not sp        # fp32 handles: all three passes in the frequency domain (conv_fft.hip, wgrad_fft.hip)
        mult = 3
        # frequency-domain passes are bound by the filter-sized spectra (F x Cin x Cout complex fp32: 6.58 GB for conv5 at 64x96), not by MFMA:
        # bytes each pass must move: W or P spectra once per producer/consumer + split activation spectra + product spectra + maps
        C5 = 512 // (4 if args.debug else 1)
        # Round 4: conv5 runs on 32 x 32 overlap-save windows (jcm_train.hip): 3 x 4 windows per image are a batch of 12 B "images" on a transform with
        # 32 * 17 = 544 frequencies -- the filter-sized spectra shrink 5.8x, the activation-sized ones grow 2.1x, plus the window tensors themselves
        WIN_F, WIN_B = 32 * 17, 12 * B
        wspec = WIN_F * C5 * C5 * 8
        xspec = yspec = WIN_F * WIN_B * C5 * 8                                            # activation / product spectra: two fp16 parts / complex fp32
        amap, awin, aval = B * 60 * 90 * C5 * 4, WIN_B * 32 * 32 * C5 * 4, WIN_B * 24 * 24 * C5 * 4      # map, gathered windows, valid regions
        fd_bytes = {'conv5': wspec + 2 * xspec + 2 * yspec + awin + aval,      # inside the layer's events: windows read, spectra written + read, filter spectra read, valid regions written
                    'dgrad:conv5': wspec + 2 * xspec + 2 * yspec + awin + aval,      # (the gather before and the scatter behind are timed with the step, not here)
                    'wgrad:conv5': 2 * wspec + amap + 2 * awin + 3 * xspec}                       # P written (spec) + read (taps); dz windows + spectra written, x and dz spectra read
        for key, what in (('conv5', 'forward, ' + ('conv_split_kernel' if sp else 'conv_fft (cgemm_split_kernel)' if fd else 'conv_igemm_f32')),
                          ('dgrad:conv5', 'data gradient, ' + ('conv_split_kernel on flipped weights' if sp else 'conv_fft on the flipped filter spectra, dz spectra shared with the weight gradient'
                                                               if fd else 'conv_igemm_f32 on flipped weights')),
                          ('wgrad:conv5', 'weight gradient, ' + ('wgrad_split_kernel<9>' if sp else 'wgrad_fft (dz transforms + wgrad_spec_kernel + wgrad_taps_*)' if fd else 'wgrad_kernel<9>'))):
            ms, n = eng.profile_read(key)
            if n:
                tf = FLOPS_DOMINANT['conv5'] / scale * B / (ms / n * 1e-3) / 1e12      # algorithmic (fp32-equivalent) FLOPs
                if sp:       # split kernels execute `mult` 16-bit MFMA FLOPs per algorithmic one: their roofline is the 16-bit peak
                    kern[key] = {'kernel': what, 'launch_ms': ms / n, 'launches': n, 'achieved': tf * mult, 'frac': tf * mult / PEAK_TFLOPS['bf16'],
                                 'fp32_equivalent_tflops': tf}
                elif fd:
                    gbs = fd_bytes[key] / (ms / n * 1e-3) / 1e9
                    kern[key] = {'kernel': what, 'launch_ms': ms / n, 'launches': n, 'bound': 'hbm', 'bytes': fd_bytes[key], 'achieved': gbs, 'unit': 'GB/s',
                                 'frac': gbs / PEAK_HBM_GBS, 'direct_equivalent_tflops': tf}
                else:
                    kern[key] = {'kernel': what, 'launch_ms': ms / n, 'launches': n, 'achieved': tf, 'frac': tf / PEAK_TFLOPS[prec]}
        out = {'metric': 'images/sec joint training step (fwd+bwd+update), part detector + spatial model', 'value': value,
               'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
               'median_ms_per_step': float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'bf16' if prec == 'bf16' else {'split16': 'f32(fp16x3)'}.get(f32_conv, 'f32(fp16x2 spectra, fft, overlap-save windows)' if fd else 'f32'), 'data': 'synthetic',
               'config': {'workload': 'configs[4]: joint training, batch=%d/GPU synthetic 480x720x3, %s%s, Adam, clip 4.0%s'
                                      % (B, ('fp32 (stride-1 layers in the frequency domain, channel products on two scaled fp16 parts per operand; conv1 on fp32 MFMA)' if fd else 'fp32 MFMA') if prec == 'fp32' else 'mixed precision: bf16 activations/gradients + bf16 MFMA, fp32 master weights / statistics / losses / spatial model / optimizer',
                                         ' operands; forward, data and weight gradients as fp16x3 split MFMA (gradients scaled per tensor by a power of two)' if f32_conv == 'split16' else '',
                                         ', DEBUG filters/4' if args.debug else ''),
                          'batch_per_gpu': B, 'global_batch': world * B, 'use_sm': use_sm,
                          'collective': 'all_reduce of %d fp32 gradients' % tr.n_elements},
               'train_tflops': value * FLOPS_TRAIN / scale / 1e12,      # images/s x direct-convolution FLOPs (the frequency-domain passes execute far fewer)
               **({} if fd else {'mfma_peak_tflops': PEAK_TFLOPS[prec], 'frac_of_mfma_peak': value * FLOPS_TRAIN / scale / 1e12 / PEAK_TFLOPS[prec]}),
               'roofline': (dict(bound='hbm', peak=PEAK_HBM_GBS, unit='GB/s', traffic=None,
                                 kernel='weight gradient of conv5 in the frequency domain on 32x32 overlap-save windows: dz window gather + transforms, wgrad_spec_kernel (K = 192 windows; writes P[f][ci][co], 1.14 GB), '
                                        'wgrad_taps_cols/rows (read it); achieved = bytes the pass must move / its time',
                                 **({'achieved': kern['wgrad:conv5']['achieved'], 'frac': kern['wgrad:conv5']['frac'],
                                     'launch_ms': kern['wgrad:conv5']['launch_ms']} if 'wgrad:conv5' in kern else {})) if fd else
                            dict(bound='mfma', peak=PEAK_TFLOPS['bf16'] if sp else PEAK_TFLOPS[prec], unit='TFLOP/s', traffic=None,
                                 **({'kernel': ('wgrad_split_kernel<9,1> on conv5 (bf16 operands, LDS transpose reads)' if prec == 'bf16' else
                                                'wgrad_split_kernel<9,2> on conv5 (2 fp16 parts per operand, 3 x fp16 MFMA per k16 step; achieved = executed fp16 MFMA FLOPs)'),
                                     'achieved': kern['wgrad:conv5']['achieved'], 'frac': kern['wgrad:conv5']['frac'],
                                     'launch_ms': kern['wgrad:conv5']['launch_ms']} if 'wgrad:conv5' in kern else {}))),
               'conv5_passes': kern,
               'loss': [float(v) for v in losses.cpu().numpy()], 'workspace_gb': eng.workspace_bytes() / 1e9}
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default: 64 fp32 / 256 bf16)')
    ap.add_argument('--global-batch', type=int, default=None,
                    help='BASELINE configs[3]: a FIXED global batch (2048) sharded over the ranks, global_batch // N images per rank '
                         '(main.py:511,516-517), walked in micro-batches inside one step; strong scaling.  bf16 unless --dtype fp32')
    ap.add_argument('--micro-batch', type=int, default=None, help='images per internal slice of jcm_forward (default 256 bf16 / 64 fp32)')
    ap.add_argument('--dtype', default=None, choices=['fp32', 'bf16'],
                    help='default: the headline line is configs[1] (fp32, batch 64) and configs[2] (bf16, batch 256) '
                         'is measured too and reported under "bf16_config2"')
    ap.add_argument('--no-sm', action='store_true', help='part detector only')
    ap.add_argument('--f32-conv', default=None, choices=['exact', 'split16'], help='arithmetic of the DIRECT fp32 kernels of the headline run; anything but the default also leaves the frequency-domain route')
    ap.add_argument('--debug', action='store_true', help='filters/4 (main.py:40-41); not the headline config, fp32 only')
    ap.add_argument('--cpu-reps', type=int, default=5, help='cpu_baseline: runs per batch size (B=1 and B=8, median; 0 = skip)')
    ap.add_argument('--cpu-budget', type=float, default=150.0, help='cpu_baseline: seconds after which no further repetition is started')
    ap.add_argument('--cpu-images', type=int, default=None, help='(deprecated) 0 = skip the cpu_baseline')
    ap.add_argument('--train', action='store_true', help='time the joint training step (configs[4]) instead of the forward; fp32, '
                                                         'default 16 images per GPU (batch 128 over 8 GPUs)')
    ap.add_argument('--overlap', action='store_true', help='--train, N > 1, RCCL: start each layer\'s gradient all-reduce during the backward pass')
    ap.add_argument('--fft-fuse', type=int, default=None, help='fp32 engines: option "fft_fuse" (3 = default: pool and merge hand-overs fused; 0 = separate kernels, the A/B arm)')
    ap.add_argument('--extras', type=int, default=1, help='0: the headline configuration only (no bf16 / chain / split16 / sustained / training lines)')
    ap.add_argument('--layer-times', action='store_true', help='print the HIP-event time of every MFMA conv layer to stderr')
    args = ap.parse_args()
    if args.cpu_images == 0 or not args.extras:
        args.cpu_reps = 0

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the
        # same command line the driver uses) and hand over.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # JCM_BENCH_BACKEND=gloo is a plumbing test of the N>1 flow on a single-GPU box (ranks share
    # cuda:0, coords cross through host memory); the real multi-GPU run is nccl = RCCL over xGMI.
    backend = os.environ.get('JCM_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1 or os.environ.get('JCM_BENCH_FORCE_DIST') == '1':      # (FORCE_DIST: a one-rank group, so that the nccl branch runs on a one-GPU box)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    use_sm = not args.no_sm

    params = synth.make_pd_params(debug=args.debug)                    # He init, BN identity (main.py:138-153)
    if use_sm:
        params.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))   # main.py:477-487

    if args.train:
        out = run_train(args, args.batch or 16, params, world, rank, local_rank, dev, use_sm)
        if rank == 0:
            emit(out)
        if dist_on():
            dist.destroy_process_group()
        return

    scaling = 'weak'
    if args.global_batch:
        # configs[3]: the global batch is fixed, every rank takes global_batch // N contiguous images (the
        # reference's tower slices, main.py:511,516-517) and one step is one pass over the rank's whole share
        head_dtype = args.dtype or 'bf16'
        head_batch = args.global_batch // world
        if head_batch < 1:
            raise SystemExit('--global-batch %d is smaller than the number of ranks %d' % (args.global_batch, world))
        scaling = 'strong'
        head = run_config(args, head_dtype, head_batch, params, world, rank, local_rank, dev, use_sm,
                          f32_conv=args.f32_conv if head_dtype == 'fp32' else None, micro_batch=args.micro_batch,
                          config_name='configs[3] (global batch %d sharded over %d rank%s)' % (args.global_batch, world, '' if world == 1 else 's'))
    else:
        head_dtype = args.dtype or 'fp32'
        head_batch = args.batch or (64 if head_dtype == 'fp32' else 256)
        head = run_config(args, head_dtype, head_batch, params, world, rank, local_rank, dev, use_sm,
                          f32_conv=args.f32_conv if head_dtype == 'fp32' else None, micro_batch=args.micro_batch)
    second = second_x2 = second_mfma = split16 = config3 = chain = sus = None
    if args.dtype is None and not args.debug and not args.global_batch and args.extras:
        # configs[2] (bf16, batch 256) with its arg-max agreement against the fp32 engine on the same images (single-rank runs: rank 0 alone
        # would hold the other ranks at the barrier while it computes the reference)
        second = run_config(args, 'bf16', args.batch or 256, params, world, rank, local_rank, dev, use_sm, agree=world == 1)
        if world == 1:      # the strict operand form of the channel GEMM (two bf16 parts, three products, fp32 row-transformed tensors) beside the default
            second_x2 = run_config(args, 'bf16', args.batch or 256, params, world, rank, local_rank, dev, use_sm, fft_single=False, fft_t16=False, agree=True,
                                   config_name='configs[2], channel GEMM on two bf16 parts (fft_single=0)')
        # configs[2] on the direct bf16 MFMA kernels only (conv_strip_bf16_kernel for the 9x9 layers): the north star's
        # "9x9 + pairwise pass on the bf16 matrix cores" with its own driver-timed roofline
        second_mfma = run_config(args, 'bf16', args.batch or 256, params, world, rank, local_rank, dev, use_sm, conv9_fft=False, agree=world == 1,
                                 config_name='configs[2], direct bf16 MFMA kernels only (conv9_fft=0)')
        # configs[3]: a FIXED global batch of 2048 images sharded over the ranks (main.py:511,516-517), bf16, micro-batches of 256;
        # present in every line so that the driver's N = 1/2/4/8 runs trace the strong-scaling curve
        gb = 2048
        if gb // world >= 1:
            config3 = run_config(args, 'bf16', gb // world, params, world, rank, local_rank, dev, use_sm,
                                 config_name='configs[3] (global batch %d sharded over %d rank%s)' % (gb, world, '' if world == 1 else 's'))
            if config3 is not None:
                config3['scaling'] = 'strong'
    if args.dtype is None and not args.debug and not args.global_batch and world == 1 and args.extras:
        # the same fp32 configuration on the exact fp32 MFMA accumulation chain (conv9_fft = 0: no reduced-precision operand anywhere); few steps, it is 18x slower
        saved = args.steps, args.warmup
        args.steps, args.warmup = min(args.steps, 5), min(args.warmup, 1)
        chain = run_config(args, 'fp32', head_batch, params, world, rank, local_rank, dev, use_sm, conv9_fft=False,
                           config_name='configs[1], exact fp32 MFMA chain (conv9_fft=0)')
        args.steps, args.warmup = saved
        sus = sustained(args, head_dtype, head_batch, params, local_rank, dev, use_sm)
        # ... and with the stride-1 layers on the direct split kernels (two fp16 parts per operand, three products: the A/B arm of the frequency-domain route)
        split16 = run_config(args, 'fp32', head_batch, params, world, rank, local_rank, dev, use_sm, f32_conv='split16')

    # configs[4] (joint training step, 16 images per GPU) beside the inference lines; single-GPU runs only, the
    # multi-GPU training flow has its own entry point (`--train`)
    train = {}
    if args.dtype is None and not args.debug and world == 1 and use_sm and not args.global_batch and args.extras:
        train['train_config4_f32'] = run_train(args, 16, params, world, rank, local_rank, dev, use_sm, prec='fp32', f32_conv=None)
        train['train_config4_bf16'] = run_train(args, 16, params, world, rank, local_rank, dev, use_sm, prec='bf16', f32_conv=None)

    if rank == 0:
        detail = {'metric': 'images/sec (720x480, K=9 joints) part-detector+spatial-model fwd',
                  'value': head['value'], 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                  'ms_per_step': head['ms_per_step'], 'ms_per_step_profiled': head.get('ms_per_step_profiled'), 'median_ms_per_step': head['median_ms_per_step'], 'value_median': head['value_median'] * world,
                  'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
                  'dtype': head['dtype'], 'data': 'synthetic', 'config': head['config'], 'path_tflops': head['path_tflops'],
                  'roofline': head['roofline'], 'device_mem_used_gb': head.get('device_mem_used_gb')}
        others = {'bf16_config2': second, 'bf16_config2_bf16x2': second_x2, 'bf16_config2_mfma': second_mfma, 'config3_gb2048': config3,
                  'f32_chain_config1': chain, 'f32_split16_config1': split16}
        for key, o in others.items():
            if o is not None:
                detail[key] = o
        if sus is not None:
            sus['vs_value'] = sus['value'] / head['value']
            detail['sustained'] = sus
        for key, tr_out in train.items():
            if tr_out is not None:
                detail[key] = {k: tr_out[k] for k in ('value', 'unit', 'ms_per_step', 'median_ms_per_step', 'dtype', 'config', 'train_tflops', 'frac_of_mfma_peak', 'roofline', 'conv5_passes', 'workspace_gb') if k in tr_out}
        if args.cpu_reps > 0 and world == 1 and not args.debug:      # the CPU baseline is a single-GPU-run item (rank 0, N = 1)
            detail['cpu_baseline'] = cpu_baseline(params, reps=args.cpu_reps, budget_s=args.cpu_budget)
        emit(detail)
    if dist_on():
        dist.destroy_process_group()


def compact_roofline(rf, klen=96):
    """The `roofline` object of the final line: the contract's keys, the kernel name cut to `klen` characters, and the step / spatial-model summaries."""
    out = {k: rf.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launch_ms')}
    out['kernel'] = (rf.get('kernel_name') or rf.get('kernel') or '')[:klen]
    if 'mfma' in rf and rf['mfma'].get('frac') is not None:
        out['mfma_frac'] = rf['mfma']['frac']
    if 'hbm' in rf and rf['hbm'].get('frac') is not None:
        out['hbm_frac'] = rf['hbm']['frac']
    if 'step' in rf:
        out['step'] = {k: rf['step'][k] for k in ('bytes_per_image', 'achieved', 'frac')}
    if 'sm' in rf:
        out['sm'] = {k: rf['sm'][k] for k in ('ms_per_call', 'algorithmic_bytes_per_image', 'traffic_bytes_per_image', 'achieved', 'unit', 'frac')}
    return out


def rnd(v, n=4):
    """Round floats (recursively) so that the final line stays small."""
    if isinstance(v, float):
        return float('%.*g' % (n + 2, v))
    if isinstance(v, dict):
        return {k: rnd(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [rnd(x, n) for x in v]
    return v


def emit(detail):
    """Verbose record -> bench_detail.json (+ stderr); ONE compact line (<= 4 KB) -> stdout, last."""
    ddir = os.path.join(ROOT, 'gpurun_out') if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else os.getcwd()
    dpath = os.path.join(ddir, 'bench_detail.json')
    try:
        with open(dpath, 'w') as fh:
            json.dump(detail, fh)
    except OSError:
        dpath = None
    print(json.dumps(detail), file=sys.stderr)
    line = {k: detail[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_profiled', 'median_ms_per_step', 'higher_is_better', 'scaling',
                                   'vs_baseline', 'dtype', 'data') if k in detail}
    cfg = detail['config']
    line['config'] = {'workload': cfg['workload'][:160], **{k: cfg[k] for k in ('batch_per_gpu', 'global_batch', 'micro_batch', 'use_sm', 'collective') if k in cfg}}
    line['roofline'] = compact_roofline(detail['roofline'])
    if 'cpu_baseline' in detail:
        cb = detail['cpu_baseline']
        line['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb['sample'][:200],
                                **{k: v['images_per_s'] for k, v in cb.items() if isinstance(v, dict) and 'images_per_s' in v}}
    configs = {}
    for key, o in detail.items():
        if not isinstance(o, dict) or 'value' not in o or key in ('cpu_baseline',):
            continue
        rf = o.get('roofline') or {}
        e = {'value': o['value'], 'ms_per_step': o['ms_per_step'], 'dtype': o['dtype'][:48] if isinstance(o.get('dtype'), str) else o.get('dtype')}
        if 'median_ms_per_step' in o:
            e['median_ms_per_step'] = o['median_ms_per_step']
        if rf.get('frac') is not None:
            e['frac'], e['bound'] = rf['frac'], rf.get('bound')
        if 'step' in rf:
            e['step_frac'] = rf['step']['frac']
        if 'argmax_agreement' in o:
            a = o['argmax_agreement']
            e['argmax_agreement'] = {'exact': a['exact'], 'within1': a['within1'], 'n_joints': a['n_joints'],
                                     'safe_exact': a['safe']['exact'], 'safe_within1': a['safe']['within1'], 'safe_n': a['safe']['n_joints']}
        if key == 'sustained':
            e = {'value': o['value'], 'ms_per_step': o['ms_per_step'], 'seconds': o['seconds'], 'vs_value': o.get('vs_value'), 'smi': o.get('smi')}
        configs[key] = e
    if configs:
        line['configs'] = configs
    if dpath:
        line['detail'] = os.path.relpath(dpath, ROOT)
    txt = json.dumps(rnd(line))
    if len(txt) > 4096:      # never expected; keep the contract keys whatever happens
        line.pop('configs', None)
        line['configs_dropped'] = 'line exceeded 4 KB, see detail'
        txt = json.dumps(rnd(line))
    sys.stderr.flush()
    print(txt)
    sys.stdout.flush()


if __name__ == '__main__':
    main()
