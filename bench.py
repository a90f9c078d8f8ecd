#!/usr/bin/env python
"""Benchmark of the DGR inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--pairs-per-step B] [--n-raw 50000]

One "step" = one pass of the hot path (FCGF x2 -> 1-NN -> 6-D inputs -> 6-D inlier net -> gate ->
weighted Procrustes -> SE(3) refinement; dgr_register_batch) over S x B synthetic 3DMatch-shaped
pairs per GPU (BASELINE.json configs[1]: 50k raw points per fragment, 5 cm voxels, conv1 k=7)
whose voxelised coordinates are already resident in HBM: S HIP streams, each driven by its own host
thread with its own library context, each registering its own batch of B pairs (pairs are
independent units, streams never exchange data; the second and third stream fill the holes the
small map-building kernels of the first leave on the chip).  With N > 1 (launched through
torch.distributed.run, one process per GPU) every rank registers its own S x B pairs (weak scaling,
no collective on the data path), the weights come from rank 0 by ONE RCCL broadcast and the results
are gathered on rank 0.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_HBM_GBPS = 8000.0          # spec


def conv_work(stats):
    """Algorithmic work of one net forward from the per-layer statistics:
    FLOP = 2 P Cin Cout; compulsory bytes = 4 (Nin Cin + Nout Cout + Kne Cin Cout) + 8 P."""
    flop = sum(2.0 * s['pairs'] * s['cin'] * s['cout'] for s in stats)
    byts = sum(4.0 * (s['n_in'] * s['cin'] + s['n_out'] * s['cout'] + s['nonempty'] * s['cin'] * s['cout'])
               + 8.0 * s['pairs'] for s in stats)
    flop_c64 = sum(2.0 * s['pairs'] * s['cin'] * s['cout'] for s in stats if max(s['cin'], s['cout']) <= 64)
    byts_c64 = sum(4.0 * (s['n_in'] * s['cin'] + s['n_out'] * s['cout'] + s['nonempty'] * s['cin'] * s['cout'])
                   + 8.0 * s['pairs'] for s in stats if max(s['cin'], s['cout']) <= 64)
    return flop, byts, flop_c64, byts_c64


def cpu_baseline(ck, n_raw, voxel, kind, full_sizes):
    """The CPU oracle (restated reference CPU path, kind "port") timed on this host's cores on a
    BOUNDED sample of the same workload: one pair generated like the benchmark pairs but with a
    quarter of the raw points; every stage runs completely on that pair and its time is scaled to
    the full-size pair (conv stacks and registration linearly in the voxel count, the brute-force
    1-NN by N0*N1).  Reported baseline, not a target."""
    from deepglobalregistration_amd import synth
    from oracle import knn as oknn, pipeline as opipe, registration as oreg, resunet as oresunet
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    xyz0, xyz1, T_gt = synth.synth_pair(0, n_raw=max(2000, n_raw // 4), kind=kind)
    t = {}
    t0 = time.time()
    p0, c0, f0 = opipe.preprocess(xyz0, voxel)
    p1, c1, f1 = opipe.preprocess(xyz1, voxel)
    t['voxelize'] = time.time() - t0
    n0, n1 = len(p0), len(p1)
    s_lin = (full_sizes[0] + full_sizes[1]) / float(n0 + n1)
    s0 = full_sizes[0] / float(n0)
    s_knn = (full_sizes[0] * full_sizes[1]) / float(n0 * n1)
    ks = ck['config']['feat_conv1_kernel_size']
    t0 = time.time()
    F0 = oresunet.resunet_forward(ck['state_dict'], c0, f0, 3, ks, True)
    F1 = oresunet.resunet_forward(ck['state_dict'], c1, f1, 3, ks, True)
    t['fcgf'] = (time.time() - t0) * s_lin
    t0 = time.time()
    idx1 = oknn.find_knn(F0, F1, nn_max_n=250).reshape(-1)
    t['knn'] = (time.time() - t0) * s_knn
    gt = synth.gt_correspondences(p0, p1, T_gt, voxel)
    idx1 = np.where(gt >= 0, gt, idx1)          # same harness override as the GPU run
    t0 = time.time()
    coords6, feats6 = opipe.inlier_inputs(p0, p1, c0, c1, np.arange(n0), idx1)
    oresunet.resunet_forward(ck['state_dict_inlier'], coords6, feats6, 6, 3, False)
    t['inlier_net'] = (time.time() - t0) * s0
    t0 = time.time()
    w, wsum, thr = opipe.confidence_gate(synth.gt_forced_logits(p0, p1[idx1], T_gt, voxel))
    if wsum >= thr:
        oreg.global_registration(p0, p1[idx1], w, break_threshold_ratio=1e-4, quantization_size=2 * voxel)
    t['registration'] = (time.time() - t0) * s0
    total = t['fcgf'] + t['knn'] + t['inlier_net'] + t['registration']
    return {'value': 1.0 / total, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
            'sample': f'1 pair with {max(2000, n_raw // 4)} raw pts/fragment ({n0}/{n1} voxels), all stages, '
                      f'times scaled to {full_sizes[0]}/{full_sizes[1]} voxels (conv/registration linear, 1-NN by N0*N1); '
                      'voxelisation excluded',
            'scaled_stage_s': {k: round(v, 3) for k, v in t.items()}}


_T0 = time.time()


def log(msg):
    if int(os.environ.get('RANK', 0)) == 0:
        print(f'[bench +{time.time() - _T0:6.1f}s] {msg}', file=sys.stderr, flush=True)


def cfg_label(args):
    """Which BASELINE.json config the chosen flags correspond to (only the default is the headline)."""
    key = (args.kind, args.n_raw, args.voxel, args.conv1_ks)
    return {('indoor', 50000, 0.05, 7): 'BASELINE configs[1]',
            ('outdoor', 120000, 0.3, 5): 'BASELINE configs[2]',
            ('indoor', 200000, 0.025, 7): 'BASELINE configs[4]'}.get(key, 'non-BASELINE configuration')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--pairs-per-step', type=int, default=4, help='pairs per stream per step')
    ap.add_argument('--n-raw', type=int, default=50000, help='raw points per fragment')
    ap.add_argument('--voxel', type=float, default=0.05)
    ap.add_argument('--kind', default='indoor', choices=['indoor', 'outdoor'])
    ap.add_argument('--conv1-ks', type=int, default=7)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-refine', action='store_true', help='ablation: stop after weighted Procrustes')
    ap.add_argument('--from-host', action='store_true', help='PCIe-inclusive variant (NOT the headline): every step '
                    'starts from the raw float64 host points (H2D copy + GPU voxelisation inside the timed region)')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams per GPU, each driven by its own host '
                    'thread with its own library context and its own batch of pairs (independent units)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    import torch.distributed as dist
    # test hooks (a 1-GPU box cannot run RCCL between two ranks): DGR_BENCH_BACKEND=gloo moves the two tiny
    # collectives to the CPU, DGR_BENCH_ONE_GPU=1 puts every rank on device 0; the driver sets neither
    backend = os.environ.get('DGR_BENCH_BACKEND', 'nccl')
    if os.environ.get('DGR_BENCH_ONE_GPU'):
        local_rank = 0
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    from deepglobalregistration_amd import _lib, dist as ddist, ops, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration

    B = args.pairs_per_step
    S = max(1, args.streams)
    ck = synth.synth_checkpoint(seed=0, voxel_size=args.voxel, feat_conv1_kernel_size=args.conv1_ks) if rank == 0 else None
    coll_dev = device if backend == 'nccl' else torch.device('cpu')
    ck = ddist.broadcast_checkpoint(ck, src=0, device=coll_dev)
    log('checkpoint ready')

    class Worker:
        """One HIP stream + one library context + one resident batch of B pairs, driven by one host
        thread.  Pairs are independent units, so streams never exchange data."""

        def __init__(self, wid):
            self.wid = wid
            self.ctx = _lib.new_ctx(device) if S > 1 else None
            self.stream = torch.cuda.Stream(device) if S > 1 else torch.cuda.current_stream(device)
            self.result = None

        def __enter__(self):
            _lib.use_ctx(self.ctx)
            self._sctx = torch.cuda.stream(self.stream)
            self._sctx.__enter__()
            return self

        def __exit__(self, *exc):
            self._sctx.__exit__(*exc)
            _lib.use_ctx(None)

        def prepare(self):
            with self:
                self.dgr = DeepGlobalRegistration({'weights': ck, 'clip_weight_thresh': 0.05}, device)
                dgr = self.dgr
                dgr.fcgf_model._handle(); dgr.inlier_model._handle()
                base = (rank * S + self.wid) * B          # this worker's pairs: seeds base .. base+B-1
                self.pairs = [synth.synth_pair(base + i, n_raw=args.n_raw, kind=args.kind) for i in range(B)]
                x0, c0, x1, c1, self.off0, self.off1 = [], [], [], [], [0], [0]
                t0 = time.time()
                for p, (a, b, _) in enumerate(self.pairs):
                    xa, ca, _ = dgr.preprocess(a, batch_index=p)
                    xb, cb, _ = dgr.preprocess(b, batch_index=p)
                    x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
                    self.off0.append(self.off0[-1] + len(xa)); self.off1.append(self.off1[-1] + len(xb))
                torch.cuda.synchronize()
                self.t_vox = (time.time() - t0) / B
                self.C0, self.X0, self.C1, self.X1 = torch.cat(c0), torch.cat(x0), torch.cat(c1), torch.cat(x1)
                off0, off1 = self.off0, self.off1
                # harness-only overrides (DESIGN.md "Synthetic workload"): untrained weights give ~0 %
                # correct matches and a meaningless confidence, so a share of the 1-NN results is replaced
                # by ground-truth matches AFTER the search ran and the logits by GT-derived ones AFTER the
                # inlier net ran.
                X0h, X1h = self.X0.cpu().numpy(), self.X1.cpu().numpy()
                self.ovr = torch.from_numpy(np.concatenate([
                    (lambda g, o: np.where(g >= 0, g + o, -1))(
                        synth.gt_correspondences(X0h[off0[p]:off0[p + 1]], X1h[off1[p]:off1[p + 1]], self.pairs[p][2],
                                                 args.voxel, seed=p), off1[p]) for p in range(B)])).to(device)
                self.forced = None
                self.step()                                # untimed: final correspondences for the forced logits
                torch.cuda.synchronize()
                self.idx1 = ops.batch_output(device, 'idx1').cpu().numpy()
                self.forced = torch.from_numpy(np.concatenate([
                    synth.gt_forced_logits(X0h[off0[p]:off0[p + 1]], X1h[self.idx1[off0[p]:off0[p + 1]]],
                                           self.pairs[p][2], args.voxel) for p in range(B)])).to(device)
                for _ in range(args.warmup):
                    self.step()
                torch.cuda.synchronize()

        def step(self):
            if args.from_host and self.forced is not None:
                x0, c0, x1, c1 = [], [], [], []
                for p, (a, b, _) in enumerate(self.pairs):
                    xa, ca, _ = self.dgr.preprocess(a, batch_index=p)
                    xb, cb, _ = self.dgr.preprocess(b, batch_index=p)
                    x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
                self.C0, self.X0, self.C1, self.X1 = torch.cat(c0), torch.cat(x0), torch.cat(c1), torch.cat(x1)
            return self.dgr.register_voxelized(self.C0, self.X0, self.off0, self.C1, self.X1, self.off1,
                                               forced_logits=self.forced, skip_refinement=args.no_refine,
                                               override_idx1=self.ovr)

        def run(self, n):
            with self:
                for _ in range(n):
                    self.result = self.step()

    workers = [Worker(w) for w in range(S)]
    for w in workers:
        w.prepare()
    log(f'{S} stream(s) ready: weights resident, inputs voxelised (N0={workers[0].off0[-1]} N1={workers[0].off1[-1]} '
        f'per batch of {B}), warm-up done')

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import threading
    barrier()
    t0 = time.perf_counter()
    if S == 1:
        workers[0].run(args.steps)
    else:
        threads = [threading.Thread(target=w.run, args=(args.steps,)) for w in workers]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    log(f'timed region done: {elapsed / args.steps * 1e3:.1f} ms/step ({S * B} pairs/step/GPU)')
    T = np.concatenate([w.result[0] for w in workers]); status = np.concatenate([w.result[1] for w in workers])
    stats = np.concatenate([w.result[2] for w in workers])
    gathered = ddist.gather_results(T, status, stats, dst=0, device=coll_dev)
    w0 = workers[0]
    dgr, pairs, off0, off1, C0, X0, C1, X1, idx1 = w0.dgr, w0.pairs, w0.off0, w0.off1, w0.C0, w0.X0, w0.C1, w0.X1, w0.idx1
    t_vox = w0.t_vox
    all_pairs = [p for w in workers for p in w.pairs]

    def step(forced=None):
        return w0.step()
    forced = w0.forced
    _lib.use_ctx(w0.ctx)

    # ---- profiled re-run of the same K steps: HIP events around every sparse-conv launch --------
    ops.set_profiling(device, True)
    prof = {}
    for _ in range(args.steps):
        step(forced)
        st = ops.stage_times(device)
        for k, v in st.items():
            prof[k] = prof.get(k, 0.0) + v
    launch_ms, gemm_ms = ops.conv_launch_times(device)   # last profiled step: FCGF layers, then the inlier net's
    ops.set_profiling(device, False)
    prof = {k: v / args.steps for k, v in prof.items()}
    log(f'profiled region done: {prof}')

    if rank == 0:
        # algorithmic work of the conv kernel per step, from the kernel maps of this very input
        fc = dgr.fcgf_model._handle()
        ones0 = torch.ones(len(C0), 1, device=device)
        fc.forward(C0, ones0); s_a = fc.layer_stats()
        fc.forward(C1, torch.ones(len(C1), 1, device=device)); s_b = fc.layer_stats()
        coords6, feats6 = ops.inlier_inputs(C0, X0, C1, X1, torch.from_numpy(idx1).to(device),
                                            dgr.inlier_feature_type)
        inl = dgr.inlier_model._handle()
        inl.forward(coords6, feats6); s_c = inl.layer_stats()
        work = [conv_work(s) for s in (s_a, s_b, s_c)]
        flop = sum(w[0] for w in work); byts = sum(w[1] for w in work)
        flop64 = sum(w[2] for w in work); byts64 = sum(w[3] for w in work)
        n_launch = max(1, int(prof['conv_launches']))
        # the layers SURVEY.md 8d calls HBM-bound (C <= 64): compulsory bytes and gather-scatter traffic
        # (4 P (Cin + 2 Cout) + 8 P + 4 Kne Cin Cout) over their measured launch durations
        c64 = dominant = None
        if len(launch_ms) == len(s_a) + len(s_c):
            per_layer = [dict(a, pairs=a['pairs'] + b['pairs'], n_in=a['n_in'] + b['n_in'], n_out=a['n_out'] + b['n_out'],
                              nonempty=max(a['nonempty'], b['nonempty'])) for a, b in zip(s_a, s_b)] + list(s_c)
            ms64 = comp64 = gath64 = 0.0
            for st, ms in zip(per_layer, launch_ms):
                if max(st['cin'], st['cout']) <= 64:
                    ms64 += ms
                    comp64 += 4.0 * (st['n_in'] * st['cin'] + st['n_out'] * st['cout'] + st['nonempty'] * st['cin'] * st['cout']) + 8.0 * st['pairs']
                    gath64 += 4.0 * st['pairs'] * (st['cin'] + 2 * st['cout']) + 8.0 * st['pairs'] + 4.0 * st['nonempty'] * st['cin'] * st['cout']
            # the dominant kernel instance: sparse_conv_mfma_v2<256, 1, 4, 2, 2, true> = every 256 -> 256 layer
            # (block4 of both nets); its average launch duration is what `rocprofv3 --kernel-trace --stats` of
            # the single-stream command reports for that kernel name (profiles/)
            dom = [(2.0 * st['pairs'] * st['cin'] * st['cout'], g) for st, g in zip(per_layer, gemm_ms)
                   if st['cin'] == 256 and st['cout'] == 256]
            if dom:
                dominant = {'name': 'sparse_conv_mfma_v2<256, 1, 4, 2, 2, true>', 'launches_per_step': len(dom),
                            'avg_launch_us': 1e3 * sum(g for _, g in dom) / len(dom),
                            'gflop_per_step': sum(f for f, _ in dom) / 1e9,
                            'achieved_tflops': sum(f for f, _ in dom) / (sum(g for _, g in dom) * 1e-3) / 1e12,
                            'share_of_conv_flop': sum(f for f, _ in dom) / flop,
                            'algorithmic_bytes_per_launch': sum(
                                4.0 * (st['n_in'] * st['cin'] + st['n_out'] * st['cout'] + st['nonempty'] * st['cin'] * st['cout'])
                                + 8.0 * st['pairs'] for st in per_layer if st['cin'] == 256 and st['cout'] == 256) / len(dom)}
            if ms64 > 0:
                c64 = {'layers': sum(1 for st in per_layer if max(st['cin'], st['cout']) <= 64), 'ms_per_step': ms64,
                       'compulsory_gbps': comp64 / ms64 / 1e6, 'gather_scatter_gbps': gath64 / ms64 / 1e6,
                       'frac_of_hbm_peak_gather_scatter': gath64 / ms64 / 1e6 / PEAK_HBM_GBPS}
        conv_ms = prof['conv_kernels']
        achieved = flop / (conv_ms * 1e-3) / 1e12
        T_all, status_all, stats_all = gathered
        te, re = [], []
        for p in range(S * B):
            if status_all[p] == 0:
                Tg = all_pairs[p][2]
                te.append(float(np.linalg.norm(T_all[p][:3, 3] - Tg[:3, 3])))
                c = (np.trace(T_all[p][:3, :3].T @ Tg[:3, :3]) - 1) / 2
                re.append(float(np.degrees(np.arccos(np.clip(c, -1, 1)))))
        # HBM traffic of the conv kernels from PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, see
        # profiles/r01_conv_hbm_traffic.json): collected in separate rocprofv3 --pmc passes on the same
        # per-stream workload (4 pairs per batch); not measurable from inside this process
        # (a) the dominant kernel instance (roofline headline) and (b) all conv launches of the batch
        traffic = traffic_all = None
        tpath = os.path.join(ROOT, 'profiles', 'r01_conv_hbm_traffic.json')
        if os.path.exists(tpath) and B == 4 and args.n_raw == 50000 and args.conv1_ks == 7:
            tj = json.load(open(tpath))
            traffic_all = tj['per_conv_launch_bytes']['total']
            for kname, kv in tj['kernels'].items():
                if dominant and dominant['name'] in kname and kv['dispatches']:
                    traffic = (2.0 * kv['FETCH_SIZE_KB'] + kv['WRITE_SIZE_KB']) * 1024.0 / kv['dispatches']
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            'metric': 'pair registrations/sec (FCGF x2 + 1-NN + 6-D inlier net + gate + weighted Procrustes + SE(3) refinement)',
            'value': world * S * B * args.steps / elapsed, 'unit': 'pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic 3DMatch-shaped pairs, seeded synthetic weights, teacher-forced matches (20% GT) and inlier logits'
                    + ('; PCIe-inclusive: raw host points -> H2D -> voxelisation inside the timed region' if args.from_host else ''),
            'config': {'workload': f'{S * B} pairs/step/GPU ({S} stream(s) x {B}), {args.n_raw} raw pts/fragment, '
                                   f'{args.kind}, voxel {args.voxel}, conv1 k={args.conv1_ks} ({cfg_label(args)})',
                       'streams_per_gpu': S,
                       'voxels_per_pair': [int(off0[-1] / B), int(off1[-1] / B)],
                       'pairs_per_step_per_gpu': S * B, 'refinement': not args.no_refine,
                       'parallelism': f'pair-sharded x{world}, no data-path collective'},
            # roofline of the DOMINANT kernel (76 % of the conv FLOPs): algorithmic FLOP per launch / its average
            # launch duration (HIP events on the launch stream); `all_conv_layers` is the same over all 46 launches
            'roofline': {'bound': 'mfma', 'achieved': dominant['achieved_tflops'] if dominant else achieved,
                         'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': (dominant['achieved_tflops'] if dominant else achieved) / PEAK_FP32_MFMA_TFLOPS,
                         'traffic': traffic,
                         'traffic_unit': 'HBM bytes per launch of the kernel (PMC: 2 x FETCH_SIZE + WRITE_SIZE)',
                         'kernel': dominant['name'] if dominant else 'sparse_conv_mfma (all instances)',
                         'launches_per_step': dominant['launches_per_step'] if dominant else n_launch,
                         'avg_launch_us': dominant['avg_launch_us'] if dominant else conv_ms * 1e3 / n_launch,
                         'gflop_per_launch': (dominant['gflop_per_step'] / dominant['launches_per_step']) if dominant else flop / 1e9 / n_launch,
                         'share_of_conv_flop': dominant['share_of_conv_flop'] if dominant else 1.0,
                         'algorithmic_bytes_per_launch': dominant['algorithmic_bytes_per_launch'] if dominant else byts / n_launch,
                         'all_conv_layers': {'achieved': achieved, 'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                                             'kernel': 'sparse_conv_mfma_v2<*> + reduce_rows (every layer launch)',
                                             'launches_per_step': n_launch, 'avg_launch_us': conv_ms * 1e3 / n_launch,
                                             'gflop_per_step': flop / 1e9, 'compulsory_gbytes_per_step': byts / 1e9,
                                             'traffic': traffic_all, 'algorithmic_bytes_per_launch': byts / n_launch,
                                             'hbm_gbps_compulsory': byts / (conv_ms * 1e-3) / 1e9},
                         'c_le_64_layers': dict({'gflop': flop64 / 1e9, 'gbytes': byts64 / 1e9}, **(c64 or {}))},
            'stage_ms_per_step': {k: round(v, 3) for k, v in prof.items() if k != 'conv_launches'},
            'te_m_mean': float(np.mean(te)) if te else None, 're_deg_mean': float(np.mean(re)) if re else None,
            'status': [int(s) for s in status_all.tolist()],
            'iterations': [int(v) for v in stats_all[:, 0].tolist()],
            'voxelize_ms_per_pair': t_vox * 1e3,
            'layers_6d': [[x['pairs'], x['nonempty'], x['n_in'], x['n_out'], x['cin'], x['cout']] for x in s_c],
            'layers_3d_cloud0': [[x['pairs'], x['nonempty'], x['n_in'], x['n_out'], x['cin'], x['cout']] for x in s_a],
        }
        log('roofline accounting done')
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N = 1 only
            out['cpu_baseline'] = cpu_baseline(ck, args.n_raw, args.voxel, args.kind, (int(off0[-1] / B), int(off1[-1] / B)))
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
