#!/usr/bin/env python
"""Benchmark of the DGR inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--pairs-per-step B] [--streams S] [--total-pairs P]

One "step" = one pass of the hot path (FCGF x2 -> 1-NN -> 6-D inputs -> 6-D inlier net -> gate ->
weighted Procrustes -> SE(3) refinement; `dgr_register_batch`) over synthetic 3DMatch-shaped pairs
(BASELINE.json configs[1]: 50k raw points per fragment, 5 cm voxels, conv1 k=7) whose voxelised
coordinates are already resident in HBM.  Per GPU: S HIP streams, each driven by its own host thread
with its own library context over ONE shared weight set, each registering batches of B pairs (pairs are independent
units, streams never exchange data).  Defaults: S = 4, B = 6, every stream on its OWN quarter of the GPU's compute units
(`dgr_ctx_create_partition_stream`: a CU-masked stream per context; `--no-cu-partition`: plain streams competing for all
CUs, the configuration of rounds 1-5 with S = 3 -- 378 against 396 pairs/s on the same box, tools/r06_runs/run42.sh).

* default (weak scaling): every rank registers its own S x B pairs per step;
* `--total-pairs P` (strong scaling, BASELINE configs[3] with P = 512): P pairs are dealt over the
  ranks by cost (N0 * N1, sorted, snake round-robin: `dist.deal_by_cost`), a step = one pass over all P.

With N > 1 the script runs one process per GPU under `torch.distributed.run` (it re-executes itself
under the launcher when started plainly with `--gpus N`); the weights come from rank 0 by ONE RCCL
broadcast, the results are gathered on rank 0, no collective on the data path.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md, dense v_mfma_f32_32x32x16_f16 / _bf16
# The conv kernels compute f32-level results on the f16 matrix pipe: every f32 operand is split into two f16 pieces
# under an exact power-of-two scale and a MAC costs THREE v_mfma_*_f16 products (conv_wide.hip, conv_dense.hip, conv_os.hip).  The
# headline `roofline` therefore prices the dominant kernel against the pipe it issues on: achieved = 3 x the
# algorithmic FLOP/s, peak = the dense f16 MFMA peak.  `roofline.f32_view` is the same launches counted as plain f32
# MACs against the f32 MFMA peak (what an exact-f32 kernel would be priced against); it can exceed 1 and is NOT the
# roofline fraction.  With DGR_EXACT_F32=1 the kernels issue v_mfma_f32_*_f32 and the f32 peak is the headline.
PRODUCTS_PER_MAC = 3
PEAK_HBM_GBPS = 8000.0          # spec

_T0 = time.time()


def log(msg):
    if int(os.environ.get('RANK', 0)) == 0:
        print(f'[bench +{time.time() - _T0:6.1f}s] {msg}', file=sys.stderr, flush=True)


def cfg_label(args):
    """Which BASELINE.json config the chosen flags correspond to (only the default is the headline)."""
    key = (args.kind, args.n_raw, args.voxel, args.conv1_ks)
    base = {('indoor', 50000, 0.05, 7): 'BASELINE configs[1]',
            ('outdoor', 120000, 0.3, 5): 'BASELINE configs[2]',
            ('indoor', 200000, 0.025, 7): 'BASELINE configs[4]'}.get(key, 'non-BASELINE configuration')
    if args.total_pairs and base == 'BASELINE configs[1]':
        return f'BASELINE configs[3] shape: {args.total_pairs} pairs sharded over the ranks'
    return base


# ----------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torch.distributed environment re-executes itself
# ----------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(n):
    one_gpu = bool(os.environ.get('DGR_BENCH_ONE_GPU')) or '--launch-check' in sys.argv
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n and not one_gpu:
        print(f'bench.py: --gpus {n} requested but this node exposes {ndev} GPU(s); no extrapolation '
              '(SURVEY.md 8e).  Run on a node with enough GPUs.', file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------
# roofline accounting
# ----------------------------------------------------------------------------------------------
def layer_bytes(s):
    """Compulsory bytes of one conv layer: every input row once, every output row once,
    every used weight slice once, one (in, out) int32 pair per map entry."""
    return 4.0 * (s['n_in'] * s['cin'] + s['n_out'] * s['cout'] + s['nonempty'] * s['cin'] * s['cout']) + 8.0 * s['pairs']


def layer_flop(s):
    return 2.0 * s['pairs'] * s['cin'] * s['cout']


def roofline_time_s(s):
    """max(compulsory bytes / HBM peak, FLOP / f32 MFMA peak) -- the model the coverage contract states."""
    return max(layer_bytes(s) / (PEAK_HBM_GBPS * 1e9), layer_flop(s) / (PEAK_FP32_MFMA_TFLOPS * 1e12))


def roofline_time_own_pipe_s(s, kind):
    """The same layer priced on the pipe its kernel ISSUES on: split-operand kernels (`f16x2`) spend three f16 MFMA
    products per MAC against the dense f16 peak; the others f32 MFMA.  A fraction of this time cannot exceed 1."""
    if 'f16x2' in kind:
        return max(layer_bytes(s) / (PEAK_HBM_GBPS * 1e9), PRODUCTS_PER_MAC * layer_flop(s) / (PEAK_F16_MFMA_TFLOPS * 1e12))
    return roofline_time_s(s)


def _group_launches(launches):
    g = {}
    for kind, us in launches:
        if us > 0:
            g.setdefault(kind, []).append(us)
    return g


def layer_gather_bytes(s):
    """SURVEY.md 8d's secondary figure: what a gather -> GEMM -> scatter-add implementation moves,
    4 P (Cin + 2 Cout) + 8 P + 4 K_ne Cin Cout."""
    return 4.0 * s['pairs'] * (s['cin'] + 2 * s['cout']) + 8.0 * s['pairs'] + 4.0 * s['nonempty'] * s['cin'] * s['cout']


# measured ceiling of the memory system for RANDOM 256-byte rows from a region beyond the 4-MB L2 of an XCD
# (tools/microbench/vmem_bw.hip, profiles/r03_vmem_load_ceiling.txt: 20 B/ns/CU from a 16-MB region = 5.1 TB/s chip-wide,
# 14 B/ns/CU = 3.6 TB/s from HBM): what bounds a kernel whose operand rows are gathered, whatever its arithmetic
RANDOM_ROW_CEILING_GBPS = 5100.0


# ----------------------------------------------------------------------------------------------
# CPU legs (rank 0, N = 1 only): parity of the timed batch's pair 0 against the oracle, and the
# oracle timed as the reported CPU baseline.  The oracle is the checker, never the measured product.
# ----------------------------------------------------------------------------------------------
def _host_checkpoint(ck):
    """The checkpoint with its state dicts as host arrays (after a broadcast they may be views of a device buffer)."""
    return {k: ({n: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for n, v in d.items()}
                if k.startswith('state_dict') and isinstance(d, dict) else d) for k, d in ck.items()}


def oracle_parity_and_baseline(ck, args, pair0, do_baseline):
    """pair0 = dict(xyz0, coords0, xyz1, coords1 [numpy, batch column 0], idx1 [local], F0, F1, logit, forced, device).
    Returns (parity dict, cpu_baseline dict | None)."""
    from oracle import knn as oknn, parity as oparity, pipeline as opipe, registration as oreg, resunet as oresunet
    from deepglobalregistration_amd import ops
    found = os.cpu_count() or 1
    threads = max(1, min(16, found))   # the oracle is many small CPU ops: beyond ~16 threads fork/join overhead wins
    torch.set_num_threads(threads)
    ks = ck['config']['feat_conv1_kernel_size']
    p0, c0, p1, c1 = pair0['xyz0'], pair0['coords0'], pair0['xyz1'], pair0['coords1']
    n0, n1 = len(p0), len(p1)
    t = {}
    net_runs = {'fcgf': [], 'inlier_net': []}
    n_runs = 3 if do_baseline else 1
    c6 = f6 = None
    if do_baseline:
        # warm-up (thread pool, allocator, the oracle's per-map caches are rebuilt every call): one FCGF forward and the 6-D
        # net on the first eighth of the correspondences
        oresunet.resunet_forward(ck['state_dict'], c1, np.ones((n1, 1), np.float32), 3, ks, True)
        c6w, f6w = opipe.inlier_inputs(p0, p1, c0, c1, np.arange(n0 // 8), pair0['idx1'][:n0 // 8])
        oresunet.resunet_forward(ck['state_dict_inlier'], c6w, f6w, 6, 3, False)
    for _ in range(n_runs):
        t0 = time.time()
        oF0 = oresunet.resunet_forward(ck['state_dict'], c0, np.ones((n0, 1), np.float32), 3, ks, True)
        oF1 = oresunet.resunet_forward(ck['state_dict'], c1, np.ones((n1, 1), np.float32), 3, ks, True)
        net_runs['fcgf'].append(time.time() - t0)
        t0 = time.time()
        c6, f6 = opipe.inlier_inputs(p0, p1, c0, c1, np.arange(n0), pair0['idx1'])
        ologit = oresunet.resunet_forward(ck['state_dict_inlier'], c6, f6, 6, 3, False).reshape(-1)
        net_runs['inlier_net'].append(time.time() - t0)
    t['fcgf'] = float(np.median(net_runs['fcgf']))
    t['inlier_net'] = float(np.median(net_runs['inlier_net']))
    parity = {'pair': pair0.get('pair', 0), 'voxels': [n0, n1],
              'dF': float(max(np.abs(pair0['F0'] - oF0).max(), np.abs(pair0['F1'] - oF1).max())),
              'dlogit_rel': float(np.abs(pair0['logit'] - ologit).max() / max(1e-12, np.abs(ologit).max())),
              'tolerance': 1e-4,
              'what': 'F0/F1 and the 6-D logits of ONE pair of the stream\'s last timed batch (HIP, dgr_register_batch) vs '
                      'oracle.resunet.resunet_forward on identical voxels / correspondences; R/t: the HIP refinement '
                      '(dgr_se3_refine) vs oracle.registration.global_registration on the same correspondences and '
                      'weights, both forced to the oracle\'s free-running iteration count (oracle/parity.py); computed '
                      'outside the timed region', 'stream': pair0.get('stream', 0)}
    # R / t (parity TE / RE): iteration-matched against the f32 reference; where that exceeds 1e-4 an F64
    # ARBITER (the reference algorithm evaluated in float64, oracle/parity.py) says which side is off -- the rule of
    # tests/helpers.py::assert_iteration_matched
    w, wsum, thr = opipe.confidence_gate(pair0['forced'])
    X, Y = p0, p1[pair0['idx1']]
    q = 2 * args.voxel
    dev = pair0['device']

    def hip_refine(Xa, Ya, wa, max_iter, max_break):
        return ops.se3_refine(torch.from_numpy(Xa).to(dev), torch.from_numpy(Ya).to(dev), torch.from_numpy(wa).to(dev),
                              q, max_iter, max_break, 1e-4)

    def hip_refine_from(Xa, Ya, wa, state, max_iter):
        return ops.se3_refine_from(torch.from_numpy(Xa).to(dev), torch.from_numpy(Ya).to(dev), torch.from_numpy(wa).to(dev),
                                   state, max_iter, q, 10 ** 9, 1e-4)
    if wsum >= thr and not args.no_refine:
        okw = dict(break_threshold_ratio=1e-4, quantization_size=q)
        Xn, Yn, wn = np.asarray(X, np.float32), np.asarray(Y, np.float32), np.asarray(w, np.float32).reshape(-1, 1)
        rp = oparity.iteration_matched(Xn, Yn, wn, hip_refine, tol=10.0, **okw)
        Ro, to = rp['R_oracle'], rp['t_oracle']
        c = (np.trace(rp['R_impl'].T @ Ro.astype(np.float64)) - 1) / 2
        d = max(rp['dR'], rp['dt'])
        parity.update({'dR': rp['dR'], 'dt': rp['dt'], 'refinement_iterations': rp['iterations'],
                       'parity_RE_deg': float(np.degrees(np.arccos(np.clip(c, -1, 1)))), 'parity_TE_m': rp['dt'] * rp['t_scale']})
        parity['rt_within_1e-4'] = bool(d <= 1e-4)
        parity['rt_ok'] = parity['rt_within_1e-4']
        if not parity['rt_within_1e-4'] or do_baseline:
            arb = oparity.f64_arbiter(Xn, Yn, wn, rp['iterations'], rp['R_impl'], rp['t_impl'], Ro, to, rp['t_scale'],
                                      family=not parity['rt_within_1e-4'], n_ulps=4, **okw)
            parity['f64_arbiter'] = dict(arb, what='|HIP - f64|, |f32 reference - f64| (and the f32 reference on a row permutation / '
                                                   '+-1..4 ulp inputs) at the same iteration count; f64 = oracle.registration.'
                                                   'global_registration(dtype=float64)')
            if not parity['rt_within_1e-4']:
                eh, e32, ef = arb['err_impl_f64'], arb['err_f32_f64'], arb['err_family_f64']
                parity['rt_not_farther_from_f64_than_reference'] = bool(eh <= max(1e-4, 1.5 * e32))
                ok = parity['rt_not_farther_from_f64_than_reference']
                if not ok:
                    rows = oparity.window_accuracy(Xn, Yn, wn, hip_refine_from, **okw)
                    a = np.array([r for r in rows if r[0] > 0], np.float64).reshape(-1, 3)
                    r32, rh = float(np.sqrt((a[:, 1] ** 2).mean())), float(np.sqrt((a[:, 2] ** 2).mean()))
                    parity['window_accuracy'] = {'rms_f32_ref_vs_f64': r32, 'rms_hip_vs_f64': rh, 'windows': len(a)}
                    ok = bool(eh <= max(1e-4, 1.5 * ef) and rh <= 2.0 * r32 + 1e-7)
                    parity['rt_note'] = ('R / t exceed 1e-4 at equal iteration counts on a chaotic input (Adam from a stationary '
                                         'start, core/registration.py:161-194): accepted iff the f32 reference on perturbed inputs '
                                         'lands as far from the f64 arbiter AND four HIP steps from any reference state are not '
                                         'farther from four f64 steps than the reference\'s own (DESIGN.md section 2)')
                parity['rt_ok'] = ok
    parity['features_logits_within_1e-4'] = bool(parity['dF'] < 1e-4 and parity['dlogit_rel'] < 1e-4)
    parity['within_1e-4'] = bool(parity['features_logits_within_1e-4'] and parity.get('rt_within_1e-4', True))
    parity['ok'] = bool(parity['features_logits_within_1e-4'] and parity.get('rt_ok', True))
    if not do_baseline:
        return parity, None
    # 1-NN: the whole search, chunked like the reference (nn_max_n = 250), one run
    t0 = time.time()
    oknn.find_knn(oF0, oF1, nn_max_n=250)
    t['knn'] = time.time() - t0
    runs = []
    for _ in range(3):
        t0 = time.time()
        if wsum >= thr:
            oreg.global_registration(p0, p1[pair0['idx1']], w, break_threshold_ratio=1e-4, quantization_size=2 * args.voxel)
        runs.append(time.time() - t0)
    t['registration'] = float(np.median(runs))
    total = sum(t.values())
    base = {'value': 1.0 / total, 'unit': 'pairs/s', 'cores': threads, 'host_cores_found': found, 'kind': 'port',
            'sample': f'pair 0 of the last timed batch at FULL size ({n0}/{n1} voxels): FCGF x2 and the 6-D net after one '
                      f'warm-up, median of {n_runs} runs each; the whole 1-NN search ({n0} x {n1} rows, chunks of 250 like '
                      f'the reference) one run; registration median of 3; voxelisation excluded; {threads} torch threads '
                      f'on a host that reports {found} cores (the oracle is thousands of small gather / mm / index_add '
                      'calls per forward: beyond ~16 threads their fork/join overhead outweighs the parallel work)',
            'stage_s': {k: round(v, 3) for k, v in t.items()},
            'stage_runs_s': dict({k: [round(x, 3) for x in v] for k, v in net_runs.items()},
                                 registration=[round(x, 3) for x in runs]),
            'reference_modules_in_container': {
                'note': 'the reference\'s own core/knn.py / core/registration.py timed on CPU tensors in the build '
                        'container (8 cores, torch 2.10; BASELINE.md section 2) -- /root/reference is absent on the GPU box',
                'find_knn_gpu_26422x24182x32_s': 25.96, 'weighted_procrustes_N26422_ms': 0.88,
                'GlobalRegistration_N26422_s': 1.19}}
    return parity, base


# ----------------------------------------------------------------------------------------------
def launch_check(args, rank, world, backend):
    """CPU-only check of the multi-rank plumbing (tests/test_bench_launcher_cpu.py): rendezvous, weight
    broadcast, cost-balanced dealing, max-over-ranks timing, result gather -- no GPU work."""
    import torch.distributed as dist
    from deepglobalregistration_amd import dist as ddist, synth
    ck = synth.synth_checkpoint(seed=0, feat_conv1_kernel_size=3, with_inlier=False) if rank == 0 else None
    ck = ddist.broadcast_checkpoint(ck, src=0, device=torch.device('cpu'))
    P = args.total_pairs or world * (args.streams or 4) * args.pairs_per_step
    lo, hi = ddist.shard_range(P, rank, world)
    rng = np.random.default_rng(1234)
    all_cost = rng.uniform(1.0, 2.0, P)                     # stand-in for N0 * N1 of the provisional block
    cost = ddist.all_gather_vector(all_cost[lo:hi], P, lo, device=torch.device('cpu'))
    mine = ddist.deal_by_cost(cost, world)[rank]
    T = np.tile(np.eye(4), (len(mine), 1, 1))
    T[:, 0, 3] = mine
    out = ddist.gather_results(T, np.zeros(len(mine), np.int32), np.zeros((len(mine), 4), np.float32), dst=0,
                               device=torch.device('cpu'))
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    cnt = torch.zeros(world, dtype=torch.int64)
    cnt[rank] = len(mine)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt)
    if rank == 0:
        ids = sorted(int(v) for v in out[0][:, 0, 3])
        print(json.dumps({'launch_check': True, 'n_gpus': world, 'requested_gpus': args.gpus, 'backend': backend,
                          'pairs': P, 'all_pairs_covered_once': ids == list(range(P)), 'max_over_ranks': float(tt.item()),
                          'pairs_per_rank': [int(v) for v in cnt.tolist()],
                          'weights_broadcast_keys': len(ck['state_dict'])}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--pairs-per-step', type=int, default=6, help='pairs per batch (one dgr_register_batch); round 5: 6 '
                    '(3 x 6 measured 365 pairs/s against 353 for 3 x 4 on one box, and every conv layer runs nearer its roofline)')
    ap.add_argument('--total-pairs', type=int, default=0, help='strong-scaling mode: this many pairs in total, dealt '
                    'over the ranks by cost; a step = one pass over all of them (BASELINE configs[3]: 512)')
    ap.add_argument('--n-raw', type=int, default=50000, help='raw points per fragment')
    ap.add_argument('--voxel', type=float, default=0.05)
    ap.add_argument('--kind', default='indoor', choices=['indoor', 'outdoor'])
    ap.add_argument('--conv1-ks', type=int, default=7)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle comparison of pair 0 (about 30 s of CPU)')
    ap.add_argument('--no-refine', action='store_true', help='ablation: stop after weighted Procrustes')
    ap.add_argument('--no-exact-leg', action='store_true', help='skip the short DGR_EXACT_F32=1 leg (a child process after the '
                    'timed region; it only runs together with the parity leg)')
    ap.add_argument('--from-host', action='store_true', help='PCIe-inclusive variant (NOT the headline): every step '
                    'starts from the raw float64 host points (H2D copy + GPU voxelisation inside the timed region)')
    ap.add_argument('--cu-shares', type=int, default=0, help='number of equal shares of the compute units (2 or 4) the contexts are '
                    'dealt over, context w on share w mod N; 0 = one share per context when S is 2 or 4 (the default)')
    ap.add_argument('--no-cu-partition', action='store_true', help='plain streams (every kernel of every stream competes for '
                    'all compute units) instead of one CU-masked stream per context on its own share (S = 2 or 4 only)')
    ap.add_argument('--streams', type=int, default=None, help='(default 4; 3 where the CU partition is not available) ''HIP streams per GPU, each driven by its own host '
                    'thread with its own library context and its own batches of pairs (independent units)')
    ap.add_argument('--full-register', action='store_true', help='register() as the reference ships it: final ICP on '
                    '(use_icp = True, core/deep_global_registration.py:78,317-322); NOT the headline configuration')
    ap.add_argument('--force-safeguard', action='store_true', help='every pair takes the safeguard branch (RANSAC over its '
                    'correspondences, :302-315): the confidence gate threshold is put out of reach')
    ap.add_argument('--rotate-sets', type=int, default=2, help='weak-scaling mode: every stream holds this many DISTINCT '
                    'batches of pairs and consecutive steps take them in turn, so that no step repeats the inputs of the '
                    'step before it (1 = every step re-registers the same pairs)')
    ap.add_argument('--dump-results', default=None, help='rank 0 writes the gathered per-pair results (pair ids, T, status, stats) '
                    'of the last timed step to this .npz (tests)')
    ap.add_argument('--launch-check', action='store_true', help='CPU-only check of the multi-rank plumbing')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    import torch.distributed as dist
    # test hooks (a 1-GPU box cannot run RCCL between two ranks): DGR_BENCH_BACKEND=gloo moves the tiny
    # collectives to the CPU, DGR_BENCH_ONE_GPU=1 puts every rank on device 0; the driver sets neither
    backend = os.environ.get('DGR_BENCH_BACKEND', 'gloo' if args.launch_check else 'nccl')
    if os.environ.get('DGR_BENCH_ONE_GPU'):
        local_rank = 0
    # DGR_BENCH_FORCE_PG=1: a one-rank run initialises the process group as well and executes every collective of the
    # multi-GPU path (object broadcast, the flat weight broadcast, result all-gather, MAX all-reduce, barriers) on a
    # one-rank communicator -- on a 1-GPU box that is the only way to run the RCCL code before an 8-GPU node does
    # (tests/test_gpu_bench_ranks.py)
    force_pg = os.environ.get('DGR_BENCH_FORCE_PG') == '1' and world == 1 and not args.launch_check
    if force_pg:
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ['DGR_DIST_FORCE_COLLECTIVES'] = '1'
    pg_up = world > 1 or force_pg
    if pg_up:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)')
    if args.launch_check:
        return launch_check(args, rank, world, backend)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    if pg_up:
        log(f'{dist.get_backend()} process group up: {dist.get_world_size()} rank(s) (backend "nccl" = RCCL on ROCm)'
            + (' -- forced one-rank group, every collective of the multi-GPU path runs' if force_pg else ''))

    from deepglobalregistration_amd import _lib, dist as ddist, ops, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration

    B = args.pairs_per_step
    S = max(1, args.streams if args.streams is not None else 4)
    if args.streams is None and not args.no_cu_partition and not args.cu_shares:
        # the default rests on CU-masked streams: where the runtime refuses one, three plain streams are the better default
        # (378 against 360 pairs/s for four; a scheduling choice -- the kernels and their results are the same)
        probe = _lib.new_ctx(device)
        _lib.use_ctx(probe)
        try:
            ops.partition_stream(device, 0, 4)
            ops.partition_stream(device, 0, 1)
        except Exception as e:
            log(f'no CU partition on this system ({e!r}): 3 plain streams')
            S, args.no_cu_partition = 3, True
        finally:
            _lib.use_ctx(None)
            _lib.load().dgr_ctx_destroy(probe)
    ck = synth.synth_checkpoint(seed=0, voxel_size=args.voxel, feat_conv1_kernel_size=args.conv1_ks) if rank == 0 else None
    coll_dev = device if backend == 'nccl' else torch.device('cpu')
    torch.cuda.synchronize()
    t_b = time.perf_counter()
    ck = ddist.broadcast_checkpoint(ck, src=0, device=coll_dev)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t_b
    log('checkpoint ready')

    # ---- which pairs does this rank register? --------------------------------------------------
    vox_cache = {}

    def voxelised(seed, keep=True):
        """Pair `seed`: raw points (kept only for --from-host), voxelised tensors on the device, ground-truth pose.
        `keep=False`: not cached (the cost pass of the strong-scaling mode looks at pairs this rank may not get)."""
        if seed in vox_cache:
            return vox_cache[seed]
        a, b, Tg = synth.synth_pair(seed, n_raw=args.n_raw, kind=args.kind)
        xa, ca, _ = ops.voxelize(a, args.voxel, 0, device)
        xb, cb, _ = ops.voxelize(b, args.voxel, 0, device)
        rec = (a if args.from_host else None, b if args.from_host else None, Tg, xa, ca, xb, cb)
        if keep:
            vox_cache[seed] = rec
        return rec

    if args.total_pairs:
        P = args.total_pairs
        lo, hi = ddist.shard_range(P, rank, world)          # provisional contiguous block: measure its costs
        if P < world:
            raise SystemExit(f'bench.py: --total-pairs {P} is fewer than the {world} ranks')
        costs = []
        for s_ in range(lo, hi):
            rec = voxelised(s_, keep=False)
            costs.append(float(len(rec[3])) * float(len(rec[5])))
            del rec
        cost = ddist.all_gather_vector(costs, P, lo, device=coll_dev)
        my_pairs = ddist.deal_by_cost(cost, world)[rank]    # sorted by N0 * N1, dealt snake round-robin
        log(f'strong scaling: {P} pairs dealt by cost, this rank {len(my_pairs)} (cost share '
            f'{cost[my_pairs].sum() / cost.sum():.4f})')
    else:
        # weak scaling: S x B own pairs per rank and step, R distinct sets of them taken in turn
        R = max(1, args.rotate_sets)
        my_pairs = [rank * S * B * R + i for i in range(S * B * R)]
    # batches of B pairs, dealt over the S streams (weak mode: stream w holds batches w, w + S, ...: one per set)
    batches = [my_pairs[i:i + B] for i in range(0, len(my_pairs), B)]
    per_stream = [batches[w::S] for w in range(S)]
    rotate = not args.total_pairs

    class Worker:
        """One HIP stream + one library context + its resident batches, driven by one host thread."""

        def __init__(self, wid, batch_ids):
            self.wid, self.batch_ids = wid, batch_ids
            self.ctx = _lib.new_ctx(device) if S > 1 else None
            self.stream = torch.cuda.Stream(device) if S > 1 else torch.cuda.current_stream(device)
            self.plain_stream, self.cus = self.stream, None
            if partition:
                # this context's own share of the compute units (a CU-masked stream owned by the library context)
                _lib.use_ctx(self.ctx)
                try:
                    self.stream = ops.partition_stream(device, wid % partition, partition)
                    self.cus = torch.cuda.get_device_properties(device).multi_processor_count // partition
                except Exception as e:   # a scheduling choice, not the product: say so and run on plain streams
                    log(f'stream {wid}: no CU partition ({e!r}); plain stream')
                _lib.use_ctx(None)
            self.results, self.result_ids, self.last_bt, self.k = [], [], None, 0

        def __enter__(self):
            _lib.use_ctx(self.ctx)
            self._sctx = torch.cuda.stream(self.stream)
            self._sctx.__enter__()
            return self

        def __exit__(self, *exc):
            self._sctx.__exit__(*exc)
            _lib.use_ctx(None)

        def prepare(self, shared_with=None):
            with self:
                # one weight set per GPU: the streams after the first share the first one's device-resident weights
                # (dgr_net_share: a net object per context over one reference-counted weight set)
                cfg = {'weights': ck, 'clip_weight_thresh': 0.05}
                if shared_with is not None:
                    cfg['share_weights_with'] = shared_with.dgr
                t_n = time.perf_counter()
                self.dgr = DeepGlobalRegistration(cfg, device)
                dgr = self.dgr
                dgr.fcgf_model._handle(); dgr.inlier_model._handle()
                torch.cuda.synchronize()
                self.net_s = time.perf_counter() - t_n     # state dict -> the kernels' operand layouts in HBM (or: share them)
                self.batches = []
                for ids in self.batch_ids:
                    x0, c0, x1, c1, off0, off1, ovr = [], [], [], [], [0], [0], []
                    for q, seed in enumerate(ids):
                        a, b, Tg, xa, ca, xb, cb = voxelised(seed)
                        ca, cb = ca.clone(), cb.clone()
                        ca[:, 0] = q; cb[:, 0] = q
                        # harness-only override (DESIGN.md "Synthetic workload"): untrained weights give ~0 %
                        # correct matches, so a share of the 1-NN results is replaced by ground-truth matches
                        # AFTER the search ran
                        g = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), Tg, args.voxel, seed=seed)
                        ovr.append(np.where(g >= 0, g + off1[-1], -1))
                        x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
                        off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
                    bt = {'ids': ids, 'C0': torch.cat(c0), 'X0': torch.cat(x0), 'C1': torch.cat(c1), 'X1': torch.cat(x1),
                          'off0': off0, 'off1': off1, 'ovr': torch.from_numpy(np.concatenate(ovr)).to(device),
                          'forced': None, 'raw': [(vox_cache[s][0], vox_cache[s][1]) for s in ids]}
                    # ... and the inlier logits by GT-derived ones AFTER the inlier net ran: one untimed call
                    # yields the final correspondences the forced logits are derived from
                    self.run_batch(bt)
                    torch.cuda.synchronize()
                    idx1 = ops.batch_output(device, 'idx1').cpu().numpy()
                    X0h, X1h = bt['X0'].cpu().numpy(), bt['X1'].cpu().numpy()
                    bt['idx1'] = idx1
                    bt['forced'] = torch.from_numpy(np.concatenate([
                        synth.gt_forced_logits(X0h[off0[q]:off0[q + 1]], X1h[idx1[off0[q]:off0[q + 1]]],
                                               vox_cache[s][2], args.voxel) for q, s in enumerate(ids)])).to(device)
                    self.batches.append(bt)
                # every input set runs at least once more, untimed, as a whole step (the preparation calls above ran
                # each of them once already, without the forced logits)
                self.warmup_steps = max(args.warmup, len(self.batches) if rotate else 1)
                for _ in range(self.warmup_steps):
                    self.step()
                torch.cuda.synchronize()
                self.k = 0   # the timed region starts with set 0

        def run_batch(self, bt):
            if args.from_host and bt['forced'] is not None:
                x0, c0, x1, c1 = [], [], [], []
                for q, (a, b) in enumerate(bt['raw']):
                    xa, ca, _ = self.dgr.preprocess(a, batch_index=q)
                    xb, cb, _ = self.dgr.preprocess(b, batch_index=q)
                    x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
                bt['C0'], bt['X0'], bt['C1'], bt['X1'] = torch.cat(c0), torch.cat(x0), torch.cat(c1), torch.cat(x1)
            forced = bt['forced']
            if args.force_safeguard and forced is not None:
                forced = bt.setdefault('forced_low', torch.full_like(forced, -20.0))   # no pair passes the confidence gate
            return self.dgr.register_voxelized(bt['C0'], bt['X0'], bt['off0'], bt['C1'], bt['X1'], bt['off1'],
                                               forced_logits=forced, skip_refinement=args.no_refine,
                                               override_idx1=bt['ovr'], safeguard=args.force_safeguard,
                                               icp=args.full_register)

        def step(self):
            """weak mode: the next of this stream's batches (set k mod R); strong mode: all of them."""
            todo = [self.batches[self.k % len(self.batches)]] if rotate else self.batches
            self.k += 1
            self.results = [self.run_batch(bt) for bt in todo]
            self.result_ids = [s for bt in todo for s in bt['ids']]
            self.last_bt = todo[-1]

        def run(self, n):
            c0 = time.thread_time()
            with self:
                for _ in range(n):
                    self.step()
            self.thread_cpu_s = time.thread_time() - c0   # CPU seconds of THIS driver thread alone (enqueue + wait)

        def run_profiled(self, n):
            """n more steps in the library's profiling mode, all workers of the rank at once: the dominant kernel stamps its
            own start and end on the device wall clock (conv_wide.hip), i.e. its execution span in the TIMED stream
            configuration -- the duration rocprofv3 --kernel-trace reports for it.  (HIP-event spans are useless here: with
            other streams on the GPU they include the wait for compute units.)"""
            self.prof_launches = []   # (kind, the kernel's own execution span in us [0 = not instrumented]) per conv launch
            with self:
                ops.set_profiling(device, True)
                for _ in range(n):
                    self.step()
                    kd = ops.conv_launch_kinds(device)
                    ku = ops.conv_launch_kernel_us(device)
                    if len(ku) == len(kd):
                        self.prof_launches += list(zip(kd, ku))
                ops.set_profiling(device, False)

    # shares of the compute units the contexts are dealt over (0: plain streams)
    partition = 0 if (args.no_cu_partition or S == 1) else (args.cu_shares or (S if S in (2, 4) else 0))
    if partition not in (0, 2, 4):
        raise SystemExit(f'bench.py: --cu-shares {partition}: 2 or 4 (dgr_ctx_create_partition_stream)')
    workers = [Worker(w, ids) for w, ids in enumerate(per_stream) if ids]
    for w in workers:
        w.prepare(shared_with=None if (w is workers[0] or os.environ.get('DGR_BENCH_PRIVATE_WEIGHTS')) else workers[0])
    # pairs this rank registers per step
    n_local = sum(len(w.batch_ids[0]) if rotate else sum(len(ids) for ids in w.batch_ids) for w in workers)
    b0 = workers[0].batches[0]
    log(f'{len(workers)} stream(s) ready: weights resident, {n_local} pairs voxelised (first batch N0={b0["off0"][-1]} '
        f'N1={b0["off1"][-1]}), warm-up done')

    def barrier():
        if pg_up:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    if len(workers) == 1:
        workers[0].run(args.steps)
    else:
        threads = [threading.Thread(target=w.run, args=(args.steps,)) for w in workers]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    barrier()
    elapsed = time.perf_counter() - t0
    # host CPU seconds (all threads of this rank) per step: the budget of N ranks x S driver threads on one node
    cpu_s_per_step = (time.process_time() - cpu0) / args.steps
    if pg_up:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    log(f'timed region done: {elapsed / args.steps * 1e3:.1f} ms/step ({n_local} pairs/step on this rank)')

    w0 = workers[0]
    st = torch.tensor([t_bcast, w0.net_s], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
    if pg_up:
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
    startup = {'checkpoint_broadcast': float(st[0]), 'weights_to_operand_layouts': float(st[1]),
               'weights_prepared_on_device_this_rank': bool(getattr(w0.dgr.inlier_model._handle(), 'created_on_device', False))}
    _lib.use_ctx(w0.ctx)
    # outputs of the LAST call of stream 0 (the last batch of the last timed step), before anything else runs there
    last_bt = w0.last_bt
    with torch.cuda.stream(w0.stream):
        hip_out = {k: ops.batch_output(device, k).cpu().numpy() for k in ('idx1', 'logit', 'F0', 'F1')}
    # ... and of every other stream's last call (the parity leg checks one pair per stream)
    stream_outs = [(w0.last_bt, hip_out)]
    for w in workers[1:]:
        _lib.use_ctx(w.ctx)
        with torch.cuda.stream(w.stream):
            stream_outs.append((w.last_bt, {k: ops.batch_output(device, k).cpu().numpy() for k in ('idx1', 'logit', 'F0', 'F1')}))
    _lib.use_ctx(w0.ctx)
    T = np.concatenate([r[0] for w in workers for r in w.results])
    status = np.concatenate([r[1] for w in workers for r in w.results])
    stats = np.concatenate([r[2] for w in workers for r in w.results])
    ids_local = [s for w in workers for s in w.result_ids]
    gathered = ddist.gather_results(T, status, stats, dst=0, device=coll_dev)
    gathered_ids = ddist.gather_results(np.tile(np.eye(4), (len(ids_local), 1, 1)) * np.asarray(ids_local)[:, None, None],
                                        np.zeros(len(ids_local), np.int32), np.zeros((len(ids_local), 4), np.float32),
                                        dst=0, device=coll_dev)

    # ---- the same stream configuration once more, profiled: every worker at once, HIP events around every conv launch on
    #      the worker's own stream (roofline.frac is quoted from HERE: the configuration the headline was timed in)
    n_prof_c = min(args.steps, 10)
    if len(workers) == 1:
        workers[0].run_profiled(n_prof_c)
    else:
        threads = [threading.Thread(target=w.run_profiled, args=(n_prof_c,)) for w in workers]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    torch.cuda.synchronize()
    timed_cfg_launches = [x for w in workers for x in w.prof_launches]
    log(f'profiled region in the timed stream configuration done ({len(timed_cfg_launches)} conv launches)')

    # ---- profiled re-run of stream 0's first batch alone: HIP events around every sparse-conv launch ----
    _lib.use_ctx(w0.ctx)
    cus_timed = w0.cus        # compute units a launch of the timed configuration ran on (None: all of them)
    if w0.cus:
        # ... alone on the WHOLE GPU: the per-kernel figures (frac_one_stream, by_kernel, the C <= 64 group, stage times,
        # the PMC passes of tools/evidence.sh) are the kernels' own, not those of a quarter of the compute units
        ops.partition_stream(device, 0, 1)
        w0.stream, w0.cus = w0.plain_stream, None
    bt = w0.batches[0]
    with torch.cuda.stream(w0.stream):
        ops.set_profiling(device, True)
        prof = {}
        n_prof = min(args.steps, 10)
        for _ in range(n_prof):
            w0.run_batch(bt)
            st = ops.stage_times(device)
            for k, v in st.items():
                prof[k] = prof.get(k, 0.0) + v
        launch_ms, gemm_ms = ops.conv_launch_times(device)   # last profiled call: FCGF layers, then the inlier net's
        kinds = ops.conv_launch_kinds(device)
        kernel_us = ops.conv_launch_kernel_us(device)          # the wide-layer kernels' own execution spans
        ops.set_profiling(device, False)
    prof = {k: v / n_prof for k, v in prof.items()}
    log(f'profiled region done: {prof}')

    if rank == 0:
        dgr = w0.dgr
        off0, off1, C0, X0, C1, X1, idx1 = bt['off0'], bt['off1'], bt['C0'], bt['X0'], bt['C1'], bt['X1'], bt['idx1']
        nb = len(bt['ids'])
        # algorithmic work of the conv kernels per batch, from the kernel maps of this very input
        with torch.cuda.stream(w0.stream):
            fc = dgr.fcgf_model._handle()
            C01 = torch.cat([C0 * torch.tensor([2, 1, 1, 1], device=device, dtype=torch.int32),
                             C1 * torch.tensor([2, 1, 1, 1], device=device, dtype=torch.int32)
                             + torch.tensor([1, 0, 0, 0], device=device, dtype=torch.int32)])
            fc.forward(C01, torch.ones(len(C01), 1, device=device)); s_a = fc.layer_stats()
            coords6, feats6 = ops.inlier_inputs(C0, X0, C1, X1, torch.from_numpy(idx1).to(device), dgr.inlier_feature_type)
            inl = dgr.inlier_model._handle()
            inl.forward(coords6, feats6); s_c = inl.layer_stats()
        per_layer = list(s_a) + list(s_c)
        flop = sum(layer_flop(s) for s in per_layer)
        byts = sum(layer_bytes(s) for s in per_layer)
        n_launch = max(1, int(prof['conv_launches']))
        groups, c64, dominant = {}, None, None
        if len(launch_ms) == len(per_layer) == len(kinds):
            # per kernel variant (name reported by the library): launches, time, algorithmic work
            for st, ms, gms, kind in zip(per_layer, launch_ms, gemm_ms, kinds):
                g = groups.setdefault(kind, {'launches': 0, 'ms': 0.0, 'main_kernel_ms': 0.0, 'gflop': 0.0, 'gbytes': 0.0,
                                             'roofline_ms': 0.0})
                g['launches'] += 1; g['ms'] += ms; g['main_kernel_ms'] += gms
                g['gflop'] += layer_flop(st) / 1e9; g['gbytes'] += layer_bytes(st) / 1e9
                g['roofline_ms'] += roofline_time_own_pipe_s(st, kind) * 1e3
            for kind, g in groups.items():
                g['pipe'] = 'f16 MFMA x3 products' if 'f16x2' in kind else 'f32 MFMA'
                g['tflops'] = g['gflop'] / max(g['ms'], 1e-9)
                g['frac_of_roofline'] = g['roofline_ms'] / max(g['ms'], 1e-9)   # on the pipe the kernel issues on: <= 1
            # the layers SURVEY.md 8d calls HBM-bound (C <= 64): COMPULSORY bytes over their measured durations
            sel = [(st, ms, kind) for st, ms, kind in zip(per_layer, launch_ms, kinds) if max(st['cin'], st['cout']) <= 64]
            if sel:
                ms64 = sum(ms for _, ms, _ in sel)
                comp64 = sum(layer_bytes(st) for st, _, _ in sel)
                gath64 = sum(layer_gather_bytes(st) for st, _, _ in sel)
                rows64 = sum(4.0 * st['pairs'] * st['cin'] for st, _, _ in sel)
                own64 = sum(roofline_time_own_pipe_s(st, kind) for st, _, kind in sel) * 1e3
                c64 = {'layers': len(sel), 'ms_per_batch': ms64, 'gflop': sum(layer_flop(st) for st, _, _ in sel) / 1e9,
                       'compulsory_gbytes': comp64 / 1e9, 'compulsory_gbps': comp64 / ms64 / 1e6,
                       'frac_of_hbm_peak_compulsory': comp64 / ms64 / 1e6 / PEAK_HBM_GBPS,
                       # SURVEY.md 8d's model (compulsory bytes at 8 TB/s | FLOP at the f32 MFMA peak): the contract's figure
                       'roofline_ms': sum(roofline_time_s(st) for st, _, _ in sel) * 1e3,
                       'frac_of_roofline': sum(roofline_time_s(st) for st, _, _ in sel) * 1e3 / ms64,
                       # the same layers on the pipes their kernels issue on (f16 x3 for the split-operand kernels)
                       'roofline_ms_own_pipe': own64, 'frac_of_roofline_own_pipe': own64 / ms64,
                       # gather view: SURVEY 8d's Bgather and the input rows alone (4 P Cin) over the measured time,
                       # against the measured rate of the memory system for random 256-byte rows beyond L2
                       'bgather_gbytes': gath64 / 1e9, 'bgather_gbps': gath64 / ms64 / 1e6,
                       'row_gather_gbytes': rows64 / 1e9, 'row_gather_gbps': rows64 / ms64 / 1e6,
                       'random_row_ceiling_gbps': RANDOM_ROW_CEILING_GBPS,
                       'bgather_frac_of_random_row_ceiling': gath64 / ms64 / 1e6 / RANDOM_ROW_CEILING_GBPS,
                       'row_gather_frac_of_random_row_ceiling': rows64 / ms64 / 1e6 / RANDOM_ROW_CEILING_GBPS}
            # the dominant kernel = the variant with the largest share of the conv time
            dk = max(groups, key=lambda k: groups[k]['main_kernel_ms'])
            sel = [(st, gms) for st, gms, kind in zip(per_layer, gemm_ms, kinds) if kind == dk]
            f = sum(layer_flop(st) for st, _ in sel)
            tms = sum(g for _, g in sel)
            dominant = {'name': dk, 'launches_per_batch': len(sel), 'avg_launch_us': 1e3 * tms / len(sel),
                        'gflop_per_launch': f / 1e9 / len(sel), 'achieved_tflops': f / (tms * 1e-3) / 1e12,
                        'share_of_conv_flop': f / flop, 'share_of_conv_time': tms / max(1e-9, sum(gemm_ms)),
                        'algorithmic_bytes_per_launch': sum(layer_bytes(st) for st, _ in sel) / len(sel)}
        conv_ms = prof['conv_kernels']
        achieved = flop / (conv_ms * 1e-3) / 1e12
        T_all, status_all, stats_all = gathered
        ids_all = [int(round(v)) for v in gathered_ids[0][:, 0, 0]]
        if args.dump_results:
            np.savez(args.dump_results, ids=np.asarray(ids_all), T=T_all, status=status_all, stats=stats_all)
        te, re = [], []
        for p, seed in enumerate(ids_all):
            if (int(status_all[p]) & 0xff) in (0, 3):   # estimated by the network path or by the safeguard RANSAC (flags masked)
                Tg = vox_cache[seed][2] if seed in vox_cache else synth.synth_pair(seed, n_raw=args.n_raw, kind=args.kind)[2]
                te.append(float(np.linalg.norm(T_all[p][:3, 3] - Tg[:3, 3])))
                c = (np.trace(T_all[p][:3, :3].T @ Tg[:3, :3]) - 1) / 2
                re.append(float(np.degrees(np.arccos(np.clip(c, -1, 1)))))
        # HBM traffic / MFMA-busy of the dominant kernel from PMC counters: separate rocprofv3 --pmc passes on
        # the same per-stream workload (tools/evidence.sh), committed under profiles/ -- not measurable from inside
        # this process; quoted only when kernel name and workload label match this run
        pmc, pmc_file = None, None
        for cand in ('r06_dominant_pmc.json', 'r05_dominant_pmc.json'):
            ppath = os.path.join(ROOT, 'profiles', cand)
            if os.path.exists(ppath):
                pj = json.load(open(ppath))
                # (a launch covers one layer of a whole batch: the counters belong to a batch size)
                if dominant and pj.get('kernel') and pj['kernel'] in dominant['name'] and \
                        pj.get('workload') in (f'{cfg_label(args)}, {B} pairs per batch',) + ((cfg_label(args),) if B == 4 else ()):
                    pmc, pmc_file = pj, cand
                    break
        split = bool(dominant and 'f16x2' in dominant['name'])     # the dominant kernel issues on the f16 pipe
        products = PRODUCTS_PER_MAC if split else 1
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        alg = dominant['achieved_tflops'] if dominant else achieved
        pairs_per_step = (args.total_pairs or world * n_local)
        ms_per_step = elapsed / args.steps * 1e3
        # the dominant kernel in the TIMED stream configuration (all streams of the rank at once, HIP events on each
        # stream): the duration roofline.frac is computed from; the one-stream figure next to it
        tc = [u for k_, u in timed_cfg_launches if dominant and k_ == dominant['name'] and u > 0]
        us_one = [u for k_, u in zip(kinds, kernel_us) if dominant and k_ == dominant['name'] and u > 0] if len(kernel_us) == len(kinds) else []
        # one stream: the kernel's own span where it stamps one (else the HIP-event span of its launch)
        us_one = float(np.mean(us_one)) if us_one else (dominant['avg_launch_us'] if dominant else None)
        us_timed = float(np.mean(tc)) if tc else us_one
        alg = (dominant['gflop_per_launch'] / (us_one * 1e-6) / 1e3) if (dominant and us_one) else alg
        gfl = dominant['gflop_per_launch'] if dominant else flop / 1e9 / n_launch
        alg_timed = gfl / (us_timed * 1e-6) / 1e3 if us_timed else alg        # TFLOP/s
        traffic = pmc.get('hbm_bytes_per_launch') if pmc else None
        alg_bytes = dominant['algorithmic_bytes_per_launch'] if dominant else byts / n_launch
        ncu_all = torch.cuda.get_device_properties(device).multi_processor_count
        # a launch of the timed configuration runs on this share of the CUs -- when its duration there is known (the kernel
        # stamped it); otherwise `us_timed` is the one-stream, whole-GPU duration and so is the peak
        share = (cus_timed / ncu_all) if (cus_timed and tc) else 1.0
        roofline = {
            # (timed configuration: `achieved` is ONE launch on its context's share of the compute units -- the other
            # contexts run theirs next to it --, `peak` the dense MFMA peak of that share; whole-GPU figures: *_one_stream)
            'bound': 'mfma', 'achieved': products * alg_timed, 'peak': peak * share, 'unit': 'TFLOP/s',
            'frac': products * alg_timed / (peak * share),
            'peak_whole_gpu': peak, 'cus_of_a_launch': int(round(share * ncu_all)), 'cus_of_the_gpu': ncu_all,
            # the same launch against the WHOLE GPU's peak (= frac x the share): what one context's launch reaches while the
            # other contexts run theirs on the other shares -- not comparable with frac_one_stream, which had all CUs
            'frac_vs_whole_gpu_peak': products * alg_timed / peak,
            'frac_note': (f'a launch of the timed configuration runs on {int(round(share * ncu_all))} of {ncu_all} compute units '
                          f'(its context\'s share, {len(workers)} contexts side by side): achieved / (peak_whole_gpu x share)')
                         if share < 1.0 else 'whole GPU',
            'traffic': traffic,
            'kernel': dominant['name'] if dominant else 'sparse conv (all variants)',
            # average duration of that kernel's launches in the timed configuration (S streams at once); the rocprofv3
            # --kernel-trace average of the same command is profiles/r06_kernel_stats_s4_b6.csv (tools/evidence.sh checks
            # that the two agree)
            'avg_launch_us': us_timed,
            'traffic_ratio': (traffic / alg_bytes) if traffic else None,           # PMC bytes / algorithmic bytes per launch
            # the C <= 64 gather/scatter layers (north_star: >= 0.40 of the HBM roofline): compulsory bytes over their
            # measured time as a fraction of 8 TB/s, and the same layers against the pipe their kernels issue on
            'c_le_64_hbm_frac': c64['frac_of_hbm_peak_compulsory'] if c64 else None,
            'c_le_64_frac_own_pipe': c64['frac_of_roofline_own_pipe'] if c64 else None,
            'c_le_64_compulsory_gbps': c64['compulsory_gbps'] if c64 else None,
            'exact_f32_pairs_per_s': None,    # filled below: the same workload with DGR_EXACT_F32=1 (v_mfma_f32_*_f32 on f32 operands)
            'products_per_mac': products,
            'pipe': 'dense f16 MFMA (v_mfma_f32_32x32x16_f16)' if split else 'f32 MFMA (v_mfma_f32_32x32x2_f32)',
            'algorithmic_bytes_per_launch': alg_bytes,
            'gflop_per_launch': gfl,
            'launches_per_batch': dominant['launches_per_batch'] if dominant else n_launch,
            'streams_when_measured': len(workers),
            'frac_one_stream': products * alg / peak,
            'avg_launch_us_one_stream': us_one if us_one else conv_ms * 1e3 / n_launch,
            'duration_source': 'stamped by the kernel itself on the device wall clock (earliest wave start to latest wave end): '
                               'the span rocprofv3 --kernel-trace reports; HIP-event span of the one-stream launch for comparison: '
                               + (f'{dominant["avg_launch_us"]:.0f} us' if dominant else 'n/a'),
            'mfma_busy_from_profiles': pmc.get('mfma_busy') if pmc else None,
            'traffic_source': (f'profiles/{pmc_file} (commit {pmc.get("commit", "not stamped")}): separate rocprofv3 --pmc passes '
                               'of the one-stream command, 2 x FETCH_SIZE (gfx950 correction, calibrated on a 2-GB read: '
                               'profiles/r05_mall_probe_counters.txt) + WRITE_SIZE, per launch; L2 <-> fabric bytes: '
                               'Infinity-Cache hits are counted') if pmc else None,
            'algorithmic_tflops': alg_timed,
            'limited_by': 'not the matrix pipe: the memory system -- one gathered KB, one product KB written and read back per '
                          'pair (rule-major tiles of 64 pairs) against 0.1 KB of algorithmic bytes; DESIGN.md 4.2'
                          if split else None,
            'share_of_conv_flop': dominant['share_of_conv_flop'] if dominant else 1.0,
            'share_of_conv_time': dominant['share_of_conv_time'] if dominant else 1.0,
            'c_le_64_ms_per_batch': c64['ms_per_batch'] if c64 else None,
            # SURVEY.md 8d's model prices these layers' FLOP at the f32 MFMA peak, which the f16-pipe kernels do not obey:
            # kept for continuity with rounds 1-5, NOT a roofline fraction (round-5 verdict, What's weak 2)
            'c_le_64_frac_vs_f32_model': c64['frac_of_roofline'] if c64 else None,
        }
        roofline_detail = {
            # the same launches counted as plain f32 MACs against the f32 MFMA peak (what an exact-f32 kernel is priced
            # against); NOT a roofline fraction of this kernel -- it can exceed 1
            'f32_view': {'achieved': alg, 'peak': PEAK_FP32_MFMA_TFLOPS, 'ratio': alg / PEAK_FP32_MFMA_TFLOPS},
            'all_conv_layers': {'achieved_algorithmic_tflops': achieved,
                                'launches_per_batch': n_launch, 'avg_launch_us': conv_ms * 1e3 / n_launch,
                                'gflop_per_batch': flop / 1e9, 'compulsory_gbytes_per_batch': byts / 1e9,
                                'hbm_gbps_compulsory': byts / (conv_ms * 1e-3) / 1e9,
                                'roofline_ms_f32_model': sum(roofline_time_s(s) for s in per_layer) * 1e3,
                                'frac_of_roofline_f32_model': sum(roofline_time_s(s) for s in per_layer) * 1e3 / conv_ms,
                                'frac_of_roofline_own_pipe': (sum(roofline_time_own_pipe_s(s_, k_) for s_, k_ in zip(per_layer, kinds)) * 1e3 / conv_ms
                                                              if len(kinds) == len(per_layer) else None)},
            'c_le_64_layers': c64,
            'by_kernel': {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in g.items()}
                          for k, g in groups.items()},
            # the instrumented (wide-layer) kernels in the timed stream configuration: launches seen, mean execution span
            'kernel_us_timed_config': {k: {'launches': len(v), 'mean_us': round(float(np.mean(v)), 1)}
                                       for k, v in _group_launches(timed_cfg_launches).items()},
            # every stamped span of the dominant kernel in that region, so that a profiler's per-launch table of the same
            # process can be matched launch by launch (tools/evidence.sh): the region is a few of the process's launches, and
            # what the other streams run next to the kernel differs from region to region
            'dominant_spans_us_timed_config': ([round(float(u), 1) for u in _group_launches(timed_cfg_launches).get(dominant['name'], [])][:96]
                                               if dominant else []),
        }
        out = {
            'metric': 'pair registrations/sec (FCGF x2 + 1-NN + 6-D inlier net + gate + weighted Procrustes + SE(3) refinement'
                      + (' + ICP' if args.full_register else '') + (', safeguard RANSAC forced' if args.force_safeguard else '') + ')',
            'value': pairs_per_step * args.steps / elapsed, 'unit': 'pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong' if args.total_pairs else 'weak', 'vs_baseline': None,
            'dtype': ('f32 results via 2 x f16 split operands (3 f16 MFMA products per MAC, f32 accumulate); k = 1 convs, '
                      'kNN re-evaluation and registration in exact f32, SVD in f64') if split else
                     'f32 (v_mfma_f32_*_f32 on the f32 operands: DGR_EXACT_F32=1); SVD in f64',
            'data': 'synthetic 3DMatch-shaped pairs, seeded synthetic weights, teacher-forced matches (20% GT) and inlier logits'
                    + ('; PCIe-inclusive: raw host points -> H2D -> voxelisation inside the timed region' if args.from_host else ''),
            'config': {'workload': f'{pairs_per_step} pairs/step ({world} GPU(s) x {len(workers)} stream(s) x batches of {B}), '
                                   f'{args.n_raw} raw pts/fragment, {args.kind}, voxel {args.voxel}, conv1 k={args.conv1_ks} '
                                   f'({cfg_label(args)})',
                       'streams_per_gpu': len(workers),
                       'cu_partition': (f'{len(workers)} contexts on {partition} shares of {cus_timed} of {ncu_all} compute units each '
                                        '(CU-masked streams, dgr_ctx_create_partition_stream)') if cus_timed else 'none (plain streams)',
                       'warmup_steps_run': workers[0].warmup_steps,
                       # weight sets resident per GPU (the streams share one: dgr_net_share) and their bytes
                       'weight_sets_per_gpu': (1 if workers[0].dgr.inlier_model._handle().sharers == len(workers) else len(workers)),
                       'weight_bytes_per_set': int(workers[0].dgr.fcgf_model._handle().param_bytes
                                                   + workers[0].dgr.inlier_model._handle().param_bytes),
                       'net_objects_sharing_the_set': int(workers[0].dgr.inlier_model._handle().sharers),
                       'voxels_per_pair': [int(off0[-1] / nb), int(off1[-1] / nb)],
                       'pairs_per_step': pairs_per_step,
                       'distinct_input_sets': (max(1, args.rotate_sets) if rotate else 1),
                       'refinement': not args.no_refine,
                       'use_icp': bool(args.full_register), 'forced_safeguard': bool(args.force_safeguard),
                       'parallelism': f'pair-sharded x{world}, no data-path collective'},
            # roofline of the DOMINANT conv kernel variant against the pipe it issues on (scalars first: the fields a reader
            # needs to reproduce `frac` by hand; the per-kernel / per-layer records are in `roofline_detail`)
            'roofline': roofline,
            'roofline_detail': roofline_detail,
            'host_cpu_s_per_step_per_rank': cpu_s_per_step,
            # ... of which the stream-driving threads themselves (enqueueing a batch's launches, then sleeping until it is
            # done: dgr_ctx_wait); the rest is the HIP runtime's own threads
            'host_cpu_s_per_step_driver_threads': sum(getattr(w, 'thread_cpu_s', 0.0) for w in workers) / args.steps,
            # one-time start-up of a rank, MAX over ranks: the RCCL broadcast of the checkpoint (ranks > 0 receive ~0.94 GB
            # into HBM and keep it there) and the preparation of the kernels' weight layouts (ranks > 0, and a forced
            # one-rank group: by HIP kernels from the broadcast buffer, dgr_net_create_device; the source rank: from its host
            # copy)
            'startup_s': startup,
            'stage_ms_per_batch': {k: round(v, 3) for k, v in prof.items() if k != 'conv_launches'},
            'te_m_mean': float(np.mean(te)) if te else None, 're_deg_mean': float(np.mean(re)) if re else None,
            'status_counts': {str(k): int((status_all == k).sum()) for k in np.unique(status_all)},
            'iterations_mean': float(stats_all[:, 0].mean()),
            'layers_6d': [[x['pairs'], x['nonempty'], x['n_in'], x['n_out'], x['cin'], x['cout']] for x in s_c],
            'layers_3d_batch': [[x['pairs'], x['nonempty'], x['n_in'], x['n_out'], x['cin'], x['cout']] for x in s_a],
        }
        log('roofline accounting done')
        out['parity'] = out['cpu_baseline'] = None
        if not args.no_parity and world == 1:
            # one pair per stream: pair (stream index mod B) of the stream's last timed batch; the CPU baseline is timed on
            # the first of them
            checks = []
            ck = _host_checkpoint(ck)
            for wi, (lb, ho) in enumerate(stream_outs):
                qi = wi % len(lb['ids'])
                s0, e0, s1, e1 = lb['off0'][qi], lb['off0'][qi + 1], lb['off1'][qi], lb['off1'][qi + 1]
                c0 = lb['C0'][s0:e0].cpu().numpy().copy(); c0[:, 0] = 0
                c1 = lb['C1'][s1:e1].cpu().numpy().copy(); c1[:, 0] = 0
                pair0 = {'xyz0': lb['X0'][s0:e0].cpu().numpy(), 'coords0': c0, 'xyz1': lb['X1'][s1:e1].cpu().numpy(), 'coords1': c1,
                         'idx1': ho['idx1'][s0:e0] - s1, 'F0': ho['F0'].reshape(-1, 32)[s0:e0],
                         'F1': ho['F1'].reshape(-1, 32)[s1:e1], 'logit': ho['logit'][s0:e0],
                         'forced': lb['forced'].cpu().numpy()[s0:e0], 'device': device, 'stream': wi, 'pair': qi}
                par, base = oracle_parity_and_baseline(ck, args, pair0, (not args.no_cpu_baseline) and wi == 0)
                checks.append(par)
                if wi == 0:
                    out['cpu_baseline'] = base
                log(f'parity (stream {wi}, pair {qi}): {par}')
            # (a pair whose R / t exceed 1e-4 carries the arbiter's numbers that decided it)
            out['parity'] = dict(checks[0], per_stream=[{k: c.get(k) for k in ('stream', 'pair', 'voxels', 'dF', 'dlogit_rel', 'dR', 'dt',
                                                                              'refinement_iterations', 'within_1e-4', 'ok',
                                                                              'rt_not_farther_from_f64_than_reference', 'window_accuracy')
                                                         + (('f64_arbiter',) if not c.get('rt_within_1e-4', True) else ())
                                                         if k in c}
                                                        for c in checks])
            # the stated tolerance next to the verdict that also accepts the reference's own chaos (f64 arbiter)
            out['config']['parity_within_1e-4'] = bool(all(c.get('within_1e-4') for c in checks))
            out['config']['parity_ok'] = bool(all(c.get('ok') for c in checks))
            out['config']['parity_pairs_checked'] = len(checks)
        # ---- the same workload on the reference's literal arithmetic (DGR_EXACT_F32=1: v_mfma_f32_*_f32 on the f32 operands),
        #      one short leg in a child process (the switch is read when the library first runs a layer); outside the timed region
        if world == 1 and not args.no_exact_leg and not args.no_parity and not os.environ.get('DGR_EXACT_F32'):
            try:
                cmd = [sys.executable, os.path.abspath(__file__), '--steps', '8', '--warmup', '2', '--no-parity', '--no-exact-leg',
                       '--pairs-per-step', str(B), '--streams', str(S), '--n-raw', str(args.n_raw), '--voxel', str(args.voxel),
                       '--kind', args.kind, '--conv1-ks', str(args.conv1_ks)] + (['--no-refine'] if args.no_refine else []) \
                    + (['--no-cu-partition'] if args.no_cu_partition else []) + ['--cu-shares', str(args.cu_shares)]
                cp = subprocess.run(cmd, env=dict(os.environ, DGR_EXACT_F32='1'), capture_output=True, text=True, timeout=300)
                line = [l for l in cp.stdout.splitlines() if l.startswith('{')][-1]
                ex = json.loads(line)
                out['roofline']['exact_f32_pairs_per_s'] = ex['value']
                out['exact_f32_leg'] = {'value': ex['value'], 'unit': 'pairs/s', 'steps': ex['steps'], 'ms_per_step': ex['ms_per_step'],
                                        'dtype': ex['dtype'], 'kernel': ex['roofline']['kernel'], 'frac_of_f32_mfma_peak': ex['roofline']['frac']}
                log(f'exact-f32 leg: {ex["value"]:.1f} pairs/s')
            except Exception as e:   # the leg is a report, not the product: say so and go on
                out['exact_f32_leg'] = {'error': repr(e)[:300]}
        print(json.dumps(out))
    if pg_up:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
