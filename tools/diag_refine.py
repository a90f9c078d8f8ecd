import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from deepglobalregistration_amd import ops, synth
from oracle import pipeline as opipe, registration as oreg
VOX = 0.05
xyz0, xyz1, T_gt = synth.synth_pair(0, n_raw=6000)
op0, oc0, _ = opipe.preprocess(xyz0, VOX); op1, oc1, _ = opipe.preprocess(xyz1, VOX)
gt = synth.gt_correspondences(op0, op1, T_gt, VOX)
from oracle import resunet as oresunet, knn as oknn
ck = synth.synth_checkpoint(seed=0, voxel_size=VOX, feat_conv1_kernel_size=7)
oF0 = oresunet.resunet_forward(ck['state_dict'], oc0, np.ones((len(oc0), 1), np.float32), 3, 7, True)
oF1 = oresunet.resunet_forward(ck['state_dict'], oc1, np.ones((len(oc1), 1), np.float32), 3, 7, True)
oi1 = oknn.find_knn(oF0, oF1, nn_max_n=250).reshape(-1)
idx = np.where(gt >= 0, gt, oi1)
forced = synth.gt_forced_logits(op0, op1[idx], T_gt, VOX)
ow, owsum, thr = opipe.confidence_gate(forced, 0.05)
X, Y, w = op0.astype(np.float32), op1[idx].astype(np.float32), ow.astype(np.float32).reshape(-1, 1)
print('N', len(X), 'inliers', int((w > 0).sum()))
Xg, Yg, wg = (torch.from_numpy(a).cuda() for a in (X, Y, w))
for k in (1, 2, 5, 10, 20, 40, 80, 150):
    Ro, to, so = oreg.global_registration(X, Y, w, max_iter=k, max_break_count=10**9, break_threshold_ratio=1e-4, quantization_size=0.1)
    R, t, st = ops.se3_refine(Xg, Yg, wg, 0.1, k, 10**9, 1e-4)
    Rp, tp, sp = oreg.global_registration(X * np.float32(1 + 2.0**-23), Y, w, max_iter=k, max_break_count=10**9, break_threshold_ratio=1e-4, quantization_size=0.1)
    print(k, 'hip-vs-oracle', max(np.abs(R - Ro).max(), np.abs(t.reshape(-1) - to.reshape(-1)).max()), 'oracle-vs-ulp', max(np.abs(Rp - Ro).max(), np.abs(tp.reshape(-1) - to.reshape(-1)).max()), 'loss', st['loss'], so['loss'], st['iterations'], so['iterations'])
print('--- spread of the reference over 1-ulp-level perturbations at k = 150 and its first-step kick signs')
k = 150
Ro, to, so = oreg.global_registration(X, Y, w, max_iter=k, max_break_count=10**9, break_threshold_ratio=1e-4, quantization_size=0.1)
R1, t1, _ = oreg.global_registration(X, Y, w, max_iter=1, max_break_count=10**9, break_threshold_ratio=1e-4, quantization_size=0.1)
Rh1, th1, _ = ops.se3_refine(Xg, Yg, wg, 0.1, 1, 10**9, 1e-4)
Rh, th, sth = ops.se3_refine(Xg, Yg, wg, 0.1, k, 10**9, 1e-4)
print('hip  k=150 dev', max(np.abs(Rh - Ro).max(), np.abs(th.reshape(-1) - to.reshape(-1)).max()), 'loss', sth['loss'], 'first-step t', th1.reshape(-1))
print('ref  first-step t', t1.reshape(-1))
devs = []
for j in range(1, 13):
    s = np.float32(1 + ((-1) ** j) * j * 2.0 ** -23)
    Rp, tp, sp = oreg.global_registration(X * s, Y * np.float32(1 + (j % 3 - 1) * 2.0 ** -23), w, max_iter=k, max_break_count=10**9, break_threshold_ratio=1e-4, quantization_size=0.1)
    R1p, t1p, _ = oreg.global_registration(X * s, Y * np.float32(1 + (j % 3 - 1) * 2.0 ** -23), w, max_iter=1, max_break_count=10**9, break_threshold_ratio=1e-4, quantization_size=0.1)
    d = max(np.abs(Rp - Ro).max(), np.abs(tp.reshape(-1) - to.reshape(-1)).max())
    devs.append(d)
    print(j, 'dev', d, 'loss', sp['loss'], 'first-step t', t1p.reshape(-1))
print('max', max(devs))
