"""Seeded synthetic inputs and weights for tests and `bench.py`.

There is no dataset and no released checkpoint offline, so the harness uses
3DMatch-/KITTI-shaped synthetic pairs and a seeded state_dict in the
reference's checkpoint layout (`core/trainer.py:527-549`; MinkowskiEngine key
names `conv1.kernel [K,Cin,Cout]`, `norm1.bn.*`, `block1.conv1.kernel`, ...,
`final.kernel`, `final.bias [1,Cout]`).  numpy only; no torch, no HIP.
"""
import numpy as np

CHANNELS = [None, 32, 64, 128, 256]      # model/resunet.py:664 (ResUNetBN2C)
TR_CHANNELS = [None, 64, 64, 64, 128]    # model/resunet.py:665


# ----------------------------------------------------------------------------
# point clouds
# ----------------------------------------------------------------------------
def _random_rotation(rng, max_angle_deg=180.0):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, max_angle_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def _sample_box_surface(rng, n, lo, hi):
    """n points area-uniformly on the 6 faces of the axis-aligned box [lo,hi]."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    s = hi - lo
    areas = np.array([s[1] * s[2], s[1] * s[2], s[0] * s[2], s[0] * s[2], s[0] * s[1], s[0] * s[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    p = lo + rng.random((n, 3)) * s
    axis = face // 2
    side = face % 2
    p[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    return p


def _indoor_scene(rng, n):
    room = np.array([4.0, 3.5, 2.6])
    boxes = [(np.zeros(3), room, 1.0)]
    for _ in range(6):
        size = rng.uniform(0.3, 1.2, size=3)
        xy = rng.uniform([0.2, 0.2], room[:2] - size[:2] - 0.2)
        lo = np.array([xy[0], xy[1], 0.0])
        boxes.append((lo, lo + size, 0.6))
    area = np.array([2 * ((h - l)[0] * (h - l)[1] + (h - l)[0] * (h - l)[2] + (h - l)[1] * (h - l)[2]) * wgt
                     for l, h, wgt in boxes])
    counts = rng.multinomial(n, area / area.sum())
    pts = [_sample_box_surface(rng, c, l, h) for (l, h, _), c in zip(boxes, counts)]
    pts = np.concatenate(pts)
    return pts + rng.normal(scale=0.005, size=pts.shape), room


def _outdoor_scan(rng, origin, yaw, scene):
    """64-ring LiDAR model hitting a ground plane and boxes out to 80 m."""
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    azim = np.linspace(-np.pi, np.pi, 1900, endpoint=False) + yaw
    e, a = np.meshgrid(elev, azim, indexing='ij')
    d = np.stack((np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)), -1).reshape(-1, 3)
    o = np.asarray(origin, float)
    t_hit = np.full(len(d), np.inf)
    # ground plane z = 0
    down = d[:, 2] < -1e-6
    t_hit[down] = -o[2] / d[down, 2]
    # axis aligned boxes: slab test
    for lo, hi in scene:
        with np.errstate(divide='ignore', invalid='ignore'):
            t1 = (lo - o) / d
            t2 = (hi - o) / d
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        ok = (tmax >= tmin) & (tmin > 0)
        t_hit = np.where(ok & (tmin < t_hit), tmin, t_hit)
    keep = np.isfinite(t_hit) & (t_hit < 80.0)
    pts = o + d[keep] * t_hit[keep, None]
    return pts + rng.normal(scale=0.02, size=pts.shape)


def synth_pair(seed, n_raw=50000, kind='indoor', max_angle_deg=180.0, grid_align=None):
    """Returns xyz0 [n_raw,3], xyz1 [n_raw,3] float64 and the 4x4 ground truth T with
    xyz1 ~= T . xyz0 on the overlap (the quantity `register()` estimates).

    The pose is a rotation of up to `max_angle_deg` about a random axis and a translation in
    [-0.5, 0.5] m (optionally snapped to multiples of `grid_align`)."""
    rng = np.random.default_rng(seed)
    if kind == 'indoor':
        pts, room = _indoor_scene(rng, 6 * n_raw)
        v0 = pts[pts[:, 0] < 0.65 * room[0]]
        v1 = pts[pts[:, 0] > 0.35 * room[0]]
        v0 = v0[rng.permutation(len(v0))[:n_raw]]
        v1 = v1[rng.permutation(len(v1))[:n_raw]]
        assert len(v0) == n_raw and len(v1) == n_raw
        R = _random_rotation(rng, max_angle_deg)
        t = rng.uniform(-0.5, 0.5, size=3)
        if grid_align:
            t = np.round(t / grid_align) * grid_align
    elif kind == 'outdoor':
        scene = []
        for _ in range(40):
            c = rng.uniform([-70, -70], [70, 70])
            s = rng.uniform([2, 2, 1.5], [12, 12, 8])
            scene.append((np.array([c[0], c[1], 0.0]), np.array([c[0] + s[0], c[1] + s[1], s[2]])))
        v0 = _outdoor_scan(rng, [0, 0, 1.7], 0.0, scene)
        yaw = np.deg2rad(rng.uniform(-5, 5))
        v1w = _outdoor_scan(rng, [10.0, 0, 1.7], yaw, scene)
        # express each scan in its own sensor frame
        c, s = np.cos(yaw), np.sin(yaw)
        Rs = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        v0 = v0 - np.array([0, 0, 1.7])
        v1 = (v1w - np.array([10.0, 0, 1.7])) @ Rs          # world -> sensor-1 frame
        R = Rs.T
        t = -Rs.T @ np.array([10.0, 0, 0])
        if n_raw is not None and n_raw > 0:
            v0 = v0[rng.permutation(len(v0))[:n_raw]]
            v1 = v1[rng.permutation(len(v1))[:n_raw]]
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        return np.ascontiguousarray(v0), np.ascontiguousarray(v1), T
    else:
        raise ValueError(kind)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    v1 = v1 @ R.T + t
    return np.ascontiguousarray(v0), np.ascontiguousarray(v1), T


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
def conv_layer_specs(D, cin, cout, conv1_ks):
    """(name, K, Cin, Cout, bn_name|None, level) for every conv of ResUNetBN2C
    (`model/resunet.py:436-596`), in forward order."""
    C, T = CHANNELS, TR_CHANNELS
    k3 = 3 ** D
    specs = [('conv1', conv1_ks ** D, cin, C[1], 'norm1', 1)]

    def block(name, c, lvl):
        return [(name + '.conv1', k3, c, c, name + '.norm1', lvl),
                (name + '.conv2', k3, c, c, name + '.norm2', lvl)]
    specs += block('block1', C[1], 1)
    specs += [('conv2', k3, C[1], C[2], 'norm2', 2)] + block('block2', C[2], 2)
    specs += [('conv3', k3, C[2], C[3], 'norm3', 4)] + block('block3', C[3], 4)
    specs += [('conv4', k3, C[3], C[4], 'norm4', 8)] + block('block4', C[4], 8)
    specs += [('conv4_tr', k3, C[4], T[4], 'norm4_tr', 4)] + block('block4_tr', T[4], 4)
    specs += [('conv3_tr', k3, C[3] + T[4], T[3], 'norm3_tr', 2)] + block('block3_tr', T[3], 2)
    specs += [('conv2_tr', k3, C[2] + T[3], T[2], 'norm2_tr', 1)] + block('block2_tr', T[2], 1)
    specs += [('conv1_tr', 1, C[1] + T[2], T[1], None, 1), ('final', 1, T[1], cout, None, 1)]
    return specs


def _k_eff(D, name, K, lvl):
    """Rough mean neighbour count so that activations stay O(1) through 23 layers
    (measured on 3DMatch-shaped synthetic clouds)."""
    if K == 1:
        return 1.0
    if D == 3:
        return {343: 70.0, 125: 35.0}.get(K, 14.0)
    return {1: 2.0, 2: 3.0, 4: 6.0, 8: 30.0}[lvl]


def synth_state_dict(D, cin, cout, conv1_ks, seed, dtype=np.float32):
    """Seeded state_dict with MinkowskiEngine 0.5.x key names and shapes."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, K, ci, co, bn, lvl in conv_layer_specs(D, cin, cout, conv1_ks):
        std = np.sqrt(2.0 / (_k_eff(D, name, K, lvl) * ci))
        shape = (ci, co) if K == 1 else (K, ci, co)
        sd[name + '.kernel'] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(dtype)
        if bn is not None:
            sd[bn + '.bn.weight'] = rng.uniform(0.5, 1.5, co).astype(dtype)
            sd[bn + '.bn.bias'] = (rng.standard_normal(co) * 0.1).astype(dtype)
            sd[bn + '.bn.running_mean'] = (rng.standard_normal(co) * 0.1).astype(dtype)
            sd[bn + '.bn.running_var'] = rng.uniform(0.5, 1.5, co).astype(dtype)
            sd[bn + '.bn.num_batches_tracked'] = np.array(1000, np.int64)
    sd['final.bias'] = (rng.standard_normal((1, cout)) * 0.1).astype(dtype)
    return sd


def synth_checkpoint(seed=0, voxel_size=0.05, feat_conv1_kernel_size=7, n_out=32,
                     inlier_feature_type='coords', with_inlier=True):
    """Dict in the layout `DeepGlobalRegistration.__init__` reads
    (`core/deep_global_registration.py:88-129`)."""
    cfg = dict(voxel_size=voxel_size, feat_model='ResUNetBN2C', feat_model_n_out=n_out,
               bn_momentum=0.05, feat_conv1_kernel_size=feat_conv1_kernel_size,
               normalize_feature=True, inlier_model='ResUNetBN2C',
               inlier_conv1_kernel_size=3, inlier_feature_type=inlier_feature_type,
               nn_max_n=250)
    ck = {'config': cfg,
          'state_dict': synth_state_dict(3, 1, n_out, feat_conv1_kernel_size, seed)}
    if with_inlier:
        cin6 = 6 if inlier_feature_type == 'coords' else 1
        ck['state_dict_inlier'] = synth_state_dict(6, cin6, 1, 3, seed + 1)
    return ck


def gt_forced_logits(xyz0_corr, xyz1_corr, T_gt, voxel_size, magnitude=4.0):
    """Teacher-forced inlier logits: +magnitude where the correspondence agrees with the
    ground-truth pose within 2 voxels, -magnitude elsewhere (random weights make the
    learned confidence meaningless)."""
    p = xyz0_corr @ T_gt[:3, :3].T + T_gt[:3, 3]
    d = np.linalg.norm(p - xyz1_corr, axis=1)
    return np.where(d < 2 * voxel_size, magnitude, -magnitude).astype(np.float32).reshape(-1, 1)


def gt_correspondences(xyz0, xyz1, T_gt, voxel_size, frac=0.5, seed=0):
    """Teacher-forced correspondences for synthetic weights: idx [N0] int64 with, for a random
    `frac` of the rows of fragment 0 that have a fragment-1 voxel within one voxel size of their
    ground-truth position, the index of that voxel; -1 elsewhere (= keep the 1-NN result).

    Untrained (seeded random) FCGF weights give descriptors that are not repeatable across views,
    so the feature matcher alone yields ~0 % correct correspondences; real checkpoints give tens of
    percent.  Overriding a share of the matches restores a realistic 6-D neighbourhood structure
    (SURVEY.md section 8d measured 23 % correct) without removing any stage from the timed path."""
    from scipy.spatial import cKDTree
    p = np.asarray(xyz0, np.float64) @ T_gt[:3, :3].T + T_gt[:3, 3]
    d, j = cKDTree(np.asarray(xyz1, np.float64)).query(p, k=1)
    rng = np.random.default_rng(seed)
    take = (d < voxel_size) & (rng.random(len(p)) < frac)
    return np.where(take, j, -1).astype(np.int64)
