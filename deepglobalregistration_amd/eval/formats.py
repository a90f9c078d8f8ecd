"""Readers / writers of the file formats the reference's data loaders consume."""
import os
import struct

import numpy as np


# ---- gt.log trajectories (3DMatch geometric-registration benchmark) -----------------------------------
def read_trajectory(filename, dim=4):
    """util/file.py:69-90: a sequence of records "i j n" + dim rows of a dim x dim pose.  Returns a list of
    (metadata [int, ...], pose [dim, dim] float64)."""
    out = []
    with open(filename, 'r') as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    if len(lines) % (dim + 1):
        raise ValueError(f'{filename}: {len(lines)} non-empty lines is not a multiple of {dim + 1}')
    for r in range(0, len(lines), dim + 1):
        meta = [int(v) for v in lines[r].split()]
        pose = np.array([[float(v) for v in lines[r + 1 + k].split()] for k in range(dim)], np.float64)
        if pose.shape != (dim, dim):
            raise ValueError(f'{filename}: malformed pose in record {r // (dim + 1)}')
        out.append((meta, pose))
    return out


def write_trajectory(filename, records):
    with open(filename, 'w') as f:
        for meta, pose in records:
            f.write('\t'.join(str(int(v)) for v in meta) + '\n')
            for row in np.asarray(pose, np.float64):
                f.write('\t'.join(f'{v:.17g}' for v in row) + '\n')


# ---- KITTI velodyne scans -----------------------------------------------------------------------------
def read_kitti_bin(filename):
    """dataloader/kitti_loader.py:132-137: float32 records (x, y, z, reflectance); returns xyz [N,3] float32."""
    raw = np.fromfile(filename, dtype=np.float32)
    if raw.size % 4:
        raise ValueError(f'{filename}: size is not a multiple of 4 floats')
    return raw.reshape(-1, 4)[:, :3].copy()


def write_kitti_bin(filename, xyz, reflectance=None):
    xyz = np.asarray(xyz, np.float32)
    r = np.zeros(len(xyz), np.float32) if reflectance is None else np.asarray(reflectance, np.float32)
    np.concatenate([xyz, r[:, None]], 1).astype(np.float32).tofile(filename)


# ---- PLY ------------------------------------------------------------------------------------------------
_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2',
              'ushort': 'u2', 'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4',
              'float': 'f4', 'float32': 'f4', 'double': 'f8', 'float64': 'f8'}


def read_ply(filename):
    """Vertex positions [N,3] float64 of an ASCII or binary PLY (what `np.asarray(o3d.io.read_point_cloud(f)
    .points)` returns).  Other vertex properties and other elements are skipped; list properties are only
    supported on elements after `vertex` (e.g. faces), which are not read."""
    with open(filename, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'{filename}: not a PLY file')
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f'{filename}: unterminated PLY header')
            tok = line.decode('ascii', 'replace').split()
            if not tok or tok[0] == 'comment' or tok[0] == 'obj_info':
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == 'property':
                elements[-1][2].append(tok[1:])
            elif tok[0] == 'end_header':
                break
        if fmt not in ('ascii', 'binary_little_endian', 'binary_big_endian'):
            raise ValueError(f'{filename}: unsupported PLY format {fmt}')
        if not elements or elements[0][0] != 'vertex':
            raise ValueError(f'{filename}: the first element must be "vertex"')
        _, n, props = elements[0]
        if any(p[0] == 'list' for p in props):
            raise ValueError(f'{filename}: list properties on vertices are not supported')
        names = [p[1] for p in props]
        if not all(k in names for k in 'xyz'):
            raise ValueError(f'{filename}: vertex element has no x/y/z')
        if fmt == 'ascii':
            rows = [f.readline().split() for _ in range(n)]
            data = np.array(rows, dtype=np.float64).reshape(n, len(names)) if n else np.zeros((0, len(names)))
            return np.stack([data[:, names.index(k)] for k in 'xyz'], 1)
        order = '<' if fmt == 'binary_little_endian' else '>'
        dt = np.dtype([(p[1], order + _PLY_TYPES[p[0]]) for p in props])
        data = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
        return np.stack([data[k].astype(np.float64) for k in 'xyz'], 1)


def write_ply(filename, xyz, binary=True):
    xyz = np.asarray(xyz, np.float64)
    with open(filename, 'wb') as f:
        f.write(b'ply\nformat ' + (b'binary_little_endian' if binary else b'ascii') + b' 1.0\n')
        f.write(f'element vertex {len(xyz)}\nproperty double x\nproperty double y\nproperty double z\nend_header\n'.encode())
        if binary:
            f.write(xyz.astype('<f8').tobytes())
        else:
            for p in xyz:
                f.write(f'{p[0]:.17g} {p[1]:.17g} {p[2]:.17g}\n'.encode())


def load_cloud(filename):
    """xyz [N,3] from .ply / .bin (KITTI) / .npy / .npz (key 'pcd' like the reference's 3DMatch pairs, else
    'xyz' or the first array) / .txt."""
    ext = os.path.splitext(filename)[1].lower()
    if ext == '.ply':
        return read_ply(filename)
    if ext == '.bin':
        return read_kitti_bin(filename)
    if ext == '.npy':
        return np.load(filename)[:, :3]
    if ext == '.npz':
        z = np.load(filename)
        key = 'pcd' if 'pcd' in z.files else ('xyz' if 'xyz' in z.files else z.files[0])
        return z[key][:, :3]
    if ext in ('.txt', '.xyz'):
        return np.loadtxt(filename)[:, :3]
    raise ValueError(f'unrecognised point-cloud file type: {filename}')
