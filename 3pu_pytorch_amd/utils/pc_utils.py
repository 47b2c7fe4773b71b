"""Point-cloud I/O and host-side normalisation -- counterpart of the parts of the reference's
utils/pc_utils.py that the `--phase test` flow uses (normalize_point_cloud :11-25,
jitter_perturbation_point_cloud :28-42, load :223-241, save_ply :244-285) and of the augmentation
helpers the training data path uses (rotate_point_cloud_and_gt :45-79,
random_scale_point_cloud_and_gt :82-97).  numpy only (the
reference needs the third-party `plyfile`; the PLY subset it reads/writes -- a `vertex` element with
float x,y,z first -- is parsed here directly)."""
import os

import numpy as np


def normalize_point_cloud(input):
    """input: pc [N, P, 3] or [P, 3] -> pc, centroid, furthest_distance  (reference :11-25)"""
    axis = 0 if input.ndim == 2 else 1
    centroid = np.mean(input, axis=axis, keepdims=True)
    input = input - centroid
    furthest_distance = np.amax(np.sqrt(np.sum(input ** 2, axis=-1, keepdims=True)), axis=axis, keepdims=True)
    return input / furthest_distance, centroid, furthest_distance


def jitter_perturbation_point_cloud(batch_data, sigma=0.005, clip=0.02, is_2D=False):
    """Per-point Gaussian jitter, clipped (reference :28-42)."""
    B, N, C = batch_data.shape
    assert(clip > 0)
    chn = 2 if is_2D else 3
    jittered = np.clip(sigma * np.random.randn(B, N, C).astype(batch_data.dtype), -clip, clip)
    jittered[:, :, chn:] = 0
    return jittered + batch_data


def rotation_matrices(batch, dtype=np.float32):
    """`batch` random rotations Rz Ry Rx, drawn exactly like the reference's augmentation draws
    them (one np.random.uniform(size=3) per element, angles in [0, 2 pi), matrices built in `dtype`;
    reference :53-65).  Returned as (batch, 3, 3); points are rotated as p @ R."""
    out = np.empty((batch, 3, 3), dtype=dtype)
    for k in range(batch):
        angles = np.random.uniform(size=(3)) * 2 * np.pi
        Rx = np.array([[1, 0, 0],
                       [0, np.cos(angles[0]), -np.sin(angles[0])],
                       [0, np.sin(angles[0]), np.cos(angles[0])]], dtype=dtype)
        Ry = np.array([[np.cos(angles[1]), 0, np.sin(angles[1])],
                       [0, 1, 0],
                       [-np.sin(angles[1]), 0, np.cos(angles[1])]], dtype=dtype)
        Rz = np.array([[np.cos(angles[2]), -np.sin(angles[2]), 0],
                       [np.sin(angles[2]), np.cos(angles[2]), 0],
                       [0, 0, 1]], dtype=dtype)
        out[k] = np.dot(Rz, np.dot(Ry, Rx))
    return out


def rotate_point_cloud_and_gt(batch_data, batch_gt=None):
    """Random rotation per batch element, the same one for input and ground truth, in place
    (reference :45-79).  (B,N,3) or (B,N,6) with normals in the last three channels."""
    R = rotation_matrices(batch_data.shape[0], batch_data.dtype)
    for k in range(batch_data.shape[0]):
        for arr in (batch_data, batch_gt):
            if arr is None:
                continue
            arr[k, ..., 0:3] = np.dot(arr[k, ..., 0:3].reshape((-1, 3)), R[k])
            if arr.shape[-1] > 3:
                arr[k, ..., 3:] = np.dot(arr[k, ..., 3:].reshape((-1, 3)), R[k])
    return batch_data, batch_gt


def random_scale_point_cloud_and_gt(batch_data, batch_gt=None, scale_low=0.5, scale_high=2):
    """Random isotropic scale per batch element (reference :82-97)."""
    B = batch_data.shape[0]
    scales = np.random.uniform(scale_low, scale_high, (B, 1, 1)).astype(batch_data.dtype)
    batch_data = np.concatenate([batch_data[:, :, :3] * scales, batch_data[:, :, 3:]], axis=-1)
    if batch_gt is not None:
        batch_gt = np.concatenate([batch_gt[:, :, :3] * scales, batch_gt[:, :, 3:]], axis=-1)
    return batch_data, batch_gt, np.squeeze(scales)


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
              "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
              "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_ply(filename, count=None):
    """(P, 3+) float32 array of the `vertex` element's properties (x, y, z first)."""
    with open(filename, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % filename)
        fmt, nvert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header in %s" % filename)
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    nvert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list property inside the vertex element is not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=nvert, ndmin=2)
            names = [p[0] for p in props]
            pts = data[:, :len(names)]
        else:
            endian = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, endian + t) for n, t in props])
            rec = np.frombuffer(f.read(dt.itemsize * nvert), dtype=dt, count=nvert)
            names = list(rec.dtype.names)
            pts = np.stack([rec[n].astype(np.float64) for n in names], axis=1)
    order = [names.index(c) for c in ("x", "y", "z")] + [i for i, n in enumerate(names) if n not in ("x", "y", "z")]
    pts = pts[:, order].astype(np.float32)
    if count is not None and count < pts.shape[0]:
        pts = downsample_points(pts, count)
    return pts


def downsample_points(pts, K):
    """Random subset when the file holds more points than asked for (reference :100-126 uses a
    greedy farthest sampler with a random start; a uniform random subset keeps the file order
    independent of the host RNG-heavy loop -- the GPU FPS is applied afterwards anyway)."""
    if pts.shape[0] <= K:
        return pts
    keep = np.sort(np.random.choice(pts.shape[0], K, replace=False))
    return pts[keep]


def load(filename, count=None):
    """.ply / whitespace-separated text -> (P, C) float32; pad by random duplication or down-sample
    to `count` (reference :223-241)."""
    if filename[-4:] == ".ply":
        return read_ply(filename, count)[:, :3].astype(np.float32)
    points = np.loadtxt(filename).astype(np.float32)
    if count is not None:
        if count > points.shape[0]:
            tmp = np.zeros((count, points.shape[1]), dtype=points.dtype)
            tmp[:points.shape[0], ...] = points
            tmp[points.shape[0]:, ...] = points[np.random.choice(points.shape[0], count - points.shape[0]), :]
            points = tmp
        elif count < points.shape[0]:
            points = downsample_points(points, count)
    return points


def save_ply(points, filename, colors=None, normals=None):
    """binary little-endian PLY with float x,y,z (+ optional nx,ny,nz / uchar colours), the layout the
    reference writes through plyfile (:244-285)."""
    points = np.asarray(points, dtype=np.float32)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    cols = [points[:, 0], points[:, 1], points[:, 2]]
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
        cols += [normals[:, 0], normals[:, 1], normals[:, 2]]
    if colors is not None:
        colors = np.asarray(colors)
        if colors.max() <= 1:
            colors = colors * 255
        names = ["red", "green", "blue", "alpha"][:colors.shape[1]]
        fields += [(n, "u1") for n in names]
        cols += [colors[:, i] for i in range(len(names))]
    rec = np.empty(points.shape[0], dtype=np.dtype(fields))
    for (n, _), c in zip(fields, cols):
        rec[n] = c
    d = os.path.dirname(filename)
    if d and not os.path.exists(d):
        os.makedirs(d)
    ply_name = {"<f4": "float", "u1": "uchar"}
    with open(filename, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(("element vertex %d\n" % points.shape[0]).encode())
        for n, t in fields:
            f.write(("property %s %s\n" % (ply_name[t], n)).encode())
        f.write(b"end_header\n")
        f.write(rec.tobytes())
