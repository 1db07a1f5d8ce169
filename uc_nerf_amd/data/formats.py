"""On-disk formats either side of the ray-marching path: what the reference's datasets
read and its trainer writes, so that real SCARED / Hamlyn scenes and reference checkpoints flow through this package.

  * COLMAP binary models (`cameras.bin`, `images.bin`, `points3D.bin`) -- the reference reads them with
    colmapUtils/read_write_model.py:128-155, :225-258, :336-365 (COLMAP's published binary layout);
  * LLFF `poses_bounds.npy` and the pose normalisation the datasets apply (data/scared.py:51-66, :108-126, :222-244);
  * the sparse depth supervision projected from the COLMAP model (data/scared.py:285-344);
  * nearest-pose selection of source views (data/scared.py:69-106);
  * DPT depth priors (`*.npz`, key 'pred', data/scared.py:353-357);
  * the trainer's checkpoint pair (train.py:403-413, network/models.py:253-266).

Host-side numpy / torch-CPU only; the records are parsed from one buffer per file with `numpy.frombuffer` /
`struct.unpack_from` instead of per-field reads, the sparse depths are projected in one vectorised pass per image.
"""
import collections
import os
import struct

import numpy as np

Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
Point3D = collections.namedtuple("Point3D", ["id", "xyz", "rgb", "error", "image_ids", "point2D_idxs"])
_ImageBase = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])


class Image(_ImageBase):
    def qvec2rotmat(self):
        return qvec2rotmat(self.qvec)


# COLMAP camera models: id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


def qvec2rotmat(q):
    """Unit quaternion (w, x, y, z) -> rotation matrix (read_write_model.py:454-465)."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def read_cameras_binary(path):
    """cameras.bin: uint64 count, then per camera int32 id, int32 model, uint64 width, uint64 height, float64 params[]."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    cams = {}
    for _ in range(n):
        cid, model, w, h = struct.unpack_from("<iiQQ", buf, off)
        off += 24
        name, k = CAMERA_MODELS[model]
        cams[cid] = Camera(cid, name, w, h, np.frombuffer(buf, "<f8", k, off).copy())
        off += 8 * k
    return cams


def read_images_binary(path):
    """images.bin: uint64 count, then per image int32 id, float64 qvec[4], tvec[3], int32 camera id, zero-terminated name,
    uint64 number of 2D points, then (float64 x, float64 y, int64 point3D id) per point."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    rec = np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
    images = {}
    for _ in range(n):
        vals = struct.unpack_from("<idddddddi", buf, off)
        off += 64
        end = buf.index(b"\x00", off)
        name = buf[off:end].decode("utf-8")
        off = end + 1
        (m,) = struct.unpack_from("<Q", buf, off)
        off += 8
        pts = np.frombuffer(buf, rec, m, off)
        off += 24 * m
        images[vals[0]] = Image(vals[0], np.array(vals[1:5]), np.array(vals[5:8]), vals[8], name,
                                np.column_stack([pts["x"], pts["y"]]).astype(np.float64).reshape(m, 2), pts["id"].astype(np.int64))
    return images


def read_points3d_binary(path):
    """points3D.bin: uint64 count, then per point uint64 id, float64 xyz[3], uint8 rgb[3], float64 error, uint64 track
    length, (int32 image id, int32 point2D index) per track element."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    pts = {}
    for _ in range(n):
        pid, x, y, z, r, g, b, err = struct.unpack_from("<QdddBBBd", buf, off)
        off += 43
        (t,) = struct.unpack_from("<Q", buf, off)
        off += 8
        tr = np.frombuffer(buf, "<i4", 2 * t, off).reshape(t, 2)
        off += 8 * t
        pts[pid] = Point3D(pid, np.array([x, y, z]), np.array([r, g, b]), np.array(err), tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64))
    return pts


def colmap_c2w(images):
    """Camera-to-world matrices in dict order (data/scared.py:274-283)."""
    out = []
    for k in images:
        w2c = np.eye(4)
        w2c[:3, :3] = images[k].qvec2rotmat()
        w2c[:3, 3] = images[k].tvec
        out.append(np.linalg.inv(w2c))
    return np.array(out)


# ------------------------------------------------------------------------------------------------ poses
def _unit(v):
    return v / np.linalg.norm(v)


def average_poses(poses):
    """data/scared.py:108-126: mean centre, mean z, x = y_mean x z, y = z x x -> [3,4]."""
    center = poses[..., 3].mean(0)
    z = _unit(poses[..., 2].mean(0))
    x = _unit(np.cross(poses[..., 1].mean(0), z))
    return np.stack([x, np.cross(z, x), z, center], 1)


def center_poses(poses, blender2opencv):
    """data/scared.py:51-66: poses [N,3,4] expressed relative to their average pose, then the axis flip."""
    avg = np.eye(4)
    avg[:3] = average_poses(poses)
    homo = np.concatenate([poses, np.tile(np.array([0, 0, 0, 1.0]), (len(poses), 1, 1))], 1)
    inv = np.linalg.inv(avg)
    return ((inv @ homo) @ blender2opencv)[:, :3], (inv @ blender2opencv)[:, :3]


BLENDER2OPENCV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1.0]])


def load_poses_bounds(path, img_wh, blender2opencv=BLENDER2OPENCV, bd_factor=0.75):
    """LLFF poses_bounds.npy [N,17] -> the scene record of data/scared.py:222-244: focal scaled to `img_wh`, axes
    (down, right, back) -> (right, up, back), translation and bounds scaled by 1 / (min bound * 0.75), poses centred."""
    pb = np.load(path) if isinstance(path, (str, os.PathLike)) else np.asarray(path)
    poses = pb[:, :15].reshape(-1, 3, 5)
    bounds = pb[:, -2:].copy()
    H, W, focal = poses[0, :, -1]
    poses = np.concatenate([poses[..., 1:2], -poses[..., :1], poses[..., 2:4]], -1)
    scale = 1.0 / (bounds.min() * bd_factor)
    bounds *= scale
    poses[..., 3] *= scale
    centred, pose_avg = center_poses(poses, blender2opencv)
    return {"poses": centred, "pose_avg": pose_avg, "bounds": bounds, "raw_bounds": pb[:, -2:].transpose([1, 0]),
            "focal": [focal * img_wh[0] / W, focal * img_wh[1] / H], "scale_factor": scale}


def get_nearest_pose_ids(tar_pose, ref_poses, num_select, tar_id=-1):
    """data/scared.py:69-106 with the 'dist' criterion the datasets use: the `num_select` closest camera centres."""
    num_select = min(num_select, len(ref_poses) - 1)
    d = np.linalg.norm(tar_pose[None, :3, 3] - ref_poses[:, :3, 3], axis=1)
    if tar_id >= 0:
        d[tar_id] = 1e3
    return np.argsort(d)[:num_select]


# ------------------------------------------------------------------------------------------------ sparse depths
def colmap_sparse_depth(images, points, bds_raw, img_wh, factor, bd_factor=0.75):
    """data/scared.py:285-344: per registered image, the depths of its triangulated keypoints along the camera axis
    (scaled like the poses), kept when inside the image's [near, far], with weight 2 exp(-(err / mean err)^2), plus
    the dense depth / weight images at `img_wh`; the list is returned in image-name order.  `bds_raw` [2,N]."""
    errs = np.array([p.error for p in points.values()])
    err_mean = errs.mean()
    c2w = colmap_c2w(images)
    bds = np.moveaxis(np.asarray(bds_raw), -1, 0).astype(np.float32)
    sc = 1.0 if bd_factor is None else 1.0 / (bds.min() * bd_factor)
    ids = sorted(points)
    lut = {pid: k for k, pid in enumerate(ids)}
    xyz = np.stack([points[i].xyz for i in ids]) if ids else np.zeros((0, 3))
    perr = np.array([float(points[i].error) for i in ids])
    names, out = [], []
    for im in range(1, len(images) + 1):
        rec = images[im]
        names.append(rec.name)
        keep = rec.point3D_ids != -1
        rows = np.array([lut[i] for i in rec.point3D_ids[keep]], dtype=np.int64)
        xy = rec.xys[keep]
        pose = c2w[im - 1]
        depth = ((xyz[rows] - pose[:3, 3]) @ pose[:3, 2]) * sc
        ok = (depth >= bds[im - 1, 0] * sc) & (depth <= bds[im - 1, 1] * sc)
        w = (xy[:, 0] / factor).astype(np.int64)                       # int(): truncation (coordinates are non-negative)
        h = (xy[:, 1] / factor).astype(np.int64)
        ok &= (w < img_wh[0]) & (h < img_wh[1])
        depth, w, h = depth[ok], w[ok], h[ok]
        weight = 2 * np.exp(-(perr[rows][ok] / err_mean) ** 2)
        depth_img = np.zeros((img_wh[1], img_wh[0]))
        weight_img = np.zeros((img_wh[1], img_wh[0]))
        weight_img[h, w] = weight                                        # later keypoints overwrite earlier ones, as in the loop
        depth_img[h, w] = depth
        out.append({"name": rec.name, "depth": depth, "coord": np.stack([h, w], 1) if len(h) else np.zeros((0, 2), np.int64),
                    "weight": weight, "depth_img": depth_img, "weight_img": weight_img} if len(depth) else None)
    order = np.argsort(names)
    kept = [o for o in out if o is not None]
    if len(kept) != len(out):
        raise ValueError("colmap_sparse_depth: an image has no usable keypoint (the reference's list would misalign here)")
    return [out[i] for i in order]


# ------------------------------------------------------------------------------------------------ priors, checkpoints
def load_dpt_prior(path):
    """data/scared.py:353-357: `np.load(file)['pred']`, a leading singleton dimension dropped (resizing is the caller's)."""
    d = np.load(path)["pred"]
    return d[0] if d.shape[0] == 1 else d


def save_checkpoint(path, network_fn, network_mvs):
    """train.py:403-413: {'network_fn_state_dict', 'network_mvs_state_dict'} through torch.save."""
    import torch
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sd = lambda m: m if isinstance(m, dict) else m.state_dict()        # noqa: E731
    torch.save({"network_fn_state_dict": sd(network_fn), "network_mvs_state_dict": sd(network_mvs)}, path)


def load_checkpoint(path, network_fn=None, network_mvs=None, map_location="cpu"):
    """network/models.py:253-266: loads the pair into the given modules (strictly) and returns the raw dict."""
    import torch
    ckpt = torch.load(path, map_location=map_location)
    if network_mvs is not None:
        network_mvs.load_state_dict(ckpt["network_mvs_state_dict"])
    if network_fn is not None:
        network_fn.load_state_dict(ckpt["network_fn_state_dict"])
    return ckpt
