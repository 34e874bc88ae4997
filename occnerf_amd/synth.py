"""Synthetic SMPL-like body, skeleton, camera and per-frame inputs.

The licensed SMPL pickle and the ZJU-MoCap / OcMotion data are not available
(reference: third_parties/smpl/models/PUT_SMPL_MODEL_HERE), so every config in
BASELINE.json is driven by seeded synthetic inputs of the same shapes:

* a 6890-vertex / 24-joint T-pose body (capsule per bone) standing in for
  ``SMPL(pose=0, betas)`` (reference third_parties/smpl/smpl_numpy.py:13-102);
* the T-pose render camera and ray set (reference core/data/occnerf/tpose.py:66-84,
  133-217) and the free-view orbit (core/data/occnerf/freeview.py:133-142,
  core/utils/camera_util.py:9-110);
* the per-frame skeleton inputs (core/utils/body_util.py:219-350).

All maths is numpy on the host: these are *inputs* to the hot path
(SURVEY.md section 8(f) rows 2 and 15-16), not part of it.
"""
import itertools
from math import cos, sin

import numpy as np

TOTAL_BONES = 24
N_VERTS = 6890

SMPL_PARENT = {
    1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7,
    11: 8, 12: 9, 13: 9, 14: 9, 15: 12, 16: 13, 17: 14, 18: 16, 19: 17,
    20: 18, 21: 19, 22: 20, 23: 21}

HEAD_JOINT = 15
TORSO_JOINTS = (0, 3, 6, 9, 13, 14)

# T-pose joint positions in metres (x: body left, y: up, z: front), SMPL order.
_TPOSE_JOINTS = np.array([
    [0.000, -0.220, 0.020],    # 0 pelvis
    [0.070, -0.310, 0.010],    # 1 left hip
    [-0.070, -0.310, 0.010],   # 2 right hip
    [0.000, -0.100, -0.010],   # 3 spine1
    [0.100, -0.700, 0.010],    # 4 left knee
    [-0.100, -0.700, 0.010],   # 5 right knee
    [0.000, 0.030, 0.000],     # 6 spine2
    [0.090, -1.100, -0.030],   # 7 left ankle
    [-0.090, -1.100, -0.030],  # 8 right ankle
    [0.000, 0.090, 0.020],     # 9 spine3
    [0.110, -1.160, 0.090],    # 10 left foot
    [-0.110, -1.160, 0.090],   # 11 right foot
    [0.000, 0.300, -0.020],    # 12 neck
    [0.080, 0.200, -0.010],    # 13 left collar
    [-0.080, 0.200, -0.010],   # 14 right collar
    [0.000, 0.380, 0.030],     # 15 head
    [0.190, 0.230, -0.020],    # 16 left shoulder
    [-0.190, 0.230, -0.020],   # 17 right shoulder
    [0.450, 0.220, -0.040],    # 18 left elbow
    [-0.450, 0.220, -0.040],   # 19 right elbow
    [0.700, 0.220, -0.040],    # 20 left wrist
    [-0.700, 0.220, -0.040],   # 21 right wrist
    [0.790, 0.210, -0.050],    # 22 left hand
    [-0.790, 0.210, -0.050],   # 23 right hand
], dtype=np.float64)

# capsule radius of the bone that ends at joint i (index 0 unused)
_BONE_RADIUS = np.array([
    0.0, 0.095, 0.095, 0.125, 0.075, 0.075, 0.130, 0.050, 0.050, 0.135,
    0.040, 0.040, 0.065, 0.090, 0.090, 0.060, 0.060, 0.060, 0.048, 0.048,
    0.038, 0.038, 0.032, 0.032])
_HEAD_RADIUS = 0.100
_HEAD_OFFSET = np.array([0.0, 0.075, 0.010])


def tpose_joints(betas=None):
    """24 canonical joints; ``betas[0]`` scales the body height a little."""
    j = _TPOSE_JOINTS.copy()
    if betas is not None and len(betas) > 0:
        j *= 1.0 + 0.02 * float(np.asarray(betas).ravel()[0])
    return j


def _capsule(a, b, radius, segs, rings):
    """Closed lat-long capsule from a to b: segs*rings + 2 verts, outward winding."""
    axis = b - a
    h = float(np.linalg.norm(axis))
    w = axis / max(h, 1e-12)
    ref = np.array([1.0, 0.0, 0.0]) if abs(w[0]) < 0.9 else np.array([0.0, 0.0, 1.0])
    u = np.cross(w, ref)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    total = np.pi * radius + h           # profile arc length pole to pole
    verts = [a - radius * w]             # south pole
    for k in range(rings):
        s = (k + 0.5) / rings * total
        if s < 0.5 * np.pi * radius:                 # south cap
            ang = s / radius
            along, rad = -radius * cos(ang), radius * sin(ang)
        elif s < 0.5 * np.pi * radius + h:           # cylinder
            along, rad = s - 0.5 * np.pi * radius, radius
        else:                                        # north cap
            ang = (s - 0.5 * np.pi * radius - h) / radius
            along, rad = h + radius * sin(ang), radius * cos(ang)
        for m in range(segs):
            phi = 2.0 * np.pi * (m + 0.5 * (k & 1)) / segs
            verts.append(a + along * w + rad * (cos(phi) * u + sin(phi) * v))
    verts.append(b + radius * w)         # north pole
    verts = np.asarray(verts)
    faces = []
    ring = lambda k, m: 1 + k * segs + (m % segs)
    north = 1 + rings * segs
    for m in range(segs):
        faces.append((0, ring(0, m + 1), ring(0, m)))
        faces.append((north, ring(rings - 1, m), ring(rings - 1, m + 1)))
    for k in range(rings - 1):
        for m in range(segs):
            p00, p01 = ring(k, m), ring(k, m + 1)
            p10, p11 = ring(k + 1, m), ring(k + 1, m + 1)
            faces.append((p00, p01, p11))
            faces.append((p00, p11, p10))
    return verts, np.asarray(faces, dtype=np.int64)


def _part_list(joints):
    parts = []
    for i in range(1, TOTAL_BONES):
        parts.append((joints[SMPL_PARENT[i]], joints[i], _BONE_RADIUS[i]))
    scale = np.linalg.norm(joints[15] - joints[12]) / np.linalg.norm(
        _TPOSE_JOINTS[15] - _TPOSE_JOINTS[12])
    c = joints[HEAD_JOINT] + _HEAD_OFFSET * scale
    parts.append((c - np.array([0, 0.02, 0]) * scale, c + np.array([0, 0.02, 0]) * scale,
                  _HEAD_RADIUS * scale))
    return parts


def _plan_tessellation(parts, n_verts):
    """Pick (segs, rings) per part: area-proportional, total exactly n_verts."""
    budget = n_verts - 2 * len(parts)
    area = np.array([2 * np.pi * r * (np.linalg.norm(b - a) + 2 * r) for a, b, r in parts])
    spacing = np.sqrt(area.sum() / budget)
    segs = np.array([max(6, int(round(2 * np.pi * r / spacing))) for _, _, r in parts])
    rings = np.maximum(4, np.round(area / area.sum() * budget / segs).astype(int))
    delta = budget - int(np.sum(segs * rings))
    # absorb the remainder by nudging ring counts of the biggest parts
    order = list(np.argsort(-area)[:4])
    best = None
    for nudge in itertools.product(range(-8, 9), repeat=len(order)):
        if sum(n * segs[p] for n, p in zip(nudge, order)) == delta:
            cost = sum(abs(n) for n in nudge)
            if best is None or cost < best[0]:
                best = (cost, nudge)
    if best is None:
        raise RuntimeError("cannot tessellate body to the requested vertex count")
    for n, p in zip(best[1], order):
        rings[p] += n
    assert int(np.sum(segs * rings)) == budget and rings.min() >= 3
    return segs, rings


def vertex_normals(verts, faces):
    """Area-weighted vertex normals (float64), the stand-in for
    ``trimesh.Trimesh(...).vertex_normals`` (reference network.py:94-98)."""
    verts = np.asarray(verts, dtype=np.float64)
    tri = verts[faces]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])   # |fn| = 2*area
    vn = np.zeros_like(verts)
    for c in range(3):
        np.add.at(vn, faces[:, c], fn)
    norm = np.linalg.norm(vn, axis=1, keepdims=True)
    return vn / np.maximum(norm, 1e-20)


class SyntheticSMPL:
    """Drop-in for ``third_parties.smpl.smpl_numpy.SMPL`` (smpl_numpy.py:13-102):
    ``model(pose, beta) -> (verts[6890,3], joints[24,3])`` plus ``.faces``.
    Only the zero pose is meaningful (it is the only one the renderer asks
    for, network.py:93)."""

    def __init__(self, sex='neutral', model_dir=None):
        self._cache = {}
        _, self.faces = self._build(np.zeros(10))

    def _build(self, beta):
        key = float(np.asarray(beta).ravel()[0]) if np.size(beta) else 0.0
        if key not in self._cache:
            joints = tpose_joints([key])
            parts = _part_list(joints)
            segs, rings = _plan_tessellation(parts, N_VERTS)
            vs, fs, base = [], [], 0
            for (a, b, r), s, k in zip(parts, segs, rings):
                v, f = _capsule(a, b, r, int(s), int(k))
                vs.append(v)
                fs.append(f + base)
                base += v.shape[0]
            verts = np.concatenate(vs, 0)
            # seeded 1 mm jitter: a mirror-symmetric body would put every sample on the
            # x=0 plane exactly equidistant from mirrored vertices (kNN ties)
            verts = verts + np.random.RandomState(6890).uniform(-1e-3, 1e-3, verts.shape)
            self._cache[key] = (verts, np.concatenate(fs, 0).astype('int32'))
        return self._cache[key]

    def __call__(self, pose, beta, trans=None):
        verts, _ = self._build(beta)
        joints = tpose_joints(np.asarray(beta).ravel()[:1])
        verts, joints = verts.copy(), joints.copy()
        if trans is not None:
            verts += np.asarray(trans).reshape(1, 3)
            joints += np.asarray(trans).reshape(1, 3)
        return verts, joints


# ---------------------------------------------------------------------------
# skeleton -> per-frame transforms (reference core/utils/body_util.py)
# ---------------------------------------------------------------------------

def rodrigues(rvec):
    """Axis-angle -> rotation matrix, body_util.py:193-216 semantics
    (axis normalised by ``norm + 1e-5``)."""
    rvec = np.asarray(rvec, dtype=np.float64).reshape(3, 1)
    theta = float(np.linalg.norm(rvec))
    r = rvec / (theta + 1e-5)
    rx, ry, rz = r.ravel()
    skew = np.array([[0, -rz, ry], [rz, 0, -rx], [-ry, rx, 0]])
    return cos(theta) * np.eye(3) + sin(theta) * skew + (1 - cos(theta)) * r.dot(r.T)


def rodrigues_exact(rvec):
    """cv2.Rodrigues(rvec)[0] equivalent (exact unit axis)."""
    rvec = np.asarray(rvec, dtype=np.float64).ravel()
    theta = float(np.linalg.norm(rvec))
    if theta < 1e-12:
        return np.eye(3)
    k = rvec / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + sin(theta) * K + (1 - cos(theta)) * K.dot(K)


def body_pose_to_body_RTs(jangles, tpose_jts):
    """body_util.py:219-246: per-joint local rotation + offset from parent."""
    jangles = np.asarray(jangles).reshape(-1, 3)
    n = jangles.shape[0]
    Rs = np.zeros((n, 3, 3), dtype='float32')
    Ts = np.zeros((n, 3), dtype='float32')
    Rs[0] = rodrigues(jangles[0])
    Ts[0] = tpose_jts[0]
    for i in range(1, n):
        Rs[i] = rodrigues(jangles[i])
        Ts[i] = tpose_jts[i] - tpose_jts[SMPL_PARENT[i]]
    return Rs, Ts


def get_canonical_global_tfms(canonical_joints):
    """body_util.py:249-271: chain of pure translations down the kinematic tree."""
    n = canonical_joints.shape[0]
    g = np.zeros((n, 4, 4), dtype='float32')

    def G(t):
        m = np.eye(4, dtype='float32')
        m[:3, 3] = t
        return m

    g[0] = G(canonical_joints[0])
    for i in range(1, n):
        g[i] = g[SMPL_PARENT[i]].dot(G(canonical_joints[i] - canonical_joints[SMPL_PARENT[i]]))
    return g


_BONE_STDS = np.array([0.03, 0.06, 0.03])
_HEAD_STDS = np.array([0.06, 0.06, 0.06])
_JOINT_STDS = np.array([0.02, 0.02, 0.02])


def _align_rotation(v1, v2):
    """Rotation taking unit(v1) to unit(v2) (body_util.py:80-112)."""
    v1 = v1 / np.clip(np.linalg.norm(v1), 1e-5, None)
    v2 = v2 / np.clip(np.linalg.norm(v2), 1e-5, None)
    n = np.cross(v1, v2)
    c = float(v1.dot(v2))
    K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]], dtype=np.float32)
    return (np.eye(3) + K + K.dot(K) * (1.0 / (1.0 + c))).astype(np.float32)


def _gaussian_volume(grid_size, bmin, bmax, center, S, R):
    sigma = R.dot(S).dot(S).dot(R.T)
    zg, yg, xg = np.meshgrid(np.linspace(bmin[2], bmax[2], grid_size),
                             np.linspace(bmin[1], bmax[1], grid_size),
                             np.linspace(bmin[0], bmax[0], grid_size), indexing='ij')
    g = np.stack([xg - center[0], yg - center[1], zg - center[2]], axis=-1)
    d = np.einsum('abci,abci->abc', np.einsum('abci,ij->abcj', g, sigma), g)
    return np.exp(-d)


def approx_gaussian_bone_volumes(tpose_jts, bbox_min_xyz, bbox_max_xyz, grid_size=32):
    """Gaussian bone-weight prior [25, G, G, G] (body_util.py:274-350)."""
    tpose_jts = tpose_jts.astype(np.float32)
    n = tpose_jts.shape[0]
    up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
    vols = []
    for j in range(n):
        vol = np.zeros((grid_size,) * 3, dtype='float32')
        is_parent = False
        for bone, parent in SMPL_PARENT.items():
            if parent != j:
                continue
            S = np.diag(1.0 / (_BONE_STDS * 2.0)).astype(np.float32)
            if j in TORSO_JOINTS:
                S[0, 0] *= 1 / 1.5
                S[2, 2] *= 1 / 1.5
            start, end = tpose_jts[parent], tpose_jts[bone]
            R = _align_rotation(up, end - start)
            vol = vol + _gaussian_volume(grid_size, bbox_min_xyz, bbox_max_xyz,
                                         (start + end) / 2.0, S, R)
            is_parent = True
        if not is_parent:
            stds = _HEAD_STDS if j == HEAD_JOINT else _JOINT_STDS
            S = np.diag(1.0 / (stds * 2.0)).astype(np.float32)
            vol = _gaussian_volume(grid_size, bbox_min_xyz, bbox_max_xyz, tpose_jts[j], S,
                                   np.eye(3, dtype='float32'))
        vols.append(vol)
    vols = np.stack(vols, 0)
    bg = 1.0 - np.sum(vols, axis=0, keepdims=True).clip(min=0.0, max=1.0)
    vols = np.concatenate([vols, bg], 0)
    return vols / np.sum(vols, axis=0, keepdims=True).clip(min=0.001)


# ---------------------------------------------------------------------------
# camera and rays (reference core/utils/camera_util.py)
# ---------------------------------------------------------------------------

def get_camrot(campos, lookat=None, inv_camera=False):
    """camera_util.py:53-82."""
    if lookat is None:
        lookat = np.zeros(3, dtype=np.float32)
    up = np.array([0.0, -1.0 if inv_camera else 1.0, 0.0], dtype=np.float32)
    fwd = lookat - campos
    fwd = fwd / np.linalg.norm(fwd)
    right = np.cross(up, fwd)
    right = right / np.linalg.norm(right)
    up = np.cross(fwd, right)
    up = up / np.linalg.norm(up)
    return np.array([right, up, fwd], dtype=np.float32)


def setup_camera(img_size, radius=6.0, focal=1250.0):
    """tpose.py:66-84; focal is scaled with the render size (1250 at 512)."""
    y = -0.25
    campos = np.array([0.0, y, radius], dtype='float32')
    camrot = get_camrot(campos, lookat=np.array([0, y, 0.0]), inv_camera=True)
    E = np.eye(4, dtype='float32')
    E[:3, :3] = camrot
    E[:3, 3] = -camrot.dot(campos)
    K = np.eye(3, dtype='float32')
    K[0, 0] = K[1, 1] = focal * img_size / 512.0
    K[:2, 2] = img_size / 2.0
    return K, E


def rotate_camera(extrinsics, angle, trans=None, rotate_axis='y'):
    """camera_util.py:9-50: orbit the camera around a world axis."""
    inv_E = np.linalg.inv(extrinsics)
    camrot, campos = inv_E[:3, :3], inv_E[:3, 3].copy()
    if trans is not None:
        campos -= trans
    if camrot.T[1, 1] < 0.0:
        angle = -angle
    rvec = np.zeros(3)
    rvec[{'x': 0, 'y': 1, 'z': 2}[rotate_axis]] = angle
    g = rodrigues_exact(rvec).astype('float32')
    rpos, rrot = g.dot(campos), g.dot(camrot)
    if trans is not None:
        rpos += trans
    E = np.identity(4)
    E[:3, :3] = rrot.T
    E[:3, 3] = -rrot.T.dot(rpos)
    return E


def get_rays_from_KRT(H, W, K, R, T):
    """camera_util.py:133-160. Directions are NOT normalised."""
    rays_o = -np.dot(R.T, T).ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32),
                       indexing='xy')
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pix_cam = np.dot(xy1, np.linalg.inv(K).T)
    pix_world = np.dot(pix_cam - T.ravel(), R)
    rays_d = pix_world - rays_o[None, None]
    return np.broadcast_to(rays_o, rays_d.shape), rays_d


def rays_intersect_3d_bbox(bounds, ray_o, ray_d):
    """camera_util.py:163-212: slab test, keeps rays hitting exactly two faces."""
    bounds = np.stack([bounds['min_xyz'], bounds['max_xyz']], 0) if isinstance(bounds, dict) \
        else np.asarray(bounds)
    bounds = bounds + np.array([-0.01, 0.01])[:, None]
    nominator = bounds[None] - ray_o[:, None]
    ray_d[np.abs(ray_d) < 1e-5] = 1e-5          # in place, as the reference: the caller's directions are clamped too
    d_int = (nominator / ray_d[:, None]).reshape(-1, 6)
    p_int = d_int[..., None] * ray_d[:, None] + ray_o[:, None]
    lo, hi = bounds[0] - 1e-6, bounds[1] + 1e-6
    at_box = np.all((p_int >= lo) & (p_int <= hi), axis=-1)
    mask = at_box.sum(-1) == 2
    p_iv = p_int[mask][at_box[mask]].reshape(-1, 2, 3)
    ro, rd = ray_o[mask], ray_d[mask]
    nrm = np.linalg.norm(rd, axis=1)
    d0 = np.linalg.norm(p_iv[:, 0] - ro, axis=1) / nrm
    d1 = np.linalg.norm(p_iv[:, 1] - ro, axis=1) / nrm
    return np.minimum(d0, d1), np.maximum(d0, d1), mask


# ---------------------------------------------------------------------------
# frame assembly: the dict Network.forward(**data) takes
# ---------------------------------------------------------------------------

def skeleton_to_bbox(joints, bbox_offset=0.3):
    return {'min_xyz': np.min(joints, 0) - bbox_offset, 'max_xyz': np.max(joints, 0) + bbox_offset}


def seeded_pose(seed, sigma=0.3):
    """72-d axis-angle pose: zero root, N(0, sigma) on the 23 body joints."""
    rng = np.random.RandomState(seed)
    pose = np.zeros(72, dtype='float32')
    pose[3:] = (rng.randn(69) * sigma).astype('float32')
    return pose


def movement_pose(idx, total_frames, seed_a=11, seed_b=12):
    """Frame idx of the synthetic movement sequence: a smooth closed walk between two seeded poses (stands in for
    the observed frames of ZJU-MoCap 387, core/data/create_dataset.py:28-33 `movement`)."""
    a, b = seeded_pose(seed_a), seeded_pose(seed_b)
    t = 0.5 - 0.5 * np.cos(2 * np.pi * idx / max(int(total_frames), 1))
    return ((1 - t) * a + t * b).astype('float32')


def posed_joints(pose72, tjoints):
    """Forward kinematics of the skeleton only (for the observation-space bbox)."""
    Rs, Ts = body_pose_to_body_RTs(pose72, tjoints)
    G = np.zeros((TOTAL_BONES, 4, 4))
    for i in range(TOTAL_BONES):
        L = np.eye(4)
        L[:3, :3], L[:3, 3] = Rs[i], Ts[i]
        G[i] = L if i == 0 else G[SMPL_PARENT[i]].dot(L)
    return G[:, :3, 3].astype('float32')


_PRIOR_CACHE = {}


def make_frame(img_size=512, pose72=None, orbit_frame=0, orbit_period=100,
               bgcolor=(255.0, 255.0, 255.0), betas=None, bbox_offset=0.3, volume_size=32,
               rotate_axis='y', with_rays=True, camera_radius=6.0, camera_focal=1250.0):
    """One frame of renderer inputs, T-pose (pose72 None/zeros, tpose.py:133-217) or a
    posed free-view orbit frame (freeview.py:177-269).  with_rays=False leaves the ray batch to the device
    (occnerf_amd/rays.py) and returns the camera and the observation-space bbox instead."""
    betas = np.zeros(10, dtype='float32') if betas is None else betas
    cjoints = tpose_joints(betas).astype('float32')
    cbbox = skeleton_to_bbox(cjoints, bbox_offset)
    pose = np.zeros(72, dtype='float32') if pose72 is None else np.asarray(pose72, 'float32')
    dst_joints = posed_joints(pose, cjoints) if np.any(pose != 0) else cjoints
    dst_bbox = skeleton_to_bbox(dst_joints, bbox_offset)

    K, E = setup_camera(img_size, radius=camera_radius, focal=camera_focal)
    if orbit_frame:
        angle = 2 * np.pi * (orbit_frame / orbit_period)
        E = rotate_camera(E, angle, rotate_axis=rotate_axis).astype('float32')
    R, T = E[:3, :3], E[:3, 3]
    if with_rays:
        rays_o, rays_d = get_rays_from_KRT(img_size, img_size, K, R, T)
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        near, far, ray_mask = rays_intersect_3d_bbox(dst_bbox, rays_o, rays_d)
        rays_o, rays_d = rays_o[ray_mask], rays_d[ray_mask]
        ray_part = {'ray_mask': ray_mask, 'rays': np.stack([rays_o, rays_d], 0).astype('float32'),
                    'near': near[:, None].astype('float32'), 'far': far[:, None].astype('float32')}
    else:
        ray_part = {'camera_K': K, 'camera_E': E, 'dst_bbox_min': dst_bbox['min_xyz'],
                    'dst_bbox_max': dst_bbox['max_xyz']}

    dst_Rs, dst_Ts = body_pose_to_body_RTs(pose, cjoints)
    # the prior depends on the canonical skeleton only: the reference's datasets build it once per subject
    # (freeview.py:49-55), not per frame
    pkey = (cjoints.tobytes(), float(bbox_offset), int(volume_size))
    prior = _PRIOR_CACHE.get(pkey)
    if prior is None:
        if len(_PRIOR_CACHE) >= 4:
            _PRIOR_CACHE.clear()
        prior = _PRIOR_CACHE[pkey] = approx_gaussian_bone_volumes(cjoints, cbbox['min_xyz'], cbbox['max_xyz'],
                                                                  grid_size=volume_size).astype('float32')
    mn, mx = cbbox['min_xyz'].astype('float32'), cbbox['max_xyz'].astype('float32')
    return {
        'img_width': img_size, 'img_height': img_size, **ray_part,
        'bgcolor': np.array(bgcolor, dtype='float32'),
        'dst_Rs': dst_Rs, 'dst_Ts': dst_Ts,
        'cnl_gtfms': get_canonical_global_tfms(cjoints),
        'motion_weights_priors': prior,
        'cnl_bbox_min_xyz': mn, 'cnl_bbox_max_xyz': mx,
        'cnl_bbox_scale_xyz': (2.0 / (mx - mn)).astype('float32'),
        'dst_posevec': (pose[3:] + 1e-2).astype('float32'),
    }
