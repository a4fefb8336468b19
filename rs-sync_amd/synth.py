"""Deterministic synthetic inputs for the PreSync/Sync path.

A rolling-shutter camera (30 fps, 11.11 ms readout) rotates with a smooth
random angular rate and translates slowly through a static point cloud; a gyro
samples the rotation at a fixed rate.  Orientations are taken from the same
natural cubic spline through the gyro samples that the solver evaluates
(core_support/minispline.cpp), so for inliers the epipolar residual is exactly
zero at the true delay ``d_true``.

Conventions follow the reference driver: gyro integration
``q_i = normalise(quat_from_aa(w_i dt) * q_{i-1})`` (core_testcode.cpp:41-46),
ray time ``ts = frame_time + readout * row_fraction`` (:144-145), rays are unit
vectors in the camera frame (:147-152).  The solver de-rotates a ray with
``R(q)^T`` (core_private.cpp:26-27), so a camera ray is ``R(q(ts + d_true))``
applied to the world direction.

numpy/scipy only; nothing here touches the oracle or the GPU.
"""
from dataclasses import dataclass

import numpy as np
from scipy.interpolate import CubicSpline

FPS = 30.0
READOUT = 0.01111  # README.md:59, thesis p.28
D_TRUE = 0.0370


def quat_mul(p, q):
    """Hamilton product, [w,x,y,z] (core_support/quat.cpp:33-38), vectorised over leading dims."""
    pw, px, py, pz = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([pw * qw - px * qx - py * qy - pz * qz,
                     pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy - px * qz + py * qw + pz * qx,
                     pw * qz + px * qy - py * qx + pz * qw], axis=-1)


def quat_from_aa(aa):
    """core_support/quat.cpp:5-17."""
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th > 0, np.sin(half) / np.where(th > 0, th, 1.0), 0.5)
    return np.concatenate([np.where(th > 0, np.cos(half), 1.0), aa * k], axis=-1)


def rotate(q, v):
    """R(q) v = vec(q (0,v) q*)."""
    w, u = q[..., :1], q[..., 1:]
    t = np.cross(u, v)
    return v + 2.0 * (w * t + np.cross(u, t))


def rotate_inv(q, v):
    """R(q)^T v = vec(q* (0,v) q)."""
    w, u = q[..., :1], q[..., 1:]
    t = np.cross(u, v)
    return v + 2.0 * (-w * t + np.cross(u, t))


def integrate_gyro(rates, dt):
    """q_0 = 1, q_i = normalise(quat_from_aa(w_i dt_i) q_{i-1}) as a parallel prefix product."""
    n = rates.shape[0]
    dq = quat_from_aa(rates * np.asarray(dt).reshape(-1, 1))
    dq[0] = [1.0, 0.0, 0.0, 0.0]
    q = dq.copy()
    shift = 1
    while shift < n:  # Hillis-Steele scan; later factors multiply on the left
        q[shift:] = quat_mul(q[shift:], q[:-shift].copy())
        shift *= 2
    return q / np.linalg.norm(q, axis=1, keepdims=True)


@dataclass
class Gyro:
    fs: float
    t0: float            # time of sample 0 (s)
    quats: np.ndarray    # (G, 4)
    times: np.ndarray    # (G,) sample times (s); uniform unless jittered
    spline: object       # natural cubic spline over the sample index
    rates: np.ndarray = None  # (G, 3) angular rate samples the quaternions were integrated from

    def orientation(self, t):
        """unit quaternion at gyro time t: componentwise natural spline over the index, renormalised
        (core_private.cpp:24-25)."""
        x = (np.asarray(t) - self.t0) * self.fs
        q = self.spline(x)
        return q / np.linalg.norm(q, axis=-1, keepdims=True)


def make_gyro(t_begin, t_end, fs=400.0, seed=0, margin=1.0):
    """Gyro covering [t_begin - margin, t_end + margin] at a fixed rate."""
    rng = np.random.default_rng(seed)
    t0 = t_begin - margin
    g = int(np.ceil((t_end - t_begin + 2 * margin) * fs)) + 1
    t = t0 + np.arange(g) / fs
    rates = np.zeros((g, 3))
    for ax in range(3):  # three sinusoids per axis, 0.3-3 Hz, total amplitude <= 2 rad/s
        for _ in range(3):
            f = rng.uniform(0.3, 3.0)
            a = rng.uniform(0.1, 2.0 / 3.0)
            ph = rng.uniform(0, 2 * np.pi)
            rates[:, ax] += a * np.sin(2 * np.pi * f * t + ph)
    rates += rng.normal(0, 0.01, size=rates.shape)
    q = integrate_gyro(rates, np.full(g, 1.0 / fs))
    spline = CubicSpline(np.arange(g, dtype=np.float64), q, axis=0, bc_type="natural")
    return Gyro(fs=fs, t0=t0, quats=q, times=t, spline=spline, rates=rates)


def make_frames(gyro, frame_begin, frame_end, n_tracks, seed=0, d_true=D_TRUE, noise=1e-3, outliers=0.10,
                chunk=256, drift=0.0, translation=0.05):
    """Yield (frame, ts_a, ts_b, rays_a, rays_b) for frames [frame_begin, frame_end).

    `drift` (seconds of delay per second of video) makes the true delay `d_true + drift * t`:
    the clock-drift scenario the reference's CSV + python/plot_sync.py evaluate.
    `translation` is the camera's displacement per frame in metres (scene depth 2-50 m).

    Every frame draws from its own generator keyed on (seed, frame), so a shard of the range
    (one rank of a multi-GPU run) sees exactly the frames the whole range would contain."""
    for f0 in range(frame_begin, frame_end, chunk):
        f1 = min(f0 + chunk, frame_end)
        nf = f1 - f0
        frames = np.arange(f0, f1)
        u = np.empty((nf, 6, n_tracks))      # uniforms: ya, cos, phi, depth, outlier mask, spare
        g = np.empty((nf, 7, n_tracks))      # normals: yb jitter, 3 noise, 3 outlier direction
        for i, fr in enumerate(frames):
            rng = np.random.default_rng([seed, int(fr)])
            u[i] = rng.uniform(size=(6, n_tracks))
            g[i] = rng.normal(size=(7, n_tracks))
        ya = u[:, 0]
        yb = np.clip(ya + 0.03 * g[:, 0], 0.0, 0.999)
        ts_a = frames[:, None] / FPS + READOUT * ya
        ts_b = (frames[:, None] + 1) / FPS + READOUT * yb
        # camera-frame ray of the current frame inside a +-55 degree cone around +z
        c55 = np.cos(np.deg2rad(55.0))
        cos_t = c55 + (1.0 - c55) * u[:, 1]
        sin_t = np.sqrt(1 - cos_t ** 2)
        phi = 2 * np.pi * u[:, 2]
        a_cam = np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], axis=-1)
        depth = (2.0 + 48.0 * u[:, 3])[..., None]
        # slowly varying translation direction, 0.05 m per frame
        ang = 0.002 * frames + 0.7 * seed
        tdir = np.stack([np.cos(ang), np.sin(ang) * np.cos(0.3 * ang), np.sin(ang) * np.sin(0.3 * ang)], axis=-1)
        trans = translation * tdir[:, None, :]  # metres per frame; 0 = pure rotation (the thesis' simplified mode)
        qa = gyro.orientation(ts_a + d_true + drift * ts_a)
        qb = gyro.orientation(ts_b + d_true + drift * ts_b)
        a_world = rotate_inv(qa, a_cam)
        X = depth * a_world                       # camera centre of frame f at the origin
        b_world = X - trans                       # seen from the next frame's centre
        b_world /= np.linalg.norm(b_world, axis=-1, keepdims=True)
        b_cam = rotate(qb, b_world)
        if noise > 0:
            b_cam = b_cam + noise * np.moveaxis(g[:, 1:4], 1, -1)
            b_cam /= np.linalg.norm(b_cam, axis=-1, keepdims=True)
        if outliers > 0:
            mask = u[:, 4] < outliers
            rnd = np.moveaxis(g[:, 4:7], 1, -1)
            rnd = rnd / np.linalg.norm(rnd, axis=-1, keepdims=True)
            b_cam = np.where(mask[..., None], rnd, b_cam)
        for i in range(nf):
            yield int(frames[i]), ts_a[i], ts_b[i], a_cam[i], b_cam[i]


def make_timestamped(gyro, jitter=0.2, seed=0):
    """Variable-rate view of a gyro track for the timestamped setter: sample times jittered by
    +-jitter of the nominal interval, quaternions taken from the smooth orientation."""
    rng = np.random.default_rng(seed)
    g = gyro.quats.shape[0]
    dt = 1.0 / gyro.fs
    t = gyro.t0 + np.arange(g) * dt + rng.uniform(-jitter, jitter, size=g) * dt * 0.5
    t = np.sort(t)
    t = np.clip(t, gyro.t0, gyro.t0 + (g - 1) * dt)
    q = gyro.orientation(t)
    ts_us = np.round(t * 1e6).astype(np.int64)
    return ts_us, q


# The 48 IMU orientations the reference's orientation-guessing block tries (core_testcode.cpp:186-190):
# position = output axis, letter = input axis, upper case = +, lower case = - (telemetry-parser's
# convention).  "XYZ" is the identity.
ORIENTATIONS = (
    "YxZ", "Xyz", "XZy", "Zxy", "zyX", "yxZ", "ZXY", "zYx", "ZYX", "yXz", "YZX", "XyZ",
    "Yzx", "zXy", "YXz", "xyz", "yZx", "XYZ", "zxy", "xYz", "XYz", "zxY", "zXY", "xZy",
    "zyx", "xyZ", "Yxz", "xzy", "yZX", "yzX", "ZYx", "xYZ", "zYX", "ZxY", "yzx", "xZY",
    "Xzy", "XzY", "YzX", "Zyx", "XZY", "yxz", "xzY", "ZyX", "YXZ", "yXZ", "YZx", "ZXy")


def orient_rates(rates, orientation):
    """Apply a signed axis permutation to an (G, 3) rate stream."""
    out = np.empty_like(rates)
    for i, ch in enumerate(orientation):
        out[:, i] = rates[:, "xyz".index(ch.lower())] * (1.0 if ch.isupper() else -1.0)
    return out


def gyro_for_orientation(gyro, orientation):
    """The quaternion track a driver would hand to SetGyroQuaternions for one candidate orientation
    (core_testcode.cpp:41-52: integrate the re-oriented rates)."""
    r = orient_rates(gyro.rates, orientation)
    return integrate_gyro(r, np.full(r.shape[0], 1.0 / gyro.fs))


# A GoPro-like fisheye preset in the reference's Lens fields (core_testcode.cpp:55-61):
# (ro, fx, fy, cx, cy, k1, k2, k3, k4) for a 2704 x 1520 image.
LENS = (READOUT, 1180.0, 1180.0, 1352.0, 760.0, 0.05, 0.01, -0.005, 0.001)
IMAGE_ROWS, IMAGE_COLS = 1520, 2704


def fisheye_distort(theta, lens):
    """theta_d = theta + k1 theta^3 + k2 theta^5 + k3 theta^7 + k4 theta^9 (the forward model whose
    inverse core_testcode.cpp:63-95 computes)."""
    _, _, _, _, _, k1, k2, k3, k4 = lens
    t2 = theta * theta
    return theta * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4))))


def fisheye_undistort_angle(theta_d, lens, iters=40):
    """Accurate inverse of fisheye_distort on (0, pi/2) (plain Newton with the true derivative)."""
    _, _, _, _, _, k1, k2, k3, k4 = lens
    th = np.array(theta_d, dtype=np.float64, copy=True)
    for _ in range(iters):
        t2 = th * th
        f = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - theta_d
        df = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)))
        th = np.clip(th - f / df, 1e-12, np.pi / 2 - 1e-9)
    return th


def project(ray_cam, lens):
    """camera-frame direction(s) (..., 3) with z > 0 -> distorted pixel position(s) (..., 2)"""
    _, fx, fy, cx, cy = lens[:5]
    r = np.linalg.norm(ray_cam[..., :2], axis=-1)
    theta = np.arctan2(r, ray_cam[..., 2])
    scale = np.where(r > 0, fisheye_distort(theta, lens) / np.where(r > 0, r, 1.0), 0.0)
    return np.stack([fx * scale * ray_cam[..., 0] + cx, fy * scale * ray_cam[..., 1] + cy], axis=-1)


def unproject(px, lens):
    """distorted pixel position(s) (..., 2) -> unit camera-frame direction(s) (..., 3)"""
    _, fx, fy, cx, cy = lens[:5]
    x_ = (px[..., 0] - cx) / fx
    y_ = (px[..., 1] - cy) / fy
    td = np.hypot(x_, y_)
    th = fisheye_undistort_angle(np.maximum(td, 1e-300), lens)
    s = np.where(td > 0, np.sin(th) / np.where(td > 0, td, 1.0), 0.0)
    return np.stack([s * x_, s * y_, np.cos(th)], axis=-1)


def make_pixel_frames(gyro, frame_begin, frame_end, n_tracks, seed=0, d_true=D_TRUE, noise_px=0.3, outliers=0.10,
                      lens=LENS, rows=IMAGE_ROWS, cols=IMAGE_COLS):
    """Yield (frame, time_a, time_b, points_a, points_b): what the reference driver has after
    optical flow and before undistortion (core_testcode.cpp:123-133) -- tracked pixel positions in
    the current and the next video frame.  The row a point lies on sets its capture time
    (frame_time + ro * y / rows, :144-145); the scene is the one make_frames uses."""
    ro = lens[0]
    for fr in range(frame_begin, frame_end):
        rng = np.random.default_rng([seed, int(fr), 7])
        u = rng.uniform(size=(5, n_tracks))
        g = rng.normal(size=(2, n_tracks))
        t_a, t_b = fr / FPS, (fr + 1) / FPS
        pa = np.stack([0.05 * cols + 0.9 * cols * u[0], 0.05 * rows + 0.9 * rows * u[1]], axis=-1)
        a_cam = unproject(pa, lens)
        ts_a = t_a + ro * (pa[:, 1] / rows)
        qa = gyro.orientation(ts_a + d_true)
        depth = (2.0 + 48.0 * u[2])[:, None]
        ang = 0.002 * fr + 0.7 * seed
        tdir = np.array([np.cos(ang), np.sin(ang) * np.cos(0.3 * ang), np.sin(ang) * np.sin(0.3 * ang)])
        b_world = depth * rotate_inv(qa, a_cam) - 0.05 * tdir
        b_world /= np.linalg.norm(b_world, axis=-1, keepdims=True)
        pb = pa.copy()
        for _ in range(4):  # the row of the point in the next frame sets the time it is seen at
            ts_b = t_b + ro * (pb[:, 1] / rows)
            pb = project(rotate(gyro.orientation(ts_b + d_true), b_world), lens)
        if noise_px > 0:
            pb = pb + noise_px * g.T
        if outliers > 0:
            mask = u[3] < outliers
            rnd = np.stack([cols * u[4], rows * rng.uniform(size=n_tracks)], axis=-1)
            pb = np.where(mask[:, None], rnd, pb)
        yield int(fr), t_a, t_b, pa, pb


def fill(problem, gyro, frame_begin, frame_end, n_tracks, seed=0, **kw):
    """Feed one problem object (SyncProblem or the oracle mirror): same calls as the reference driver."""
    problem.SetGyroQuaternions(gyro.quats, gyro.fs, gyro.t0)
    for fr, ta, tb, ra, rb in make_frames(gyro, frame_begin, frame_end, n_tracks, seed=seed, **kw):
        problem.SetTrackResult(fr, ta, tb, ra, rb)


def config(index):
    """The BASELINE.json configurations: frames, tracks, sweep parameters."""
    cfgs = {
        1: dict(frames=64, tracks=256, step=0.002, radius=0.2),
        2: dict(frames=1024, tracks=1024, step=0.0005, radius=0.2),
        3: dict(frames=4096, tracks=2048, step=0.0005, radius=0.2),
    }
    return cfgs[index]
