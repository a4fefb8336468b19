"""ctypes mirror of oracle/rssync_oracle.h -- TEST INFRASTRUCTURE ONLY.

Same method names as ``rssync_amd.SyncProblem`` (which are the reference's
``ISyncProblem`` names) so parity tests read the same on both sides.  Only
tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this.  PARITY UNPINNED: see rssync_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STREAM_SYNC_INIT = 0x80000000
STREAM_DEBUG = 0x40000000

_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int)


def library_path():
    return os.path.join(_HERE, "_build", "librssync_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(library_path()):
        build()
    lib = C.CDLL(library_path())
    lib.ora_create.restype = C.c_void_p
    lib.ora_last_error.restype = C.c_char_p
    lib.ora_last_error.argtypes = [C.c_void_p]
    lib.ora_sample_rate.restype = C.c_double
    lib.ora_quats_start.restype = C.c_double
    lib.ora_gyro_count.restype = C.c_size_t
    lib.ora_frame_count.restype = C.c_size_t
    lib.ora_frame_tracks.restype = C.c_size_t
    for n in ("ora_sample_rate", "ora_quats_start", "ora_gyro_count", "ora_frame_count"):
        getattr(lib, n).argtypes = [C.c_void_p]
    lib.ora_frame_tracks.argtypes = [C.c_void_p, C.c_int64]
    lib.ora_destroy.argtypes = [C.c_void_p]
    lib.ora_set_seed.argtypes = [C.c_void_p, C.c_uint64]
    lib.ora_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.ora_set_max_outer_iters.argtypes = [C.c_void_p, C.c_int]
    lib.ora_set_faithful.argtypes = [C.c_void_p, C.c_int]
    lib.ora_set_verbose.argtypes = [C.c_void_p, C.c_int]
    lib.ora_set_lbfgs_reeval.argtypes = [C.c_void_p, C.c_int]
    lib.ora_lbfgs_best_not_last.argtypes = [C.c_void_p]
    lib.ora_lbfgs_best_not_last.restype = C.c_long
    lib.ora_gyro_knots.argtypes = [C.c_void_p, _PD]
    lib.ora_set_gyro_quaternions.argtypes = [C.c_void_p, _PD, C.c_size_t, C.c_double, C.c_double]
    lib.ora_set_gyro_quaternions_ts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), _PD, C.c_size_t]
    lib.ora_set_track_result.argtypes = [C.c_void_p, C.c_int64, _PD, _PD, _PD, _PD, C.c_size_t]
    lib.ora_presync.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD]
    lib.ora_sync.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD]
    lib.ora_debug_presync.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, _PD, _PD, C.c_int]
    lib.ora_spline_eval.argtypes = [C.c_void_p, C.c_double, _PD]
    lib.ora_spline_deriv.argtypes = [C.c_void_p, C.c_double, _PD]
    lib.ora_quat_slerp.argtypes = [_PD, _PD, C.c_double, _PD]
    lib.ora_undistort_point.argtypes = [_PD, C.c_double, C.c_double, _PD]
    lib.ora_pixels_to_tracks.argtypes = [_PD, C.c_double, C.c_double, C.c_double, _PD, _PD, C.c_size_t, _PD, _PD, _PD,
                                         _PD]
    lib.ora_quat_from_aa.argtypes = [_PD, _PD]
    lib.ora_integrate_gyro.argtypes = [_PD, _PD, C.c_size_t, _PD, C.POINTER(C.c_int64)]
    lib.ora_orient_rates.argtypes = [_PD, C.c_size_t, C.c_char_p, _PD]
    lib.ora_compute_problem.argtypes = [C.c_void_p, C.c_int64, C.c_double, _PD]
    lib.ora_sample_pair.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ora_guess_motion.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_uint32, _PD, _PI, _PD]
    lib.ora_frame_presync_cost.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_uint32, _PD, _PI]
    lib.ora_presync_curve.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD,
                                      C.c_int, _PI, _PD, _PI]
    lib.ora_loss.argtypes = [C.c_void_p, C.c_int64, C.c_double, _PD, C.c_double, _PD, _PD, _PD, _PD]
    lib.ora_lbfgs_motion.argtypes = [C.c_void_p, C.c_int64, C.c_double, _PD, C.c_double, _PI, _PI, _PD]
    lib.ora_sync_trace.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD,
                                   _PD, C.c_int, _PI]
    lib.ora_set_init_override.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t]
    lib.ora_last_init_winners.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t]
    lib.ora_last_init_winners.restype = C.c_size_t
    lib.ora_sync_state.argtypes = [C.c_void_p, _PD, _PD, C.c_int, _PI]
    lib.ora_sync_simplified_trace.argtypes = lib.ora_sync_trace.argtypes
    lib.ora_loss_simplified.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, _PD, _PD, _PD]
    _LIB = lib
    return lib


class OracleError(RuntimeError):
    pass


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=_PD):
    return a.ctypes.data_as(t)


def sample_pair(seed, frame, stream, h, n):
    lib = load()
    i0, i1 = C.c_uint32(), C.c_uint32()
    lib.ora_sample_pair(seed, frame, stream, h, n, C.byref(i0), C.byref(i1))
    return i0.value, i1.value


def quat_slerp(p, q, t):
    lib = load()
    out = np.zeros(4)
    lib.ora_quat_slerp(_p(_d(p)), _p(_d(q)), float(t), _p(out))
    return out


def undistort_point(lens, px, py):
    """core_testcode.cpp:63-95; lens = (ro, fx, fy, cx, cy, k1, k2, k3, k4)"""
    out = np.zeros(2)
    load().ora_undistort_point(_p(_d(lens)), float(px), float(py), _p(out))
    return out


def pixels_to_tracks(lens, time_a, time_b, rows, points_a, points_b):
    """core_testcode.cpp:135-152 -> (ts_a, ts_b, rays_a, rays_b) as SetTrackResult takes them"""
    a, b = _d(points_a), _d(points_b)
    n = a.shape[0]
    ts_a, ts_b, ra, rb = np.zeros(n), np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3))
    load().ora_pixels_to_tracks(_p(_d(lens)), float(time_a), float(time_b), float(rows), _p(a), _p(b), n, _p(ts_a),
                                _p(ts_b), _p(ra), _p(rb))
    return ts_a, ts_b, ra, rb


def quat_from_aa(aa):
    out = np.zeros(4)
    load().ora_quat_from_aa(_p(_d(aa)), _p(out))
    return out


def integrate_gyro(timestamps_s, rates, orientation=None):
    """core_testcode.cpp:36-51 -> (quats (n, 4), timestamps_us int64)"""
    t, r = _d(timestamps_s), _d(rates)
    n = r.shape[0]
    if orientation is not None:
        o = np.zeros_like(r)
        if load().ora_orient_rates(_p(r), n, orientation.encode(), _p(o)):
            raise ValueError("bad orientation string")
        r = o
    q, us = np.zeros((n, 4)), np.zeros(n, np.int64)
    load().ora_integrate_gyro(_p(t), _p(r), n, _p(q), us.ctypes.data_as(C.POINTER(C.c_int64)))
    return q, us


class OracleProblem:
    def __init__(self, seed=None, max_outer_iters=None, threads=1, faithful=True, verbose=False, lbfgs_reeval=False):
        self._lib = load()
        self._h = self._lib.ora_create()
        if seed is not None:
            self._lib.ora_set_seed(self._h, int(seed))
        if max_outer_iters is not None:
            self._lib.ora_set_max_outer_iters(self._h, int(max_outer_iters))
        self._lib.ora_set_threads(self._h, int(threads))
        self._lib.ora_set_faithful(self._h, 1 if faithful else 0)
        self._lib.ora_set_verbose(self._h, 1 if verbose else 0)
        self._lib.ora_set_lbfgs_reeval(self._h, 1 if lbfgs_reeval else 0)

    def lbfgs_best_not_last(self):
        """line searches so far whose best step was not the last one tried"""
        return int(self._lib.ora_lbfgs_best_not_last(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ora_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise OracleError(self._lib.ora_last_error(self._h).decode())

    # ---- ISyncProblem names --------------------------------------------------
    def SetGyroQuaternions(self, data, sample_rate, first_timestamp):
        q = _d(data).reshape(-1, 4)
        self._check(self._lib.ora_set_gyro_quaternions(self._h, _p(q), q.shape[0], float(sample_rate),
                                                       float(first_timestamp)))

    def SetGyroQuaternionsTimestamped(self, timestamps_us, quats):
        ts = np.ascontiguousarray(timestamps_us, dtype=np.int64)
        q = _d(quats).reshape(-1, 4)
        self._check(self._lib.ora_set_gyro_quaternions_ts(self._h, _p(ts, C.POINTER(C.c_int64)), _p(q), q.shape[0]))

    def SetTrackResult(self, frame, ts_a, ts_b, rays_a, rays_b):
        ta, tb, ra, rb = _d(ts_a), _d(ts_b), _d(rays_a).reshape(-1, 3), _d(rays_b).reshape(-1, 3)
        self._check(self._lib.ora_set_track_result(self._h, int(frame), _p(ta), _p(tb), _p(ra), _p(rb), ta.shape[0]))

    def PreSync(self, initial_delay, frame_begin, frame_end, search_step, search_radius):
        c, d = C.c_double(), C.c_double()
        self._check(self._lib.ora_presync(self._h, initial_delay, frame_begin, frame_end, search_step, search_radius,
                                          C.byref(c), C.byref(d)))
        return c.value, d.value

    def Sync(self, initial_delay, frame_begin, frame_end, search_center, search_radius):
        c, d = C.c_double(), C.c_double()
        self._check(self._lib.ora_sync(self._h, initial_delay, frame_begin, frame_end, search_center, search_radius,
                                       C.byref(c), C.byref(d)))
        return c.value, d.value

    def DebugPreSync(self, initial_delay, frame_begin, frame_end, search_radius, point_count):
        delays, costs = np.zeros(point_count), np.zeros(point_count)
        self._check(self._lib.ora_debug_presync(self._h, initial_delay, frame_begin, frame_end, search_radius,
                                                _p(delays), _p(costs), point_count))
        return delays, costs

    # ---- introspection -------------------------------------------------------
    def set_seed(self, seed):
        self._lib.ora_set_seed(self._h, int(seed))

    def set_max_outer_iters(self, n):
        self._lib.ora_set_max_outer_iters(self._h, int(n))

    def gyro_info(self):
        return (self._lib.ora_sample_rate(self._h), self._lib.ora_quats_start(self._h),
                self._lib.ora_gyro_count(self._h))

    def gyro_knots(self):
        out = np.zeros((self._lib.ora_gyro_count(self._h), 4))
        self._lib.ora_gyro_knots(self._h, _p(out))
        return out

    def spline_eval(self, x):
        out = np.zeros(4)
        self._lib.ora_spline_eval(self._h, float(x), _p(out))
        return out

    def spline_deriv(self, x):
        out = np.zeros(4)
        self._lib.ora_spline_deriv(self._h, float(x), _p(out))
        return out

    def frame_tracks(self, frame):
        return self._lib.ora_frame_tracks(self._h, int(frame))

    def problem_matrix(self, frame, delay):
        n = self.frame_tracks(frame)
        P = np.zeros((n, 3))
        self._check(self._lib.ora_compute_problem(self._h, int(frame), float(delay), _p(P)))
        return P

    def guess_motion(self, frame, delay, max_iters, stream):
        M, bh, bm = np.zeros(3), C.c_int(), C.c_double()
        self._check(self._lib.ora_guess_motion(self._h, int(frame), float(delay), int(max_iters), int(stream), _p(M),
                                               C.byref(bh), C.byref(bm)))
        return M, bh.value, bm.value

    def frame_presync_cost(self, frame, delay, stream):
        c, bh = C.c_double(), C.c_int()
        rc = self._lib.ora_frame_presync_cost(self._h, int(frame), float(delay), int(stream), C.byref(c), C.byref(bh))
        if rc < 0:
            raise OracleError("unknown frame")
        return c.value, bh.value, rc

    def presync_curve(self, initial_delay, frame_begin, frame_end, search_step, search_radius, per_frame=False):
        cap = int(2 * search_radius / search_step) + 8
        delays, costs, n = np.zeros(cap), np.zeros(cap), C.c_int()
        fc = bh = None
        if per_frame:
            nf = int(per_frame)
            fc = np.zeros((cap, nf))
            bh = np.zeros((cap, nf), dtype=np.int32)
        self._check(self._lib.ora_presync_curve(self._h, initial_delay, frame_begin, frame_end, search_step,
                                                search_radius, _p(delays), _p(costs), cap, C.byref(n),
                                                _p(fc) if per_frame else None, _p(bh, _PI) if per_frame else None))
        k = n.value
        if per_frame:
            return delays[:k], costs[:k], fc[:k], bh[:k]
        return delays[:k], costs[:k]

    def loss(self, frame, delay, M, k):
        """-> loss, numeric d/d-delay (reference), analytic d/d-delay, dL/dM"""
        L, dn, da, g = C.c_double(), C.c_double(), C.c_double(), np.zeros(3)
        self._check(self._lib.ora_loss(self._h, int(frame), float(delay), _p(_d(M)), float(k), C.byref(L),
                                       C.byref(dn), C.byref(da), _p(g)))
        return L.value, dn.value, da.value, g

    def lbfgs_motion(self, frame, delay, M, k):
        Mo = _d(M).copy()
        it, ev, fl = C.c_int(), C.c_int(), C.c_double()
        self._check(self._lib.ora_lbfgs_motion(self._h, int(frame), float(delay), _p(Mo), float(k), C.byref(it),
                                               C.byref(ev), C.byref(fl)))
        return Mo, it.value, ev.value, fl.value

    def sync_trace(self, initial_delay, frame_begin, frame_end, search_center, search_radius, cap=512):
        c, d, n = C.c_double(), C.c_double(), C.c_int()
        tr = np.zeros((cap, 6))
        self._check(self._lib.ora_sync_trace(self._h, initial_delay, frame_begin, frame_end, search_center,
                                             search_radius, C.byref(c), C.byref(d), _p(tr), cap, C.byref(n)))
        return c.value, d.value, tr[:min(n.value, cap)].copy()

    def sync_simplified_trace(self, initial_delay, frame_begin, frame_end, search_center, search_radius, cap=512):
        """Sync in the thesis' simplified (no-translation) mode -> (cost, delay, trace)"""
        c, d, n = C.c_double(), C.c_double(), C.c_int()
        tr = np.zeros((cap, 6))
        self._check(self._lib.ora_sync_simplified_trace(self._h, initial_delay, frame_begin, frame_end, search_center,
                                                        search_radius, C.byref(c), C.byref(d), _p(tr), cap, C.byref(n)))
        return c.value, d.value, tr[:min(n.value, cap)].copy()

    def loss_simplified(self, frame, delay_k, delay):
        """one frame: (k chosen at delay_k, loss at delay, central-difference d loss / d delay)"""
        k, L, g = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.ora_loss_simplified(self._h, int(frame), float(delay_k), float(delay), C.byref(k),
                                                  C.byref(L), C.byref(g)))
        return k.value, L.value, g.value

    def set_init_override(self, winners):
        """test hook: these hypothesis indices instead of GuessMotion's search in the next Sync"""
        w = np.ascontiguousarray(winners, np.int32)
        self._lib.ora_set_init_override(self._h, w.ctypes.data_as(C.POINTER(C.c_int32)), w.size)

    def last_init_winners(self):
        n = self._lib.ora_last_init_winners(self._h, None, 0)
        out = np.zeros(n, np.int32)
        self._lib.ora_last_init_winners(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def sync_state(self, cap=1 << 16):
        M, k, n = np.zeros((cap, 3)), np.zeros(cap), C.c_int()
        self._lib.ora_sync_state(self._h, _p(M), _p(k), cap, C.byref(n))
        return M[:n.value].copy(), k[:n.value].copy()
