"""Python mirror of the reference's ``ISyncProblem`` (src/core/public/rssync.h:9-29).

Method names, argument order, units and the end-exclusive (PreSync) /
end-inclusive (Sync) frame ranges are the reference's.  Everything is a thin
ctypes call into ``librssync_core.so`` through the flat C-ABI declared in
``include/rssync_c.h``; the library is required (no Python or CPU fallback).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RsSyncError(RuntimeError):
    """A 'panic' of the library (invalid input, device failure) in status mode."""


def library_path():
    """librssync_core.so next to this file.  RSSYNC_LIB names another BUILD of the same library (kernel A/B
    scripts select their variants this way instead of copying over the product .so)."""
    return os.environ.get("RSSYNC_LIB") or os.path.join(_HERE, "librssync_core.so")


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_void_p)

_PD = C.POINTER(C.c_double)
_PF = C.POINTER(C.c_float)
_PI64 = C.POINTER(C.c_int64)
_PI32 = C.POINTER(C.c_int32)
_PU64 = C.POINTER(C.c_uint64)

# name -> (restype, argtypes): every symbol include/rssync_c.h declares
SIGNATURES = {
    "rssync_create": (C.c_void_p, []),
    "rssync_ext_borrow": (C.c_void_p, [C.c_void_p]),
    "rssync_destroy": (None, [C.c_void_p]),
    "rssync_last_error": (C.c_char_p, []),
    "rssync_set_panic_mode": (None, [C.c_int]),
    "rssync_set_gyro_quaternions": (C.c_int, [C.c_void_p, _PD, C.c_size_t, C.c_double, C.c_double]),
    "rssync_set_gyro_quaternions_ts": (C.c_int, [C.c_void_p, _PI64, _PD, C.c_size_t]),
    "rssync_set_track_result": (C.c_int, [C.c_void_p, C.c_int64, _PD, _PD, _PD, _PD, C.c_size_t]),
    "rssync_pre_sync": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD]),
    "rssync_sync": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD]),
    "rssync_debug_pre_sync": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, _PD, _PD, C.c_int]),
    "rssync_ext_set_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rssync_ext_set_max_outer_iters": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_set_verbose": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_set_lbfgs_reeval": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_lbfgs_best_not_last": (C.c_int, [C.c_void_p, _PU64]),
    "rssync_ext_set_host_loop": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_set_hook_device_loop": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rssync_ext_set_devices": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "rssync_ext_device_count": (C.c_int, [C.c_void_p]),
    "rssync_ext_set_reduce_hook": (C.c_int, [C.c_void_p, REDUCE_FN, C.c_void_p]),
    "rssync_ext_rccl_preflight": (C.c_int, [C.c_void_p]),
    "rssync_ext_rccl_library": (C.c_char_p, [C.c_void_p]),
    "rssync_ext_rccl_unique_id": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rssync_ext_rccl_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rssync_ext_rccl_shutdown": (C.c_int, [C.c_void_p]),
    "rssync_ext_set_tracks_hint": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rssync_ext_exchange_stats": (C.c_int, [C.c_void_p, _PU64, _PU64]),
    "rssync_ext_window_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rssync_ext_set_executor_check": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_set_executor_check_every": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rssync_ext_executor_stats": (C.c_int, [C.c_void_p, _PU64, _PU64, C.POINTER(C.c_uint32)]),
    "rssync_ext_executor_mismatches": (C.c_int, [C.c_void_p, _PU64]),
    "rssync_ext_near_static_stats": (C.c_int, [C.c_void_p, _PU64, _PU64, _PU64]),
    "rssync_ext_debug_residuals": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    "rssync_ext_debug_residuals_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32)]),
    "rssync_ext_record_init_winners": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_last_init_winners": (C.c_int, [C.c_void_p, _PI32, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rssync_ext_set_init_override": (C.c_int, [C.c_void_p, _PI32, C.c_size_t]),
    "rssync_ext_debug_math64": (C.c_int, [C.c_void_p, C.c_int, _PD, _PD, _PD, C.c_size_t]),
    "rssync_ext_upload": (C.c_int, [C.c_void_p]),
    "rssync_ext_sample_rate": (C.c_int, [C.c_void_p, _PD, _PD, C.POINTER(C.c_size_t)]),
    "rssync_ext_gyro_knots": (C.c_int, [C.c_void_p, _PD, C.c_size_t]),
    "rssync_ext_gyro_table": (C.c_int, [C.c_void_p, _PD, C.c_size_t]),
    "rssync_ext_presync_curve": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                           _PD, _PD, C.c_int, C.POINTER(C.c_int), _PD, _PI32, C.POINTER(C.c_int)]),
    "rssync_ext_problem_matrix": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, _PF, _PF, C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "rssync_ext_init_motion": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, _PD, _PD, C.c_int,
                                         C.POINTER(C.c_int)]),
    "rssync_ext_opt_motion": (C.c_int, [C.c_void_p, C.c_double, _PD, _PD, C.c_int, C.POINTER(C.c_int), _PU64, _PU64]),
    "rssync_ext_set_motion": (C.c_int, [C.c_void_p, _PD, _PD, C.c_int]),
    "rssync_ext_loss": (C.c_int, [C.c_void_p, _PD, C.c_int, _PD, _PD]),
    "rssync_ext_sync_simplified": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD]),
    "rssync_ext_init_k_simplified": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, _PD, C.c_int,
                                               C.POINTER(C.c_int)]),
    "rssync_ext_loss_simplified": (C.c_int, [C.c_void_p, _PD, C.c_int, _PD, _PD]),
    "rssync_ext_problem_matrix64": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, _PD, _PD, C.c_size_t,
                                              C.POINTER(C.c_size_t)]),
    "rssync_ext_set_track_pixels": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double, _PD, _PD, C.c_size_t,
                                              C.c_void_p, C.c_double]),
    "rssync_ext_set_gyro_rates": (C.c_int, [C.c_void_p, _PD, _PD, C.c_size_t, C.c_char_p]),
    "rssync_ext_orientation_sweep": (C.c_int, [C.c_void_p, _PD, _PD, C.c_size_t, C.POINTER(C.c_char_p), C.c_int,
                                               C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, _PD, _PD]),
    "rssync_ext_frame_rays": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t)]),
    "rssync_ext_pre_sync_windows": (C.c_int, [C.c_void_p, C.c_double, _PI64, _PI64, C.c_int, C.c_double, C.c_double,
                                              _PD, _PD]),
    "rssync_ext_sync_windows": (C.c_int, [C.c_void_p, _PD, _PI64, _PI64, C.c_int, C.c_double, C.c_double, _PD, _PD]),
    "rssync_ext_sync_points": (C.c_int, [C.c_void_p, _PI64, C.c_int, C.c_int64, C.c_double, C.c_int, C.c_double,
                                         C.c_double, C.c_int, _PD, _PD]),
    "rssync_ext_window_trace": (C.c_int, [C.c_void_p, C.c_int, _PD, C.c_int, C.POINTER(C.c_int)]),
    "rssync_ext_sync_trace": (C.c_int, [C.c_void_p, _PD, C.c_int, C.POINTER(C.c_int)]),
    "rssync_ext_device_context": (C.c_void_p, [C.c_void_p]),
    "rssync_ext_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "rssync_ext_profile_get": (C.c_int, [C.c_void_p, C.c_int, _PU64, _PD]),
    "rssync_ext_profile_reset": (C.c_int, [C.c_void_p]),
}


def bind(lib):
    """Attach the C-ABI signatures of include/rssync_c.h to a loaded library."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library():
    """dlopen librssync_core.so and bind every C-ABI symbol; raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RsSyncError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C rs-sync_amd/csrc); there is no fallback implementation")
    _LIB = bind(C.CDLL(path))
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=_PD):
    return a.ctypes.data_as(t)


class SyncProblem:
    """``CreateSyncProblem()`` + the six ``ISyncProblem`` methods (rssync.h:9-31)."""

    def __init__(self, seed=None, max_outer_iters=None, verbose=False, _lib=None):
        # _lib: tests pass a build of the same host code linked against a CPU test double of the
        # device ABI (tests/cpu_device); the package itself only ever loads librssync_core.so
        self._lib = _lib if _lib is not None else load_library()
        self._lib.rssync_set_panic_mode(1)  # report panics as exceptions instead of exit(1)
        self._h = self._lib.rssync_create()
        if not self._h:
            raise RsSyncError("rssync_create failed: " + self._lib.rssync_last_error().decode())
        self._hook = None
        self._hook_error = None
        self._lib.rssync_ext_set_verbose(self._h, 1 if verbose else 0)
        if seed is not None:
            self._lib.rssync_ext_set_seed(self._h, int(seed))
        if max_outer_iters is not None:
            self._lib.rssync_ext_set_max_outer_iters(self._h, int(max_outer_iters))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rssync_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            err = RsSyncError(self._lib.rssync_last_error().decode())
            cause, self._hook_error = self._hook_error, None
            if cause is not None:
                raise err from cause  # the exception a Python reduce hook raised
            raise err

    # ---- ISyncProblem -------------------------------------------------------
    def SetGyroQuaternions(self, data, sample_rate, first_timestamp):
        """rssync.h:13-14: `data` is (count, 4) [w,x,y,z] at a fixed rate (Hz), first sample time in s."""
        q = _d(data).reshape(-1, 4)
        self._check(self._lib.rssync_set_gyro_quaternions(self._h, _p(q), q.shape[0], float(sample_rate),
                                                          float(first_timestamp)))

    def SetGyroQuaternionsTimestamped(self, timestamps_us, quats):
        """rssync.h:15-16 (the int64-timestamp overload)."""
        ts = np.ascontiguousarray(timestamps_us, dtype=np.int64)
        q = _d(quats).reshape(-1, 4)
        assert ts.shape[0] == q.shape[0]
        self._check(self._lib.rssync_set_gyro_quaternions_ts(self._h, _p(ts, _PI64), _p(q), q.shape[0]))

    def SetTrackResult(self, frame, ts_a, ts_b, rays_a, rays_b):
        """rssync.h:17-18: rays are (count, 3) unit vectors, timestamps in seconds."""
        ta, tb, ra, rb = _d(ts_a), _d(ts_b), _d(rays_a).reshape(-1, 3), _d(rays_b).reshape(-1, 3)
        n = ta.shape[0]
        assert tb.shape[0] == n and ra.shape[0] == n and rb.shape[0] == n
        self._check(self._lib.rssync_set_track_result(self._h, int(frame), _p(ta), _p(tb), _p(ra), _p(rb), n))

    def PreSync(self, initial_delay, frame_begin, frame_end, search_step, search_radius):
        """rssync.h:19-21 -> (cost, delay); frame_end exclusive."""
        c, d = C.c_double(), C.c_double()
        self._check(self._lib.rssync_pre_sync(self._h, initial_delay, frame_begin, frame_end, search_step,
                                              search_radius, C.byref(c), C.byref(d)))
        return c.value, d.value

    def Sync(self, initial_delay, frame_begin, frame_end, search_center, search_radius):
        """rssync.h:22-24 -> (cost, delay); frame_end inclusive."""
        c, d = C.c_double(), C.c_double()
        self._check(self._lib.rssync_sync(self._h, initial_delay, frame_begin, frame_end, search_center,
                                          search_radius, C.byref(c), C.byref(d)))
        return c.value, d.value

    def DebugPreSync(self, initial_delay, frame_begin, frame_end, search_radius, point_count):
        """rssync.h:26-28 -> (delays, costs)."""
        delays = np.zeros(point_count)
        costs = np.zeros(point_count)
        self._check(self._lib.rssync_debug_pre_sync(self._h, initial_delay, frame_begin, frame_end, search_radius,
                                                    _p(delays), _p(costs), point_count))
        return delays, costs

    # ---- extensions (include/rssync_c.h, rssync_ext_*) ----------------------
    def set_seed(self, seed):
        self._lib.rssync_ext_set_seed(self._h, int(seed))

    def set_max_outer_iters(self, n):
        self._lib.rssync_ext_set_max_outer_iters(self._h, int(n))

    def set_lbfgs_reeval(self, reeval):
        """False (default): a line search whose best step is not its last leaves value and gradient as
        the last trial computed them (published ens::L_BFGS); True: re-evaluate at the best step."""
        self._check(self._lib.rssync_ext_set_lbfgs_reeval(self._h, 1 if reeval else 0))

    def set_host_loop(self, host_loop=True):
        """keep Sync's outer loop on the host (it runs on the device where one GPU holds all frames)"""
        self._lib.rssync_ext_set_host_loop(self._h, 1 if host_loop else 0)

    def set_hook_device_loop(self, on=True):
        """with a reduce hook: Sync's loop on the device, the hook called between the kernels on the window sums"""
        self._lib.rssync_ext_set_hook_device_loop(self._h, 1 if on else 0)

    def lbfgs_best_not_last(self):
        n = C.c_uint64()
        self._lib.rssync_ext_lbfgs_best_not_last(self._h, C.byref(n))
        return n.value

    def set_devices(self, device_ids):
        """Spread this object's frames over several GPUs of the process (list of device ordinals)."""
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self._check(self._lib.rssync_ext_set_devices(self._h, ids, len(device_ids)))

    def device_count(self):
        return int(self._lib.rssync_ext_device_count(self._h))

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.rssync_ext_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def set_reduce_hook(self, fn):
        """fn(np.ndarray float64) must sum the array in place over all ranks; None = single rank."""
        if fn is None:
            self._hook = None
            self._lib.rssync_ext_set_reduce_hook(self._h, C.cast(None, REDUCE_FN), None)
            return

        def tramp(buf, n, _user):
            # ctypes swallows exceptions raised inside a callback: report them as a status instead,
            # which the library turns into a panic (a skipped all-reduce must never go unnoticed)
            try:
                arr = np.ctypeslib.as_array(buf, shape=(n,))
                fn(arr)
                return 0
            except BaseException as exc:  # noqa: BLE001
                self._hook_error = exc
                return 1

        self._hook = REDUCE_FN(tramp)  # keep the trampoline alive
        self._lib.rssync_ext_set_reduce_hook(self._h, self._hook, None)

    def rccl_preflight(self):
        """raises unless this process can use RCCL (library + entry points resolve); no communication"""
        self._check(self._lib.rssync_ext_rccl_preflight(self._h))

    def rccl_library(self):
        return (self._lib.rssync_ext_rccl_library(self._h) or b"").decode()

    def rccl_unique_id(self):
        """128 opaque bytes from ncclGetUniqueId (call on rank 0, hand them to every rank)."""
        buf = C.create_string_buffer(128)
        self._check(self._lib.rssync_ext_rccl_unique_id(self._h, buf))
        return buf.raw

    def rccl_init(self, unique_id, rank, world_size):
        """Join the library's own RCCL communicator; replaces the reduce hook."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of rccl_unique_id()")
        self._check(self._lib.rssync_ext_rccl_init(self._h, C.create_string_buffer(unique_id, 128), int(rank),
                                                   int(world_size)))

    def rccl_shutdown(self):
        """Leave the library's RCCL communicator (collective)."""
        self._check(self._lib.rssync_ext_rccl_shutdown(self._h))

    def set_tracks_hint(self, max_tracks_all_ranks):
        """A no-op since round 5 (kept for callers written against round 4, when the kernel shapes followed the problem's
        largest frame and ranks had to agree on it): a frame's kernels follow its own track count."""
        self._lib.rssync_ext_set_tracks_hint(self._h, int(max_tracks_all_ranks))

    def exchange_stats(self):
        """(calls, doubles) exchanged with other ranks so far, through the hook or the native communicator."""
        a, b = C.c_uint64(), C.c_uint64()
        self._lib.rssync_ext_exchange_stats(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def window_info(self):
        """-> dict: how the kernels' LDS spline windows were laid out for this problem's gyro rate"""
        q = (C.c_uint32 * 8)()
        self._check(self._lib.rssync_ext_window_info(self._h, q))
        return dict(frame_span_knots=q[0], fp64_window_knots=q[1], presync_window_knots=q[2] or 80, presync_window_dynamic=bool(q[2]),
                    presync_chunk=q[3], init_window_knots=q[4] or 80, trial_delays_per_pass=q[5], frame_ends_knots=q[6],
                    fp64_window_compact=bool(q[7]))

    def lmeds_shapes(self):
        """-> list of 6: rows / 256 of the LMedS tile the last PreSync sweep used per size class (include/rssync_hip.h:
        rship_lmeds_shapes; 0 = class absent or not the tile kernel)"""
        q = (C.c_uint32 * 6)()
        fn = self._lib.rship_lmeds_shapes
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]
        self._check(fn(C.c_void_p(self.device_context()), q))
        return list(q)

    def set_executor_check(self, on=True):
        """debug mode: every call the window executor runs is re-run by the launch chain and must give the same bits"""
        self._check(self._lib.rssync_ext_set_executor_check(self._h, 1 if on else 0))

    def set_executor_check_every(self, every):
        """with the check mode off: verify one executor call in `every` against the launch chain (process-wide count; 0 = never)"""
        self._lib.rssync_ext_set_executor_check_every(self._h, int(every))

    def executor_stats(self):
        """-> dict(runs, checked, head, tail, ring_cells, waves) of the window executor on this object"""
        runs, chk, q = C.c_uint64(), C.c_uint64(), (C.c_uint32 * 4)()
        self._check(self._lib.rssync_ext_executor_stats(self._h, C.byref(runs), C.byref(chk), q))
        mm = C.c_uint64()
        self._lib.rssync_ext_executor_mismatches(self._h, C.byref(mm))
        return dict(runs=runs.value, checked=chk.value, head=q[0], tail=q[1], ring_cells=q[2], waves=q[3], mismatches=mm.value)

    def debug_residuals(self, on=True, cap_rows=0):
        """TEST-VARIANTS build only: later sweeps also store the |residuals| their LMedS selection worked on"""
        self._check(self._lib.rssync_ext_debug_residuals(self._h, 1 if on else 0, int(cap_rows)))

    def debug_residuals_get(self):
        """-> float32 [candidates][frames][hypotheses][cap_rows] of the last sweep (NaN where there is no row)"""
        dims = (C.c_uint32 * 4)()
        self._check(self._lib.rssync_ext_debug_residuals_get(self._h, None, 0, dims))
        out = np.zeros(tuple(int(x) for x in dims), dtype=np.uint32)
        self._check(self._lib.rssync_ext_debug_residuals_get(self._h, _p(out, C.POINTER(C.c_uint32)), out.size, dims))
        return out.view(np.float32)

    def near_static_stats(self):
        """-> dict(pairs, sweeps): (frame, candidate) pairs of PreSync sweeps recomputed with fp64 rows so far (near-static
        frames, |P| below ~2e-4: core_private.cpp:19-28,45-46 are double), and the sweeps that needed it.  0 on ordinary scenes."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.rssync_ext_near_static_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(pairs=a.value, sweeps=b.value, searches=c.value)

    def record_init_winners(self, on=True):
        self._lib.rssync_ext_record_init_winners(self._h, 1 if on else 0)

    def last_init_winners(self):
        """GuessMotion's winning hypothesis per slot of the last Sync-type call (needs record_init_winners)."""
        n = C.c_size_t()
        self._lib.rssync_ext_last_init_winners(self._h, None, 0, C.byref(n))
        out = np.zeros(n.value, np.int32)
        self._lib.rssync_ext_last_init_winners(self._h, _p(out, _PI32), out.size, C.byref(n))
        return out

    def set_init_override(self, winners):
        """Install these winners instead of the search's in the next Sync-type call (bit-exactness tests)."""
        w = np.ascontiguousarray(winners, np.int32)
        self._lib.rssync_ext_set_init_override(self._h, _p(w, _PI32), w.size)

    def debug_math64(self, op, a, b=None):
        """The Sync kernels' fp64 building blocks on arrays: 0 a/b, 1 sqrt, 2 (log1p, 1/(1+a)), 3 fma(a,b,a), 4 wave sums, 5 a/3 (div3_exact)."""
        a = _d(a)
        b = _d(b) if b is not None else None
        n = a.size
        out = np.zeros(2 * n if op == 2 else ((n + 63) // 64 if op == 4 else n))
        self._check(self._lib.rssync_ext_debug_math64(self._h, int(op), _p(a), _p(b) if b is not None else None, _p(out), n))
        return out.reshape(n, 2) if op == 2 else out

    def upload(self):
        """Pack tracks + spline and copy them to HBM now (otherwise lazy)."""
        self._check(self._lib.rssync_ext_upload(self._h))

    def gyro_info(self):
        fs, st, n = C.c_double(), C.c_double(), C.c_size_t()
        self._lib.rssync_ext_sample_rate(self._h, C.byref(fs), C.byref(st), C.byref(n))
        return fs.value, st.value, n.value

    def gyro_knots(self):
        n = self.gyro_info()[2]
        out = np.zeros((n, 4))
        self._check(self._lib.rssync_ext_gyro_knots(self._h, _p(out), out.size))
        return out

    def gyro_table(self):
        """The device's spline table, (n_knots, 4, 4): [knot][y, b, c, d][w, x, y, z]."""
        n = self.gyro_info()[2]
        out = np.zeros((n, 4, 4))
        self._check(self._lib.rssync_ext_gyro_table(self._h, _p(out), out.size))
        return out

    def presync_curve(self, initial_delay, frame_begin, frame_end, search_step, search_radius, per_frame=False,
                      cap=None):
        if cap is None:
            cap = int(2 * search_radius / search_step) + 8
        delays, costs = np.zeros(cap), np.zeros(cap)
        n, nf = C.c_int(), C.c_int()
        fc = bh = None
        if per_frame:
            if per_frame is True:
                raise ValueError('per_frame must be the number of frames in the range')
            nf_max = int(per_frame)
            fc = np.zeros((cap, nf_max))
            bh = np.zeros((cap, nf_max), dtype=np.int32)
        self._check(self._lib.rssync_ext_presync_curve(
            self._h, initial_delay, frame_begin, frame_end, search_step, search_radius, _p(delays), _p(costs), cap,
            C.byref(n), _p(fc) if per_frame else None, _p(bh, _PI32) if per_frame else None, C.byref(nf)))
        k = n.value
        if per_frame:
            f = nf.value
            fc = fc.reshape(-1)[:k * f].reshape(k, f)
            bh = bh.reshape(-1)[:k * f].reshape(k, f)
            return delays[:k], costs[:k], fc, bh
        return delays[:k], costs[:k]

    def problem_matrix(self, frame, delay, n_tracks, deriv=False):
        P = np.zeros((n_tracks, 3), dtype=np.float32)
        dP = np.zeros((n_tracks, 3), dtype=np.float32) if deriv else None
        n = C.c_size_t()
        self._check(self._lib.rssync_ext_problem_matrix(self._h, int(frame), float(delay), _p(P, _PF),
                                                        _p(dP, _PF) if deriv else None, n_tracks, C.byref(n)))
        return (P[:n.value], dP[:n.value]) if deriv else P[:n.value]

    def problem_matrix64(self, frame, delay, n_tracks, deriv=False):
        """P (and dP/d-delay) as the Sync kernels compute it, fp64"""
        P = np.zeros((n_tracks, 3))
        dP = np.zeros((n_tracks, 3)) if deriv else None
        n = C.c_size_t()
        self._check(self._lib.rssync_ext_problem_matrix64(self._h, int(frame), float(delay), _p(P),
                                                          _p(dP) if deriv else None, n_tracks, C.byref(n)))
        return (P[:n.value], dP[:n.value]) if deriv else P[:n.value]

    def SyncSimplified(self, initial_delay, frame_begin, frame_end, search_center, search_radius):
        """Sync with translation neglected (thesis section 2.11 eq. (12)) -> (cost, delay); frame_end inclusive."""
        c, d = C.c_double(), C.c_double()
        self._check(self._lib.rssync_ext_sync_simplified(self._h, initial_delay, frame_begin, frame_end, search_center,
                                                         search_radius, C.byref(c), C.byref(d)))
        return c.value, d.value

    def init_k_simplified(self, delay, frame_begin, frame_end, cap=1 << 16):
        k, n = np.zeros(cap), C.c_int()
        self._check(self._lib.rssync_ext_init_k_simplified(self._h, delay, frame_begin, frame_end, _p(k), cap, C.byref(n)))
        return k[:n.value].copy()

    def loss_simplified(self, delays, grad=False):
        d = _d(np.atleast_1d(delays))
        out = np.zeros(d.shape[0])
        g = np.zeros(d.shape[0]) if grad else None
        self._check(self._lib.rssync_ext_loss_simplified(self._h, _p(d), d.shape[0], _p(out), _p(g) if grad else None))
        return (out, g) if grad else out

    def init_motion(self, delay, frame_begin, frame_end, cap=1 << 16):
        M, k, n = np.zeros((cap, 3)), np.zeros(cap), C.c_int()
        self._check(self._lib.rssync_ext_init_motion(self._h, delay, frame_begin, frame_end, _p(M), _p(k), cap,
                                                     C.byref(n)))
        return M[:n.value].copy(), k[:n.value].copy()

    def opt_motion(self, delay, cap=1 << 16):
        M, k, n = np.zeros((cap, 3)), np.zeros(cap), C.c_int()
        it, ev = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rssync_ext_opt_motion(self._h, delay, _p(M), _p(k), cap, C.byref(n), C.byref(it),
                                                    C.byref(ev)))
        return M[:n.value].copy(), k[:n.value].copy(), it.value, ev.value

    def set_motion(self, M, k):
        M, k = _d(M).reshape(-1, 3), _d(k)
        self._check(self._lib.rssync_ext_set_motion(self._h, _p(M), _p(k), k.shape[0]))

    def loss(self, delays, grad=False):
        d = _d(np.atleast_1d(delays))
        out = np.zeros(d.shape[0])
        g = np.zeros(d.shape[0]) if grad else None
        self._check(self._lib.rssync_ext_loss(self._h, _p(d), d.shape[0], _p(out), _p(g) if grad else None))
        return (out, g) if grad else out

    def set_track_pixels(self, frame, frame_time_a, frame_time_b, points_a, points_b, lens, image_rows):
        """Tracked pixel positions (n x 2 each) of one frame pair; undistortion, normalisation and
        row times (core_testcode.cpp:135-158) happen on the device.  lens = (ro, fx, fy, cx, cy,
        k1, k2, k3, k4)."""
        a = np.ascontiguousarray(points_a, np.float64)
        b = np.ascontiguousarray(points_b, np.float64)
        if a.ndim != 2 or a.shape[1] != 2 or a.shape != b.shape:
            raise ValueError("points_a / points_b must both be (n, 2)")
        L = np.ascontiguousarray(lens, np.float64)
        if L.shape != (9,):
            raise ValueError("lens = (ro, fx, fy, cx, cy, k1, k2, k3, k4)")
        self._check(self._lib.rssync_ext_set_track_pixels(self._h, int(frame), float(frame_time_a),
                                                          float(frame_time_b), _p(a), _p(b), a.shape[0],
                                                          L.ctypes.data, float(image_rows)))

    def set_gyro_rates(self, timestamps_s, rates, orientation=None):
        """Angular rates (n x 3, rad/s) at timestamps (s): integrated as the reference driver does
        (core_testcode.cpp:36-52) and passed to the timestamped gyro setter."""
        t = np.ascontiguousarray(timestamps_s, np.float64)
        r = np.ascontiguousarray(rates, np.float64)
        if r.ndim != 2 or r.shape[1] != 3 or t.shape != (r.shape[0],):
            raise ValueError("rates must be (n, 3) and timestamps (n,)")
        self._check(self._lib.rssync_ext_set_gyro_rates(self._h, _p(t), _p(r), r.shape[0],
                                                        orientation.encode() if orientation else None))

    def orientation_sweep(self, timestamps_s, rates, orientations, initial_delay, frame_begin, frame_end,
                          search_step, search_radius):
        """core_testcode.cpp:186-224: PreSync under every candidate IMU orientation ->
        (costs[n], delays[n]); the true orientation has the lowest cost."""
        t = np.ascontiguousarray(timestamps_s, np.float64)
        r = np.ascontiguousarray(rates, np.float64)
        if r.ndim != 2 or r.shape[1] != 3 or t.shape != (r.shape[0],):
            raise ValueError("rates must be (n, 3) and timestamps (n,)")
        names = (C.c_char_p * len(orientations))(*[o.encode() for o in orientations])
        costs, delays = np.zeros(len(orientations)), np.zeros(len(orientations))
        self._check(self._lib.rssync_ext_orientation_sweep(self._h, _p(t), _p(r), r.shape[0], names,
                                                           len(orientations), float(initial_delay), int(frame_begin),
                                                           int(frame_end), float(search_step), float(search_radius),
                                                           _p(costs), _p(delays)))
        return costs, delays

    def frame_rays(self, frame, cap=2048):
        """The packed device streams of one frame: ({ax,bx,ay,by}, {az,bz,ta,tb}) as (n, 4) float32."""
        a4, b4, n = np.zeros((cap, 4), np.float32), np.zeros((cap, 4), np.float32), C.c_size_t()
        self._check(self._lib.rssync_ext_frame_rays(self._h, int(frame), a4.ctypes.data, b4.ctypes.data, cap,
                                                    C.byref(n)))
        return a4[:n.value].copy(), b4[:n.value].copy()

    def pre_sync_windows(self, initial_delay, frame_begins, frame_ends, search_step, search_radius):
        """PreSync on W windows [begin[w], end[w]) in one sweep -> (costs[W], delays[W])."""
        b = np.ascontiguousarray(frame_begins, np.int64)
        e = np.ascontiguousarray(frame_ends, np.int64)
        if b.shape != e.shape or b.ndim != 1:
            raise ValueError("frame_begins / frame_ends must be 1-D and of equal length")
        costs, delays = np.zeros(b.size), np.zeros(b.size)
        self._check(self._lib.rssync_ext_pre_sync_windows(self._h, float(initial_delay), _p(b, _PI64), _p(e, _PI64), b.size,
                                                          float(search_step), float(search_radius), _p(costs),
                                                          _p(delays)))
        return costs, delays

    def sync_windows(self, initial_delays, frame_begins, frame_ends, search_center, search_radius):
        """Sync on W windows [begin[w], end[w]] advanced in lock-step -> (costs[W], delays[W])."""
        b = np.ascontiguousarray(frame_begins, np.int64)
        e = np.ascontiguousarray(frame_ends, np.int64)
        d0 = np.ascontiguousarray(np.broadcast_to(np.asarray(initial_delays, np.float64), b.shape))
        if b.shape != e.shape or b.ndim != 1:
            raise ValueError("frame_begins / frame_ends must be 1-D and of equal length")
        costs, delays = np.zeros(b.size), np.zeros(b.size)
        self._check(self._lib.rssync_ext_sync_windows(self._h, _p(d0), _p(b, _PI64), _p(e, _PI64), b.size,
                                                      float(search_center), float(search_radius), _p(costs),
                                                      _p(delays)))
        return costs, delays

    def sync_points(self, positions, sync_window, initial_delay, presync_step=None, presync_radius=None, repeats=4):
        """The reference driver's loop (core_testcode.cpp:303-316) over all sync points at once:
        optional PreSync, then `repeats` chained Sync calls per position -> (costs[W], delays[W])."""
        pos = np.ascontiguousarray(positions, np.int64)
        if pos.ndim != 1:
            raise ValueError("positions must be 1-D")
        use = presync_step is not None and presync_radius is not None
        costs, delays = np.zeros(pos.size), np.zeros(pos.size)
        self._check(self._lib.rssync_ext_sync_points(self._h, _p(pos, _PI64), pos.size, int(sync_window),
                                                     float(initial_delay), 1 if use else 0,
                                                     float(presync_step) if use else 0.0,
                                                     float(presync_radius) if use else 0.0, int(repeats), _p(costs),
                                                     _p(delays)))
        return costs, delays

    def window_trace(self, window, cap=2048):
        t, n = np.zeros((cap, 6)), C.c_int()
        self._check(self._lib.rssync_ext_window_trace(self._h, int(window), _p(t), cap, C.byref(n)))
        return t[:min(n.value, cap)].copy()

    def sync_trace(self, cap=512):
        t, n = np.zeros((cap, 6)), C.c_int()
        self._lib.rssync_ext_sync_trace(self._h, _p(t), cap, C.byref(n))
        return t[:min(n.value, cap)].copy()

    def device_context(self):
        """rship_ctx* of this problem (include/rssync_hip.h), for kernel-level tools."""
        return self._lib.rssync_ext_device_context(self._h)

    def profile(self, enable=True):
        self._check(self._lib.rssync_ext_profile(self._h, 1 if enable else 0))

    def profile_reset(self):
        self._check(self._lib.rssync_ext_profile_reset(self._h))

    def profile_get(self):
        names = ["lmeds", "loss", "motion", "reduce", "init", "pixels", "gyro", "loss_grad"]
        out = {}
        for i, nm in enumerate(names):
            n, ms = C.c_uint64(), C.c_double()
            self._check(self._lib.rssync_ext_profile_get(self._h, i, C.byref(n), C.byref(ms)))
            out[nm] = (n.value, ms.value)
        return out
