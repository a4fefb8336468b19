"""Result path of the reference driver and its accuracy metric.

The driver writes one `pos,delay_ms` CSV row per sync point (core_testcode.cpp:282,315) and a
`debug.csv` of the DebugPreSync cost curve (:285-301); python/plot_sync.py:19-20,46 then fits a
straight line (clock drift) through the delays and reports the standard deviation of the
residuals as "RMSE".  Host-side only: nothing here touches the device."""
import numpy as np


def linear_fit_rmse(positions, delays_ms):
    """(slope, intercept, rmse): least-squares line through (pos, delay_ms) and the standard
    deviation of the residuals (python/plot_sync.py:19-20,46 — `np.std`, population form)."""
    x = np.asarray(positions, np.float64)
    y = np.asarray(delays_ms, np.float64)
    if x.size < 2 or np.ptp(x) == 0:
        raise ValueError("need at least two distinct sync points")
    xm, ym = x.mean(), y.mean()
    slope = np.sum((x - xm) * (y - ym)) / np.sum((x - xm) ** 2)
    intercept = ym - slope * xm
    fit = intercept + slope * x
    return float(slope), float(intercept), float(np.std(fit - y))


def write_sync_csv(path, positions, delays_s):
    """core_testcode.cpp:315: `pos,1000*delay` per sync point."""
    with open(path, "w") as f:
        for p, d in zip(positions, delays_s):
            f.write("%d,%.9g\n" % (int(p), 1000.0 * float(d)))


def read_sync_csv(path):
    a = np.loadtxt(path, delimiter=",", ndmin=2)
    return a[:, 0].astype(np.int64), a[:, 1]


def write_debug_csv(path, delays, costs):
    """core_testcode.cpp:297-300: `delay,cost` rows of the DebugPreSync curve."""
    with open(path, "w") as f:
        for d, c in zip(delays, costs):
            f.write("%.9g,%.9g\n" % (float(d), float(c)))


def sync_points_auto(frame_start, frame_end, sync_window, syncpoint_distance):
    """core_testcode.cpp:270-272 ("auto" sync point format)."""
    return list(range(int(frame_start), int(frame_end) - int(sync_window), int(syncpoint_distance)))
