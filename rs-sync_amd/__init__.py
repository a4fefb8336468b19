"""rs-sync PreSync/Sync hot path on MI355X (gfx950).

The product is ``librssync_core.so`` (hand-written HIP kernels + C++ host
solver, built in-tree from ``csrc/``).  This package is only the Python-side
mirror of the reference's ``ISyncProblem`` surface (ctypes over the flat C-ABI
in ``include/rssync_c.h``), the synthetic-input generator used by the tests
and the benchmark, and the torch.distributed reduce hook for frame-sharded
multi-GPU runs.  There is no CPU fallback: importing works anywhere, creating
a ``SyncProblem`` needs a HIP device.
"""
from .problem import SyncProblem, RsSyncError, load_library, library_path  # noqa: F401
from . import synth  # noqa: F401
from . import dist  # noqa: F401
from . import quality  # noqa: F401
