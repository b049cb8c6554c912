"""daala_amd - MI355X (gfx950) back-end for Daala's per-block transform + PVQ hot
path.  The product is the C-ABI library daala_amd/libdaala_hip.so
(include/daala_hip.h); this package is only the thin Python host mirror used by
the tests, bench.py and __graft_entry__.py.  There is no CPU fallback."""
from .binding import DaalaHip, HipError, lib_path, load  # noqa: F401
