"""kernel_times.py against tools/diag/libvgpa_stamps.so (an experimental build of the library)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from vgpa_amd import _lib
dbg = ctypes.CDLL(os.path.join(ROOT, "tools/diag/libvgpa_stamps.so"))
for name, args in _lib.SIGNATURES.items():
    fn = getattr(dbg, name); fn.restype = ctypes.c_int; fn.argtypes = args
_lib.lib = dbg
sys.argv = [os.path.join(ROOT, "tools/diag/kernel_times.py")] + sys.argv[1:]
exec(open(sys.argv[0]).read())
