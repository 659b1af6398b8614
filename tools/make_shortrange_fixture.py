"""Extract the calibrated TreePM short-range window table from the reference's own data file.

The reference ships the table as a C initialiser (libgadget/shortrange-kernel.c, 512 rows of
x, w_pot, w_force, w_pot_erf, w_force_erf).  oracle/Makefile compiles that file, as it lies in
/root/reference, into oracle/_ref/libshortrange_ref.so; this script reads the exported array
and writes it as raw little-endian float64 [512][5] into shenqi_amd/data/.  The data file is a
numeric calibration table (a fixture), not source text.
Run in the build container only:  python tools/make_shortrange_fixture.py
"""
import ctypes
import os
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "oracle", "_ref", "libshortrange_ref.so"))
arr = (ctypes.c_double * (512 * 5)).in_dll(lib, "shortrange_force_kernels")
tab = np.frombuffer(arr, dtype=np.float64).reshape(512, 5).copy()
assert tab[0, 0] == 0.0 and abs(tab[-1, 0] - 15.0) < 1e-12
out = os.path.join(root, "shenqi_amd", "data", "shortrange_force_kernels.f64")
tab.tofile(out)
print("wrote", out, tab.shape, "dx =", repr(tab[1, 0]))
