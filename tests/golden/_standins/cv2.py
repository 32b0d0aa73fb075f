"""Build-owned stand-in for the one OpenCV function the reference's pHash calls.

OpenCV is not installed in the build container and is not part of the reference tree, so
`cv2.dct` (src/sig/phash.py:38) is substituted by SciPy's orthonormal DCT-II on float32,
which is the transform OpenCV documents for `cv2.dct`.  Used ONLY by make_golden.py.
"""
import numpy as np
from scipy.fft import dctn


def dct(src):
    a = np.asarray(src, dtype=np.float32)
    return dctn(a, type=2, norm="ortho").astype(np.float32)
