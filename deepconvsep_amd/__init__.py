"""deepconvsep_amd -- the MTG/DeepConvSep separation path on MI355X (gfx950).

Host side in Python (the reference's language), arithmetic in hand-written HIP
kernels behind the C ABI of ``include/dcs.h`` (``libdcs.so``, bound with
ctypes).  Importing the package does not need a GPU; creating a context does.
"""
from . import _lib
from .arch import (ARCHS, EPS_A, EPS_B, TIE_ALL, TIE_FIRST, TILER_LIBRARY, TILER_SCRIPT)
from .transform import TransformFFT, Transforms, compute_file, compute_inverse, sinebell, transformFFT
from .separation import (PredictFunction, Separator, blackmanharris, generate_overlapadd, load_model,
                         overlapadd, overlapadd_multi, save_model, train_auto)

__all__ = ["ARCHS", "EPS_A", "EPS_B", "TIE_ALL", "TIE_FIRST", "TILER_LIBRARY", "TILER_SCRIPT", "TransformFFT",
           "Transforms", "transformFFT", "compute_file", "compute_inverse", "sinebell", "PredictFunction",
           "Separator", "blackmanharris", "generate_overlapadd", "load_model", "save_model", "overlapadd",
           "overlapadd_multi", "train_auto"]
