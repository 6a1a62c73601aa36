"""sola_amd - MI355X (gfx950) implementation of SOLA's track-selection hot path.

Python mirrors of the reference interfaces (module/module.py, tools/loss.py, track_generation/seg_utils.py)
over the C ABI of libsola_hip.so.  Importing the package does not load the library; the first compute call does,
and it raises if the library is missing (there is no CPU / PyTorch fallback).
"""
from ._lib import SolaError, SolaLibraryError  # noqa: F401

__all__ = ["SolaError", "SolaLibraryError"]
