"""MI355X-native stage-II candidate re-ranking (ViT-B/16 + stage-I BERT/MED + two-branch BERT).

Host side of the drop-in: mirrors the reference's `blip_stage2` model surface and its
`validate_stage2` scoring loop; all arithmetic runs in hand-written gfx950 HIP kernels behind the
C ABI declared in `include/cirrank.h` (`libcirrank.so`, loaded with ctypes by `.lib`).
Importing the package does not load the library; the first op does and fails loudly if it is absent.
"""
from .config import BertGeometry, VitGeometry, HEAD_DIM  # noqa: F401

__version__ = "0.1.0"
ABI_VERSION = 15  # CIR_ABI_VERSION in include/cirrank.h
