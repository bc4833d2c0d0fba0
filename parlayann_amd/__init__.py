"""parlayann_amd -- MI355X-native hot path of ParlayANN (beam search, robustPrune, leaf kNN).

Host-side mirror of the reference's operator surface for that path; all compute is in
libpann.so (hand-written HIP for gfx950) behind the C-ABI of include/pann.h.
"""
from ._capi import (PANN_BF16, PANN_F16, PANN_F32, PANN_I8, PANN_L2, PANN_MIPS, PANN_U8, PannError,  # noqa: F401
                    QueryParams)
from .bf16 import bfloat16, from_bf16, to_bf16  # noqa: F401
from .index import DeviceIndex, dtype_code  # noqa: F401

__all__ = ["DeviceIndex", "QueryParams", "PannError", "dtype_code", "bfloat16", "to_bf16", "from_bf16",
           "PANN_U8", "PANN_I8", "PANN_F32", "PANN_F16", "PANN_BF16", "PANN_L2", "PANN_MIPS"]
