"""sz_amd -- MI355X-native SZ 2.1 hot path (predict + quantise + Huffman, and the inverse).

The product is the C-ABI shared library ``sz_amd/csrc/libszhip.so`` (HIP kernels for gfx950 + host C
exporting the reference's ``SZ_*`` API).  This package is only the Python mirror of that interface
used by tests and bench.py: same function names, argument meaning and error behaviour as the
reference's ``sz.h`` (see include/sz.h).  There is no CPU fallback: without the built library or
without a GPU the calls raise.
"""
from .api import (  # noqa: F401
    ABS, REL, VR_REL, ABS_AND_REL, ABS_OR_REL, PSNR, NORM, PW_REL,
    SZ_FLOAT, SZ_DOUBLE, SZ_BEST_SPEED, SZ_BEST_COMPRESSION, SZ_DEFAULT_COMPRESSION,
    SZ_SCES, SZ_NSCS, sz_params, szhip_stats, szhip_params,
    lib, lib_path, build_library, SZError,
    SZ_Init, SZ_Init_Params, SZ_Finalize, SZ_compress, SZ_compress_args, SZ_decompress, SZ_hip_last_stats,
    conf_params, HipContext, HipPool, make_meta,
)
