"""nquant.android_amd -- MI355X (gfx950) implementation of the PNN / PNN-LAB colour-quantizer hot path of
mcychan/nQuant.android, behind the reference's own PnnQuantizer / PnnLABQuantizer.convert() interface.

The product is the C-ABI shared library libnquant_hip.so (include/nquant_abi.h, csrc/); this package is the
Python host-side mirror used by the tests and bench.py.  It never imports the CPU oracle and has no CPU fallback:
constructing a quantizer without a usable HIP device raises."""
from .host import (NQ_KIND_RGB, NQ_KIND_LAB, MODE_REFERENCE_SEQUENTIAL, MODE_PARALLEL_TILED, MODE_LOOKUP_ONLY,
                   NqError, Params, PnnQuantizer, PnnLABQuantizer, QuantizedImage, load_library, library_path,
                   abi_symbols, convert_batch_device, convert_batch_host)
from .build import build as build_library

__all__ = ["NQ_KIND_RGB", "NQ_KIND_LAB", "MODE_REFERENCE_SEQUENTIAL", "MODE_PARALLEL_TILED", "MODE_LOOKUP_ONLY",
           "NqError", "Params", "PnnQuantizer", "PnnLABQuantizer", "QuantizedImage", "load_library", "library_path",
           "abi_symbols", "build_library", "convert_batch_device", "convert_batch_host"]
