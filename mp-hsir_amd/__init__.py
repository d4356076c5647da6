"""mp-hsir_amd: MI355X-native (gfx950) implementation of the MP-HSIR forward/backward hot path.

Python here is glue: tensors, streams and torch.distributed.  The compute lives in libmphsir.so
(mp-hsir_amd/csrc, C ABI in include/mphsir.h).  Importing the package does not need a GPU; running
any op does, and fails loudly if libmphsir.so is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
