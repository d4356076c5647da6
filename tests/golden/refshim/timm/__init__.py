"""Test-only stand-in for the `timm` package (absent in this image).

Only used by tests/golden/make_golden.py so that /root/reference/net/MP_HSIR.py can be
imported on CPU to generate golden vectors.  Restates the three helpers the reference
imports (MP_HSIR.py:11) from their published semantics (timm 0.9.12).  Not product code.
"""
