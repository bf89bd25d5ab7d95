"""smallhardface_amd — MI355X-native multi-scale face-detection inference hot path.

Host side is Python (as in the reference, lib/test.py) over a C-ABI HIP runtime
(``csrc/`` -> ``libshf_hip.so``, declared in ``include/shf_hip.h``).
"""
__version__ = "0.1.0"
