"""lanemapping_amd — MI355X-native inference hot path of WHU-USI3DV/LaneMapping.

HIP kernels + C-ABI live in ``csrc/`` (built into ``liblanemap_hip.so``); the Python side mirrors the
reference's registry/config interface (``boundary``) and is plumbing only.
"""
__version__ = '0.1.0'
