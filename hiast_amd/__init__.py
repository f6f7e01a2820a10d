"""hiast_amd — MI355X-native implementation of HIAST's self-training hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's `sseg` / `utils` / `workflows`
interfaces (same registries, class names, cfg keys, dict keys, state-dict keys).
Device side: hiast_amd/csrc/libhiast_hip.so, hand-written gfx950 kernels behind the C ABI
declared in include/hiast_hip.h, loaded by hiast_amd._lib.
"""
__version__ = "0.1.0"
