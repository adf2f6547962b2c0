"""alicevision_amd — MI355X-native (gfx950) implementation of AliceVision's depth-map estimation hot path.

  abi        ctypes mirror of include/avdm.h (the C ABI of csrc/libavdm.so, hand-written HIP kernels)
  pipeline   per-tile sequencing of the ABI calls (SGM + Refine) on torch device buffers
  synthetic  seeded analytic multi-view scenes
  build      hipcc build of csrc/libavdm.so (in-tree)
"""
__all__ = ["abi", "pipeline", "synthetic", "build"]
