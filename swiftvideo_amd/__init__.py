"""swiftvideo_amd — MI355X-native compute backend for SwiftVideo's picture path.

`chipvideo` is the ctypes binding of the C ABI (include/chipvideo.h, implemented
by csrc/ as hand-written HIP for gfx950); `compute` mirrors the reference's
operator surface (ComputeContext / runComputeKernel / applyComputeImage /
VideoMixer / GPUBarrierUpload...) on top of it.
"""
from . import chipvideo  # noqa: F401

__version__ = "0.1.0"
