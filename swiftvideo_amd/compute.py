"""Host-side mirror of SwiftVideo's compute/picture operator surface over CHIPVideo.

Same names, argument meaning and error behaviour as the reference's backend
contract (Sources/SwiftVideo/compute.swift, compute.cl.swift, mix.video.swift,
sample.pict.linux.swift), so that tests read like the reference's call sites:

    ctx = makeComputeContext(forType="GPU")
    gpu = uploadComputePicture(ctx, pict)
    ctx = usingContext(ctx, lambda c: applyComputeImage(c, image=gpu, target=backing, kernel=k))
    out = downloadComputePicture(ctx, backing)

All pixel work is done by libchipvideo.so on the GPU; this module only builds
descriptors (and does the 4x4 matrix algebra applyComputeImage does on the host,
compute.swift:149-161).
"""
import ctypes as C
import enum
import uuid

import numpy as np

from . import chipvideo as cv
from .chipvideo import ComputeError

__all__ = [
    "ComputeError", "ComputeKernel", "ComputeDevice", "ComputeContext", "ComputeBuffer", "PixelFormat",
    "Plane", "ImageBuffer", "PictureSample", "ImageUniforms", "defaultComputeKernelFromString",
    "availableComputeDevices", "hasAvailableComputeDevices", "makeComputeContext", "createComputeContext",
    "destroyComputeContext", "beginComputePass", "endComputePass", "usingContext", "runComputeKernel",
    "applyComputeImage", "uploadComputePicture", "downloadComputePicture", "uploadComputeBuffer",
    "downloadComputeBuffer", "createPictureSample", "GPUBarrierUpload", "GPUBarrierDownload", "VideoMixer",
    "compositeTick", "scaleLanczos", "LanczosBatch", "PictureFilter", "CustomKernel", "buildComputeKernel", "TickBatch", "VideoMixerGroup", "BufferImage",
]


# ---- enums --------------------------------------------------------------------
class ComputeKernel(enum.IntEnum):
    """`enum ComputeKernel`, compute.swift:49-74 (+ the BGRA-target kernels)."""
    img_nv12_nv12 = cv.K_IMG_NV12_NV12
    img_bgra_nv12 = cv.K_IMG_BGRA_NV12
    img_rgba_nv12 = cv.K_IMG_RGBA_NV12
    img_bgra_bgra = cv.K_IMG_BGRA_BGRA
    img_y420p_y420p = cv.K_IMG_Y420P_Y420P
    img_y420p_nv12 = cv.K_IMG_Y420P_NV12
    img_clear_nv12 = cv.K_IMG_CLEAR_NV12
    img_clear_yuvs = cv.K_IMG_CLEAR_YUVS
    img_clear_bgra = cv.K_IMG_CLEAR_BGRA
    img_clear_y420p = cv.K_IMG_CLEAR_Y420P
    img_clear_rgba = cv.K_IMG_CLEAR_RGBA
    img_rgba_y420p = cv.K_IMG_RGBA_Y420P
    img_bgra_y420p = cv.K_IMG_BGRA_Y420P
    snd_s16i_s16i = cv.K_SND_S16I_S16I
    me_fullsearch = cv.K_ME_FULLSEARCH
    img_nv12_bgra = cv.K_IMG_NV12_BGRA
    img_y420p_bgra = cv.K_IMG_Y420P_BGRA
    img_bgra_bgra_tx = cv.K_IMG_BGRA_BGRA_TX
    img_rgba_bgra_tx = cv.K_IMG_RGBA_BGRA_TX
    # integer BT.601/709 RGB -> YUV onto 4:2:0 canvases (the encoder side)
    img_bgra_nv12_int = cv.K_IMG_BGRA_NV12_INT
    img_rgba_nv12_int = cv.K_IMG_RGBA_NV12_INT
    img_bgra_y420p_int = cv.K_IMG_BGRA_Y420P_INT
    img_rgba_y420p_int = cv.K_IMG_RGBA_Y420P_INT

    def __str__(self):  # String(describing:)
        return self.name


def defaultComputeKernelFromString(name):
    """compute.swift:90-110; throws ComputeError.invalidValue for unknown names."""
    return ComputeKernel(cv.kernel_from_string(name))


class PixelFormat(enum.IntEnum):
    """`enum PixelFormat`, sample.pict.swift:20-33"""
    nv12 = cv.FMT_NV12
    nv21 = cv.FMT_NV21
    yuvs = cv.FMT_YUVS
    zvuy = cv.FMT_ZVUY
    y420p = cv.FMT_Y420P
    y422p = cv.FMT_Y422P
    y444p = cv.FMT_Y444P
    RGBA = cv.FMT_RGBA
    BGRA = cv.FMT_BGRA
    invalid = cv.FMT_INVALID


# ---- devices / contexts ---------------------------------------------------------
class ComputeDevice:
    """compute.cl.swift:36-44"""

    def __init__(self, info):
        self.deviceId = info.index
        self.available = bool(info.available)
        self.deviceType = "GPU"
        self.vendorId = info.vendor_id
        self.vendorName = info.name.decode()
        self.arch = info.arch.decode()
        self.supportsImages = bool(info.supports_images)
        self.computeUnits = info.compute_units
        self.totalMemory = info.total_memory


def availableComputeDevices():
    """compute.cl.swift:107-109"""
    try:
        n = cv.device_count()
    except ComputeError:
        return []
    return [ComputeDevice(cv.device_info(i)) for i in range(n)]


def hasAvailableComputeDevices(forType="GPU"):
    return any(d.deviceType == forType and d.available for d in availableComputeDevices())


class ComputeContext:
    """compute.cl.swift:75-105: device + shared state + one command queue (HIP stream)."""

    def __init__(self, handle, device):
        self._h = C.c_void_p(handle)
        self.device = device
        self.logger = None

    @property
    def handle(self):
        if not self._h:
            raise ComputeError(3, "context destroyed")
        return self._h


def createComputeContext(device=None, logger=None, sharing=None):
    """createComputeContext(_:logger:) / createComputeContext(sharing:), compute.cl.swift:111-145"""
    lib = cv.load()
    out = C.c_void_p()
    if sharing is not None:
        cv.check(lib.chv_context_share(sharing.handle, C.byref(out)))
        return ComputeContext(out.value, sharing.device)
    cv.check(lib.chv_context_create(device.deviceId, C.byref(out)))
    ctx = ComputeContext(out.value, device)
    ctx.logger = logger
    return ctx


def makeComputeContext(forType="GPU", index=0):
    """compute.swift:121-129; throws deviceNotAvailable when no device of that type is usable."""
    devices = [d for d in availableComputeDevices() if d.deviceType == forType and d.available]
    if index >= len(devices):
        raise ComputeError(8, f"no available {forType} compute device #{index}")
    return createComputeContext(devices[index])


def destroyComputeContext(ctx):
    cv.check(cv.load().chv_context_destroy(ctx.handle))
    ctx._h = C.c_void_p()


def beginComputePass(ctx):
    cv.check(cv.load().chv_pass_begin(ctx.handle))
    return ctx


def endComputePass(ctx, waitForCompletion):
    cv.check(cv.load().chv_pass_end(ctx.handle, 1 if waitForCompletion else 0))
    return ctx


def usingContext(ctx, fun):
    """compute.swift:131-134.  (When `fun` throws, the reference never reaches endComputePass; here the bracket is closed on the way out — the
    library holds a pass's kernels until its end, and a bracket left open would keep holding whatever the caller issues next.)"""
    ctx = beginComputePass(ctx)
    try:
        ctx = fun(ctx)
    except BaseException:
        cv.load().chv_pass_end(ctx.handle, 0)
        raise
    return endComputePass(ctx, True)


# ---- buffers / samples ------------------------------------------------------------
class ComputeBuffer:
    """compute.cl.swift:46-58; device memory freed when the last reference goes away."""

    def __init__(self, handle, size, pitch=0):
        self._h = C.c_void_p(handle)
        self.size = size
        self.pitch = pitch

    def __del__(self):
        try:
            if self._h:
                cv.load().chv_buffer_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class Plane:
    """sample.pict.swift:47-58"""

    def __init__(self, size, stride, bitDepth, components):
        self.size = (int(size[0]), int(size[1]))
        self.stride = int(stride)
        self.bitDepth = bitDepth
        self.components = list(components)


def planesForFormat(fmt, size):
    """sample.pict.linux.swift:275-294"""
    w, h = int(size[0]), int(size[1])
    if fmt == PixelFormat.nv12:
        return [Plane((w, h), w, 8, ["y"]), Plane((w // 2, h // 2), w, 8, ["cb", "cr"])]
    if fmt in (PixelFormat.BGRA, PixelFormat.RGBA):
        return [Plane((w, h), w * 4, 8, ["r", "g", "b", "a"])]
    if fmt == PixelFormat.y420p:
        return [Plane((w, h), w, 8, ["y"]), Plane((w // 2, h // 2), w // 2, 8, ["cb"]),
                Plane((w // 2, h // 2), w // 2, 8, ["cr"])]
    raise ComputeError(5, "Invalid pixel format")


class ImageBuffer:
    """sample.pict.linux.swift:23-72"""

    def __init__(self, pixelFormat, bufferType, size, computeTextures=(), buffers=(), planes=(), gpuPitches=(),
                 gpuOffsets=()):
        if not computeTextures and not buffers:
            raise ComputeError(5, "Must provide either compute textures or buffers")
        self.pixelFormat = pixelFormat
        self.bufferType = bufferType          # "cpu" | "gpu"
        self.size = (int(size[0]), int(size[1]))
        self.computeTextures = list(computeTextures)
        self.buffers = list(buffers)          # one uint8 ndarray (rows, stride) per plane
        self.planes = list(planes)
        self.gpuPitches = list(gpuPitches)
        # HIP backend: the planes of one picture live in ONE device allocation (computeTextures holds the same
        # ComputeBuffer for each plane) at these byte offsets, so that a picture whose host planes are contiguous
        # (sample.pict.linux.swift:296-311) goes up in one copy instead of one per plane
        self.gpuOffsets = list(gpuOffsets) if gpuOffsets else [0] * len(self.computeTextures)

    def withChanges(self, **kw):
        d = dict(pixelFormat=self.pixelFormat, bufferType=self.bufferType, size=self.size,
                 computeTextures=self.computeTextures, buffers=self.buffers, planes=self.planes,
                 gpuPitches=self.gpuPitches, gpuOffsets=self.gpuOffsets)
        d.update(kw)
        return ImageBuffer(**d)


_IDENT = np.eye(4, dtype=np.float64)


class PictureSample:
    """sample.pict.linux.swift:105-249 (the fields the picture path reads)."""

    def __init__(self, img, assetId="", workspaceId="", time=0.0, pts=0.0, matrix=None, textureMatrix=None,
                 borderMatrix=None, fillColor=(0, 0, 0, 0), opacity=1.0, zIndex=0, revision=None, eventInfo=None):
        self.img = img
        self._assetId, self._workspaceId = assetId, workspaceId
        self._time, self._pts = time, pts
        self._matrix = _IDENT if matrix is None else np.asarray(matrix, dtype=np.float64)
        self._textureMatrix = _IDENT if textureMatrix is None else np.asarray(textureMatrix, dtype=np.float64)
        self._borderMatrix = self._matrix if borderMatrix is None else np.asarray(borderMatrix, dtype=np.float64)
        self._fillColor = tuple(float(v) for v in fillColor)
        self._opacity = float(opacity)
        self._zIndex = zIndex
        self._revision = revision or uuid.uuid4().hex
        self._info = eventInfo

    def derive(self, **kw):
        d = dict(img=self.img, assetId=self._assetId, workspaceId=self._workspaceId, time=self._time,
                 pts=self._pts, matrix=self._matrix, textureMatrix=self._textureMatrix,
                 borderMatrix=self._borderMatrix, fillColor=self._fillColor, opacity=self._opacity,
                 zIndex=self._zIndex, revision=self._revision, eventInfo=self._info)
        d.update(kw)
        return PictureSample(**d)

    def imageBuffer(self): return self.img
    def pixelFormat(self): return self.img.pixelFormat
    def bufferType(self): return self.img.bufferType
    def size(self): return self.img.size
    def matrix(self): return self._matrix
    def textureMatrix(self): return self._textureMatrix
    def borderMatrix(self): return self._borderMatrix
    def fillColor(self): return self._fillColor
    def opacity(self): return self._opacity
    def zIndex(self): return self._zIndex
    def revision(self): return self._revision
    def assetId(self): return self._assetId
    def workspaceId(self): return self._workspaceId
    def time(self): return self._time
    def pts(self): return self._pts
    def info(self): return self._info


def createPictureSample(size, fmt, assetId="", workspaceId="", workspaceToken=None):
    """sample.pict.linux.swift:254-273: CPU sample with zeroed planes."""
    if not (size[0] > 0 and size[1] > 0):
        raise ComputeError(10, "size must be positive")
    planes = planesForFormat(fmt, size)
    buffers = [np.zeros((max(p.size[1], 1), p.stride), dtype=np.uint8) for p in planes]
    img = ImageBuffer(fmt, "cpu", size, buffers=buffers, planes=planes)
    return PictureSample(img, assetId=assetId, workspaceId=workspaceId)


def pictureFromArrays(fmt, size, arrays, **kw):
    """Convenience: wrap existing plane arrays (rows, stride) as a CPU PictureSample."""
    planes = planesForFormat(fmt, size)
    bufs = []
    for p, a in zip(planes, arrays):
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(max(p.size[1], 1), -1)
        bufs.append(a)
        p.stride = a.shape[1]
    return PictureSample(ImageBuffer(fmt, "cpu", size, buffers=bufs, planes=planes), **kw)


# ---- transfers ----------------------------------------------------------------------
def _plane_comps(plane):
    comps = len(plane.components)
    return 4 if comps >= 3 else comps


def _createTexture(ctx, image, maxPlanes=3):
    """compute.cl.swift:532-581 creates one device image per plane (R8 / RG8 / RGBA8); here the planes of a picture are
    linear pitched regions of ONE allocation (128-byte aligned pitches, each plane starting where the previous one ends)."""
    if image.bufferType != "cpu":
        return image.computeTextures, image.gpuPitches, image.gpuOffsets
    n = len(image.planes)
    if not (0 < n <= 3):
        raise ComputeError(5, "Input image must have 1, 2, or 3 planes")
    if n != len(image.buffers):
        raise ComputeError(5, f"Input image must have the same number of buffers as planes: {len(image.buffers)} vs. {n}")
    pitches, offsets, total = [], [], 0
    for p in image.planes[: min(n, maxPlanes)]:
        if p.size[0] <= 0 or p.size[1] <= 0:
            raise ComputeError(10, f"plane of size {p.size[0]}x{p.size[1]}")
        pitch = (p.size[0] * _plane_comps(p) + 127) // 128 * 128
        pitches.append(pitch)
        offsets.append(total)
        total += pitch * p.size[1]
    h = C.c_void_p()
    cv.check(cv.load().chv_buffer_alloc(ctx.handle, total, C.byref(h)))
    tex = ComputeBuffer(h.value, total)
    return [tex] * len(pitches), pitches, offsets


def _upload_regions(image, pitches, offsets):
    """(dst_offset, dst_pitch, src_address, src_pitch, width_bytes, rows) per copy: planes that are adjacent with equal
    pitches and widths on both sides (Y + interleaved chroma of NV12; the two chroma planes of y420p) travel as one."""
    regions = []
    for off, pitch, buf, plane in zip(offsets, pitches, image.buffers, image.planes):
        wb, rows, src = plane.size[0] * _plane_comps(plane), plane.size[1], buf.ctypes.data
        if regions:
            o, p, s0, sp, w, r = regions[-1]
            if p == pitch and sp == plane.stride and w == wb and o + p * r == off and s0 + sp * r == src:
                regions[-1] = (o, p, s0, sp, w, r + rows)
                continue
        regions.append((off, pitch, src, plane.stride, wb, rows))
    return regions


def uploadComputePicture(ctx, pict, maxPlanes=3, retainCpuBuffer=True, asynchronous=False):
    """compute.cl.swift:421-459"""
    if pict.bufferType() != "cpu":
        return pict
    image = pict.imageBuffer()
    if image is None:
        raise ComputeError(5, "Missing image buffer")
    texs, pitches, offsets = _createTexture(ctx, image, maxPlanes)
    lib = cv.load()
    beginComputePass(ctx)
    for off, pitch, src, src_pitch, wb, rows in _upload_regions(image, pitches, offsets):
        cv.check(lib.chv_upload(ctx.handle, texs[0]._h, off, pitch, src, src_pitch, wb, rows, 1 if asynchronous else 0))
    # asynchronous: the bytes are already staged in pinned memory and the copies are ordered on this
    # context's stream; kernels on other contexts wait on the planes' upload events, so no host stall
    endComputePass(ctx, not asynchronous)
    img = image.withChanges(computeTextures=texs, gpuPitches=pitches, gpuOffsets=offsets,
                            buffers=image.buffers if retainCpuBuffer else [], bufferType="gpu")
    return pict.derive(img=img)


class PictureSlab:
    """`count` device pictures of one size and format in ONE allocation, frame after frame — the device side of an upload ring.
    A decoder that writes its frames back to back into pinned host memory (same plane order, rows packed: the layout of
    sample.pict.linux.swift:296-311) gets `frames` of them onto the device with ONE linear copy (`upload`): on this link a
    3 MiB copy reaches 48 GB/s, copies of 8 MiB and more 55-57 (tools/h2d_probe.py).  `pictures[i]` are ordinary GPU
    PictureSamples (views with plane offsets); packed rows need width x components to be a multiple of 128 bytes."""

    def __init__(self, ctx, size, fmt, count):
        proto = createPictureSample(size, fmt).imageBuffer()
        pitches, offsets, total = [], [], 0
        for p in proto.planes:
            pitch = p.size[0] * _plane_comps(p)
            if pitch % 128:
                raise ComputeError(1, f"PictureSlab needs packed rows of a multiple of 128 bytes, got {pitch}")
            pitches.append(pitch); offsets.append(total); total += pitch * p.size[1]
        self.frameBytes, self.count = total, count
        h = C.c_void_p()
        cv.check(cv.load().chv_buffer_alloc(ctx.handle, total * count, C.byref(h)))
        self.buffer = ComputeBuffer(h.value, total * count)
        self.pictures = []
        for i in range(count):
            img = proto.withChanges(computeTextures=[self.buffer] * len(pitches), gpuPitches=pitches,
                                    gpuOffsets=[i * total + o for o in offsets], buffers=[], bufferType="gpu")
            self.pictures.append(PictureSample(img))

    def upload(self, ctx, first, frames, host_address, mode=2):
        """frames [first, first + frames) from `frames` packed host frames at host_address, one linear copy on ctx's stream
        (mode 2: caller-pinned memory, chv_host_alloc; 1: staged; 0: synchronous)"""
        n = self.frameBytes * frames
        cv.check(cv.load().chv_upload(ctx.handle, self.buffer._h, first * self.frameBytes, n, host_address, n, n, 1, mode))

    def download(self, ctx, first, frames, host_address):
        """frames [first, first + frames) into `frames` packed host frames at host_address (pinned: chv_host_alloc), one linear copy on ctx's
        stream, NOT waited for (chv_download_async): the bytes are there once ctx's stream has passed the copy"""
        n = self.frameBytes * frames
        cv.check(cv.load().chv_download_async(ctx.handle, host_address, n, self.buffer._h, first * self.frameBytes, n, n, 1))


def downloadComputePicture(ctx, pict, retainGpuBuffer=False):
    """compute.cl.swift:461-498"""
    if pict.bufferType() != "gpu":
        return pict
    image = pict.imageBuffer()
    if image is None:
        raise ComputeError(5, "Missing image buffer")
    lib = cv.load()
    beginComputePass(ctx)
    bufs = []
    for idx, tex in enumerate(image.computeTextures):
        plane = image.planes[idx]
        comps = len(plane.components)
        comps = 4 if comps >= 3 else comps
        buf = image.buffers[idx] if idx < len(image.buffers) else np.zeros((max(plane.size[1], 1), plane.stride), dtype=np.uint8)
        cv.check(lib.chv_download(ctx.handle, buf.ctypes.data, plane.stride, tex._h, image.gpuOffsets[idx], image.gpuPitches[idx],
                                  plane.size[0] * comps, plane.size[1]))
        bufs.append(buf)
    endComputePass(ctx, True)
    img = image.withChanges(computeTextures=image.computeTextures if retainGpuBuffer else [],
                            gpuPitches=image.gpuPitches if retainGpuBuffer else [],
                            gpuOffsets=image.gpuOffsets if retainGpuBuffer else [], buffers=bufs, bufferType="cpu")
    return pict.derive(img=img)


def uploadComputeBuffer(ctx, src, dst=None):
    """compute.cl.swift:361-379"""
    data = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8))
    lib = cv.load()
    if dst is None:
        h = C.c_void_p()
        cv.check(lib.chv_buffer_alloc(ctx.handle, data.size, C.byref(h)))
        dst = ComputeBuffer(h.value, data.size)
    if dst.size < data.size:
        raise ComputeError(5, "Compute buffer needs to be >= to data.count")
    cv.check(lib.chv_upload(ctx.handle, dst._h, 0, data.size, data.ctypes.data, data.size, data.size, 1, 0))
    return dst


def downloadComputeBuffer(ctx, src, dst=None):
    """compute.cl.swift:381-396"""
    out = np.zeros(src.size, dtype=np.uint8) if dst is None else dst
    if out.size < src.size:
        raise ComputeError(5, "Destination data buffer must be >= buffer.size")
    cv.check(cv.load().chv_download(ctx.handle, out.ctypes.data, src.size, src._h, 0, src.size, src.size, 1))
    return out


# ---- kernels ------------------------------------------------------------------------------
class ImageUniforms:
    """compute.swift:76-86.  `blob()` is the 236 bytes the Swift struct occupies."""

    def __init__(self, transform, textureTransform, borderMatrix, fillColor, inputSize, outputSize,
                 opacity, imageTime=0.0, targetTime=0.0):
        self.transform, self.textureTransform, self.borderMatrix = transform, textureTransform, borderMatrix
        self.fillColor, self.inputSize, self.outputSize = fillColor, inputSize, outputSize
        self.opacity, self.imageTime, self.targetTime = opacity, imageTime, targetTime

    def blob(self):
        u = np.zeros(59, dtype=np.float32)
        u[0:16] = np.asarray(self.transform, dtype=np.float32).reshape(-1)
        u[16:32] = np.asarray(self.textureTransform, dtype=np.float32).reshape(-1)
        u[32:48] = np.asarray(self.borderMatrix, dtype=np.float32).reshape(-1)
        u[48:52] = self.fillColor
        u[52:54] = self.inputSize
        u[54:56] = self.outputSize
        u[56], u[57], u[58] = self.opacity, self.imageTime, self.targetTime
        return u


class BufferImage:
    """A ComputeBuffer bound where a kernel expects an image: one plane of `width` x `height` texels of `components` bytes.  The two kernels of
    `enum ComputeKernel` that work on buffers rather than pictures take these — snd_s16i_s16i (kernels.cl.swift:534-562: interleaved-stereo
    int16 samples = 2-byte texels) — and me_fullsearch's output (one RGBA8 texel per block) may be one as well."""

    def __init__(self, buffer, width, height=1, components=2, pitch=None, offset=0):
        self.buffer, self.width, self.height, self.components = buffer, int(width), int(height), int(components)
        self.pitch = int(pitch) if pitch else self.width * self.components
        self.offset = int(offset)


def _image_desc(sample, maxPlanes=3):
    if isinstance(sample, BufferImage):
        d = cv.Image()
        d.format = cv.FMT_INVALID
        d.width, d.height, d.n_planes = sample.width, sample.height, 1
        d.planes[0] = cv.Plane(sample.buffer._h, sample.offset, sample.width, sample.height, sample.pitch, sample.components)
        return d
    image = sample.imageBuffer()
    if image is None or image.bufferType != "gpu" or not image.computeTextures:
        return None
    d = cv.Image()
    d.format = int(image.pixelFormat)
    d.width, d.height = image.size
    n = min(len(image.computeTextures), maxPlanes)
    d.n_planes = n
    for i in range(n):
        p = image.planes[i]
        comps = len(p.components)
        comps = 4 if comps >= 3 else comps
        d.planes[i] = cv.Plane(image.computeTextures[i]._h, image.gpuOffsets[i], p.size[0], p.size[1], image.gpuPitches[i], comps)
    return d


def _uniform_blob(uniforms, raw=False):
    """The bytes bound as the kernel's uniforms (MemoryLayout<T>.size of them, compute.cl.swift:508-512).  Picture kernels take ImageUniforms:
    anything array-like is its 59 floats and is converted to float32 (a list or an int64 array of matrix rows stays what it always was).
    `raw` (the two buffer kernels and .custom kernels, whose uniforms are other value types — BufferUniforms, MotionEstimationUniforms, a
    user's struct): uint8 / int32 / uint32 arrays pass as the bytes they are.  bytes / bytearray always pass unchanged."""
    if uniforms is None:
        return None
    if isinstance(uniforms, ImageUniforms):
        return uniforms.blob()
    if isinstance(uniforms, (bytes, bytearray)):
        return np.frombuffer(bytes(uniforms), dtype=np.uint8)
    if raw and isinstance(uniforms, np.ndarray) and uniforms.dtype in (np.uint8, np.int32, np.uint32):
        return np.ascontiguousarray(uniforms).reshape(-1)
    return np.ascontiguousarray(uniforms, dtype=np.float32).reshape(-1)


class CustomKernel:
    """`ComputeKernel.custom(name:)`, compute.swift:72-73: a kernel built at run time with buildComputeKernel."""

    def __init__(self, name):
        self.name = name

    def __str__(self):
        return self.name


def buildComputeKernel(ctx, name, source):
    """compute.cl.swift:153-195: compile `source` (HIP C++ here, prefixed with cv.custom_prelude()) and put its
    kernel `name` into the context's library; throws ComputeError.badInputData with the build log on failure."""
    cv.check(cv.load().chv_kernel_build(ctx.handle, name.encode(), source.encode()))
    return ctx


def runComputeKernel(ctx, images, target, kernel, maxPlanes=3, requiredMemory=None, uniforms=None,
                     blends=False, colorspace=cv.CSC_BT601_LIMITED):
    """Both overloads of runComputeKernel, compute.cl.swift:250-344."""
    tdesc = _image_desc(target)
    if tdesc is None:
        raise ComputeError(4, "target has no GPU image buffer")
    descs = (cv.Image * max(1, len(images)))()
    for i, im in enumerate(images):
        d = _image_desc(im, maxPlanes)
        if d is None:
            raise ComputeError(5, "Bad input image")
        descs[i] = d
    if isinstance(kernel, CustomKernel):      # ComputeKernel.custom(name:), compute.swift:72-73
        # uniforms: any value type in the reference (MemoryLayout<T>.size bytes are bound, compute.cl.swift:508-512)
        if isinstance(uniforms, (bytes, bytearray)):
            u = np.frombuffer(bytes(uniforms), dtype=np.uint8)
        else:
            u = _uniform_blob(uniforms, raw=True)
        cv.check(cv.load().chv_run_custom(ctx.handle, kernel.name.encode(), C.byref(tdesc), descs, len(images),
                                          u.ctypes.data if u is not None else None,
                                          u.nbytes if u is not None else 0, 1 if blends else 0))
        return ctx
    u = _uniform_blob(uniforms, raw=int(kernel) in (int(ComputeKernel.snd_s16i_s16i), int(ComputeKernel.me_fullsearch)))
    opts = cv.KernelOpts(colorspace=int(colorspace))
    cv.check(cv.load().chv_run_kernel(ctx.handle, int(kernel), C.byref(tdesc), descs, len(images),
                                      u.ctypes.data if u is not None else None,
                                      u.nbytes if u is not None else 0, 1 if blends else 0, C.byref(opts)))
    return ctx


def imageUniformsFor(image, target):
    """The uniforms applyComputeImage builds, compute.swift:147-161: each matrix is
    M.inverse.transpose in VectorMath's layout, i.e. kernel row i = row i of M^-1."""
    inv = lambda m: np.linalg.inv(np.asarray(m, dtype=np.float64))
    return ImageUniforms(transform=inv(image.matrix()), textureTransform=inv(image.textureMatrix()),
                         borderMatrix=inv(image.borderMatrix()), fillColor=image.fillColor(),
                         inputSize=image.size(), outputSize=target.size(), opacity=image.opacity(),
                         imageTime=float(image.time()), targetTime=float(target.time()))


def applyComputeImage(ctx, image, target, kernel, colorspace=cv.CSC_BT601_LIMITED):
    """compute.swift:145-170"""
    return runComputeKernel(ctx, images=[image], target=target, kernel=kernel, maxPlanes=3,
                            uniforms=imageUniformsFor(image, target), blends=True, colorspace=colorspace)


def _layer_array(layers):
    """layers: [(kernel, PictureSample, uniforms blob or ImageUniforms, colorspace)]"""
    arr = (cv.Layer * max(1, len(layers)))()
    for i, (kernel, sample, uniforms, csc) in enumerate(layers):
        d = _image_desc(sample)
        if d is None:
            raise ComputeError(5, "Bad input image")
        arr[i].kernel = int(kernel)
        arr[i].image = d
        u = _uniform_blob(uniforms)
        C.memmove(C.byref(arr[i].uniforms), u.ctypes.data, 236)
        arr[i].opts = cv.KernelOpts(colorspace=int(csc))
    return arr


def compositeTick(ctx, target, layers, clearFirst=True):
    """One mixer tick in one launch (chv_composite); byte-identical to the clear +
    per-layer applyComputeImage sequence of mix.video.swift:116-124."""
    tdesc = _image_desc(target)
    if tdesc is None:
        raise ComputeError(4, "target has no GPU image buffer")
    arr = _layer_array(layers)
    cv.check(cv.load().chv_composite(ctx.handle, C.byref(tdesc), 1 if clearFirst else 0, arr, len(layers)))
    return ctx


class TickBatch:
    """Many independent mixer ticks issued as ONE launch (chv_batch_*): what a host with several mixers / streams on a
    device (composer.swift:203-224) uses instead of one chv_composite per tick.  Byte-identical to running the ticks
    one by one.  ticks: [(target PictureSample, clearFirst, [(kernel, PictureSample, uniforms, colorspace)])].
    The batch keeps the descriptors on the device; the pictures it refers to must stay alive until it is destroyed."""

    def __init__(self, ctx, ticks):
        lib = cv.load()
        arr = (cv.Tick * max(1, len(ticks)))()
        self._keep = [arr]
        for i, (target, clear, layers) in enumerate(ticks):
            tdesc = _image_desc(target)
            if tdesc is None:
                raise ComputeError(4, "target has no GPU image buffer")
            la = _layer_array(layers)
            self._keep += [la, target, [l[1] for l in layers]]
            arr[i].target = tdesc
            arr[i].clear_first = 1 if clear else 0
            arr[i].n_layers = len(layers)
            arr[i].layers = la
        self._h = C.c_void_p()
        cv.check(lib.chv_batch_create(ctx.handle, arr, len(ticks), C.byref(self._h)))
        name = C.create_string_buffer(128)
        launches = C.c_int(0)
        cv.check(lib.chv_batch_describe(self._h, name, 128, C.byref(launches)))
        self.kernelName, self.launches, self.count = name.value.decode(), launches.value, len(ticks)

    def run(self, ctx):
        """Enqueue the batch on ctx's stream (inside a compute pass, like runComputeKernel)."""
        cv.check(cv.load().chv_batch_run(ctx.handle, self._h))
        return ctx

    def destroy(self):
        if self._h:
            cv.check(cv.load().chv_batch_destroy(self._h))
            self._h = C.c_void_p()
            self._keep = []

    def __del__(self):
        try:
            self.destroy()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


def scaleLanczos(ctx, dst, src):
    d, s = _image_desc(dst), _image_desc(src)
    if d is None:
        raise ComputeError(4, "target has no GPU image buffer")
    if s is None:
        raise ComputeError(5, "Bad input image")
    cv.check(cv.load().chv_scale_lanczos(ctx.handle, C.byref(d), C.byref(s)))
    return ctx


class LanczosBatch:
    """n Lanczos-3 resizes of one geometry issued as one launch per 64 pairs (CHV_LANCZOS_BATCH_CHUNK; chv_scale_lanczos_batch): what a host with
    several streams per device uses per tick instead of n launches.  pairs: [(dst PictureSample, src PictureSample)]; the
    descriptors are built once, `run` can be called every tick (canvas rings make the same pairs recur)."""

    def __init__(self, pairs):
        n = len(pairs)
        self.n = n
        self._d, self._s = (cv.Image * max(1, n))(), (cv.Image * max(1, n))()
        self._keep = list(pairs)
        for i, (dst, src) in enumerate(pairs):
            d, s = _image_desc(dst), _image_desc(src)
            if d is None:
                raise ComputeError(4, "target has no GPU image buffer")
            if s is None:
                raise ComputeError(5, "Bad input image")
            self._d[i], self._s[i] = d, s

    def run(self, ctx):
        if self.n == 0:          # an empty list is a no-op, as in the C++ host (scaleLanczos(ctx, pairs))
            return ctx
        cv.check(cv.load().chv_scale_lanczos_batch(ctx.handle, self._d, self._s, self.n))
        return ctx


# ---- pipeline operators -----------------------------------------------------------------------
class GPUBarrierUpload:
    """Tx<PictureSample, PictureSample>, compute.swift:175-198: owns a context sharing
    the given one; passes GPU samples through; errors become ("error", EventError-like)."""

    def __init__(self, context, retainCpuBuffer=True):
        self.context = createComputeContext(sharing=context)
        self.retainCpuBuffer = retainCpuBuffer

    def __call__(self, sample):
        if sample.bufferType() == "cpu":
            try:
                return ("just", uploadComputePicture(self.context, sample, retainCpuBuffer=self.retainCpuBuffer))
            except ComputeError as e:
                return ("error", ("barrier.upload", -1, str(e), sample.assetId()))
        return ("just", sample)


class GPUBarrierDownload:
    """compute.swift:232-255"""

    def __init__(self, context, retainGpuBuffer=True):
        self.context = createComputeContext(sharing=context)
        self.retainGpuBuffer = retainGpuBuffer

    def __call__(self, sample):
        if sample.bufferType() == "gpu":
            try:
                return ("just", downloadComputePicture(self.context, sample, retainGpuBuffer=self.retainGpuBuffer))
            except ComputeError as e:
                return ("error", ("barrier.download", -1, str(e), sample.assetId()))
        return ("just", sample)


class PictureFilter:
    """Tx<PictureSample, PictureSample> that converts a picture to `outputFormat` at `outputSize` on the
    device — the operator the reference sketches and leaves commented out (filter.pict.swift:20-47: same
    constructor shape: a context of its own, sharing the given one).  One full-canvas layer through the
    composite kernels: colour conversion + bilinear scale in one launch (`scaler="bilinear"`, any format
    pair the kernel table has), or a separable Lanczos-3 resample (`scaler="lanczos"`, BGRA -> BGRA).
    CPU samples are uploaded first; the sample's time stamps, ids and transform state are carried over.
    Results land in a ring of `numberBackingImages` device images like the mixer's (mix.video.swift:148-167)."""

    numberBackingImages = 10

    def __init__(self, outputSize, outputFormat=PixelFormat.BGRA, computeContext=None, scaler="bilinear",
                 colorspace=cv.CSC_BT601_LIMITED, integerMatrix=True):
        """integerMatrix: an RGB picture converted to a 4:2:0 format goes through the integer BT.601/709 matrix of `colorspace`
        (img_*_int, DESIGN.md 4.5: what an encoder expects); False selects the reference's own float kernels (img_bgra_nv12 ...,
        full range, kernels.cl.swift:96-99)."""
        self.integerMatrix = integerMatrix
        if scaler not in ("bilinear", "lanczos"):
            raise ComputeError(0, f"unknown scaler {scaler!r}")
        try:
            self.context = createComputeContext(sharing=computeContext) if computeContext is not None \
                else makeComputeContext(forType="GPU")
        except ComputeError:
            self.context = None                     # filter.pict.swift:31-33
        self.outputSize, self.outputFormat = outputSize, outputFormat
        self.scaler, self.colorspace = scaler, colorspace
        self.backing, self.currentBacking = [], 0

    def findKernel(self, image):
        """same naming rule as VideoMixer.findKernel (mix.video.swift:142-146)"""
        inp, outp = str(image.pixelFormat().name).lower(), str(PixelFormat(self.outputFormat).name).lower()
        name = f"img_{inp}_{outp}"
        if outp == "bgra" and inp in ("bgra", "rgba"):
            name += "_tx"
        if outp in ("nv12", "y420p") and inp in ("bgra", "rgba") and self.integerMatrix:
            name += "_int"
        return defaultComputeKernelFromString(name)

    def _backing(self, like):
        if len(self.backing) < self.numberBackingImages:
            image = createPictureSample(self.outputSize, self.outputFormat, assetId=like.assetId(),
                                        workspaceId=like.workspaceId())
            self.backing.append(uploadComputePicture(self.context, image))
            return self.backing[-1]
        image = self.backing[self.currentBacking]
        self.currentBacking = (self.currentBacking + 1) % len(self.backing)
        return image

    def __call__(self, sample):
        if self.context is None:
            return ("error", ("filter.pict", -1, "No Compute Context", sample.assetId()))
        ctx = self.context
        try:
            src = uploadComputePicture(ctx, sample) if sample.bufferType() == "cpu" else sample
            dst = self._backing(sample)
            beginComputePass(ctx)
            if self.scaler == "lanczos":
                if src.pixelFormat() != PixelFormat.BGRA or self.outputFormat != PixelFormat.BGRA:
                    raise ComputeError(9, "lanczos: BGRA -> BGRA only")
                scaleLanczos(ctx, dst, src)
            else:
                # a full-canvas opaque layer: identity placement, no border, no fill
                full = src.derive(matrix=_unit_quad_to_ndc(), textureMatrix=np.eye(4), borderMatrix=_unit_quad_to_ndc(),
                                  fillColor=(0.0, 0.0, 0.0, 0.0), opacity=1.0)
                compositeTick(ctx, dst, [(self.findKernel(src), full, imageUniformsFor(full, dst), self.colorspace)],
                              clearFirst=True)
            endComputePass(ctx, True)
            return ("just", dst.derive(pts=sample.pts(), time=sample.time(), assetId=sample.assetId(),
                                       workspaceId=sample.workspaceId(), matrix=sample.matrix(),
                                       textureMatrix=sample.textureMatrix(), borderMatrix=sample.borderMatrix(),
                                       fillColor=sample.fillColor(), opacity=sample.opacity(), zIndex=sample.zIndex(),
                                       revision=sample.revision()))
        except ComputeError as e:
            return ("error", ("filter.pict", -2, f"Compute error {e}", sample.assetId()))


class VideoMixerGroup:
    """Several VideoMixers of one device ticked together: their canvases are composed by one launch per canvas format
    (TickBatch) and one host wait, instead of one launch and one host wait per mixer (mix.video.swift:116-124 per mixer).  Each mixer keeps its own samples,
    backing ring and z-order; the result of `mix(at)` is the list of what every mixer's own `mix(at)` would return."""

    def __init__(self, mixers):
        if not mixers:
            raise ComputeError(0, "empty mixer group")
        self.mixers = list(mixers)
        self.context = self.mixers[0].clContext

    def mix(self, at=0.0):
        ticks, backings = [], []
        try:
            for m in self.mixers:
                backing = m.getBacking()
                merged = dict(m.samples[1])
                merged.update(m.samples[0])
                images = sorted(merged.values(), key=lambda s: s.zIndex())
                ticks.append((backing, True, [(m.findKernel(im, backing), im, imageUniformsFor(im, backing), m.colorspace)
                                              for im in images]))
                backings.append(backing)
            # a batch is one kernel family: one batch per canvas format, all enqueued in one pass
            by_format = {}
            for tick in ticks:
                by_format.setdefault(int(tick[0].pixelFormat()), []).append(tick)
            batches = [TickBatch(self.context, group) for group in by_format.values()]
            try:
                def body(c):
                    for batch in batches:
                        c = batch.run(c)
                    return c
                usingContext(self.context, body)
            finally:
                for batch in batches:
                    batch.destroy()
            out = [b.derive(pts=at, time=at, assetId=m.assetId()) for b, m in zip(backings, self.mixers)]
            for m in self.mixers:
                m.result = ("nothing", None)
            return out
        except ComputeError as e:
            for m in self.mixers:
                m.result = ("error", ("mix.video", -2, f"Compute error {e}", at, m.idAsset))
            return None
        finally:
            for m in self.mixers:
                m.samples[1] = m.samples[0]
                m.samples[0] = dict()


def _unit_quad_to_ndc():
    """the unit quad [0,1]^2 stretched over the whole canvas in NDC [-1,1]^2 (what PictureAnimator
    produces for a picture at (0,0) with the canvas' size: ortho * T(0) * S(canvas))"""
    m = np.eye(4)
    m[0, 0], m[1, 1], m[0, 3], m[1, 3], m[2, 3] = 2.0, 2.0, -1.0, -1.0, 1.0
    return m


class VideoMixer:
    """mix.video.swift:21-184 without the clock: `push(sample)` is the Source's set
    closure (:57-75), `mix(at)` is one tick (:95-140).  `fused=True` issues the tick as
    one chv_composite launch; `fused=False` replays the reference's clear + per-layer
    launches.  Both give the same bytes."""

    numberBackingImages = 10  # mix.video.swift:167

    def __init__(self, workspaceId, frameDuration, outputSize, outputFormat=PixelFormat.nv12,
                 computeContext=None, assetId=None, fused=True, bgraKernelFamily="reference",
                 colorspace=cv.CSC_BT601_LIMITED):
        self.clContext = createComputeContext(sharing=computeContext) if computeContext is not None \
            else makeComputeContext(forType="GPU")
        self.frameDuration = frameDuration
        self.backing, self.currentBacking = [], 0
        self.backingSize, self.backingFormat = outputSize, outputFormat
        self.idWorkspace, self.idAsset = workspaceId, assetId or uuid.uuid4().hex
        self.samples = [dict(), dict()]
        self.fused, self.bgraKernelFamily, self.colorspace = fused, bgraKernelFamily, colorspace
        self.result = None

    def assetId(self): return self.idAsset
    def workspaceId(self): return self.idWorkspace
    def computeContext(self): return self.clContext

    def push(self, pic):
        if self.clContext is None:
            return ("error", ("mix.video", -1, "No Compute Context"))
        if pic.assetId() != self.assetId():
            self.samples[0][pic.revision()] = pic
            return ("nothing", pic.info())
        return ("just", pic)

    def findKernel(self, image, target):
        """mix.video.swift:142-146: "img_" + (input format | "clear") + "_" + target format, resolved through
        defaultComputeKernelFromString.  bgraKernelFamily = "reference" (default) is the unchanged Swift VideoMixer: a
        BGRA layer on a BGRA canvas resolves to img_bgra_bgra, the Metal kernel's semantics (kernels.metal:52-62: nearest,
        no transform, no opacity); "tx" is the optional mix.video.swift hunk of INTEGRATION.md section 1: BGRA / RGBA
        layers on a BGRA canvas take the transform- and opacity-aware img_*_bgra_tx kernels."""
        inp = str(image.pixelFormat().name).lower() if image is not None else "clear"
        outp = str(target.pixelFormat().name).lower()
        name = f"img_{inp}_{outp}"
        if outp == "bgra" and image is not None and inp in ("bgra", "rgba") and self.bgraKernelFamily == "tx":
            name += "_tx"
        return defaultComputeKernelFromString(name)

    def getBacking(self):
        """mix.video.swift:148-165"""
        if self.clContext is None:
            raise ComputeError(11, "No context")
        if len(self.backing) < self.numberBackingImages:
            image = createPictureSample(self.backingSize, self.backingFormat, assetId=self.assetId(),
                                        workspaceId=self.workspaceId())
            gpu = uploadComputePicture(self.clContext, image)
            self.backing.append(gpu)
            return gpu
        image = self.backing[self.currentBacking]
        self.currentBacking = (self.currentBacking + 1) % len(self.backing)
        return image

    def mix(self, at=0.0):
        ctx = self.clContext
        try:
            backing = self.getBacking()
            merged = dict(self.samples[1])
            merged.update(self.samples[0])  # lhs wins, mix.video.swift:114
            images = sorted(merged.values(), key=lambda s: s.zIndex())
            if self.fused:
                layers = [(self.findKernel(im, backing), im, imageUniformsFor(im, backing), self.colorspace)
                          for im in images]
                beginComputePass(ctx)
                compositeTick(ctx, backing, layers, clearFirst=True)
                endComputePass(ctx, True)
            else:
                def body(c):
                    c = runComputeKernel(c, images=[], target=backing, kernel=self.findKernel(None, backing))
                    for im in images:
                        c = applyComputeImage(c, image=im, target=backing, kernel=self.findKernel(im, backing),
                                              colorspace=self.colorspace)
                    return c
                usingContext(ctx, body)
            sample = backing.derive(pts=at, time=at, assetId=self.assetId())
            self.result = ("nothing", None)
            return sample
        except ComputeError as e:
            self.result = ("error", ("mix.video", -2, f"Compute error {e}", at, self.idAsset))
            return None
        finally:
            self.samples[1] = self.samples[0]
            self.samples[0] = dict()
