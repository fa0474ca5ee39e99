/*
   compute.hip.swift — SwiftVideo compute backend over CHIPVideo (AMD MI355X / gfx950).

   Drop-in for Sources/SwiftVideo/compute.cl.swift: it defines the same types
   (ComputeDevice, ComputeBuffer, ComputeContext) and the same free functions, so
   mix.video.swift and sample.pict.linux.swift compile unchanged when the package is built
   with the GPGPU_HIP define; compute.swift gets four enum cases and five kernelMap entries
   (INTEGRATION.md section 1 has that hunk and the Package.swift one).  All pixel work happens in libchipvideo.so; this file only marshals
   PictureSample / ImageBuffer values into chv_image descriptors and maps status codes
   to ComputeError.

   NOTE: written against the backend contract of compute.cl.swift:36-499; it cannot be
   compiled in the container this repository is developed in (no Swift toolchain), so the
   call sequences it relies on are exercised through the same C ABI by tests/ instead.
*/
#if GPGPU_HIP
import Foundation
import CHIPVideo
import VectorMath
import Logging

// MARK: - Types (compute.cl.swift:36-105)

struct ComputeDevice {
    let deviceId: Int32
    let available: Bool
    let deviceType: ComputeDeviceType?
    let vendorId: Int?
    let vendorName: String?
    let supportsImages: Bool
}

/// One device allocation (chv_buffer); freed when the last ComputeBuffer that views it goes away.
private final class DeviceAllocation {
    let handle: OpaquePointer
    init(_ handle: OpaquePointer) { self.handle = handle }
    deinit {
        // callable from any thread: the library makes the owning device current
        _ = chv_buffer_free(handle)
    }
}

/// ComputeBuffer, compute.cl.swift:46-58.  A picture's planes are views (offset, pitch) of ONE allocation, so that planes
/// which are adjacent on both sides travel as one pitched copy (INTEGRATION.md section 3).
public class ComputeBuffer {
    private let allocation: DeviceAllocation
    fileprivate var handle: OpaquePointer { return allocation.handle }
    fileprivate let offset: Int
    fileprivate let size: Int
    fileprivate let pitch: Int
    fileprivate init(_ handle: OpaquePointer, size: Int, pitch: Int = 0) {
        self.allocation = DeviceAllocation(handle)
        self.offset = 0
        self.size = size
        self.pitch = pitch
    }
    fileprivate init(viewOf other: ComputeBuffer, offset: Int, size: Int, pitch: Int) {
        self.allocation = other.allocation
        self.offset = offset
        self.size = size
        self.pitch = pitch
    }
}

public struct ComputeContext {
    fileprivate let handle: OpaquePointer
    let device: ComputeDevice
    let logger: Logger
    fileprivate init(_ handle: OpaquePointer, device: ComputeDevice, logger: Logger) {
        self.handle = handle
        self.device = device
        self.logger = logger
    }
}

// MARK: - Error mapping (the role of checkCLError, compute.cl.swift:683-702)

private func check(_ status: Int32, kernel: ComputeKernel? = nil) throws {
    guard status != 0 else { return }
    let detail = String(cString: chv_last_error_detail())
    switch chv_status(UInt32(status)) {
    case CHV_ERR_INVALID_VALUE: throw ComputeError.invalidValue
    case CHV_ERR_OUT_OF_MEMORY: throw ComputeError.outOfMemory
    case CHV_ERR_INVALID_CONTEXT: throw ComputeError.invalidContext
    case CHV_ERR_BAD_TARGET: throw ComputeError.badTarget
    case CHV_ERR_BAD_INPUT: throw ComputeError.badInputData(description: detail)
    case CHV_ERR_NOT_IMPLEMENTED: throw ComputeError.notImplemented
    case CHV_ERR_KERNEL_NOT_FOUND: throw ComputeError.computeKernelNotFound(kernel ?? .custom(name: detail))
    case CHV_ERR_DEVICE_NOT_AVAILABLE: throw ComputeError.deviceNotAvailable
    case CHV_ERR_INVALID_DEVICE: throw ComputeError.invalidDevice
    case CHV_ERR_INVALID_OPERATION: throw ComputeError.invalidOperation
    case CHV_ERR_BAD_CONTEXT_STATE: throw ComputeError.badContextState(description: detail)
    case CHV_ERR_INVALID_PLATFORM: throw ComputeError.invalidPlatform
    default: throw ComputeError.unknownError
    }
}

/// chv_status -> ComputeError for the other HIP-only source files (mixgroup.hip.swift); nil for success.
func computeError(fromStatus status: Int32) -> Error {
    do { try check(status) } catch let error { return error }
    return ComputeError.unknownError
}

/// ComputeKernel -> chv_kernel.  Every case's name is its own description ("img_nv12_bgra", ...), and the library's name
/// table (chv_kernel_from_string) is defaultComputeKernelFromString's table plus the cases the compute.swift hunk adds, so
/// one lookup serves the reference's thirteen cases and the four new ones.  A `.custom(name:)` whose name the table knows
/// is accepted as well (hosts that cannot patch compute.swift can spell `.custom(name: "img_nv12_bgra")`).
private func kernelId(_ kernel: ComputeKernel) throws -> Int32 {
    var id: Int32 = -1
    if case .custom(let name) = kernel {
        try check(chv_kernel_from_string(name, &id), kernel: kernel)
        return id
    }
    try check(chv_kernel_from_string(String(describing: kernel), &id), kernel: kernel)
    return id
}

func kernelIdentifier(_ kernel: ComputeKernel) throws -> Int32 { try kernelId(kernel) }      // (for mixgroup.hip.swift)

// MARK: - Devices and contexts (compute.cl.swift:107-151)

func availableComputeDevices() -> [ComputeDevice] {
    var count: Int32 = 0
    guard chv_device_count(&count) == 0 else { return [] }
    return (0..<count).compactMap { idx in
        var info = chv_device_info()
        guard chv_device_info_get(idx, &info) == 0 else { return nil }
        let name = withUnsafePointer(to: &info.name) {
            $0.withMemoryRebound(to: CChar.self, capacity: 128) { String(cString: $0) }
        }
        return ComputeDevice(deviceId: idx, available: info.available != 0, deviceType: .GPU,
                             vendorId: Int(info.vendor_id), vendorName: name,
                             supportsImages: info.supports_images != 0)
    }
}

func createComputeContext(sharing ctx: ComputeContext) -> ComputeContext? {
    var out: OpaquePointer?
    guard chv_context_share(ctx.handle, &out) == 0, let handle = out else { return nil }
    return ComputeContext(handle, device: ctx.device, logger: ctx.logger)
}

func createComputeContext(_ device: ComputeDevice,
                          logger: Logger = Logger(label: "SwiftVideo")) throws -> ComputeContext? {
    var out: OpaquePointer?
    try check(chv_context_create(device.deviceId, &out))
    // all kernels are part of the library: nothing to build here (the OpenCL backend compiles
    // every OpenCLKernel case at this point, compute.cl.swift:139-144)
    return out.map { ComputeContext($0, device: device, logger: logger) }
}

func destroyComputeContext( _ context: ComputeContext) throws {
    try check(chv_context_destroy(context.handle))
}

/// compute.cl.swift:153-195.  `source` is HIP C++ here (compiled with hipRTC for the context's device, prefixed
/// with chv_custom_prelude()) and must define `extern "C" __global__ void <name>(chv_custom_args a)`.
/// The library lives in the native context; contexts created with createComputeContext(sharing:) afterwards
/// inherit it, as `ComputeContext(other:)` copies `library` in the OpenCL backend (:82-87).
func buildComputeKernel(_ context: ComputeContext, name: String, source: String) throws -> ComputeContext {
    context.logger.info("buildComputeKernel")
    let status = chv_kernel_build(context.handle, name, source)
    if status != 0, let detail = chv_last_error_detail() {
        context.logger.info("Build log:\n\(String(cString: detail))")
    }
    try check(status)
    return context
}

/// name of a user kernel: a `.custom` whose name the built-in table does not know
private func userKernelName(_ kernel: ComputeKernel) -> String? {
    guard case .custom(let name) = kernel else { return nil }
    var id: Int32 = -1
    return chv_kernel_from_string(name, &id) == 0 ? nil : name
}

// MARK: - Passes and kernels (compute.cl.swift:234-359)

func beginComputePass(_ context: ComputeContext) -> ComputeContext {
    _ = chv_pass_begin(context.handle)
    return context
}

// The library HOLDS the picture kernels issued since beginComputePass and launches them here as one fused tick (include/chipvideo.h,
// "compute passes"): a launch error of those kernels surfaces in this call.  The reference's endComputePass cannot throw and drops the result
// of clFinish / clFlush (compute.cl.swift:346-359), where OpenCL reports asynchronous launch errors too; the signature stays, the error is logged.
func endComputePass(_ context: ComputeContext, _ waitForCompletion: Bool) -> ComputeContext {
    let status = chv_pass_end(context.handle, waitForCompletion ? 1 : 0)
    if status != 0 {
        context.logger.error("endComputePass: \(String(cString: chv_error_string(status))) (\(String(cString: chv_last_error_detail())))")
    }
    return context
}

private func describe(_ image: ImageBuffer, maxPlanes: Int) -> chv_image? {
    guard image.bufferType == .gpu, image.computeTextures.count > 0 else { return nil }
    var desc = chv_image()
    desc.format = Int32(pixelFormatCode(image.pixelFormat))
    desc.width = Int32(image.size.x)
    desc.height = Int32(image.size.y)
    let count = min(image.computeTextures.count, maxPlanes)
    desc.n_planes = Int32(count)
    withUnsafeMutablePointer(to: &desc.planes) {
        $0.withMemoryRebound(to: chv_plane.self, capacity: 3) { planes in
            for idx in 0..<count {
                let plane = image.planes[idx]
                let comps = plane.components.count >= 3 ? 4 : plane.components.count
                planes[idx] = chv_plane(buffer: image.computeTextures[idx].handle, offset: image.computeTextures[idx].offset,
                                        width: Int32(plane.size.x), height: Int32(plane.size.y),
                                        pitch: Int32(image.computeTextures[idx].pitch),
                                        components: Int32(comps))
            }
        }
    }
    return desc
}

func describeImage(_ image: ImageBuffer, maxPlanes: Int) -> chv_image? { describe(image, maxPlanes: maxPlanes) }      // (for mixgroup.hip.swift)

private func pixelFormatCode(_ fmt: PixelFormat) -> Int {
    switch fmt {
    case .nv12: return Int(CHV_FMT_NV12.rawValue)
    case .nv21: return Int(CHV_FMT_NV21.rawValue)
    case .yuvs: return Int(CHV_FMT_YUVS.rawValue)
    case .zvuy: return Int(CHV_FMT_ZVUY.rawValue)
    case .y420p: return Int(CHV_FMT_Y420P.rawValue)
    case .y422p: return Int(CHV_FMT_Y422P.rawValue)
    case .y444p: return Int(CHV_FMT_Y444P.rawValue)
    case .RGBA: return Int(CHV_FMT_RGBA.rawValue)
    case .BGRA: return Int(CHV_FMT_BGRA.rawValue)
    default: return Int(CHV_FMT_INVALID.rawValue)
    }
}

func runComputeKernel(_ context: ComputeContext,
                      images: [PictureSample],
                      target: PictureSample,
                      kernel: ComputeKernel,
                      maxPlanes: Int = 3,
                      requiredMemory: Int? = nil) throws -> ComputeContext {
    return try runComputeKernel(context, images: images, target: target, kernel: kernel,
                                maxPlanes: maxPlanes, uniforms: Void?.none)
}

func runComputeKernel<T>(_ context: ComputeContext,
                         images: [PictureSample],
                         target: PictureSample,
                         kernel: ComputeKernel,
                         maxPlanes: Int = 3,
                         requiredMemory: Int? = nil,
                         uniforms: T? = nil,
                         blends: Bool = false) throws -> ComputeContext {
    guard let targetImage = target.imageBuffer(), var targetDesc = describe(targetImage, maxPlanes: 3) else {
        throw ComputeError.badTarget
    }
    var inputs = try images.map { sample -> chv_image in
        // a CPU sample is uploaded on the fly, as createTexture does (compute.cl.swift:280-286)
        let gpu = try uploadComputePicture(context, pict: sample, maxPlanes: maxPlanes)
        guard let image = gpu.imageBuffer(), let desc = describe(image, maxPlanes: maxPlanes) else {
            throw ComputeError.badInputData(description: "Bad input image")
        }
        return desc
    }
    let status: Int32
    if let user = userKernelName(kernel) {
        // getComputeKernel's `.custom` branch (compute.cl.swift:218-232): a kernel from the context's library
        if var uniforms = uniforms {
            status = withUnsafeBytes(of: &uniforms) { raw in
                chv_run_custom(context.handle, user, &targetDesc, &inputs, Int32(inputs.count),
                               raw.baseAddress, MemoryLayout<T>.size, blends ? 1 : 0)
            }
        } else {
            status = chv_run_custom(context.handle, user, &targetDesc, &inputs, Int32(inputs.count), nil, 0, blends ? 1 : 0)
        }
        try check(status, kernel: kernel)
        return context
    }
    let id = try kernelId(kernel)
    if var uniforms = uniforms {
        status = withUnsafeBytes(of: &uniforms) { raw in
            chv_run_kernel(context.handle, id, &targetDesc, &inputs, Int32(inputs.count),
                           raw.baseAddress, MemoryLayout<T>.size, blends ? 1 : 0, nil)
        }
    } else {
        status = chv_run_kernel(context.handle, id, &targetDesc, &inputs, Int32(inputs.count),
                                nil, 0, blends ? 1 : 0, nil)
    }
    try check(status, kernel: kernel)
    return context
}

// MARK: - The two buffer kernels of `ComputeKernel` (compute.swift:67,70; kernels.cl.swift:534-562, kernels.metal:129-267)
//
// Nothing in the reference dispatches `.snd_s16i_s16i` / `.me_fullsearch`; CHIPVideo runs them through chv_run_kernel in the reference's bind
// order [outputs][inputs][uniforms].  A ComputeBuffer is bound where the kernel expects an image: one plane of `width` x `height` texels of
// `components` bytes (interleaved-stereo Int16 samples = 2-byte texels; a luma plane = 1; me_fullsearch's output = one RGBA8 texel per block).
struct BufferImage {
    let buffer: ComputeBuffer
    let width: Int
    var height: Int = 1
    var components: Int = 2
    var offset: Int = 0
    func describe() -> chv_image {
        var d = chv_image()
        d.format = Int32(CHV_FMT_INVALID.rawValue)
        d.width = Int32(width); d.height = Int32(height); d.n_planes = 1
        d.planes.0 = chv_plane(buffer: buffer.handle, offset: offset, width: Int32(width), height: Int32(height),
                               pitch: Int32(width * components), components: Int32(components))
        return d
    }
}
typealias BufferUniforms = chv_snd_uniforms                // kernels.cl.swift:536-541 (100 bytes)
typealias MotionEstimationUniforms = chv_me_uniforms       // kernels.metal:33-37 (24 bytes)

func runComputeKernel<T>(_ context: ComputeContext,
                         buffers: [BufferImage],
                         target: BufferImage,
                         kernel: ComputeKernel,
                         uniforms: T) throws -> ComputeContext {
    var targetDesc = target.describe()
    var inputs = buffers.map { $0.describe() }
    var u = uniforms
    let id = try kernelId(kernel)
    let status = withUnsafeBytes(of: &u) { raw in
        chv_run_kernel(context.handle, id, &targetDesc, &inputs, Int32(inputs.count), raw.baseAddress, MemoryLayout<T>.size, 0, nil)
    }
    try check(status, kernel: kernel)
    return context
}

// MARK: - Transfers (compute.cl.swift:361-498)

func uploadComputeBuffer(_ ctx: ComputeContext, src: Data, dst: ComputeBuffer?) throws -> ComputeBuffer {
    let buffer: ComputeBuffer
    if let dst = dst {
        buffer = dst
    } else {
        var out: OpaquePointer?
        try check(chv_buffer_alloc(ctx.handle, src.count, &out))
        buffer = ComputeBuffer(out!, size: src.count)
    }
    guard buffer.size >= src.count else {
        throw ComputeError.badInputData(description: "Compute buffer needs to be >= to data.count")
    }
    try src.withUnsafeBytes {
        try check(chv_upload(ctx.handle, buffer.handle, buffer.offset, src.count, $0.baseAddress, src.count, src.count, 1, 0))
    }
    return buffer
}

func downloadComputeBuffer(_ ctx: ComputeContext, src: ComputeBuffer, dst: Data?) throws -> Data {
    var dst = dst ?? Data(count: src.size)
    guard dst.count >= src.size else {
        throw ComputeError.badInputData(description: "Destination data buffer must be >= buffer.size")
    }
    try dst.withUnsafeMutableBytes {
        try check(chv_download(ctx.handle, $0.baseAddress, src.size, src.handle, src.offset, src.size, src.size, 1))
    }
    return dst
}

func uploadComputePicture(_ ctx: ComputeContext,
                          pict: PictureSample,
                          maxPlanes: Int = 3,
                          retainCpuBuffer: Bool = true) throws -> PictureSample {
    guard pict.bufferType() == .cpu else { return pict }
    guard let imageBuffer = pict.imageBuffer() else {
        throw ComputeError.badInputData(description: "Missing image buffer")
    }
    let planeCount = imageBuffer.planes.count
    guard 3 >= planeCount && 0 < planeCount else {
        throw ComputeError.badInputData(description: "Input image must have 1, 2, or 3 planes")
    }
    guard planeCount == imageBuffer.buffers.count else {
        throw ComputeError.badInputData(description: "Input image must have the same number of buffers as planes")
    }
    // One allocation for the whole picture: 128-byte aligned pitches, each plane starting where the previous one ends
    // (the role of createTexture, compute.cl.swift:532-581, which makes one image per plane).
    let count = min(planeCount, maxPlanes)
    var pitches = [Int](), offsets = [Int](), total = 0
    for idx in 0..<count {
        let plane = imageBuffer.planes[idx]
        guard plane.size.x > 0 && plane.size.y > 0 else { throw ComputeError.invalidOperation }
        let comps = plane.components.count >= 3 ? 4 : plane.components.count
        let pitch = (Int(plane.size.x) * comps + 127) / 128 * 128
        pitches.append(pitch)
        offsets.append(total)
        total += pitch * Int(plane.size.y)
    }
    var out: OpaquePointer?
    try check(chv_buffer_alloc(ctx.handle, total, &out))
    let whole = ComputeBuffer(out!, size: total)
    let textures = (0..<count).map { idx in
        ComputeBuffer(viewOf: whole, offset: offsets[idx], size: pitches[idx] * Int(imageBuffer.planes[idx].size.y), pitch: pitches[idx])
    }
    // Planes that are adjacent with equal pitch and width on BOTH sides go as one pitched copy: luma + interleaved chroma
    // of NV12, the two chroma planes of y420p (buffersForPlanes slices one contiguous Data, sample.pict.linux.swift:296-311).
    // 3 MiB copies reach 48 GB/s on this link, separate 2 MiB + 1 MiB copies 37 GB/s.
    var idx = 0
    while idx < count {
        let plane = imageBuffer.planes[idx]
        let comps = plane.components.count >= 3 ? 4 : plane.components.count
        let widthBytes = Int(plane.size.x) * comps
        var rows = Int(plane.size.y)
        var last = idx
        try imageBuffer.buffers[idx].withUnsafeBytes { first in
            while last + 1 < count {
                let next = imageBuffer.planes[last + 1]
                let nextComps = next.components.count >= 3 ? 4 : next.components.count
                let contiguous = imageBuffer.buffers[last + 1].withUnsafeBytes {
                    $0.baseAddress == first.baseAddress.map { $0 + rows * plane.stride }
                }
                guard pitches[last + 1] == pitches[idx], next.stride == plane.stride,
                      Int(next.size.x) * nextComps == widthBytes, contiguous else { break }
                rows += Int(next.size.y)
                last += 1
            }
            // async = 1: the bytes are staged into pinned memory before the call returns; the copy is ordered on the
            // context's stream, and kernels or downloads of ANY context of the device wait for it (per-buffer event)
            try check(chv_upload(ctx.handle, whole.handle, offsets[idx], pitches[idx], first.baseAddress, plane.stride,
                                 widthBytes, rows, 1))
        }
        idx = last + 1
    }
    let image = ImageBuffer(imageBuffer, computeTextures: textures,
                            buffers: !retainCpuBuffer ? [] : nil, bufferType: .gpu)
    return PictureSample(pict, img: image)
}

func downloadComputePicture(_ ctx: ComputeContext,
                            pict: PictureSample,
                            retainGpuBuffer: Bool = false) throws -> PictureSample {
    guard pict.bufferType() == .gpu else { return pict }
    guard let imageBuffer = pict.imageBuffer() else {
        throw ComputeError.badInputData(description: "Missing image buffer")
    }
    let buffers = try (0..<imageBuffer.computeTextures.count).map { idx -> Data in
        let plane = imageBuffer.planes[idx]
        let comps = plane.components.count >= 3 ? 4 : plane.components.count
        var buffer = imageBuffer.buffers[safe: idx] ?? Data(count: Int(plane.size.y) * plane.stride)
        let texture = imageBuffer.computeTextures[idx]
        try buffer.withUnsafeMutableBytes {
            try check(chv_download(ctx.handle, $0.baseAddress, plane.stride, texture.handle, texture.offset, texture.pitch,
                                   Int(plane.size.x) * comps, Int(plane.size.y)))
        }
        return buffer
    }
    let image = ImageBuffer(imageBuffer, computeTextures: !retainGpuBuffer ? [] : nil,
                            buffers: buffers, bufferType: .cpu)
    return PictureSample(pict, img: image)
}

/// The D2H half of a download barrier that runs on a context of its own (GPUBarrierDownload, compute.swift:217-255) without the host
/// wait: the planes of `pict` are copied into `pinned` (chv_host_alloc; plane after plane, rows packed) on `ctx`'s stream and the call
/// returns.  The bytes are the picture's once that stream has passed the copy (`endComputePass(ctx, true)` on it, or an event).  Work of
/// the mixer's context that writes `pict` is ordered in front of the copy by the caller: chv_event_record there, chv_event_wait here —
/// the read-back of tick t then overlaps the kernels of tick t + 1 (bench.py leg pipeline_e2e).
func downloadComputePictureAsync(_ ctx: ComputeContext, pict: PictureSample, pinned: UnsafeMutableRawPointer) throws {
    guard pict.bufferType() == .gpu, let imageBuffer = pict.imageBuffer() else {
        throw ComputeError.badInputData(description: "Missing device image")
    }
    var offset = 0
    for idx in 0..<imageBuffer.computeTextures.count {
        let plane = imageBuffer.planes[idx]
        let comps = plane.components.count >= 3 ? 4 : plane.components.count
        let rowBytes = Int(plane.size.x) * comps
        let texture = imageBuffer.computeTextures[idx]
        try check(chv_download_async(ctx.handle, pinned + offset, rowBytes, texture.handle, texture.offset, texture.pitch,
                                     rowBytes, Int(plane.size.y)))
        offset += rowBytes * Int(plane.size.y)
    }
}

// MARK: - One mixer tick in one launch (optional fast path for VideoMixer.mix)

/// Equivalent of `usingContext { clear; images.reduce { applyComputeImage } }` (mix.video.swift:116-124)
/// issued as a single chv_composite launch; byte-identical output.
func compositeTick(_ context: ComputeContext,
                   layers: [(PictureSample, ComputeKernel, ImageUniforms)],
                   target: PictureSample) throws -> ComputeContext {
    guard let targetImage = target.imageBuffer(), var targetDesc = describe(targetImage, maxPlanes: 3) else {
        throw ComputeError.badTarget
    }
    var descs = try layers.map { (sample, kernel, uniforms) -> chv_layer in
        guard let image = sample.imageBuffer(), let desc = describe(image, maxPlanes: 3) else {
            throw ComputeError.badInputData(description: "Bad input image")
        }
        var layer = chv_layer()
        layer.kernel = try kernelId(kernel)
        layer.image = desc
        var u = uniforms
        withUnsafeBytes(of: &u) { src in
            withUnsafeMutableBytes(of: &layer.uniforms) { $0.copyMemory(from: UnsafeRawBufferPointer(rebasing: src[0..<236])) }
        }
        return layer
    }
    try check(chv_composite(context.handle, &targetDesc, 1, &descs, Int32(descs.count)))
    return context
}

// MARK: - Lanczos-3 resample, BGRA -> BGRA (no reference counterpart; used by filter.pict.hip.swift)

func scaleLanczos(_ context: ComputeContext, src: PictureSample, target: PictureSample) throws -> ComputeContext {
    guard let targetImage = target.imageBuffer(), var targetDesc = describe(targetImage, maxPlanes: 1) else {
        throw ComputeError.badTarget
    }
    guard let image = src.imageBuffer(), var desc = describe(image, maxPlanes: 1) else {
        throw ComputeError.badInputData(description: "Bad input image")
    }
    try check(chv_scale_lanczos(context.handle, &targetDesc, &desc))
    return context
}

/// n resizes of one geometry in one launch per 64 pairs (chv_scale_lanczos_batch): several PictureFilters / streams per tick
func scaleLanczos(_ context: ComputeContext, pairs: [(src: PictureSample, target: PictureSample)]) throws -> ComputeContext {
    var targets = [chv_image](), sources = [chv_image]()
    for pair in pairs {
        guard let targetImage = pair.target.imageBuffer(), let targetDesc = describe(targetImage, maxPlanes: 1) else {
            throw ComputeError.badTarget
        }
        guard let image = pair.src.imageBuffer(), let desc = describe(image, maxPlanes: 1) else {
            throw ComputeError.badInputData(description: "Bad input image")
        }
        targets.append(targetDesc)
        sources.append(desc)
    }
    try check(chv_scale_lanczos_batch(context.handle, &targets, &sources, Int32(pairs.count)))
    return context
}
#endif
