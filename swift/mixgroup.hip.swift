/*
   mixgroup.hip.swift — many mixer ticks in ONE launch for the HIP backend (source only: no Swift toolchain in the build
   container; the same two classes are exercised through swiftvideo_amd/compute.py::TickBatch / VideoMixerGroup and
   swiftvideo_amd/host/swiftvideo_hip.hpp, tests/test_gpu_mixer.py, tests/cpp/).

   Why it exists.  VideoMixer.mix (mix.video.swift:95-140) issues, per mixer and per tick, a clear, one kernel per layer
   and one blocking wait (usingContext, compute.swift:131-134).  On an MI355X one 720p tick is ~7 us of device work behind
   a ~14 us launch + wait floor (tools/tick_latency.py): a composer with many mixers on one device
   (composer.swift:203-224) leaves the device idle most of the time.  With the two hunks of INTEGRATION.md section 1
   a mixer (a) issues its tick as one chv_composite launch, or (b) — when it belongs to a VideoMixerGroup — hands the
   tick to the group, which composes the ticks of all its members with ONE chv_batch launch per canvas format and ONE
   host wait.  Current figures: INTEGRATION.md section 1 and the `workloads` of the default bench.py line (round 3: 256
   ticks per launch = 4.9 us of device time per tick; one tick at a time 23-25 us fused, 54-56 us as the unchanged
   clear + 4 launches sequence: legs pipeline_per_tick / pipeline_reference_sequence).  The per-tick figure of a group does
   not include building the TickBatch: VideoMixerGroup.flush below makes a new one every tick (a device allocation and a
   descriptor copy, ~10 us); a group whose members and canvas rings recur should keep its batches, keyed by the
   (canvas, sources) tuple, as swiftvideo_amd/compute.py::LanczosBatch does for resizes.
*/
#if GPGPU_HIP
import Foundation
import Dispatch
import VectorMath
import CHIPVideo

/// One mixer tick: clear the target, then the layers in z order (mix.video.swift:116-124).
public struct MixTick {
    public let target: PictureSample
    public let clearFirst: Bool
    public let layers: [(PictureSample, ComputeKernel, ImageUniforms)]
    public init(target: PictureSample, clearFirst: Bool = true, layers: [(PictureSample, ComputeKernel, ImageUniforms)]) {
        self.target = target
        self.clearFirst = clearFirst
        self.layers = layers
    }
}

/// N independent ticks of one canvas format as ONE kernel launch (chv_batch_create / chv_batch_run).  The descriptors
/// live on the device until `deinit`; the pictures a batch refers to are retained by it.  Byte-identical to running the
/// ticks one by one through compositeTick or through clear + applyComputeImage.
public final class TickBatch {
    public let count: Int
    public private(set) var kernelName = ""
    private var handle: OpaquePointer?
    private let retained: [PictureSample]

    public init(_ context: ComputeContext, ticks: [MixTick]) throws {
        guard !ticks.isEmpty else {
            throw ComputeError.invalidValue
        }
        // chv_tick holds a pointer to its layers: keep every tick's layer array alive, at a stable address, until
        // chv_batch_create has copied the descriptors to the device
        var layerStorage = [UnsafeMutablePointer<chv_layer>]()
        defer { layerStorage.forEach { $0.deallocate() } }
        var descs = [chv_tick]()
        var keep = [PictureSample]()
        for tick in ticks {
            guard let targetImage = tick.target.imageBuffer(), let targetDesc = describeImage(targetImage, maxPlanes: 3) else {
                throw ComputeError.badTarget
            }
            let storage = UnsafeMutablePointer<chv_layer>.allocate(capacity: max(tick.layers.count, 1))
            layerStorage.append(storage)
            for (index, (sample, kernel, uniforms)) in tick.layers.enumerated() {
                storage[index] = try makeLayer(sample, kernel, uniforms)
                keep.append(sample)
            }
            var desc = chv_tick()
            desc.target = targetDesc
            desc.clear_first = tick.clearFirst ? 1 : 0
            desc.n_layers = Int32(tick.layers.count)
            desc.layers = UnsafePointer(storage)
            descs.append(desc)
            keep.append(tick.target)
        }
        var batch: OpaquePointer?
        try checkStatus(chv_batch_create(context.handle, &descs, Int32(descs.count), &batch))
        self.handle = batch
        self.count = ticks.count
        self.retained = keep
        var name = [CChar](repeating: 0, count: 128)
        if chv_batch_describe(batch, &name, 128, nil) == 0 {
            self.kernelName = String(cString: name)
        }
    }

    /// Enqueue the whole batch on the context's stream (inside a compute pass, like runComputeKernel).
    public func run(_ context: ComputeContext) throws -> ComputeContext {
        try checkStatus(chv_batch_run(context.handle, handle))
        return context
    }

    deinit {
        if let batch = handle {
            _ = chv_batch_destroy(batch)      // safe from any thread; selects the device itself
        }
    }
}

/// Several VideoMixers of one device ticked together.  Members keep their own samples, backing ring and z order
/// (mix.video.swift:57-75,148-165); what changes is WHO launches: a member that belongs to a group calls
/// `submit(...)` from its `mix(at:)` (the mix.video.swift hunk in INTEGRATION.md section 1) instead of launching, and the
/// group composes all ticks submitted for one clock time with one launch per canvas format and one host wait, then
/// runs every member's completion (which emits the member's PictureSample exactly as the unchanged code does).
public final class VideoMixerGroup {
    public init(_ context: ComputeContext, members: Int, flushAfter: DispatchTimeInterval = .milliseconds(2)) {
        self.context = createComputeContext(sharing: context)
        self.members = members
        self.flushAfter = flushAfter
        self.queue = DispatchQueue(label: "mix.video.group")
    }

    /// Called by a member mixer on its own queue.  `completion` receives nil on success or the error that made the
    /// group's launch fail (the member turns it into EventError("mix.video", -2, …) as mix.video.swift:133-137 does).
    public func submit(time: TimePoint, tick: MixTick, completion: @escaping (Error?) -> Void) {
        queue.async { [weak self] in
            guard let strongSelf = self else {
                return
            }
            strongSelf.pending[time.value, default: []].append((tick, completion))
            if strongSelf.pending[time.value]?.count == strongSelf.members {
                strongSelf.flush(time.value)
            } else if strongSelf.pending[time.value]?.count == 1 {
                // a member that misses a tick (no context, torn down) must not hold the others back
                strongSelf.queue.asyncAfter(deadline: .now() + strongSelf.flushAfter) { [weak self] in
                    self?.flush(time.value)
                }
            }
        }
    }

    private func flush(_ time: Int64) {
        guard let ticks = pending.removeValue(forKey: time), let ctx = context else {
            return
        }
        var failure: Error?
        do {
            // a batch is one kernel family: one batch per canvas format, all enqueued in one pass, one wait
            var byFormat = [String: [MixTick]]()
            for (tick, _) in ticks {
                byFormat[String(describing: tick.target.pixelFormat()), default: []].append(tick)
            }
            let batches = try byFormat.values.map { try TickBatch(ctx, ticks: $0) }
            context = try usingContext(ctx) { try batches.reduce($0) { try $1.run($0) } }
        } catch let error {
            failure = error
        }
        ticks.forEach { $0.1(failure) }
    }

    deinit {
        if let ctx = context {
            try? destroyComputeContext(ctx)
        }
    }

    private var context: ComputeContext?
    private let members: Int
    private let flushAfter: DispatchTimeInterval
    private let queue: DispatchQueue
    private var pending = [Int64: [(MixTick, (Error?) -> Void)]]()
}

// MARK: - shared with compute.hip.swift (which keeps its own private copies of describe / check for the contract functions)

/// `ImageUniforms` of a layer exactly as applyComputeImage builds them (compute.swift:149-161).
public func imageUniforms(for image: PictureSample, target: PictureSample) -> ImageUniforms {
    ImageUniforms(transform: image.matrix().inverse.transpose,
                  textureTransform: image.textureMatrix().inverse.transpose,
                  borderMatrix: image.borderMatrix().inverse.transpose,
                  fillColor: image.fillColor(),
                  inputSize: Vector2([image.size().x, image.size().y]),
                  outputSize: Vector2([target.size().x, target.size().y]),
                  opacity: image.opacity(),
                  imageTime: seconds(image.time()),
                  targetTime: seconds(target.time()))
}

private func checkStatus(_ status: Int32) throws {
    if status != 0 {
        throw computeError(fromStatus: status)          // compute.hip.swift: chv_status -> ComputeError, case by case
    }
}

private func makeLayer(_ sample: PictureSample, _ kernel: ComputeKernel, _ uniforms: ImageUniforms) throws -> chv_layer {
    guard let image = sample.imageBuffer(), let desc = describeImage(image, maxPlanes: 3) else {
        throw ComputeError.badInputData(description: "Bad input image")
    }
    var layer = chv_layer()
    layer.kernel = try kernelIdentifier(kernel)         // compute.hip.swift: ComputeKernel -> chv_kernel by name
    layer.image = desc
    var u = uniforms
    withUnsafeBytes(of: &u) { src in
        withUnsafeMutableBytes(of: &layer.uniforms) { $0.copyMemory(from: UnsafeRawBufferPointer(rebasing: src[0..<236])) }
    }
    return layer
}
#endif
