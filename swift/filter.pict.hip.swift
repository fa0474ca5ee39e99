/*
   filter.pict.hip.swift — PictureFilter for the HIP backend (source only: no Swift toolchain in the
   build container; the same operator is exercised through swiftvideo_amd/compute.py::PictureFilter and
   swiftvideo_amd/host/swiftvideo_hip.hpp::PictureFilter, tests/test_gpu_mixer.py, tests/cpp/).

   The reference declares this operator and leaves it commented out
   (Sources/SwiftVideo/filter.pict.swift:20-47): a Tx<PictureSample, PictureSample> with a compute
   context of its own.  This file gives it a body: convert a picture to `outputFormat` at `outputSize`
   on the device — one full-canvas layer through the composite kernels (colour conversion + bilinear
   scale in one launch) or, for BGRA -> BGRA, a separable Lanczos-3 resample (chv_scale_lanczos).
   It replaces filter.pict.swift when GPGPU_HIP is defined.
*/
#if GPGPU_HIP
import Foundation
import VectorMath
import CHIPVideo

public enum PictureScaler {
    case bilinear
    case lanczos
}

public class PictureFilter: Tx<PictureSample, PictureSample> {
    public init(_ clock: Clock,
                outputSize: Vector2,
                outputFormat: PixelFormat = .BGRA,
                scaler: PictureScaler = .bilinear,
                integerMatrix: Bool = true,
                computeContext: ComputeContext? = nil) {
        self.clock = clock
        self.outputSize = outputSize
        self.outputFormat = outputFormat
        self.scaler = scaler
        self.integerMatrix = integerMatrix
        do {
            if let context = computeContext {
                self.context = createComputeContext(sharing: context)
            } else {
                self.context = try makeComputeContext(forType: .GPU)
            }
        } catch {
            self.context = nil
        }
        super.init()
        super.set { [weak self] sample in
            guard let strongSelf = self else {
                return .gone
            }
            guard let ctx = strongSelf.context else {
                return .error(EventError("filter.pict", -1, "No Compute Context", assetId: sample.assetId()))
            }
            do {
                // CPU samples are uploaded first (what GPUBarrierUpload would do, compute.swift:175-198)
                let src = sample.bufferType() == .cpu ? try uploadComputePicture(ctx, pict: sample) : sample
                let dst = try strongSelf.getBacking(ctx, like: sample)
                strongSelf.context = try usingContext(ctx) {
                    switch strongSelf.scaler {
                    case .lanczos:
                        return try scaleLanczos($0, src: src, target: dst)
                    case .bilinear:
                        // the unit quad stretched over the whole canvas (what PictureAnimator produces for a
                        // picture at the origin with the canvas' size), no border, no fill, opaque
                        let quad = Matrix4(outputSize) * Matrix4(scale: Vector3(outputSize.x, outputSize.y, 1))
                        let full = PictureSample(src, matrix: quad, textureMatrix: Matrix4.identity,
                                                 borderMatrix: quad, fillColor: Vector4(0, 0, 0, 0), opacity: 1.0)
                        let cleared = try runComputeKernel($0, images: [PictureSample](), target: dst,
                                                           kernel: strongSelf.findKernel(nil))
                        return try applyComputeImage(cleared, image: full, target: dst,
                                                     kernel: strongSelf.findKernel(src))
                    }
                }
                // time stamps, ids and transform state of the incoming sample carry over; only the image changes
                return .just(PictureSample(sample, img: dst.imageBuffer()))
            } catch let error {
                return .error(EventError("filter.pict", -2, "Compute error \(error)", assetId: sample.assetId()))
            }
        }
    }

    // same naming rule as VideoMixer.findKernel (mix.video.swift:142-146); BGRA targets take the
    // transform-aware kernels of the HIP backend
    private func findKernel(_ image: PictureSample?) throws -> ComputeKernel {
        let inp = image <??> { String(describing: $0.pixelFormat()).lowercased() } <|> "clear"
        let outp = String(describing: outputFormat).lowercased()
        let rgbIn = image != nil && (inp == "bgra" || inp == "rgba")
        // BGRA targets: the transform-aware kernels; 4:2:0 targets from RGB pictures: the integer BT.601/709 matrix
        // (img_*_int, what an encoder expects; the default, as in swiftvideo_amd/compute.py::PictureFilter and
        // host/swiftvideo_hip.hpp) — or, with integerMatrix: false, the reference's own float full-range rows
        // (kernels.cl.swift:96-99), byte for byte what its OpenCL family writes
        let to420 = outp == "nv12" || outp == "y420p"
        let suffix = (rgbIn && outp == "bgra") ? "_tx" : (rgbIn && to420 && integerMatrix) ? "_int" : ""
        return try defaultComputeKernelFromString("img_\(inp)_\(outp)\(suffix)")
    }

    // ring of device images, as VideoMixer.getBacking (mix.video.swift:148-167)
    private func getBacking(_ ctx: ComputeContext, like: PictureSample) throws -> PictureSample {
        if backing.count < numberBackingImages {
            let image = try createPictureSample(outputSize, outputFormat,
                                                assetId: like.assetId(), workspaceId: like.workspaceId())
            let gpuImage = try uploadComputePicture(ctx, pict: image)
            backing.append(gpuImage)
            return gpuImage
        }
        let image = backing[currentBacking]
        currentBacking = (currentBacking + 1) % backing.count
        return image
    }

    private let numberBackingImages = 10
    private var backing = [PictureSample]()
    private var currentBacking = 0
    let clock: Clock
    let outputSize: Vector2
    let outputFormat: PixelFormat
    let scaler: PictureScaler
    let integerMatrix: Bool
    var context: ComputeContext?
}
#endif
