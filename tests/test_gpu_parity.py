"""HIP path (through the C ABI) == oracle, bit for bit, on identical seeded inputs."""
import zlib

import numpy as np
import pytest

import gpuutil as G
import scenarios as S
import util
from oracle import oracle as O
from swiftvideo_amd import compute as sv

pytestmark = pytest.mark.gpu

ALL_LAYER = S.LAYER_KERNELS_REF + S.LAYER_KERNELS_OWN + S.LAYER_KERNELS_INT


@pytest.mark.parametrize("scenario", list(S.SCENARIOS))
@pytest.mark.parametrize("kernel", ALL_LAYER)
def test_layer_kernel_matches_oracle(ctx, kernel, scenario):
    cw, ch, iw, ih, _ = S.SCENARIOS[scenario]
    u = S.uniforms_for(scenario)
    seed = (zlib.crc32(f"{kernel}/{scenario}".encode()) & 0xFFFF) + 1
    got, exp = G.run_both(ctx, kernel, cw, ch, iw, ih, u, seed)
    G.assert_same(got, exp, f"{kernel}/{scenario}")


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
@pytest.mark.parametrize("kernel", ["img_nv12_bgra", "img_y420p_bgra"])
def test_yuv_to_bgra_colorspaces(ctx, kernel, csc):
    u = S.uniforms_for("downscale")
    got, exp = G.run_both(ctx, kernel, 64, 36, 96, 54, u, seed=77 + csc, csc=csc, clear_first=True)
    G.assert_same(got, exp, f"{kernel}/csc{csc}")


@pytest.mark.parametrize("csc", [0, 1, 2, 3])
@pytest.mark.parametrize("kernel", S.LAYER_KERNELS_INT)
def test_rgb_to_yuv_int_colorspaces(ctx, kernel, csc):
    for scenario in ("downscale", "rect_fill"):
        cw, ch, iw, ih, _ = S.SCENARIOS[scenario]
        got, exp = G.run_both(ctx, kernel, cw, ch, iw, ih, S.uniforms_for(scenario), seed=91 + csc, csc=csc)
        G.assert_same(got, exp, f"{kernel}/{scenario}/csc{csc}")


@pytest.mark.parametrize("fmt", ["nv12", "y420p", "bgra"])
@pytest.mark.parametrize("size", [(64, 36), (7, 5), (1, 1), (130, 3), (1030, 9), (2050, 2)])
def test_clear_kernels(ctx, fmt, size):
    w, h = size
    if fmt != "bgra" and min(w, h) < 2:
        # a 4:2:0 canvas needs at least one chroma sample: size/2 == 0 cannot be allocated
        # (the reference's clCreateImage fails the same way, compute.cl.swift:570-579)
        with pytest.raises(sv.ComputeError):
            G.to_gpu(ctx, fmt, w, h, util.alloc_image(fmt, w, h, seed=5))
        return
    canvas0 = util.alloc_image(fmt, w, h, seed=5)
    exp = util.copy_image(canvas0)
    assert O.run_kernel(f"img_clear_{fmt}", exp) == 0
    g = G.to_gpu(ctx, fmt, w, h, canvas0)
    k = sv.defaultComputeKernelFromString(f"img_clear_{fmt}")
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, images=[], target=g, kernel=k))
    G.assert_same(G.from_gpu(ctx, g, fmt, w, h), exp, f"clear {fmt} {size}")


@pytest.mark.parametrize("insize", [(64, 36), (96, 54), (20, 11)])
def test_metal_bgra_bgra(ctx, insize):
    iw, ih = insize
    u = util.full_canvas_uniforms((64, 36), (iw, ih))
    got, exp = G.run_both(ctx, "img_bgra_bgra", 64, 36, iw, ih, u, seed=31)
    G.assert_same(got, exp, "img_bgra_bgra (Metal semantics)")


def _tick_layers(dst_fmt, cw, ch):
    """Four layers of mixed formats/geometry for a fused-vs-sequential check."""
    if dst_fmt == "bgra":
        kernels = ["img_nv12_bgra", "img_bgra_bgra_tx", "img_y420p_bgra", "img_rgba_bgra_tx"]
    else:
        kernels = [f"img_nv12_{dst_fmt}" if dst_fmt == "nv12" else "img_y420p_y420p",
                   f"img_bgra_{dst_fmt}", f"img_y420p_{dst_fmt}", f"img_rgba_{dst_fmt}"]
    geos = [dict(), dict(rect=(5, 3, 40, 25), opacity=0.7, border=(2, 2, 2, 2), fill=(0.3, 0.1, 0.8, 0.9)),
            dict(rect=(30, 10, 30, 20), rotation=0.3, opacity=0.5),
            dict(rect=(-4, 20, 50, 14), opacity=0.9, fill=(0.5, 0.5, 0.5, 0.4))]
    sizes = [(96, 54), (40, 30), (32, 18), (50, 14)]
    out = []
    for i, (k, g, (iw, ih)) in enumerate(zip(kernels, geos, sizes)):
        s, _ = G.kernel_formats(k)
        out.append((k, s, iw, ih, util.make_uniforms((cw, ch), in_size=(iw, ih), **g), 400 + i))
    return out


@pytest.mark.parametrize("dst_fmt", ["nv12", "y420p", "bgra"])
def test_fused_tick_equals_sequential_and_oracle(ctx, dst_fmt):
    cw, ch = 64, 36
    layers = _tick_layers(dst_fmt, cw, ch)
    # oracle: clear + one kernel per layer (mix.video.swift:116-124)
    exp = util.alloc_image(dst_fmt, cw, ch, seed=9)
    assert O.run_kernel(f"img_clear_{dst_fmt}", exp) == 0
    srcs = []
    for k, s, iw, ih, u, seed in layers:
        src = util.alloc_image(s, iw, ih, seed=seed)
        srcs.append(src)
        assert O.run_kernel(k, exp, src, u) == 0
    gl = [(sv.defaultComputeKernelFromString(k), G.to_gpu(ctx, s, iw, ih, src), u, 0)
          for (k, s, iw, ih, u, _), src in zip(layers, srcs)]
    # fused: one launch
    fused = G.to_gpu(ctx, dst_fmt, cw, ch, util.alloc_image(dst_fmt, cw, ch, seed=9))
    sv.usingContext(ctx, lambda c: sv.compositeTick(c, fused, gl, True))
    G.assert_same(G.from_gpu(ctx, fused, dst_fmt, cw, ch), exp, f"fused {dst_fmt}")
    # sequential: the reference's call sequence
    seq = G.to_gpu(ctx, dst_fmt, cw, ch, util.alloc_image(dst_fmt, cw, ch, seed=9))

    def body(c):
        c = sv.runComputeKernel(c, images=[], target=seq, kernel=sv.defaultComputeKernelFromString(f"img_clear_{dst_fmt}"))
        for k, im, u, csc in gl:
            c = sv.runComputeKernel(c, images=[im], target=seq, kernel=k, uniforms=u, blends=True)
        return c
    sv.usingContext(ctx, body)
    G.assert_same(G.from_gpu(ctx, seq, dst_fmt, cw, ch), exp, f"sequential {dst_fmt}")


def test_pitched_planes_and_padding_untouched(ctx):
    """Device planes are 256-byte pitched; host planes may carry their own stride."""
    cw, ch, iw, ih = 50, 22, 36, 20
    src = util.alloc_image("nv12", iw, ih, seed=3, pad=12)
    canvas0 = util.alloc_image("nv12", cw, ch, seed=4, pad=6)
    u = util.make_uniforms((cw, ch), rect=(3, 2, 40, 18), opacity=0.8, in_size=(iw, ih))
    exp = util.copy_image(canvas0)
    assert O.run_kernel("img_nv12_nv12", exp, src, u) == 0
    gs, gd = G.to_gpu(ctx, "nv12", iw, ih, src), G.to_gpu(ctx, "nv12", cw, ch, canvas0)
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, images=[gs], target=gd,
                                                       kernel=sv.ComputeKernel.img_nv12_nv12, uniforms=u, blends=True))
    G.assert_same(G.from_gpu(ctx, gd, "nv12", cw, ch), exp, "pitched nv12")


def test_error_behaviour(ctx):
    """Error codes follow ComputeError (compute.swift:22-39); nothing aborts."""
    bgra = G.to_gpu(ctx, "bgra", 16, 8, util.alloc_image("bgra", 16, 8, seed=1))
    nv12 = G.to_gpu(ctx, "nv12", 16, 8, util.alloc_image("nv12", 16, 8, seed=2))
    u = util.full_canvas_uniforms((16, 8), (16, 8))
    # unknown name -> invalidValue (compute.swift:106-108)
    with pytest.raises(sv.ComputeError) as e:
        sv.defaultComputeKernelFromString("img_nv21_nv12")
    assert e.value.case == "invalidValue"
    # enum case without a kernel -> computeKernelNotFound
    with pytest.raises(sv.ComputeError) as e:
        sv.runComputeKernel(ctx, images=[], target=nv12, kernel=sv.ComputeKernel.img_clear_yuvs)
    assert e.value.case == "computeKernelNotFound"
    # wrong target plane structure -> badTarget
    with pytest.raises(sv.ComputeError) as e:
        sv.runComputeKernel(ctx, images=[bgra], target=bgra, kernel=sv.ComputeKernel.img_bgra_nv12, uniforms=u, blends=True)
    assert e.value.case == "badTarget"
    # wrong input plane structure -> badInputData
    with pytest.raises(sv.ComputeError) as e:
        sv.runComputeKernel(ctx, images=[nv12], target=nv12, kernel=sv.ComputeKernel.img_bgra_nv12, uniforms=u, blends=True)
    assert e.value.case == "badInputData"
    # CPU-resident target -> badTarget (compute.cl.swift:277-279)
    cpu = sv.createPictureSample((16, 8), sv.PixelFormat.nv12)
    with pytest.raises(sv.ComputeError) as e:
        sv.runComputeKernel(ctx, images=[], target=cpu, kernel=sv.ComputeKernel.img_clear_nv12)
    assert e.value.case == "badTarget"
    # composite kernel without uniforms -> invalidValue
    with pytest.raises(sv.ComputeError) as e:
        sv.runComputeKernel(ctx, images=[bgra], target=nv12, kernel=sv.ComputeKernel.img_bgra_nv12, blends=True)
    assert e.value.case == "invalidValue"
    # the context is still usable afterwards
    sv.usingContext(ctx, lambda c: sv.runComputeKernel(c, images=[], target=nv12, kernel=sv.ComputeKernel.img_clear_nv12))


def test_upload_download_roundtrip_and_barriers(ctx):
    """GPUBarrierUpload/Download (compute.swift:175-255): idempotent pass-through, own shared context."""
    src = util.alloc_image("y420p", 38, 22, seed=11)
    pict = sv.pictureFromArrays(sv.PixelFormat.y420p, (38, 22), src)
    up, down = sv.GPUBarrierUpload(ctx), sv.GPUBarrierDownload(ctx, retainGpuBuffer=False)
    tag, gpu = up(pict)
    assert tag == "just" and gpu.bufferType() == "gpu"
    assert up(gpu)[1] is gpu                      # already on the GPU: passes through
    tag, cpu = down(gpu.derive(img=gpu.imageBuffer().withChanges(buffers=[])))
    assert tag == "just" and cpu.bufferType() == "cpu" and cpu.imageBuffer().computeTextures == []
    for a, b in zip(cpu.imageBuffer().buffers, src):
        assert np.array_equal(a[:, : b.shape[1]], b)
    assert down(cpu)[1] is cpu
    # async upload stages the bytes before returning: the source may be overwritten immediately
    big = util.alloc_image("bgra", 320, 200, seed=12)
    keep = big[0].copy()
    p2 = sv.pictureFromArrays(sv.PixelFormat.BGRA, (320, 200), big)
    g2 = sv.uploadComputePicture(ctx, p2, asynchronous=True, retainCpuBuffer=False)
    p2.imageBuffer().buffers[0][:] = 0
    back = sv.downloadComputePicture(ctx, g2).imageBuffer().buffers[0]
    assert np.array_equal(back.reshape(200, 320, 4), keep)


def test_async_upload_on_a_side_context_orders_before_kernels(ctx):
    """hipMemcpyAsync on the upload barrier's own stream, kernels on the mixer's stream: the planes'
    upload events order them without a host wait (north star: H2D overlaps the kernel chain)."""
    up_ctx = sv.createComputeContext(sharing=ctx)
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    u = util.full_canvas_uniforms((dw, dh), (sw, sh))
    canvas = G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh))
    for i in range(6):
        src = util.alloc_image("nv12", sw, sh, seed=900 + i)
        gs = sv.uploadComputePicture(up_ctx, sv.pictureFromArrays(sv.PixelFormat.nv12, (sw, sh), src), asynchronous=True)
        for p in src:        # the host bytes were staged: scribbling over them must not matter
            keep = p.copy()
            p[...] = 0
            p[...] = keep
        sv.usingContext(ctx, lambda c: sv.compositeTick(c, canvas, [(sv.ComputeKernel.img_nv12_bgra, gs, u, 0)], True))
        exp = util.alloc_image("bgra", dw, dh)
        assert O.run_kernel("img_clear_bgra", exp, threads=8) == 0
        assert O.run_kernel("img_nv12_bgra", exp, src, u, threads=8) == 0
        G.assert_same(G.from_gpu(ctx, canvas, "bgra", dw, dh), exp, f"async frame {i}")
    sv.destroyComputeContext(up_ctx)


def test_descriptor_validation_through_the_c_abi(ctx):
    """Malformed descriptors are rejected with the ComputeError case the reference would raise; no launch happens."""
    import ctypes as C
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    bgra = G.to_gpu(ctx, "bgra", 32, 16, util.alloc_image("bgra", 32, 16, seed=1))
    nv12 = G.to_gpu(ctx, "nv12", 32, 16, util.alloc_image("nv12", 32, 16, seed=2))
    u = util.full_canvas_uniforms((32, 16), (32, 16))
    t = sv._image_desc(nv12)
    i = sv._image_desc(bgra)

    def run(target, inp, kernel=cv.K_IMG_BGRA_NV12, uniforms=u, size=236, blends=1):
        arr = (cv.Image * 1)(inp)
        return lib.chv_run_kernel(ctx.handle, kernel, C.byref(target), arr, 1, uniforms.ctypes.data if uniforms is not None else None,
                                  size, blends, None)

    assert run(t, i) == 0
    bad = sv._image_desc(bgra); bad.planes[0].pitch = 32 * 4 - 4
    assert run(t, bad) == 5                                  # pitch smaller than a row -> badInputData
    bad = sv._image_desc(bgra); bad.planes[0].height = 4000
    assert run(t, bad) == 5                                  # extent exceeds the device buffer
    bad = sv._image_desc(bgra); bad.planes[0].offset = 2
    assert run(t, bad) == 5                                  # 4-component plane not 4-byte aligned
    bad = sv._image_desc(bgra); bad.planes[0].buffer = None
    assert run(t, bad) == 5
    badt = sv._image_desc(nv12); badt.planes[1].components = 1
    assert run(badt, i) == 4                                 # chroma plane must be 2 components -> badTarget
    badt = sv._image_desc(nv12); badt.n_planes = 1
    assert run(badt, i) == 4
    assert run(t, i, size=200) == 1                          # not the 236-byte ImageUniforms -> invalidValue
    assert run(t, i, uniforms=None, size=0) == 1
    assert run(t, i, blends=0) == 10                         # composite kernels read the target -> invalidOperation
    assert run(t, i, kernel=99) == 7                         # unknown id -> computeKernelNotFound
    # upload / download regions are checked against the buffer
    buf = bgra.imageBuffer().computeTextures[0]
    host = np.zeros(64, dtype=np.uint8)
    assert lib.chv_upload(ctx.handle, buf._h, buf.size - 8, 64, host.ctypes.data, 64, 64, 1, 0) == 5
    assert lib.chv_download(ctx.handle, host.ctypes.data, 64, buf._h, 0, 16, 64, 1) == 5   # pitch < row bytes
    # mixed target formats in a batch (a tick deeper than CHV_MAX_LAYERS is fine: chv_composite splits it,
    # tests/test_gpu_mixpath.py::test_composite_more_than_sixteen_layers)
    layer = (sv.ComputeKernel.img_bgra_nv12, bgra, u, 0)
    sv.compositeTick(ctx, nv12, [layer] * 17, True)
    with pytest.raises(sv.ComputeError) as e:
        G.make_batch(ctx, [(nv12, True, [layer]), (bgra, True, [])])
    assert e.value.case == "badTarget"
    # everything still works afterwards
    assert run(t, i) == 0
    sv.endComputePass(ctx, True)


def test_wrapped_decoder_surface_with_plane_offsets(ctx):
    """Zero-copy hand-off of a decoder-style frame: ONE device allocation holding the Y and the interleaved
    chroma plane with a 64-byte-aligned linesize (how FFmpeg lays out an AVFrame, dec.video.ffmpeg.swift:187-221),
    adopted with chv_buffer_wrap and described by plane offsets."""
    import ctypes as C
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    sw, sh, dw, dh = 200, 90, 320, 144
    linesize = 256                                             # > width, 64-byte aligned
    src = util.alloc_image("nv12", sw, sh, seed=55)
    frame = np.zeros(linesize * (sh + sh // 2), dtype=np.uint8)
    frame[: linesize * sh].reshape(sh, linesize)[:, :sw] = src[0]
    frame[linesize * sh:].reshape(sh // 2, linesize)[:, :sw] = src[1].reshape(sh // 2, sw)
    owner = C.c_void_p()
    cv.check(lib.chv_buffer_alloc(ctx.handle, frame.size, C.byref(owner)))
    cv.check(lib.chv_upload(ctx.handle, owner, 0, frame.size, frame.ctypes.data, frame.size, frame.size, 1, 0))
    devptr = C.c_void_p()
    cv.check(lib.chv_buffer_info(owner, C.byref(devptr), None))
    wrapped = C.c_void_p()
    cv.check(lib.chv_buffer_wrap(ctx.handle, devptr, frame.size, C.byref(wrapped)))
    img = cv.Image()
    img.format, img.width, img.height, img.n_planes = cv.FMT_NV12, sw, sh, 2
    img.planes[0] = cv.Plane(wrapped.value, 0, sw, sh, linesize, 1)
    img.planes[1] = cv.Plane(wrapped.value, linesize * sh, sw // 2, sh // 2, linesize, 2)
    canvas = G.to_gpu(ctx, "bgra", dw, dh, util.alloc_image("bgra", dw, dh, seed=56))
    u = util.make_uniforms((dw, dh), rect=(10, 7, 280, 120), opacity=0.9, in_size=(sw, sh))
    tdesc = sv._image_desc(canvas)
    arr = (cv.Image * 1)(img)
    cv.check(lib.chv_pass_begin(ctx.handle))
    cv.check(lib.chv_run_kernel(ctx.handle, cv.K_IMG_NV12_BGRA, C.byref(tdesc), arr, 1, u.ctypes.data, 236, 1, None))
    cv.check(lib.chv_pass_end(ctx.handle, 1))
    exp = util.alloc_image("bgra", dw, dh, seed=56)
    assert O.run_kernel("img_nv12_bgra", exp, src, u) == 0
    G.assert_same(G.from_gpu(ctx, canvas, "bgra", dw, dh), exp, "wrapped surface")
    cv.check(lib.chv_buffer_free(wrapped))                     # does not free the adopted memory
    back = np.zeros(16, dtype=np.uint8)
    cv.check(lib.chv_download(ctx.handle, back.ctypes.data, 16, owner, 0, 16, 16, 1))
    assert np.array_equal(back, frame[:16])
    cv.check(lib.chv_buffer_free(owner))


@pytest.mark.parametrize("iw,ih,ow,oh", [(3, 5, 7, 9),          # narrower than one 16-byte vector: texel-by-texel staging
                                         (200, 120, 20, 12),    # 10:1, 60 taps: weights from memory, no register prefetch
                                         (130, 70, 40, 22),     # 3.25:1, 20 taps in registers, rectangle too tall to prefetch
                                         (257, 131, 129, 66),   # ~2:1 with odd sizes: prefetch path, edge vectors re-ordered
                                         (16, 16, 16, 16),      # identity size
                                         # exact 2:1 (12 taps, every output two source texels on): the wave-per-strip kernel
                                         (256, 128, 128, 64),   # two full strips, one row chunk
                                         (146, 20, 73, 10),     # narrowest source it takes; second strip partial, both strips on a picture edge
                                         (640, 360, 320, 180),  # interior strips + edge strips, several row chunks
                                         (1000, 200, 500, 100), # 8 strips, the last one partial
                                         (160, 1000, 80, 500),  # tall: many row chunks, tail chunk shorter than the others
                                         (200, 8, 100, 4),      # fewer output rows than one window
                                         (256, 90, 128, 60),    # 2:1 across only (10 taps down): stays on the tile kernel
                                         # equal tap counts on both axes, staged row of at most 64 vectors: the general wave-per-strip kernel,
                                         # one case per tap count it is instantiated for (6 and 12 and 20 are above: 16x16, 257x131, 130x70)
                                         (240, 120, 200, 100),  # 1.2:1, 8 taps
                                         (300, 150, 200, 100),  # 3:2, 10 taps: one or two new source rows per output row
                                         (440, 220, 200, 100),  # 2.2:1, 14 taps
                                         (500, 250, 200, 100),  # 2.5:1, 16 taps
                                         (600, 300, 200, 100),  # 3:1, 18 taps
                                         (700, 140, 200, 40),   # 3.5:1, 22 taps: 63 staged vectors
                                         (100, 50, 333, 171),   # enlargement by 3.33 / 3.42, odd sizes: output rows sharing all their source rows
                                         (8, 8, 5, 5),          # narrower than one strip, every vector on an edge
                                         (1920, 1080, 1280, 720)])   # a real 3:2 size: many strips, row chunks with and without a tail
def test_lanczos_paths_match_oracle(ctx, iw, ih, ow, oh):
    src = util.alloc_image("bgra", iw, ih, seed=iw * 7 + oh)
    exp = util.alloc_image("bgra", ow, oh)
    assert O.lanczos_bgra(exp[0], src[0]) == 0
    gs = G.to_gpu(ctx, "bgra", iw, ih, src)
    gd = G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh))
    sv.usingContext(ctx, lambda c: sv.scaleLanczos(c, gd, gs))
    G.assert_same(G.from_gpu(ctx, gd, "bgra", ow, oh), exp, f"lanczos {iw}x{ih} -> {ow}x{oh}")


@pytest.mark.parametrize("seed", range(48))
def test_lanczos_random_geometries(ctx, seed):
    """Seeded random sizes: every kernel of the family (2:1 strip, general strip for each tap count, tiles) with strips and row chunks
    that end anywhere, enlargements and reductions mixed across the axes."""
    rng = np.random.default_rng(5200 + seed)
    iw, ih = int(rng.integers(4, 700)), int(rng.integers(1, 260))
    if seed % 3 == 0:                                   # same ratio on both axes: the strip kernels
        r = float(rng.uniform(0.3, 3.4))
        ow, oh = max(1, int(round(iw / r))), max(1, int(round(ih / r)))
    elif seed % 3 == 1:                                 # exact 2:1
        iw, ih = 2 * int(rng.integers(2, 350)), 2 * int(rng.integers(1, 130))
        ow, oh = iw // 2, ih // 2
    else:                                               # anything
        ow, oh = int(rng.integers(1, 400)), int(rng.integers(1, 200))
    src = util.alloc_image("bgra", iw, ih, seed=seed + 1)
    exp = util.alloc_image("bgra", ow, oh)
    if O.lanczos_bgra(exp[0], src[0]) != 0:
        pytest.skip("ratio beyond the oracle's tap limit")
    gs = G.to_gpu(ctx, "bgra", iw, ih, src)
    gd = G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh, seed=9))
    try:
        sv.usingContext(ctx, lambda c: sv.scaleLanczos(c, gd, gs))
    except sv.ComputeError:
        # the library refuses reductions whose 8 x 4 tile needs more than 160 KB of staged source (about 24:1 on one axis, about 17:1 on
        # both at once: include/chipvideo.h); nothing else may fail
        assert max(iw / ow, ih / oh) > 20 or min(iw / ow, ih / oh) > 14
        return
    G.assert_same(G.from_gpu(ctx, gd, "bgra", ow, oh), exp, f"lanczos {iw}x{ih} -> {ow}x{oh}")


@pytest.mark.parametrize("iw,ih,ow,oh,n", [(96, 54, 48, 27, 5), (64, 36, 128, 72, 3), (200, 120, 75, 45, 70), (33, 17, 20, 10, 2), (40, 24, 20, 12, 130), (288, 96, 144, 48, 9), (300, 150, 200, 100, 7)])
def test_lanczos_batch_equals_single_calls(ctx, iw, ih, ow, oh, n):
    """chv_scale_lanczos_batch: n resizes of one geometry in one launch per 64 pairs == the oracle, image by image"""
    srcs = [util.alloc_image("bgra", iw, ih, seed=900 + i) for i in range(n)]
    pairs, exps = [], []
    for s in srcs:
        exp = util.alloc_image("bgra", ow, oh)
        assert O.lanczos_bgra(exp[0], s[0]) == 0
        exps.append(exp)
        pairs.append((G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh, seed=3)), G.to_gpu(ctx, "bgra", iw, ih, s)))
    batch = sv.LanczosBatch(pairs)
    sv.usingContext(ctx, lambda c: batch.run(c))
    sv.usingContext(ctx, lambda c: batch.run(c))          # replayable
    for i, ((gd, _), exp) in enumerate(zip(pairs, exps)):
        G.assert_same(G.from_gpu(ctx, gd, "bgra", ow, oh), exp, f"batched lanczos, image {i} of {n}")


def test_lanczos_empty_batch_is_a_noop(ctx):
    assert sv.LanczosBatch([]).run(ctx) is ctx


def test_lanczos_table_cache_eviction_keeps_results_exact(ctx):
    """More live geometries than the table cache holds (64 per device, evicted tables freed 32 at a time): an animated resize
    makes a new (in, out) pair per frame.  Every result stays exact, including a geometry that was evicted and comes back."""
    src = util.alloc_image("bgra", 96, 40, seed=77)
    gs = G.to_gpu(ctx, "bgra", 96, 40, src)
    sizes = [(20 + i, 9 + (i % 7)) for i in range(110)] + [(20, 9), (21, 10)]
    for ow, oh in sizes:
        gd = G.to_gpu(ctx, "bgra", ow, oh, util.alloc_image("bgra", ow, oh))
        sv.usingContext(ctx, lambda c: sv.scaleLanczos(c, gd, gs))
        if ow % 10 == 0 or (ow, oh) in ((21, 10),):
            exp = util.alloc_image("bgra", ow, oh)
            assert O.lanczos_bgra(exp[0], src[0]) == 0
            G.assert_same(G.from_gpu(ctx, gd, "bgra", ow, oh), exp, f"lanczos 96x40 -> {ow}x{oh} after evictions")


def test_lanczos_batch_rejects_mixed_geometry(ctx):
    a = (G.to_gpu(ctx, "bgra", 32, 18, util.alloc_image("bgra", 32, 18)), G.to_gpu(ctx, "bgra", 64, 36, util.alloc_image("bgra", 64, 36, seed=1)))
    b = (G.to_gpu(ctx, "bgra", 32, 18, util.alloc_image("bgra", 32, 18)), G.to_gpu(ctx, "bgra", 80, 36, util.alloc_image("bgra", 80, 36, seed=2)))
    with pytest.raises(sv.ComputeError) as e:
        sv.LanczosBatch([a, b]).run(ctx)
    assert e.value.case == "invalidValue"
