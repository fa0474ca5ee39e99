// tests/cpp/test_host_mirror.cpp — the C++ host mirror (swiftvideo_amd/host/swiftvideo_hip.hpp)
// driven the way the reference's Swift call sites drive the backend, checked against the oracle.
//   ./test_host_mirror cpu   — no device needed: name table, matrices, uniforms, layouts, error on no device
//   ./test_host_mirror gpu   — mixer ticks, barriers, kernels == oracle byte for byte
// Built and run by tests/test_cpp_host_mirror.py.  The oracle is linked here as the checker only.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../oracle/ref_kernels.h"
#include "../../swiftvideo_amd/host/swiftvideo_hip.hpp"

static int g_fail = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } } while (0)

// splitmix64 low bytes, as tests/util.py
static void fill(sv::Data &d, uint64_t seed) {
    uint64_t x = seed;
    for (auto &b : d) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        b = (uint8_t)z;
    }
}

static sv::PictureSample randomPicture(sv::PixelFormat f, int w, int h, uint64_t seed) {
    sv::PictureSample s = sv::createPictureSample({ (float)w, (float)h }, f, "cam" + std::to_string(seed));
    for (size_t i = 0; i < s.img->buffers.size(); i++) fill(*s.img->buffers[i], seed * 16 + i);
    s.revision = "rev" + std::to_string(seed);
    return s;
}

static std::vector<orc_plane> oraclePlanes(const sv::PictureSample &s) {
    std::vector<orc_plane> out;
    for (size_t i = 0; i < s.img->planes.size(); i++) {
        const sv::Plane &p = s.img->planes[i];
        out.push_back(orc_plane{ s.img->buffers[i]->data(), (int)p.size.x, (int)p.size.y, p.stride, sv::planeComponents(p) });
    }
    return out;
}

static bool samePlanes(const sv::PictureSample &a, const sv::PictureSample &b) {
    for (size_t i = 0; i < a.img->planes.size(); i++) {
        const sv::Plane &p = a.img->planes[i];
        size_t row = (size_t)p.size.x * sv::planeComponents(p);
        for (int y = 0; y < (int)p.size.y; y++)
            if (std::memcmp(a.img->buffers[i]->data() + (size_t)y * p.stride, b.img->buffers[i]->data() + (size_t)y * p.stride, row)) return false;
    }
    return true;
}

static void cpuTests() {
    // computeTests.swift:9-39
    const char *names[] = { "img_nv12_nv12", "img_bgra_nv12", "img_rgba_nv12", "img_bgra_bgra", "img_y420p_y420p", "img_y420p_nv12",
                            "img_clear_nv12", "img_clear_yuvs", "img_clear_bgra", "img_clear_rgba", "img_rgba_y420p", "img_bgra_y420p",
                            "img_clear_y420p" };
    for (const char *n : names) {
        std::string got = sv::describing(sv::defaultComputeKernelFromString(n));
        EXPECT(got == (std::string(n) == "img_clear_rgba" ? "img_clear_bgra" : n));
    }
    try { sv::defaultComputeKernelFromString("img_nv21_nv12"); EXPECT(false); }
    catch (const sv::ComputeError &e) { EXPECT(e.caseName == "invalidValue"); }
    // matrices: M = ortho * T * S, kernel rows = rows of M^-1 (compute.swift:151-155)
    sv::Matrix4 M = sv::Matrix4::ortho(1280, 720) * sv::Matrix4::translation(100, 50) * sv::Matrix4::scale(640, 360);
    float rows[16];
    M.kernelRows(rows);
    auto apply = [&](double px, double py, int r) {
        double v[4] = { px / 1280 * 2 - 1, py / 720 * 2 - 1, 0, 1 }, s = 0;
        for (int j = 0; j < 4; j++) s += rows[r * 4 + j] * v[j];
        return s;
    };
    EXPECT(std::fabs(apply(100, 50, 0)) < 1e-5 && std::fabs(apply(100, 50, 1)) < 1e-5);
    EXPECT(std::fabs(apply(740, 410, 0) - 1) < 1e-5 && std::fabs(apply(740, 410, 1) - 1) < 1e-5);
    sv::Matrix4 I = M * M.inverse();
    for (int i = 0; i < 16; i++) EXPECT(std::fabs(I.m[i] - (i % 5 == 0 ? 1.0 : 0.0)) < 1e-9);
    sv::Matrix4 full = sv::Matrix4::ortho(64, 36) * sv::Matrix4::scale(64, 36);
    full.kernelRows(rows);
    const float expect[16] = { .5f, 0, 0, .5f, 0, .5f, 0, .5f, 0, 0, 1, -1, 0, 0, 0, 1 };
    for (int i = 0; i < 16; i++) EXPECT(std::fabs(rows[i] - expect[i]) < 1e-6);
    // layouts
    auto p = sv::planesForFormat(sv::PixelFormat::nv12, { 1920, 1080 });
    EXPECT(p.size() == 2 && p[1].stride == 1920 && p[1].size.x == 960 && p[1].components == 2);
    auto q = sv::planesForFormat(sv::PixelFormat::y420p, { 1280, 720 });
    EXPECT(q.size() == 3 && q[2].stride == 640);
    EXPECT(sizeof(sv::ImageUniforms) == 236);
    try { sv::createPictureSample({ 0, 4 }, sv::PixelFormat::nv12); EXPECT(false); }
    catch (const sv::ComputeError &e) { EXPECT(e.caseName == "invalidOperation"); }
    // PictureAnimator (animator.pic.swift:207-272): aspect fit of a 4:3 picture in a 16:9 rect, border rect
    {
        sv::ElementState st;
        st.picPos[0] = 100; st.picPos[1] = 50; st.size[0] = 640; st.size[1] = 360;
        st.borderSize[0] = 4; st.borderSize[1] = 2; st.borderSize[2] = 6; st.borderSize[3] = 8;
        st.picAspect = sv::AspectMode::aspectFit; st.transparency = 0.25; st.hasFillColor = true; st.fillColor[0] = 1; st.fillColor[3] = 0.5;
        sv::ComputedPictureState cs = sv::computePictureState({ 640, 480 }, st);
        double sxs = (640.0 / 480.0) / (640.0 / 360.0);
        EXPECT(std::fabs(cs.textureMatrix.m[0] - sxs) < 1e-6 && std::fabs(cs.textureMatrix.m[3] - (1 - sxs) / 2) < 1e-6 && cs.textureMatrix.m[5] == 1.0);
        EXPECT(cs.opacity == 0.75f && cs.fillColor.w == 0.5f);
        EXPECT(cs.matrix.m[3] == 100 && cs.matrix.m[7] == 50 && cs.matrix.m[0] == 640 && cs.matrix.m[5] == 360);
        EXPECT(cs.borderMatrix.m[3] == 96 && cs.borderMatrix.m[7] == 48 && cs.borderMatrix.m[0] == 650 && cs.borderMatrix.m[5] == 370);
        sv::PictureSample pic = sv::createPictureSample({ 640, 480 }, sv::PixelFormat::nv12);
        auto stamped = sv::PictureAnimator({ 1280, 720 }, st, "cam@1")(pic);
        EXPECT(stamped.kind == stamped.just && stamped.value.revision == "cam@1" && stamped.value.opacity == 0.75f);
        sv::ImageUniforms u = sv::imageUniformsFor(stamped.value, sv::createPictureSample({ 1280, 720 }, sv::PixelFormat::nv12));
        // kernel rows map the rect's corners to tx = (0,0) / (1,1)
        auto tx = [&](double px, double py, int r) { double v[4] = { px / 1280 * 2 - 1, py / 720 * 2 - 1, 0, 1 }, s = 0; for (int j = 0; j < 4; j++) s += u.transform[r * 4 + j] * v[j]; return s; };
        EXPECT(std::fabs(tx(100, 50, 0)) < 1e-5 && std::fabs(tx(740, 410, 1) - 1) < 1e-5);
        st.hidden = true;
        EXPECT(sv::PictureAnimator({ 64, 36 }, st)(pic).kind == sv::EventBox<sv::PictureSample>::nothing);
    }
    // parent anchors (animator.pic.swift:149-193): corners follow the parent's corners when it grows by (30, 16)
    {
        const double base[2] = { 10, 20 }, size[2] = { 100, 50 }, ppos[2] = { 5, 7 }, d[2] = { 30, 16 };
        struct { unsigned a; double x, y, w, h; } cases[] = {
            { sv::anchorTopLeft, 15, 27, 100, 50 }, { sv::anchorTopRight, 45, 27, 100, 50 },
            { sv::anchorBottomLeft, 15, 43, 100, 50 }, { sv::anchorBottomRight, 45, 43, 100, 50 },
            { sv::anchorTopLeft | sv::anchorTopRight, 15, 27, 130, 50 }, { sv::anchorTopLeft | sv::anchorBottomLeft, 15, 27, 100, 66 },
            { sv::anchorTopLeft | sv::anchorBottomRight, 15, 27, 130, 66 }, { sv::anchorTopRight | sv::anchorBottomRight, 45, 27, 100, 66 },
            { sv::anchorBottomLeft | sv::anchorBottomRight, 15, 43, 130, 50 },
        };
        for (auto &c : cases) {
            double pos[2], sz[2];
            sv::computePositionSize(base, size, ppos, d, c.a, pos, sz);
            EXPECT(pos[0] == c.x && pos[1] == c.y && sz[0] == c.w && sz[1] == c.h);
        }
        sv::ElementState ps; ps.picPos[0] = 200; ps.picPos[1] = 100; ps.size[0] = 400; ps.size[1] = 300; ps.transparency = 0.5;
        sv::ElementState cs; cs.picPos[0] = 10; cs.picPos[1] = 20; cs.size[0] = 100; cs.size[1] = 50; cs.transparency = 0.2;
        cs.parentAnchor = sv::anchorBottomRight;
        sv::PictureAnimator parent({ 1280, 720 }, ps);
        sv::PictureAnimator child({ 1280, 720 }, sv::ElementState(), "", &parent);
        child.setState(cs);
        sv::PictureSample pic = sv::createPictureSample({ 64, 64 }, sv::PixelFormat::BGRA);
        sv::Matrix4 inv = sv::Matrix4::ortho(1280, 720).inverse();
        auto origin = [&](const sv::PictureSample &s, double out[2]) { sv::Matrix4 m = inv * s.matrix; out[0] = m.m[3]; out[1] = m.m[7]; };
        double o[2];
        auto first = child(pic);       // reference quirk: computed before the attachment state exists (:115-117)
        origin(first.value, o);
        EXPECT(std::fabs(o[0] - 610) < 1e-6 && std::fabs(o[1] - 420) < 1e-6 && std::fabs(first.value.opacity - 0.4f) < 1e-6);
        auto second = child(pic);
        origin(second.value, o);
        EXPECT(std::fabs(o[0] - 210) < 1e-6 && std::fabs(o[1] - 120) < 1e-6);
        ps.picPos[0] = 220; ps.picPos[1] = 90; ps.size[0] = 460; ps.size[1] = 330;
        parent.setState(ps);
        auto third = child(pic);       // rides the parent's bottom-right corner: +20 +60, -10 +30
        origin(third.value, o);
        EXPECT(std::fabs(o[0] - 290) < 1e-6 && std::fabs(o[1] - 140) < 1e-6);
    }
    if (!sv::hasAvailableComputeDevices(sv::ComputeDeviceType::GPU)) {
        try { sv::makeComputeContext(sv::ComputeDeviceType::GPU); EXPECT(false); }
        catch (const sv::ComputeError &e) { EXPECT(e.caseName == "deviceNotAvailable"); }
    }
}

static void gpuTests() {
    auto ctx = sv::makeComputeContext(sv::ComputeDeviceType::GPU);
    const int CW = 96, CH = 54;
    for (sv::PixelFormat fmt : { sv::PixelFormat::nv12, sv::PixelFormat::y420p }) {
        const char *fname = fmt == sv::PixelFormat::nv12 ? "nv12" : "y420p";
        sv::GPUBarrierUpload up(ctx);
        sv::GPUBarrierDownload down(ctx, true);
        sv::VideoMixer fused("ws", { (float)CW, (float)CH }, fmt, ctx, "mixer", true);
        sv::VideoMixer seq("ws", { (float)CW, (float)CH }, fmt, ctx, "mixer", false);
        struct Spec { sv::PixelFormat f; const char *name; int w, h; uint64_t seed; double x, y, rw, rh; int z; float op; };
        Spec specs[] = { { sv::PixelFormat::BGRA, "bgra", 40, 30, 11, 30, 10, 50, 30, 2, 0.8f },
                         { fmt, fname, 64, 36, 12, 0, 0, 96, 54, 0, 1.0f },
                         { sv::PixelFormat::RGBA, "rgba", 20, 20, 13, 5, 5, 30, 30, 1, 0.6f } };
        std::vector<sv::PictureSample> cpu;
        for (auto &sp : specs) {
            sv::PictureSample s = randomPicture(sp.f, sp.w, sp.h, sp.seed);
            s.matrix = sv::Matrix4::ortho(CW, CH) * sv::Matrix4::translation(sp.x, sp.y) * sv::Matrix4::scale(sp.rw, sp.rh);
            s.borderMatrix = s.matrix; s.opacity = sp.op; s.zIndex = sp.z;
            cpu.push_back(s);
            auto g = up(s);
            EXPECT(g.kind == g.just && g.value.bufferType() == sv::BufferType::gpu);
            EXPECT(fused.push(g.value).kind == sv::EventBox<sv::PictureSample>::nothing);
            seq.push(g.value);
        }
        // oracle: clear + layers in z order (mix.video.swift:114-124)
        sv::PictureSample exp = sv::createPictureSample({ (float)CW, (float)CH }, fmt);
        auto tp = oraclePlanes(exp);
        EXPECT(orc_run_kernel(fmt == sv::PixelFormat::nv12 ? ORC_IMG_CLEAR_NV12 : ORC_IMG_CLEAR_Y420P, tp.data(), (int)tp.size(), nullptr, 0, nullptr, 0, 1) == 0);
        int order[] = { 1, 2, 0 };
        for (int i : order) {
            sv::ImageUniforms u = sv::imageUniformsFor(cpu[i], exp);
            auto ip = oraclePlanes(cpu[i]);
            int kid = (int)sv::defaultComputeKernelFromString(std::string("img_") + specs[i].name + "_" + fname);
            EXPECT(orc_run_kernel(kid, tp.data(), (int)tp.size(), ip.data(), (int)ip.size(), (const orc_uniforms *)&u, 0, 2) == 0);
        }
        auto a = fused.mix(1.0), b = seq.mix(1.0);
        EXPECT(a.kind == a.just && b.kind == b.just);
        if (a.kind == a.just && b.kind == b.just) {
            auto da = down(a.value), db = down(b.value);
            EXPECT(da.kind == da.just && samePlanes(da.value, exp));
            EXPECT(db.kind == db.just && samePlanes(db.value, exp));
            // the same canvas through the asynchronous download: packed planes in pinned memory, complete after the pass's wait
            void *pinned = nullptr;
            EXPECT(chv_host_alloc(ctx.get(), (size_t)CW * CH * 2, &pinned) == 0);
            if (pinned) {
                std::memset(pinned, 0xA5, (size_t)CW * CH * 2);
                sv::beginComputePass(ctx);
                const size_t n = sv::downloadComputePictureAsync(ctx, a.value, pinned);
                sv::endComputePass(ctx, true);
                EXPECT(n == (size_t)CW * CH * 3 / 2);
                bool same = true;
                size_t off = 0;
                for (size_t i = 0; i < exp.img->planes.size(); i++) {
                    const sv::Plane &p = exp.img->planes[i];
                    const size_t rb = (size_t)p.size.x * (size_t)sv::planeComponents(p);
                    for (int y = 0; y < (int)p.size.y; y++)
                        same = same && std::memcmp((const uint8_t *)pinned + off + (size_t)y * rb, exp.img->buffers[i]->data() + (size_t)y * p.stride, rb) == 0;
                    off += rb * (size_t)p.size.y;
                }
                EXPECT(same);
                EXPECT(chv_host_free(ctx.get(), pinned) == 0);
            }
        }
        // a layer format the target has no kernel for surfaces as an event error (mix.video.swift:133-137)
        if (fmt == sv::PixelFormat::y420p) {
            auto g = up(randomPicture(sv::PixelFormat::nv12, 16, 8, 99));
            fused.push(g.value);
            auto r = fused.mix(2.0);
            EXPECT(r.kind == r.error && r.err.source == "mix.video" && r.err.code == -2);
        }
    }
    // NV12 -> BGRA convert + scale through applyComputeImage on a cleared canvas
    {
        sv::PictureSample src = randomPicture(sv::PixelFormat::nv12, 192, 108, 21);
        src.matrix = sv::Matrix4::ortho(128, 72) * sv::Matrix4::scale(128, 72);
        src.borderMatrix = src.matrix;
        auto gsrc = sv::uploadComputePicture(ctx, src);
        auto canvas = sv::uploadComputePicture(ctx, sv::createPictureSample({ 128, 72 }, sv::PixelFormat::BGRA));
        ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) {
            c = sv::runComputeKernel(c, {}, canvas, sv::ComputeKernel::img_clear_bgra);
            return sv::applyComputeImage(c, gsrc, canvas, sv::ComputeKernel::img_nv12_bgra);
        });
        auto got = sv::downloadComputePicture(ctx, canvas, true);
        sv::PictureSample exp = sv::createPictureSample({ 128, 72 }, sv::PixelFormat::BGRA);
        auto tp = oraclePlanes(exp); auto ip = oraclePlanes(src);
        sv::ImageUniforms u = sv::imageUniformsFor(src, exp);
        orc_run_kernel(ORC_IMG_CLEAR_BGRA, tp.data(), 1, nullptr, 0, nullptr, 0, 1);
        orc_run_kernel(ORC_IMG_NV12_BGRA, tp.data(), 1, ip.data(), 2, (const orc_uniforms *)&u, 0, 1);
        EXPECT(samePlanes(got, exp));
        // error behaviour: wrong target structure -> badTarget; context stays usable
        try { sv::applyComputeImage(ctx, gsrc, gsrc, sv::ComputeKernel::img_nv12_bgra); EXPECT(false); }
        catch (const sv::ComputeError &e) { EXPECT(e.caseName == "badTarget"); }
        try { sv::runComputeKernel(ctx, {}, canvas, sv::ComputeKernel::img_clear_yuvs); EXPECT(false); }
        catch (const sv::ComputeError &e) { EXPECT(e.caseName == "computeKernelNotFound"); }
        ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) { return sv::runComputeKernel(c, {}, canvas, sv::ComputeKernel::img_clear_bgra); });
    }
    // PictureFilter (filter.pict.swift:20-47): CPU y420p picture -> 128x72 BGRA, and BGRA -> NV12, against the oracle
    {
        sv::PictureSample src = randomPicture(sv::PixelFormat::y420p, 192, 108, 31);
        src.time = 2.0; src.pts = 2.5; src.assetId = "cam"; src.zIndex = 4;
        sv::PictureFilter filt({ 128, 72 }, sv::PixelFormat::BGRA, ctx);
        auto out = filt(src);
        EXPECT(out.kind == out.just && out.value.bufferType() == sv::BufferType::gpu && out.value.pixelFormat() == sv::PixelFormat::BGRA);
        EXPECT(out.value.assetId == "cam" && out.value.time == 2.0 && out.value.pts == 2.5 && out.value.zIndex == 4);
        sv::PictureSample exp = sv::createPictureSample({ 128, 72 }, sv::PixelFormat::BGRA);
        sv::PictureSample full = src;
        full.matrix = sv::Matrix4::ortho(128, 72) * sv::Matrix4::scale(128, 72); full.borderMatrix = full.matrix;
        sv::ImageUniforms u = sv::imageUniformsFor(full, exp);
        auto tp = oraclePlanes(exp); auto ip = oraclePlanes(src);
        orc_run_kernel(ORC_IMG_CLEAR_BGRA, tp.data(), 1, nullptr, 0, nullptr, 0, 1);
        orc_run_kernel(ORC_IMG_Y420P_BGRA, tp.data(), 1, ip.data(), 3, (const orc_uniforms *)&u, 0, 1);
        if (out.kind == out.just) EXPECT(samePlanes(sv::downloadComputePicture(ctx, out.value, true), exp));
        sv::PictureFilter toNv12({ 96, 54 }, sv::PixelFormat::nv12, ctx);
        sv::PictureSample rgb = randomPicture(sv::PixelFormat::BGRA, 160, 90, 32);
        auto o2 = toNv12(sv::uploadComputePicture(ctx, rgb));
        sv::PictureSample e2 = sv::createPictureSample({ 96, 54 }, sv::PixelFormat::nv12);
        sv::PictureSample f2 = rgb;
        f2.matrix = sv::Matrix4::ortho(96, 54) * sv::Matrix4::scale(96, 54); f2.borderMatrix = f2.matrix;
        sv::ImageUniforms u2 = sv::imageUniformsFor(f2, e2);
        auto tp2 = oraclePlanes(e2); auto ip2 = oraclePlanes(rgb);
        orc_run_kernel(ORC_IMG_CLEAR_NV12, tp2.data(), 2, nullptr, 0, nullptr, 0, 1);
        // an RGB picture to a 4:2:0 format: the integer BT.601 matrix by default (img_bgra_nv12_int) ...
        orc_run_kernel(ORC_IMG_BGRA_NV12_INT, tp2.data(), 2, ip2.data(), 1, (const orc_uniforms *)&u2, 0, 1);
        EXPECT(o2.kind == o2.just);
        if (o2.kind == o2.just) EXPECT(samePlanes(sv::downloadComputePicture(ctx, o2.value, true), e2));
        // ... the reference's float full-range kernel on request
        toNv12.integerMatrix = false;
        auto o2f = toNv12(sv::uploadComputePicture(ctx, rgb));
        sv::PictureSample e2f = sv::createPictureSample({ 96, 54 }, sv::PixelFormat::nv12);
        auto tp2f = oraclePlanes(e2f);
        orc_run_kernel(ORC_IMG_CLEAR_NV12, tp2f.data(), 2, nullptr, 0, nullptr, 0, 1);
        orc_run_kernel(ORC_IMG_BGRA_NV12, tp2f.data(), 2, ip2.data(), 1, (const orc_uniforms *)&u2, 0, 1);
        EXPECT(o2f.kind == o2f.just);
        if (o2f.kind == o2f.just) EXPECT(samePlanes(sv::downloadComputePicture(ctx, o2f.value, true), e2f));
        // no kernel for the pair -> error event, not an exception
        sv::PictureFilter bad({ 96, 54 }, sv::PixelFormat::RGBA, ctx);
        auto o3 = bad(src);
        EXPECT(o3.kind == o3.error && o3.err.source == "filter.pict");
    }
    // Lanczos-3: three resizes of one geometry as one launch == the same three one by one
    {
        std::vector<std::pair<sv::PictureSample, sv::PictureSample>> pairs;
        std::vector<sv::PictureSample> single;
        for (int k = 0; k < 3; k++) {
            sv::PictureSample src = sv::uploadComputePicture(ctx, randomPicture(sv::PixelFormat::BGRA, 96, 54, 90 + k));
            sv::PictureSample a = sv::uploadComputePicture(ctx, sv::createPictureSample({ 40, 22 }, sv::PixelFormat::BGRA));
            sv::PictureSample b = sv::uploadComputePicture(ctx, sv::createPictureSample({ 40, 22 }, sv::PixelFormat::BGRA));
            pairs.push_back({ a, src });
            ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) { return sv::scaleLanczos(c, b, src); });
            single.push_back(b);
        }
        ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) { return sv::scaleLanczos(c, pairs); });
        for (int k = 0; k < 3; k++)
            EXPECT(samePlanes(sv::downloadComputePicture(ctx, pairs[k].first, true), sv::downloadComputePicture(ctx, single[k], true)));
    }
    // VideoMixerGroup / TickBatch: three mixers ticked by one launch per canvas format == each mixer's own mix()
    {
        std::vector<std::unique_ptr<sv::VideoMixer>> grouped, solo;
        sv::PixelFormat fmts[3] = { sv::PixelFormat::nv12, sv::PixelFormat::BGRA, sv::PixelFormat::BGRA };
        for (int k = 0; k < 3; k++) for (auto *dest : { &grouped, &solo }) {
            dest->emplace_back(new sv::VideoMixer("ws", { 80, 44 }, fmts[k], ctx, "mixer" + std::to_string(k)));
            sv::PictureSample s = randomPicture(sv::PixelFormat::nv12, 48, 30, 60 + k);
            s.matrix = sv::Matrix4::ortho(80, 44) * sv::Matrix4::scale(80, 44); s.borderMatrix = s.matrix; s.revision = "cam";
            dest->back()->push(sv::uploadComputePicture(ctx, s));
            sv::PictureSample o = randomPicture(sv::PixelFormat::BGRA, 20, 16, 70 + k);
            o.matrix = sv::Matrix4::ortho(80, 44) * sv::Matrix4::translation(8, 6) * sv::Matrix4::scale(30, 20); o.borderMatrix = o.matrix;
            o.opacity = 0.7f; o.zIndex = 1; o.revision = "logo";
            dest->back()->push(sv::uploadComputePicture(ctx, o));
        }
        sv::VideoMixerGroup group({ grouped[0].get(), grouped[1].get(), grouped[2].get() });
        auto outs = group.mix(1.0);
        EXPECT(outs.size() == 3);
        for (int k = 0; k < 3; k++) {
            auto ref = solo[k]->mix(1.0);
            EXPECT(outs[k].kind == outs[k].just && ref.kind == ref.just && outs[k].value.assetId == "mixer" + std::to_string(k));
            if (outs[k].kind == outs[k].just && ref.kind == ref.just)
                EXPECT(samePlanes(sv::downloadComputePicture(ctx, outs[k].value, true), sv::downloadComputePicture(ctx, ref.value, true)));
        }
    }
    // .custom kernels through hipRTC: swap B and R of a picture, scaled by a float uniform
    {
        const char *src_text =
            "extern \"C\" __global__ void swap_br(chv_custom_args a) {\n"
            "    const chv_dev_plane &d = a.target.planes[0]; CHV_GUARD(d);\n"
            "    const chv_dev_plane &s = a.inputs[0].planes[0];\n"
            "    float gain = *(const float *)a.uniforms;\n"
            "    int x = CHV_GID_X, y = CHV_GID_Y;\n"
            "    chv_write(d, x, y, 0, chv_read(s, x, y, 2) * gain); chv_write(d, x, y, 1, chv_read(s, x, y, 1) * gain);\n"
            "    chv_write(d, x, y, 2, chv_read(s, x, y, 0) * gain); chv_write(d, x, y, 3, 1.0f);\n"
            "}\n";
        sv::PictureSample pic = randomPicture(sv::PixelFormat::BGRA, 50, 20, 41);
        auto gsrc = sv::uploadComputePicture(ctx, pic);
        auto gdst = sv::uploadComputePicture(ctx, sv::createPictureSample({ 50, 20 }, sv::PixelFormat::BGRA));
        sv::CustomKernel k{ "swap_br" };
        try { sv::runComputeKernel<float>(ctx, { gsrc }, gdst, k); EXPECT(false); }
        catch (const sv::ComputeError &e) { EXPECT(e.caseName == "computeKernelNotFound"); }
        try { sv::buildComputeKernel(ctx, "bad", "extern \"C\" __global__ void bad(chv_custom_args a) { oops }"); EXPECT(false); }
        catch (const sv::ComputeError &e) { EXPECT(e.caseName == "badInputData"); }
        ctx = sv::buildComputeKernel(ctx, "swap_br", src_text);
        float gain = 0.5f;
        ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) { return sv::runComputeKernel<float>(c, { gsrc }, gdst, k, 3, &gain); });
        auto got = sv::downloadComputePicture(ctx, gdst, true);
        const uint8_t *in = pic.img->buffers[0]->data(), *out = got.img->buffers[0]->data();
        int stride_in = pic.img->planes[0].stride, stride_out = got.img->planes[0].stride;
        bool ok = true;
        for (int y = 0; y < 20 && ok; y++) for (int x = 0; x < 50 && ok; x++) {
            const uint8_t *p = in + y * stride_in + x * 4, *q = out + y * stride_out + x * 4;
            for (int c = 0; c < 3; c++) {
                float f = ((float)p[2 - c] / 255.0f) * 0.5f;
                ok = ok && q[c] == orc_store_unorm8(f);
            }
            ok = ok && q[3] == 255;
        }
        EXPECT(ok);
    }
    // the two buffer kernels of the enum (compute.swift:67,70) through runComputeKernel's bind order, against the oracle
    {
        const int n = 1003;
        std::vector<int16_t> out(n), in0(n), in1(n), exp;
        for (int i = 0; i < n; i++) { out[i] = (int16_t)(i * 37 - 9000); in0[i] = (int16_t)(i * 131 + 77); in1[i] = (int16_t)(30000 - i * 59); }
        exp = out;
        sv::BufferUniforms u{};
        u.input_count = 2; u.input_gains[0] = 0.8f; u.input_gains[1] = -3.5f; u.input_fade[0] = 0.25f; u.input_fade[1] = 0.9f;
        const int16_t *ins[2] = { in0.data(), in1.data() };
        orc_snd_uniforms ou;
        std::memcpy(&ou, &u, sizeof ou);
        EXPECT(orc_snd_s16i_s16i(exp.data(), n, ins, &ou) == 0);
        auto gout = sv::uploadComputeBuffer(ctx, out.data(), n * 2), g0 = sv::uploadComputeBuffer(ctx, in0.data(), n * 2), g1 = sv::uploadComputeBuffer(ctx, in1.data(), n * 2);
        EXPECT(sv::defaultComputeKernelFromString("snd_s16i_s16i") == sv::ComputeKernel::snd_s16i_s16i);
        ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) {
            return sv::runComputeKernel(c, { sv::BufferImage{ g0, n }, sv::BufferImage{ g1, n } }, sv::BufferImage{ gout, n }, sv::ComputeKernel::snd_s16i_s16i, u);
        });
        sv::downloadComputeBuffer(ctx, gout, out.data());
        EXPECT(out == exp);

        const int W = 96, H = 64;
        std::vector<uint8_t> ref((size_t)W * H), cur((size_t)W * H);
        for (int i = 0; i < W * H; i++) { ref[i] = (uint8_t)((i * 2654435761u) >> 24); cur[i] = (uint8_t)(((i + 3 * W + 2) * 2654435761u) >> 24); }
        sv::MotionEstimationUniforms mu{ { 16, 16 }, { 32, 32 }, { W, H } };
        std::vector<uint8_t> mv((size_t)(W / 16) * (H / 16) * 4), mexp(mv.size());
        orc_plane po{ mexp.data(), W / 16, H / 16, (W / 16) * 4, 4 }, pr{ ref.data(), W, H, W, 1 }, pc{ cur.data(), W, H, W, 1 };
        orc_me_uniforms omu;
        std::memcpy(&omu, &mu, sizeof omu);
        EXPECT(orc_me_fullsearch(&po, &pr, &pc, &omu) == 0);
        auto gr = sv::uploadComputeBuffer(ctx, ref.data(), ref.size()), gc = sv::uploadComputeBuffer(ctx, cur.data(), cur.size()), gm = sv::uploadComputeBuffer(ctx, mv.data(), mv.size());
        ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) {
            return sv::runComputeKernel(c, { sv::BufferImage{ gr, W, H, 1 }, sv::BufferImage{ gc, W, H, 1 } }, sv::BufferImage{ gm, W / 16, H / 16, 4 },
                                        sv::ComputeKernel::me_fullsearch, mu);
        });
        sv::downloadComputeBuffer(ctx, gm, mv.data());
        EXPECT(mv == mexp);
    }
    sv::destroyComputeContext(ctx);
}

int main(int argc, char **argv) {
    std::string mode = argc > 1 ? argv[1] : "cpu";
    try {
        cpuTests();
        if (mode == "gpu") gpuTests();
    } catch (const std::exception &e) {
        std::printf("FAIL uncaught: %s\n", e.what());
        g_fail++;
    }
    std::printf("%s: %s (%d failures)\n", mode.c_str(), g_fail ? "FAILED" : "ok", g_fail);
    return g_fail ? 1 : 0;
}
