// tests/stubhip/abi_stress.cpp — the C ABI's host logic under sanitizers (tests/test_sanitizers.py): libchipvideo's chipvideo.cpp compiled for
// the CPU against the stand-in runtime of this directory, driven the way the Swift host drives it (a VideoMixer on its serial queue,
// upload / download barriers on contexts of their own, ComputeBuffer.deinit from any thread) plus what a hostile caller does — descriptor-ring
// wrap without a host wait, launches that fail after their descriptor slot was taken, ticks deeper than a launch, bad arguments.
// No pixels are checked here (the stub kernels touch memory, they do not compute): that is tests/ -m gpu.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "chipvideo.h"

void stubhip_fail_launch_after(int n);      // stub_runtime.cpp
long stubhip_launches();                    // kernel launches issued so far (all threads)

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s:%d %s -> %s (%s)\n", __FILE__, __LINE__, #x, chv_error_string(rc_), chv_last_error_detail()); exit(2); } } while (0)
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d expectation failed: %s\n", __FILE__, __LINE__, #cond); exit(3); } } while (0)

struct Pic { chv_buffer *buf = nullptr; chv_image img; };

static Pic make_pic(chv_context *c, int fmt, int w, int h) {
    Pic p; memset(&p.img, 0, sizeof p.img);
    p.img.format = fmt; p.img.width = w; p.img.height = h;
    if (fmt == CHV_FMT_NV12) {
        CK(chv_buffer_alloc(c, (size_t)w * h * 3 / 2, &p.buf));
        p.img.n_planes = 2;
        p.img.planes[0] = chv_plane{ p.buf, 0, w, h, w, 1 };
        p.img.planes[1] = chv_plane{ p.buf, (size_t)w * h, w / 2, h / 2, w, 2 };
    } else if (fmt == CHV_FMT_Y420P) {
        CK(chv_buffer_alloc(c, (size_t)w * h * 3 / 2, &p.buf));
        p.img.n_planes = 3;
        p.img.planes[0] = chv_plane{ p.buf, 0, w, h, w, 1 };
        p.img.planes[1] = chv_plane{ p.buf, (size_t)w * h, w / 2, h / 2, w / 2, 1 };
        p.img.planes[2] = chv_plane{ p.buf, (size_t)w * h * 5 / 4, w / 2, h / 2, w / 2, 1 };
    } else {
        size_t pitch = 0;
        CK(chv_plane_alloc(c, w, h, 4, &p.buf, &pitch));
        p.img.n_planes = 1;
        p.img.planes[0] = chv_plane{ p.buf, 0, w, h, (int32_t)pitch, 4 };
    }
    return p;
}
static chv_uniforms full_canvas(float opacity) {
    chv_uniforms u; memset(&u, 0, sizeof u);
    const float t[16] = { .5f, 0, 0, .5f, 0, .5f, 0, .5f, 0, 0, 1, -1, 0, 0, 0, 1 }, id[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    memcpy(u.transform, t, sizeof t); memcpy(u.border_matrix, t, sizeof t); memcpy(u.texture_transform, id, sizeof id);
    u.opacity = opacity; u.output_size[0] = 64; u.output_size[1] = 32; u.input_size[0] = 64; u.input_size[1] = 32;
    return u;
}
static chv_layer layer_of(int kernel, const Pic &p, float opacity) {
    chv_layer l; memset(&l, 0, sizeof l);
    l.kernel = kernel; l.image = p.img; l.uniforms = full_canvas(opacity);
    return l;
}

static void single_thread(chv_context *c) {
    const int W = 256, H = 64;
    std::vector<uint8_t> host((size_t)W * H * 4, 7);
    Pic nv = make_pic(c, CHV_FMT_NV12, W, H), yp = make_pic(c, CHV_FMT_Y420P, W, H), rgb = make_pic(c, CHV_FMT_BGRA, W, H);
    Pic canvas = make_pic(c, CHV_FMT_BGRA, W, H), canvas420 = make_pic(c, CHV_FMT_NV12, W, H), small = make_pic(c, CHV_FMT_BGRA, W / 2, H / 2);
    // uploads: synchronous, staged, pinned; then a download of each flavour
    void *pinned = nullptr;
    CK(chv_host_alloc(c, (size_t)W * H * 4, &pinned));
    memset(pinned, 9, (size_t)W * H * 4);
    CK(chv_upload(c, nv.buf, 0, W, host.data(), W, W, (size_t)H * 3 / 2, 0));
    CK(chv_upload(c, yp.buf, 0, W, host.data(), W, W, (size_t)H * 3 / 2, 1));
    CK(chv_upload(c, rgb.buf, 0, rgb.img.planes[0].pitch, pinned, (size_t)W * 4, (size_t)W * 4, H, 2));
    { std::vector<uint8_t> gone((size_t)W * H * 3 / 2, 3); CK(chv_upload(c, nv.buf, 0, W, gone.data(), W, W, (size_t)H * 3 / 2, 1)); }   // borrowed for the call only
    CK(chv_download(c, host.data(), (size_t)W * 4, rgb.buf, 0, rgb.img.planes[0].pitch, (size_t)W * 4, H));
    CK(chv_download_async(c, pinned, (size_t)W * 4, rgb.buf, 0, rgb.img.planes[0].pitch, (size_t)W * 4, H));
    CK(chv_download_async(c, host.data(), (size_t)W * 4, rgb.buf, 0, rgb.img.planes[0].pitch, (size_t)W * 4, H));       // pageable: completes in the call
    CK(chv_pass_end(c, 1));
    // ticks: by-value route (videos), ring route (an RGB layer), a clear, 4:2:0 canvas, a tick deeper than one launch
    chv_layer vids[4] = { layer_of(CHV_K_IMG_NV12_BGRA, nv, 1.f), layer_of(CHV_K_IMG_NV12_BGRA, nv, .75f), layer_of(CHV_K_IMG_NV12_BGRA, nv, .5f), layer_of(CHV_K_IMG_NV12_BGRA, nv, .25f) };
    chv_layer mixed[3] = { layer_of(CHV_K_IMG_NV12_BGRA, nv, 1.f), layer_of(CHV_K_IMG_Y420P_BGRA, yp, .5f), layer_of(CHV_K_IMG_BGRA_BGRA_TX, rgb, .5f) };
    chv_layer yuv[2] = { layer_of(CHV_K_IMG_NV12_NV12, nv, 1.f), layer_of(CHV_K_IMG_BGRA_NV12, rgb, .5f) };
    CK(chv_pass_begin(c));
    CK(chv_composite(c, &canvas.img, 1, vids, 4));
    CK(chv_composite(c, &canvas.img, 0, mixed, 3));
    CK(chv_composite(c, &canvas420.img, 1, yuv, 2));
    CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
    chv_uniforms u = full_canvas(.5f);
    CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &u, sizeof u, 1, nullptr));
    std::vector<chv_layer> deep(40, mixed[2]);
    CK(chv_composite(c, &canvas.img, 1, deep.data(), (int)deep.size()));
    CK(chv_pass_end(c, 0));
    // the descriptor ring wraps: 300 ring-route launches without a host wait, each with descriptors of its own
    for (int i = 0; i < 300; i++) { mixed[2].uniforms.opacity = i / 300.f; CK(chv_composite(c, &canvas.img, 0, mixed, 3)); }
    CK(chv_pass_end(c, 1));
    // batches: plain, split ("videos, then something else"), run several times, destroyed with work in flight behind them
    chv_layer split[5] = { vids[0], vids[1], vids[2], vids[3], mixed[2] };
    chv_tick ticks[6]; memset(ticks, 0, sizeof ticks);
    for (int i = 0; i < 6; i++) { ticks[i].target = canvas.img; ticks[i].clear_first = 1; ticks[i].n_layers = 5; ticks[i].layers = split; }
    chv_batch *b = nullptr, *b2 = nullptr;
    CK(chv_batch_create(c, ticks, 6, &b));
    for (int i = 0; i < 6; i++) { ticks[i].n_layers = 3; ticks[i].layers = mixed; ticks[i].clear_first = 0; }
    CK(chv_batch_create(c, ticks, 6, &b2));
    char name[128]; int nl = 0;
    CK(chv_batch_describe(b, name, sizeof name, &nl));
    EXPECT(nl == 2);
    for (int r = 0; r < 20; r++) { CK(chv_batch_run(c, b)); CK(chv_batch_run(c, b2)); }
    CK(chv_batch_destroy(b)); CK(chv_batch_destroy(b2));
    // Lanczos: one, and a batch of 130 pairs (three descriptor slots)
    CK(chv_scale_lanczos(c, &small.img, &canvas.img));
    std::vector<chv_image> ds(130, small.img), ss(130, canvas.img);
    CK(chv_scale_lanczos_batch(c, ds.data(), ss.data(), 130));
    // the two buffer kernels
    chv_buffer *a0 = nullptr, *a1 = nullptr;
    CK(chv_buffer_alloc(c, 2000, &a0)); CK(chv_buffer_alloc(c, 2000, &a1));
    chv_image out; memset(&out, 0, sizeof out); out.n_planes = 1; out.planes[0] = chv_plane{ a0, 0, 1000, 1, 2000, 2 };
    chv_image in = out; in.planes[0].buffer = a1;
    chv_snd_uniforms su; memset(&su, 0, sizeof su); su.input_count = 1; su.input_gains[0] = 1.f; su.input_fade[0] = .5f;
    CK(chv_run_kernel(c, CHV_K_SND_S16I_S16I, &out, &in, 1, &su, sizeof su, 0, nullptr));
    chv_me_uniforms mu = { { 16, 16 }, { 32, 32 }, { W, H } };
    chv_image me_in[2] = { nv.img, yp.img };
    chv_image me_out; memset(&me_out, 0, sizeof me_out); me_out.n_planes = 1; me_out.planes[0] = chv_plane{ small.buf, 0, W / 16, H / 16, small.img.planes[0].pitch, 4 };
    CK(chv_run_kernel(c, CHV_K_ME_FULLSEARCH, &me_out, me_in, 2, &mu, sizeof mu, 0, nullptr));
    // custom kernels through the stand-in hipRTC: a build, a run, a failing build
    CK(chv_kernel_build(c, "mine", "extern \"C\" __global__ void mine(chv_custom_args a) {}"));
    CK(chv_run_custom(c, "mine", &canvas.img, &rgb.img, 1, &u, sizeof u, 1));
    EXPECT(chv_kernel_build(c, "broken", "int x;") != CHV_OK);
    EXPECT(chv_run_custom(c, "absent", &canvas.img, nullptr, 0, nullptr, 0, 0) != CHV_OK);
    // events
    chv_event *e0 = nullptr, *e1 = nullptr; float ms = -1.f;
    CK(chv_event_create(c, &e0)); CK(chv_event_create(c, &e1));
    CK(chv_event_record(c, e0)); CK(chv_composite(c, &canvas.img, 1, vids, 4)); CK(chv_event_record(c, e1));
    CK(chv_event_synchronize(e1)); CK(chv_event_elapsed_ms(e0, e1, &ms)); EXPECT(ms >= 0.f);
    CK(chv_event_destroy(e0)); CK(chv_event_destroy(e1));
    // launches that FAIL, at every position of a short sequence: the error comes back, the context stays usable, no ring slot is lost
    for (int k = 1; k <= 8; k++) {
        stubhip_fail_launch_after(k);
        int failures = 0;
        for (int i = 0; i < 6; i++) {
            failures += chv_composite(c, &canvas.img, 0, mixed, 3) != CHV_OK;
            failures += chv_composite(c, &canvas.img, 1, vids, 4) != CHV_OK;
            failures += chv_scale_lanczos_batch(c, ds.data(), ss.data(), 70) != CHV_OK;
        }
        EXPECT(failures == 1);
        stubhip_fail_launch_after(0);
        for (int i = 0; i < 200; i++) CK(chv_composite(c, &canvas.img, 0, mixed, 3));       // three times round the ring
        CK(chv_pass_end(c, 1));
    }
    // ---- deferred passes: picture kernels issued inside a pass are held and leave as ONE launch at its end (an unchanged VideoMixer tick,
    // mix.video.swift:116-124: clear + N x runComputeKernel + endComputePass)
    {
        chv_uniforms us[4] = { full_canvas(1.f), full_canvas(.75f), full_canvas(.5f), full_canvas(.25f) };
        long l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        for (int l = 0; l < 4; l++) CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[l], sizeof us[l], 1, nullptr));
        EXPECT(stubhip_launches() == l0);                                  // nothing has been launched yet
        CK(chv_pass_end(c, 1));
        EXPECT(stubhip_launches() == l0 + 1);                              // ... and the five kernels were one launch
        // outside a pass every kernel launches at once, as before
        l0 = stubhip_launches();
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[0], sizeof us[0], 1, nullptr));
        EXPECT(stubhip_launches() == l0 + 2);
        // two canvases in one pass: two launches, in issue order; a clear after layers starts a new launch
        l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[0], sizeof us[0], 1, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_NV12, &canvas420.img, nullptr, 0, nullptr, 0, 0, nullptr));
        EXPECT(stubhip_launches() == l0 + 1);                              // (the first canvas went out when the second one appeared)
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_NV12, &canvas420.img, &nv.img, 1, &us[0], sizeof us[0], 1, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_NV12, &canvas420.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_BGRA_NV12, &canvas420.img, &rgb.img, 1, &us[1], sizeof us[1], 1, nullptr));
        CK(chv_pass_end(c, 0));
        EXPECT(stubhip_launches() == l0 + 3);
        // a ComputeBuffer released between runComputeKernel and endComputePass (deinit may run the moment the call returns,
        // compute.cl.swift:55-57): held until the pass has been launched, then freed; a second free of it is an error, not a crash
        Pic tmp = make_pic(c, CHV_FMT_NV12, W, H), tmp_canvas = make_pic(c, CHV_FMT_BGRA, W, H);
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &tmp_canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &tmp_canvas.img, &tmp.img, 1, &us[0], sizeof us[0], 1, nullptr));
        CK(chv_buffer_free(tmp.buf));
        EXPECT(chv_buffer_free(tmp.buf) != CHV_OK);
        CK(chv_buffer_free(tmp_canvas.buf));                               // (the target too)
        CK(chv_pass_end(c, 1));
        // an argument error comes back from the call that made it; the pass goes on without that kernel
        l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        EXPECT(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[0], 17, 1, nullptr) == CHV_ERR_INVALID_VALUE);
        EXPECT(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &rgb.img, 1, &us[0], sizeof us[0], 1, nullptr) != CHV_OK);      // BGRA planes for an NV12 kernel
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[0], sizeof us[0], 1, nullptr));
        // a download in the middle of the pass sees what was issued before it
        CK(chv_download(c, host.data(), (size_t)W * 4, canvas.buf, 0, canvas.img.planes[0].pitch, (size_t)W * 4, H));
        EXPECT(stubhip_launches() == l0 + 1);
        CK(chv_run_kernel(c, CHV_K_IMG_Y420P_BGRA, &canvas.img, &yp.img, 1, &us[2], sizeof us[2], 1, nullptr));
        CK(chv_pass_end(c, 1));
        EXPECT(stubhip_launches() == l0 + 2);
        // a pass deeper than one launch (CHV_MAX_LAYERS): chunks, like chv_composite
        l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        for (int l = 0; l < 40; l++) CK(chv_run_kernel(c, CHV_K_IMG_BGRA_BGRA_TX, &canvas.img, &rgb.img, 1, &us[l & 3], sizeof us[0], 1, nullptr));
        CK(chv_pass_end(c, 1));
        EXPECT(stubhip_launches() == l0 + 3);
        // a layer that reads the held canvas sends it out first; a pass of thousands of layers goes out in instalments
        l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_BGRA_BGRA_TX, &canvas.img, &rgb.img, 1, &us[0], sizeof us[0], 1, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_BGRA_BGRA_TX, &canvas.img, &canvas.img, 1, &us[1], sizeof us[1], 1, nullptr));       // reads what it writes
        EXPECT(stubhip_launches() == l0 + 1);
        CK(chv_pass_end(c, 1));
        EXPECT(stubhip_launches() == l0 + 2);
        CK(chv_pass_begin(c));
        for (int l = 0; l < 2100; l++) CK(chv_run_kernel(c, CHV_K_IMG_BGRA_BGRA_TX, &canvas.img, &rgb.img, 1, &us[l & 3], sizeof us[0], 1, nullptr));
        CK(chv_pass_end(c, 1));
        // the launch fails when the pass ends: the error comes back from chv_pass_end, nothing stays held, the context stays usable
        stubhip_fail_launch_after(1);
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[0], sizeof us[0], 1, nullptr));
        EXPECT(chv_pass_end(c, 1) != CHV_OK);
        stubhip_fail_launch_after(0);
        CK(chv_pass_end(c, 1));
        // brackets nest (uploadComputePicture brackets its copies, compute.cl.swift:433,453 — inside the mixer's pass in a host that uploads there): the
        // inner end launches what is held so far, the outer pass goes on holding
        l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        CK(chv_pass_begin(c));
        CK(chv_upload(c, yp.buf, 0, W, host.data(), W, W, (size_t)H * 3 / 2, 0));
        CK(chv_pass_end(c, 1));
        EXPECT(stubhip_launches() == l0 + 1);
        CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[0], sizeof us[0], 1, nullptr));
        CK(chv_run_kernel(c, CHV_K_IMG_Y420P_BGRA, &canvas.img, &yp.img, 1, &us[1], sizeof us[1], 1, nullptr));
        EXPECT(stubhip_launches() == l0 + 1);                              // (still inside the outer bracket)
        CK(chv_pass_end(c, 1));
        EXPECT(stubhip_launches() == l0 + 2);
        CK(chv_pass_end(c, 1));                                            // (an end without a begin: a plain wait, as before)
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        EXPECT(stubhip_launches() == l0 + 3);                              // outside every bracket: at once
        // the escape switch: every kernel of a pass launches at once again
        CK(chv_debug_set_switch("CHV_PASS_FUSE", "0"));
        l0 = stubhip_launches();
        CK(chv_pass_begin(c));
        CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
        for (int l = 0; l < 4; l++) CK(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &us[l], sizeof us[l], 1, nullptr));
        EXPECT(stubhip_launches() == l0 + 5);
        CK(chv_pass_end(c, 1));
        CK(chv_debug_set_switch("CHV_PASS_FUSE", nullptr));
    }
    // bad arguments: codes, not crashes
    EXPECT(chv_composite(c, nullptr, 1, vids, 4) != CHV_OK);
    EXPECT(chv_composite(c, &canvas420.img, 1, vids, 4) != CHV_OK);                       // BGRA-target kernels on a 4:2:0 canvas
    EXPECT(chv_run_kernel(c, CHV_K_IMG_NV12_BGRA, &canvas.img, &nv.img, 1, &u, 17, 1, nullptr) != CHV_OK);
    EXPECT(chv_run_kernel(c, CHV_K_IMG_CLEAR_YUVS, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr) != CHV_OK);
    EXPECT(chv_upload(c, nv.buf, (size_t)1 << 40, W, host.data(), W, W, 1, 0) != CHV_OK);
    EXPECT(chv_scale_lanczos_batch(c, ds.data(), ss.data(), 0) != CHV_OK);
    chv_image bad = canvas.img; bad.planes[0].pitch = 8;
    EXPECT(chv_composite(c, &bad, 1, vids, 4) != CHV_OK);
    CK(chv_pass_end(c, 1));
    CK(chv_host_free(c, pinned));
    for (Pic *p : { &nv, &yp, &rgb, &canvas, &canvas420, &small }) CK(chv_buffer_free(p->buf));
    CK(chv_buffer_free(a0)); CK(chv_buffer_free(a1));
}

// what the Swift host's threads do: a mixer per thread on a context of its own (mix.video.swift:55,99), an uploader and a downloader on theirs
// (compute.swift:177,234), buffers released from whatever thread drops the last reference (compute.cl.swift:55-57)
static void many_threads(chv_context *parent, int n_threads) {
    std::vector<std::thread> th;
    std::vector<std::atomic<chv_buffer *>> orphans((size_t)n_threads * 8);
    for (auto &o : orphans) o.store(nullptr);
    std::atomic<int> ready{0};
    for (int t = 0; t < n_threads; t++) th.emplace_back([&, t] {
        chv_context *c = nullptr, *up = nullptr;
        CK(chv_context_share(parent, &c)); CK(chv_context_share(parent, &up));
        const int W = 128, H = 32;
        std::vector<uint8_t> frame((size_t)W * H * 3 / 2, (uint8_t)t);
        void *pinned = nullptr;
        CK(chv_host_alloc(c, (size_t)W * H * 4, &pinned));
        chv_event *ev = nullptr;
        CK(chv_event_create(c, &ev));
        for (int it = 0; it < 60; it++) {
            Pic src = make_pic(up, CHV_FMT_NV12, W, H), canvas = make_pic(c, CHV_FMT_BGRA, W, H), ov = make_pic(c, CHV_FMT_BGRA, W, H);
            CK(chv_upload(up, src.buf, 0, W, frame.data(), W, W, (size_t)H * 3 / 2, 1));          // the upload barrier's context
            chv_layer ls[2] = { layer_of(CHV_K_IMG_NV12_BGRA, src, 1.f), layer_of(CHV_K_IMG_BGRA_BGRA_TX, ov, .5f) };
            CK(chv_pass_begin(c));
            bool src_freed = false;
            if (it % 3 == 2 && it >= 8) {
                // the unchanged mixer sequence (held, launched by whatever touches the stream next: here the event record), its source picture
                // dropped before the pass ends
                CK(chv_run_kernel(c, CHV_K_IMG_CLEAR_BGRA, &canvas.img, nullptr, 0, nullptr, 0, 0, nullptr));
                for (int l = 0; l < ((it & 1) ? 2 : 1); l++) CK(chv_run_kernel(c, ls[l].kernel, &canvas.img, &ls[l].image, 1, &ls[l].uniforms, sizeof ls[l].uniforms, 1, nullptr));
                CK(chv_buffer_free(src.buf)); src_freed = true;
            } else
                CK(chv_composite(c, &canvas.img, 1, ls, (it & 1) ? 2 : 1));                        // waits for the upload on its own stream
            CK(chv_event_record(c, ev));
            CK(chv_pass_end(c, 0));
            CK(chv_event_wait(up, ev));                                                            // the download barrier's side
            CK(chv_download_async(up, pinned, (size_t)W * 4, canvas.buf, 0, canvas.img.planes[0].pitch, (size_t)W * 4, H));
            if (it % 7 == 0) CK(chv_pass_end(up, 1));
            // hand some buffers to another thread to free, free the rest here — with work still queued behind them
            if (it < 8) { orphans[(size_t)t * 8 + it].store(src.buf, std::memory_order_release); ready++; } else if (!src_freed) CK(chv_buffer_free(src.buf));
            CK(chv_buffer_free(canvas.buf)); CK(chv_buffer_free(ov.buf));
        }
        CK(chv_pass_end(c, 1)); CK(chv_pass_end(up, 1));
        CK(chv_event_destroy(ev));
        CK(chv_host_free(c, pinned));
        CK(chv_context_destroy(up)); CK(chv_context_destroy(c));
    });
    std::thread reaper([&] {
        size_t freed = 0;
        while (freed < orphans.size()) {
            for (auto &b : orphans) { chv_buffer *p = b.exchange(nullptr, std::memory_order_acq_rel); if (p) { CK(chv_buffer_free(p)); freed++; } }
            std::this_thread::yield();
        }
    });
    for (auto &t : th) t.join();
    reaper.join();
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8;
    int n = 0;
    CK(chv_device_count(&n));
    EXPECT(n >= 2);
    chv_device_info info;
    CK(chv_device_info_get(1, &info));
    chv_context *c0 = nullptr, *c1 = nullptr, *bad = nullptr;
    CK(chv_context_create(0, &c0)); CK(chv_context_create(1, &c1));
    EXPECT(chv_context_create(99, &bad) != CHV_OK);
    single_thread(c0);
    single_thread(c1);                          // (a second device: every entry sets its device first)
    many_threads(c0, threads);
    // a buffer of device 0 handed to a context of device 1: refused, not dereferenced
    Pic p0 = make_pic(c0, CHV_FMT_BGRA, 64, 32), p1 = make_pic(c1, CHV_FMT_BGRA, 64, 32);
    chv_layer l = layer_of(CHV_K_IMG_BGRA_BGRA_TX, p0, 1.f);
    EXPECT(chv_composite(c1, &p1.img, 1, &l, 1) != CHV_OK);
    CK(chv_buffer_free(p0.buf)); CK(chv_buffer_free(p1.buf));
    CK(chv_context_destroy(c0)); CK(chv_context_destroy(c1));
    EXPECT(chv_context_destroy(nullptr) != CHV_OK);
    printf("abi_stress: ok (%d threads)\n", threads);
    return 0;
}
