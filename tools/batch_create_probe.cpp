// tools/batch_create_probe.cpp — what a host that groups its mixers (VideoMixerGroup: one launch for N mixers' ticks) pays per GROUP TICK when it
// builds the batch for that tick, runs it once and frees it — the real use: every tick has new pictures — against running one prebuilt batch again
// (what bench.py times): chv_batch_create / chv_batch_run / chv_pass_end(wait) / chv_batch_destroy in microseconds, for groups of 8, 64 and 256
// headline ticks (4 x 1080p NV12 -> 720p BGRA).  Native host over the C ABI.
//   g++ -std=c++17 -O2 tools/batch_create_probe.cpp -Iinclude -Lswiftvideo_amd -lchipvideo -Wl,-rpath,$PWD/swiftvideo_amd -o tools/batch_create_probe.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "chipvideo.h"
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: %s (%s)\n", #x, chv_error_string(rc_), chv_last_error_detail()); exit(2); } } while (0)
static const int SW = 1920, SH = 1080, DW = 1280, DH = 720, LAYERS = 4, NSRC = 8, NCAN = 256;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const bool json = argc > 1 && !strcmp(argv[1], "--json");          // (bench.py --full: the leg group_tick_built_fresh)
    const bool mixer = argc > 1 && !strcmp(argv[1], "--mixer");        // the reference-default mixer tick instead (1080p y420p canvas <- video + 2 BGRA overlays: strip kernel)
    const int device = argc > 2 ? atoi(argv[2]) : 0;
    chv_context *ctx = nullptr;
    CK(chv_context_create(device, &ctx));
    if (json) printf("{");
    bool first_group = true;
    chv_uniforms u;
    memset(&u, 0, sizeof u);
    const float rows[16] = { .5f, 0, 0, .5f, 0, .5f, 0, .5f, 0, 0, 1, -1, 0, 0, 0, 1 };
    memcpy(u.transform, rows, sizeof rows); memcpy(u.border_matrix, rows, sizeof rows);
    for (int i = 0; i < 4; i++) u.texture_transform[5 * i] = 1.f;
    u.input_size[0] = SW; u.input_size[1] = SH; u.output_size[0] = DW; u.output_size[1] = DH;
    chv_kernel_opts opts; memset(&opts, 0, sizeof opts); opts.colorspace = CHV_CSC_BT601_LIMITED;
    std::vector<chv_image> simg(NSRC), cimg(NCAN);
    std::vector<unsigned char> host((size_t)SW * SH * 3 / 2);
    for (int l = 0; l < NSRC; l++) {
        chv_buffer *b = nullptr;
        CK(chv_buffer_alloc(ctx, host.size(), &b));
        for (size_t i = 0; i < host.size(); i++) host[i] = (unsigned char)((i * 2654435761u + l * 97u) >> 13);
        CK(chv_upload(ctx, b, 0, SW, host.data(), SW, SW, (size_t)SH * 3 / 2, 0));
        memset(&simg[l], 0, sizeof(chv_image));
        simg[l].format = CHV_FMT_NV12; simg[l].width = SW; simg[l].height = SH; simg[l].n_planes = 2;
        simg[l].planes[0] = chv_plane{ b, 0, SW, SH, SW, 1 };
        simg[l].planes[1] = chv_plane{ b, (size_t)SW * SH, SW / 2, SH / 2, SW, 2 };
    }
    for (int r = 0; r < NCAN; r++) {
        chv_buffer *b = nullptr; size_t pitch = 0;
        CK(chv_plane_alloc(ctx, DW, DH, 4, &b, &pitch));
        memset(&cimg[r], 0, sizeof(chv_image));
        cimg[r].format = CHV_FMT_BGRA; cimg[r].width = DW; cimg[r].height = DH; cimg[r].n_planes = 1;
        cimg[r].planes[0] = chv_plane{ b, 0, DW, DH, (int32_t)pitch, 4 };
    }
    const float op[4] = { 1.f, .75f, .5f, .25f };
    // --mixer: y420p canvases and sources, two 640x360 BGRA overlays
    std::vector<chv_image> ysrc(NSRC), ycan(mixer ? 128 : 0), ov(2);
    chv_uniforms uy = u, uo[2];
    if (mixer) {
        uy.input_size[0] = 1920; uy.input_size[1] = 1080; uy.output_size[0] = 1920; uy.output_size[1] = 1080;
        auto y420 = [&](chv_image &im) {
            chv_buffer *b = nullptr;
            CK(chv_buffer_alloc(ctx, (size_t)1920 * 1080 * 3 / 2, &b));
            memset(&im, 0, sizeof im);
            im.format = CHV_FMT_Y420P; im.width = 1920; im.height = 1080; im.n_planes = 3;
            im.planes[0] = chv_plane{ b, 0, 1920, 1080, 1920, 1 };
            im.planes[1] = chv_plane{ b, (size_t)1920 * 1080, 960, 540, 960, 1 };
            im.planes[2] = chv_plane{ b, (size_t)1920 * 1080 * 5 / 4, 960, 540, 960, 1 };
        };
        for (auto &im : ysrc) y420(im);
        for (auto &im : ycan) y420(im);
        for (int k = 0; k < 2; k++) {
            chv_buffer *b = nullptr; size_t pitch = 0;
            CK(chv_plane_alloc(ctx, 640, 360, 4, &b, &pitch));
            memset(&ov[k], 0, sizeof(chv_image));
            ov[k].format = CHV_FMT_BGRA; ov[k].width = 640; ov[k].height = 360; ov[k].n_planes = 1;
            ov[k].planes[0] = chv_plane{ b, 0, 640, 360, (int32_t)pitch, 4 };
            // a 640 x 360 rectangle at (px, py) of the 1920 x 1080 canvas: border / transform rows map its pixels to [0, 1]
            const float px = k ? 1200.f : 64.f, py = k ? 640.f : 64.f, sx = 1920.f / 640.f, sy = 1080.f / 360.f;
            memset(&uo[k], 0, sizeof(chv_uniforms));
            const float r[16] = { .5f * sx, 0, 0, .5f * sx - px / 640.f, 0, .5f * sy, 0, .5f * sy - py / 360.f, 0, 0, 1, -1, 0, 0, 0, 1 };
            memcpy(uo[k].transform, r, sizeof r); memcpy(uo[k].border_matrix, r, sizeof r);
            for (int i = 0; i < 4; i++) uo[k].texture_transform[5 * i] = 1.f;
            uo[k].input_size[0] = 640; uo[k].input_size[1] = 360; uo[k].output_size[0] = 1920; uo[k].output_size[1] = 1080; uo[k].opacity = k ? .6f : .8f;
        }
    }
    const int NL = mixer ? 3 : LAYERS;
    for (int G : { 8, 64, mixer ? 128 : 256 }) {
        std::vector<chv_layer> layers((size_t)G * LAYERS);
        std::vector<chv_tick> ticks(G);
        for (int rep = 0; rep < 2; rep++) {            // rep 0: warm-up
            double t_create = 0, t_run = 0, t_wait = 0, t_destroy = 0, t_rerun = 0;
            const int N = 30;
            for (int it = 0; it < N; it++) {
                for (int t = 0; t < G; t++) {
                    for (int l = 0; l < NL; l++) {
                        chv_layer &L = layers[(size_t)t * LAYERS + l];
                        memset(&L, 0, sizeof L);
                        if (!mixer) { L.kernel = CHV_K_IMG_NV12_BGRA; L.image = simg[(t + l + it) % NSRC]; L.uniforms = u; L.uniforms.opacity = op[l]; }
                        else if (l == 0) { L.kernel = CHV_K_IMG_Y420P_Y420P; L.image = ysrc[(t + it) % NSRC]; L.uniforms = uy; L.uniforms.opacity = 1.f; }
                        else { L.kernel = CHV_K_IMG_BGRA_Y420P; L.image = ov[l - 1]; L.uniforms = uo[l - 1]; }
                        L.opts = opts;
                    }
                    memset(&ticks[t], 0, sizeof(chv_tick));
                    ticks[t].target = mixer ? ycan[(t + it) % 128] : cimg[(t + it) % NCAN]; ticks[t].clear_first = 1; ticks[t].n_layers = NL; ticks[t].layers = &layers[(size_t)t * LAYERS];
                }
                chv_batch *b = nullptr;
                double a = now();
                CK(chv_batch_create(ctx, ticks.data(), G, &b));
                double c = now();
                CK(chv_pass_begin(ctx)); CK(chv_batch_run(ctx, b));
                double d = now();
                CK(chv_pass_end(ctx, 1));
                double e = now();
                CK(chv_pass_begin(ctx)); CK(chv_batch_run(ctx, b)); CK(chv_pass_end(ctx, 1));
                double f = now();
                CK(chv_batch_destroy(b));
                double g = now();
                t_create += c - a; t_run += d - c; t_wait += e - d; t_rerun += f - e; t_destroy += g - f;
            }
            if (rep && json) {
                printf("%s\"%d\": {\"create_us\": %.1f, \"run_us\": %.1f, \"wait_us\": %.1f, \"destroy_us\": %.1f, \"built_fresh_us\": %.1f, \"run_again_us\": %.1f}", first_group ? "" : ", ", G,
                       t_create / N, t_run / N, t_wait / N, t_destroy / N, (t_create + t_run + t_wait + t_destroy) / N, t_rerun / N);
                first_group = false;
            } else if (rep) printf("group of %3d ticks: create %7.1f us, run (enqueue) %6.1f, wait %7.1f, destroy %6.1f  => %7.1f us per group tick built fresh; the same batch run again + wait: %7.1f us\n",
                            G, t_create / N, t_run / N, t_wait / N, t_destroy / N, (t_create + t_run + t_wait + t_destroy) / N, t_rerun / N);
        }
    }
    if (json) printf("}\n");
    return 0;
}
