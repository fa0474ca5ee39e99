// tools/tick_threads.cpp — the one-tick-at-a-time path of a host with SEVERAL mixers on one device, from native threads over the C ABI
// (what a Swift composer's mixer queues do, composer.swift:203-224, mix.video.swift:55,99): T threads, each with its own context
// (chv_context_share), four 1080p NV12 sources and a ring of ten 720p BGRA canvases, issue the headline tick and wait for it, for a fixed
// time; ticks per second for T = 1, 2, 4, 8.  Modes: fused (one chv_composite + chv_pass_end(wait)) and sequence (img_clear_bgra +
// 4 x chv_run_kernel + chv_pass_end(wait): an unchanged mix.video.swift:116-124).  bench.py runs it beside its own Python threads so that
// the Python host's interpreter lock can be told apart from what the HIP runtime serialises (the leg `per_tick_thread_scaling`).
//
//   tick_threads <uniforms.bin: 4 x 236 bytes> <seconds per point> [device]
//
// Prints one JSON object.  Build: g++ -std=c++17 -O2 -pthread tools/tick_threads.cpp -Iinclude -Lswiftvideo_amd -lchipvideo
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "chipvideo.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: %s (%s)\n", #x, chv_error_string(rc_), chv_last_error_detail()); exit(2); } } while (0)

static const int SW = 1920, SH = 1080, DW = 1280, DH = 720, RING = 10, LAYERS = 4;

struct Mixer {
    chv_context *ctx = nullptr;
    chv_buffer *src[LAYERS] = {}, *canvas[RING] = {};
    chv_image simg[LAYERS], cimg[RING];
    chv_layer layers[LAYERS][LAYERS];       // tick t: sources rotate
    chv_uniforms uni[LAYERS];
    chv_kernel_opts opts;
    long n = 0;

    void init(chv_context *parent, const chv_uniforms *u) {
        CK(chv_context_share(parent, &ctx));
        memcpy(uni, u, sizeof uni);
        memset(&opts, 0, sizeof opts);
        opts.colorspace = CHV_CSC_BT601_LIMITED;
        std::vector<unsigned char> host((size_t)SW * SH * 3 / 2);
        for (int l = 0; l < LAYERS; l++) {
            // luma and chroma in ONE allocation, planes adjacent (what the hosts of this repository do)
            CK(chv_buffer_alloc(ctx, (size_t)SW * SH * 3 / 2, &src[l]));
            for (size_t i = 0; i < host.size(); i++) host[i] = (unsigned char)((i * 2654435761u + l * 97u) >> 13);
            CK(chv_upload(ctx, src[l], 0, SW, host.data(), SW, SW, (size_t)SH * 3 / 2, 0));
            memset(&simg[l], 0, sizeof(chv_image));
            simg[l].format = CHV_FMT_NV12; simg[l].width = SW; simg[l].height = SH; simg[l].n_planes = 2;
            simg[l].planes[0] = chv_plane{ src[l], 0, SW, SH, SW, 1 };
            simg[l].planes[1] = chv_plane{ src[l], (size_t)SW * SH, SW / 2, SH / 2, SW, 2 };
        }
        for (int r = 0; r < RING; r++) {
            size_t pitch = 0;
            CK(chv_plane_alloc(ctx, DW, DH, 4, &canvas[r], &pitch));
            memset(&cimg[r], 0, sizeof(chv_image));
            cimg[r].format = CHV_FMT_BGRA; cimg[r].width = DW; cimg[r].height = DH; cimg[r].n_planes = 1;
            cimg[r].planes[0] = chv_plane{ canvas[r], 0, DW, DH, (int32_t)pitch, 4 };
        }
        for (int t = 0; t < LAYERS; t++)
            for (int l = 0; l < LAYERS; l++) {
                memset(&layers[t][l], 0, sizeof(chv_layer));
                layers[t][l].kernel = CHV_K_IMG_NV12_BGRA;
                layers[t][l].image = simg[(t + l) % LAYERS];
                layers[t][l].uniforms = uni[l];
                layers[t][l].opts = opts;
            }
    }
    void tick_fused() {
        const long t = n++;
        chv_pass_begin(ctx);
        CK(chv_composite(ctx, &cimg[t % RING], 1, layers[t % LAYERS], LAYERS));
        CK(chv_pass_end(ctx, 1));
    }
    void tick_sequence() {
        const long t = n++;
        chv_pass_begin(ctx);
        CK(chv_run_kernel(ctx, CHV_K_IMG_CLEAR_BGRA, &cimg[t % RING], nullptr, 0, nullptr, 0, 0, nullptr));
        for (int l = 0; l < LAYERS; l++)
            CK(chv_run_kernel(ctx, CHV_K_IMG_NV12_BGRA, &cimg[t % RING], &simg[(t + l) % LAYERS], 1, &uni[l], sizeof(chv_uniforms), 1, &opts));
        CK(chv_pass_end(ctx, 1));
    }
    void destroy() {
        for (auto b : src) chv_buffer_free(b);
        for (auto b : canvas) chv_buffer_free(b);
        chv_context_destroy(ctx);
    }
};

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: tick_threads <uniforms.bin> <seconds> [device]\n"); return 1; }
    const double seconds = atof(argv[2]);
    const int device = argc > 3 ? atoi(argv[3]) : 0;
    chv_uniforms u[LAYERS];
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(u, sizeof(chv_uniforms), LAYERS, f) != (size_t)LAYERS) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);
    chv_context *root = nullptr;
    CK(chv_context_create(device, &root));
    const int counts[4] = { 1, 2, 4, 8 };
    std::vector<Mixer> mixers(8);
    for (auto &m : mixers) m.init(root, u);
    printf("{");
    for (int mode = 0; mode < 2; mode++) {
        printf("%s\"%s\": {", mode ? ", " : "", mode ? "sequence" : "fused");
        for (int ci = 0; ci < 4; ci++) {
            const int T = counts[ci];
            for (int i = 0; i < T; i++) for (int k = 0; k < 20; k++) { if (mode) mixers[i].tick_sequence(); else mixers[i].tick_fused(); }
            std::atomic<int> go{0};
            std::vector<long> done(T, 0);
            std::vector<std::thread> th;
            const auto t0 = std::chrono::steady_clock::now();
            const auto until = t0 + std::chrono::duration<double>(seconds);
            for (int i = 0; i < T; i++)
                th.emplace_back([&, i] {
                    while (!go.load()) { }
                    long k = 0;
                    while (std::chrono::steady_clock::now() < until) { if (mode) mixers[i].tick_sequence(); else mixers[i].tick_fused(); k++; }
                    done[i] = k;
                });
            go.store(1);
            for (auto &t : th) t.join();
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            long total = 0;
            for (long k : done) total += k;
            printf("%s\"%d\": %.1f", ci ? ", " : "", T, total / el);
        }
        printf("}");
    }
    printf("}\n");
    for (auto &m : mixers) m.destroy();
    chv_context_destroy(root);
    return 0;
}
