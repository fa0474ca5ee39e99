#!/usr/bin/env python3
"""bench.py — throughput of the picture hot path on MI355X.

Workload at every N (weak scaling, one process per GPU, no collective on the data
path): BASELINE.json configs[1] — 1920x1080 NV12 -> BGRA (integer BT.601) with
bilinear downscale to 1280x720 — over a batch of `--frames` distinct device-resident
frames per GPU (the per-device PictureSample buses of configs[3]); one step = one
pass of the path over the batch = one chv_batch_run launch.

Prints ONE JSON line on rank 0 (see the contract in the task description):
  value     = target pixels written per second, whole job, inputs resident in HBM
  roofline  = algorithmic bytes per launch / mean launch duration (HIP events on the
              context's stream) against the 8 TB/s HBM peak
  cpu_baseline = the oracle (CPU restatement of the same kernels, "port") timed on the
              host cores of this box on a bounded sample of the same workload (N=1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

WORKLOADS = {
    # name: (src fmt, src w, h, dst w, h, n_layers, algorithmic bytes per tick)
    "cfg2": dict(desc="1920x1080 NV12 -> BGRA (BT.601 int) + bilinear downscale to 1280x720",
                 sw=1920, sh=1080, dw=1280, dh=720, layers=1, bytes=3110400 + 3686400),
    "cfg2_y420p": dict(desc="cfg2 with a planar source: 1920x1080 y420p -> BGRA (BT.601 int) + bilinear downscale to 1280x720",
                 sw=1920, sh=1080, dw=1280, dh=720, layers=1, bytes=3110400 + 3686400, src="y420p"),
    "cfg3": dict(desc="4 x 1080p BGRA layers (opacity 1/.75/.5/.25) alpha-composited onto a 1080p BGRA canvas",
                 sw=1920, sh=1080, dw=1920, dh=1080, layers=4, bytes=4 * 8294400 + 8294400),
    "mixer_y420p": dict(desc="reference-default canvas: 1080p y420p canvas <- full-canvas 1080p y420p layer + two 640x360 BGRA overlays (opacity .8/.6)",
                 sw=1920, sh=1080, dw=1920, dh=1080, layers=3, bytes=3110400 + 3110400 + 2 * 921600, mixer="y420p"),
    "cfg5": dict(desc="8 x 3840x2160 BGRA layers composited onto a 2160p canvas, then Lanczos-3 down to 1920x1080",
                 sw=3840, sh=2160, dw=3840, dh=2160, layers=8, bytes=8 * 33177600 + 8294400, lanczos=(1920, 1080)),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=256, help="distinct frames (ticks) per launch per GPU")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--alias", default="none", choices=("none", "src", "dst", "both"),
                    help="DIAGNOSTIC (cfg2 only): every tick reads frame 0's source and/or writes frame 0's canvas, so that "
                         "side of the traffic stays in cache; the line is marked and is not a benchmark result")
    ap.add_argument("--device", type=int, default=None,
                    help="device index for every rank (default: LOCAL_RANK); lets the N>1 path be exercised on a 1-GPU box")
    ap.add_argument("--with-upload", action="store_true",
                    help="end-to-end mode: every step also uploads its source frames from pinned host memory on a side "
                         "stream (PCIe-inclusive rate; reported for DESIGN.md, never the headline value)")
    return ap.parse_args()


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        # control plane only (barrier, max over ranks): gloo on the host; the pixel path has no collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist = dist_mod
    return rank, local, world, dist


def reduce_max(dist, value):
    """MAX over ranks of a host float (gloo all-reduce); identity when not distributed."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def whole_job_gpix(n_gpus, px_per_step_per_gpu, steps, elapsed_max):
    """Weak scaling: every rank processes the same per-GPU batch; value = all pixels / slowest rank's time."""
    return n_gpus * px_per_step_per_gpu * steps / elapsed_max / 1e9


def stream_to_device(stream_id, n_gpus):
    """configs[3]: picture bus s is bound to device s mod n_gpus."""
    return stream_id % n_gpus


def build_workload(sv, ctx, wl, frames, seed_base, alias="none"):
    """Device-resident source frames, canvases and the batch descriptor."""
    import util
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    sw, sh, dw, dh = wl["sw"], wl["sh"], wl["dw"], wl["dh"]
    distinct = 4
    host_src = []
    keep = []  # keep PictureSamples alive (they own the device memory)
    ticks = (cv.Tick * frames)()
    layer_arrays = []
    lanczos_pairs = []
    if "mixer" in wl:
        fmt = wl["mixer"]
        pf = sv.PixelFormat.y420p if fmt == "y420p" else sv.PixelFormat.nv12
        for i in range(distinct):
            host_src.append(util.alloc_image(fmt, sw, sh, seed=seed_base + i))
        ov = [util.alloc_image("bgra", 640, 360, seed=seed_base + 100 + i) for i in range(2)]
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh)),
              util.make_uniforms((dw, dh), rect=(64, 64, 640, 360), opacity=0.8, in_size=(640, 360)),
              util.make_uniforms((dw, dh), rect=(1200, 640, 640, 360), opacity=0.6, in_size=(640, 360))]
        k_main = sv.defaultComputeKernelFromString(f"img_{fmt}_{fmt}")
        k_ov = sv.defaultComputeKernelFromString(f"img_bgra_{fmt}")
        govs = [sv.uploadComputePicture(ctx, sv.pictureFromArrays(sv.PixelFormat.BGRA, (640, 360), o), retainCpuBuffer=False) for o in ov]
        keep += govs
        for f in range(frames):
            src = sv.uploadComputePicture(ctx, sv.pictureFromArrays(pf, (sw, sh), host_src[f % distinct]), retainCpuBuffer=False)
            dst = sv.uploadComputePicture(ctx, sv.createPictureSample((dw, dh), pf), retainCpuBuffer=False)
            keep += [src, dst]
            arr = sv._layer_array([(k_main, src, us[0], 0), (k_ov, govs[0], us[1], 0), (k_ov, govs[1], us[2], 0)])
            layer_arrays.append(arr)
            ticks[f].target = sv._image_desc(dst)
            ticks[f].clear_first = 1
            ticks[f].n_layers = 3
            ticks[f].layers = arr
        verify = None
    elif wl["layers"] == 1:
        sfmt = wl.get("src", "nv12")
        spf = sv.PixelFormat.y420p if sfmt == "y420p" else sv.PixelFormat.nv12
        skernel = sv.ComputeKernel.img_y420p_bgra if sfmt == "y420p" else sv.ComputeKernel.img_nv12_bgra
        for i in range(distinct):
            host_src.append(util.alloc_image(sfmt, sw, sh, seed=seed_base + i))
        u = util.full_canvas_uniforms((dw, dh), (sw, sh))
        for f in range(frames):
            src = sv.uploadComputePicture(ctx, sv.pictureFromArrays(spf, (sw, sh), host_src[f % distinct]),
                                          retainCpuBuffer=False)
            dst = sv.uploadComputePicture(ctx, sv.createPictureSample((dw, dh), sv.PixelFormat.BGRA), retainCpuBuffer=False)
            keep += [src, dst]
            if f > 0 and alias in ("src", "both"):
                src = keep[0]
            if f > 0 and alias in ("dst", "both"):
                dst = keep[1]
            arr = sv._layer_array([(skernel, src, u, cv.CSC_BT601_LIMITED)])
            layer_arrays.append(arr)
            ticks[f].target = sv._image_desc(dst)
            ticks[f].clear_first = 1
            ticks[f].n_layers = 1
            ticks[f].layers = arr
        verify = (f"img_{sfmt}_bgra", host_src, [u])
    else:
        nl = wl["layers"]
        for i in range(distinct):
            host_src.append(util.alloc_image("bgra", sw, sh, seed=seed_base + i))
        ops = (1.0, 0.75, 0.5, 0.25) if nl <= 4 else (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3)
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=o) for o in ops[:nl]]
        for f in range(frames):
            layers = []
            for l in range(nl):
                src = sv.uploadComputePicture(ctx, sv.pictureFromArrays(sv.PixelFormat.BGRA, (sw, sh), host_src[(f + l) % distinct]),
                                              retainCpuBuffer=False)
                keep.append(src)
                layers.append((sv.ComputeKernel.img_bgra_bgra_tx, src, us[l], 0))
            dst = sv.uploadComputePicture(ctx, sv.createPictureSample((dw, dh), sv.PixelFormat.BGRA), retainCpuBuffer=False)
            keep.append(dst)
            if "lanczos" in wl:
                small = sv.uploadComputePicture(ctx, sv.createPictureSample(wl["lanczos"], sv.PixelFormat.BGRA), retainCpuBuffer=False)
                lanczos_pairs.append((small, dst))
            arr = sv._layer_array(layers)
            layer_arrays.append(arr)
            ticks[f].target = sv._image_desc(dst)
            ticks[f].clear_first = 1
            ticks[f].n_layers = nl
            ticks[f].layers = arr
        verify = ("img_bgra_bgra_tx", host_src, us)
    batch = C.c_void_p()
    cv.check(lib.chv_batch_create(ctx.handle, ticks, frames, C.byref(batch)))
    name = C.create_string_buffer(128)
    cv.check(lib.chv_batch_describe(batch, name, 128, None))
    return dict(batch=batch, keep=keep, layer_arrays=layer_arrays, ticks=ticks, kernel=name.value.decode(), verify=verify,
                lanczos=lanczos_pairs)


def verify_frame(sv, ctx, wl, w, frame=0):
    """Frame `frame` of the batch output == oracle (outside any timed region)."""
    import util
    from oracle import oracle as O
    kernel, host_src, us = w["verify"]
    dw, dh = wl["dw"], wl["dh"]
    exp = util.alloc_image("bgra", dw, dh)
    assert O.run_kernel("img_clear_bgra", exp, threads=os.cpu_count()) == 0
    distinct = len(host_src)
    for l, u in enumerate(us):
        src = host_src[(frame + l) % distinct] if len(us) > 1 else host_src[frame % distinct]
        assert O.run_kernel(kernel, exp, src, u, threads=os.cpu_count()) == 0
    per_frame = 1 + len(us) if len(us) > 1 else 2
    dst = w["keep"][frame * per_frame + per_frame - 1]
    got = sv.downloadComputePicture(ctx, dst, retainGpuBuffer=True).imageBuffer().buffers[0]
    return bool(np.array_equal(got[:, : dw * 4].reshape(dh, dw, 4), exp[0]))


def cpu_baseline(wl, w, budget_s):
    """The oracle on this box's host cores, same kernels/uniforms, bounded sample.
    Stream-parallel like the GPU path: every worker thread composites whole ticks on its
    own canvas (ctypes releases the GIL), so no per-call thread start-up is measured."""
    import threading
    import util
    from oracle import oracle as O
    kernel, host_src, us = w["verify"]
    dw, dh = wl["dw"], wl["dh"]
    cores = os.cpu_count() or 1
    O.lib()
    counts = [0] * cores
    t_end = [0.0]

    def worker(i):
        canvas = util.alloc_image("bgra", dw, dh)
        n = 0
        while time.perf_counter() < t_end[0]:
            O.run_kernel("img_clear_bgra", canvas, threads=1)
            for l, u in enumerate(us):
                O.run_kernel(kernel, canvas, host_src[(n + l) % len(host_src)], u, threads=1)
            n += 1
        counts[i] = n

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    t_end[0] = t0 + budget_s
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    el = time.perf_counter() - t0
    n = sum(counts)
    gpix = n * dw * dh / el / 1e9
    return {"value": gpix, "unit": "Gpix/s", "cores": cores, "kind": "port",
            "sample": f"{n} ticks of the same workload in {el:.1f} s: oracle/ref_kernels.c, {cores} threads each "
                      f"compositing whole ticks (clear + {len(us)} layer kernel(s) per tick, as the reference issues them)"}


def run_with_upload(args, sv, cv, lib, ctx, wl, rank, n_gpus, dist):
    """PCIe-inclusive pipeline for cfg2: two frame sets; while set A is converted on the compute
    context's stream, set B's NV12 planes are uploaded (hipMemcpy2DAsync from pinned memory) on a
    sharing context's stream.  Ordering: per-buffer upload events (kernel waits for its inputs) and a
    per-set 'batch done' event (the next upload into the set waits for the kernel that read it)."""
    import util
    assert wl["layers"] == 1, "--with-upload is implemented for the convert+scale workload"
    up = sv.createComputeContext(sharing=ctx)
    sw, sh = wl["sw"], wl["sh"]
    ysz, csz = sw * sh, sw * sh // 2
    distinct = 4
    pinned = C.c_void_p()
    cv.check(lib.chv_host_alloc(up.handle, distinct * (ysz + csz), C.byref(pinned)))
    host = np.ctypeslib.as_array((C.c_uint8 * (distinct * (ysz + csz))).from_address(pinned.value))
    for i in range(distinct):
        img = util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 32 + i)
        host[i * (ysz + csz): i * (ysz + csz) + ysz] = img[0].reshape(-1)
        host[i * (ysz + csz) + ysz: (i + 1) * (ysz + csz)] = img[1].reshape(-1)
    sets = [build_workload(sv, ctx, wl, args.frames, seed_base=0x5EED0000 + 32) for _ in range(2)]
    done = []
    for _ in sets:
        e = C.c_void_p()
        cv.check(lib.chv_event_create(ctx.handle, C.byref(e)))
        cv.check(lib.chv_event_record(ctx.handle, e))
        done.append(e)

    def upload_set(k):
        cv.check(lib.chv_event_wait(up.handle, done[k]))          # the kernel that last read this set is finished
        keep = sets[k]["keep"]
        for f in range(args.frames):
            src = keep[2 * f].imageBuffer()
            base = pinned.value + (f % distinct) * (ysz + csz)
            # luma + interleaved chroma are adjacent with equal pitch on both sides: one pitched copy per frame
            # (3.1 MB copies reach ~48 GB/s on this link, separate 2 MB + 1 MB copies ~37 GB/s; tools/h2d_probe.py)
            assert src.gpuPitches[0] == src.gpuPitches[1] and src.gpuOffsets[1] == src.gpuPitches[0] * sh
            cv.check(lib.chv_upload(up.handle, src.computeTextures[0]._h, src.gpuOffsets[0], src.gpuPitches[0], base, sw, sw, sh + sh // 2, 2))

    def convert_set(k):
        cv.check(lib.chv_batch_run(ctx.handle, sets[k]["batch"]))  # waits for the set's upload events on its stream
        cv.check(lib.chv_event_record(ctx.handle, done[k]))

    def sync():
        cv.check(lib.chv_device_synchronize(ctx.handle))

    for w in range(args.warmup):
        upload_set(w % 2); convert_set(w % 2)
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        upload_set(k % 2); convert_set(k % 2)
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = reduce_max(dist, time.perf_counter() - t0)
    if rank == 0:
        px = args.frames * wl["dw"] * wl["dh"]
        h2d = args.frames * (ysz + csz) * args.steps / elapsed / 1e9
        print(json.dumps({
            "metric": "Gpix/s + achieved HBM GB/s, 1080p NV12→BGRA+scale+4-layer composite, 1/2/4/8 GPU",
            "value": whole_job_gpix(n_gpus, px, args.steps, elapsed), "unit": "Gpix/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "mode": "END-TO-END incl. H2D upload of every source frame "
                       "from pinned host memory on a side stream (not the headline mode)", "frames_per_step_per_gpu": args.frames,
                       "h2d_GBps_per_gpu": h2d, "kernel": sets[0]["kernel"]}}), flush=True)
    cv.check(lib.chv_host_free(up.handle, pinned))


def main():
    args = parse_args()
    rank, local, world, dist = dist_setup(args.gpus)
    if world != args.gpus and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    n_gpus = max(world, 1)

    from swiftvideo_amd import chipvideo as cv
    from swiftvideo_amd import compute as sv
    lib = cv.load()
    ctx = sv.makeComputeContext(forType="GPU", index=local if args.device is None else args.device)
    wl = WORKLOADS[args.workload]
    if args.with_upload:
        run_with_upload(args, sv, cv, lib, ctx, wl, rank, n_gpus, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    w = build_workload(sv, ctx, wl, args.frames, seed_base=0x5EED0000 + 16 * 2 + rank, alias=args.alias)

    def step():
        cv.check(lib.chv_batch_run(ctx.handle, w["batch"]))
        for small, big in w["lanczos"]:
            sv.scaleLanczos(ctx, small, big)

    def sync():
        cv.check(lib.chv_device_synchronize(ctx.handle))
        try:
            import torch
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
        except Exception:
            pass

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    verified = None
    if not args.no_verify and rank == 0 and w["verify"] is not None:
        verified = verify_frame(sv, ctx, wl, w, frame=0) and verify_frame(sv, ctx, wl, w, frame=args.frames - 1)

    # per-launch HIP events on the stream the kernels run on
    evs = []
    for _ in range(args.steps + 1):
        e = C.c_void_p()
        cv.check(lib.chv_event_create(ctx.handle, C.byref(e)))
        evs.append(e)

    barrier()
    sync()
    t0 = time.perf_counter()
    cv.check(lib.chv_event_record(ctx.handle, evs[0]))
    for k in range(args.steps):
        step()
        cv.check(lib.chv_event_record(ctx.handle, evs[k + 1]))
    sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = reduce_max(dist, t1 - t0)

    durs = []
    for k in range(args.steps):
        ms = C.c_float()
        cv.check(lib.chv_event_elapsed_ms(evs[k], evs[k + 1], C.byref(ms)))
        durs.append(ms.value)
    launch_ms = float(np.mean(durs))

    if rank == 0:
        px_per_step = args.frames * wl["dw"] * wl["dh"]
        value = whole_job_gpix(n_gpus, px_per_step, args.steps, elapsed)
        bytes_per_launch = args.frames * wl["bytes"]
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pmc_latest.json"
        if pmc.exists():
            try:
                j = json.loads(pmc.read_text())
                if j.get("workload") == args.workload and j.get("frames") == args.frames:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Gpix/s + achieved HBM GB/s, 1080p NV12→BGRA+scale+4-layer composite, 1/2/4/8 GPU",
            "value": value, "unit": "Gpix/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "frames_per_step_per_gpu": args.frames,
                       "gpix_counts": "target pixels written", "source_mpix_per_step_per_gpu": args.frames * wl["sw"] * wl["sh"] * wl["layers"] / 1e6,
                       "parallelism": f"{n_gpus} independent per-device picture buses, no collective",
                       "kernel": w["kernel"], "verified_vs_oracle": verified},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": w["kernel"], "launch_ms": launch_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if args.alias != "none":
            out["data"] = f"DIAGNOSTIC --alias {args.alias}: ticks share frame 0's buffers, cache-resident traffic; not a benchmark result"
            out["roofline"]["frac"] = None
        if n_gpus == 1 and not args.no_cpu_baseline and w["verify"] is not None:
            out["cpu_baseline"] = cpu_baseline(wl, w, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
