#!/usr/bin/env python3
"""bench.py — throughput of the picture hot path on MI355X.

Headline workload ("pipeline", the literal metric of BASELINE.json: NV12 -> BGRA + scale + 4-layer composite):
one mixer tick = four distinct 1920x1080 NV12 streams, converted (integer BT.601), bilinearly scaled to 1280x720
and alpha-composited (opacities 1 / .75 / .5 / .25, z order 0..3) onto one cleared 720p BGRA canvas
(mix.video.swift:114-124 with findKernel -> img_nv12_bgra).  `--frames` such ticks — each with its own four
device-resident source frames and its own canvas, i.e. the ticks of `--frames` independent PictureSample buses —
form one batch = one chv_batch_run launch.

One step = `launches_per_step` passes of the path over the batch.  The count is calibrated after warm-up so that the
timed region lasts >= --min-seconds whatever --steps is (a 256-tick launch is ~1 ms); the K steps are timed exactly as
the contract says (barrier + device sync on both sides, max over ranks).

The other BASELINE configs (cfg2 convert+scale, cfg3 4 x BGRA composite, cfg5 4K 8-layer + Lanczos, the reference's own 4:2:0 mixer canvas, the
encoder-side frame, the planar / grid / logo variants of the pipeline) are timed the same way right after; `--full` adds the legs that are not
kernel throughput (cfg2 and the whole chain end to end over PCIe, one tick at a time, thread scaling, route regret, clock and power of every
workload).

Multi-GPU: weak scaling, one process per GPU, no collective on the data path (stream s -> device s mod N).  When started without a launcher
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script spawns its N ranks itself; under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE.  The control plane (barrier, max over ranks) is gloo on the host.

Output (rank 0): the LAST stdout line is ONE compact JSON object (< 4 KB, compact_line()):
  value        = target pixels written per second, whole job, inputs resident in HBM
  config       = the workload, launches and frames per step, kernel, oracle verdict, per-GPU rates, every workload's [fraction of 8 TB/s, ms per launch]
  roofline     = algorithmic bytes per launch / mean launch duration (HIP events on the context's stream) against the 8 TB/s HBM peak, the counted
                 HBM traffic of the same kernel (two rocprofv3 --pmc child passes), the limiter the round's probes name, clock and power
  cpu_baseline = the oracle (CPU restatement of the same kernels, "port") timed on the host cores of this box on a bounded sample (N=1 only)
and everything else — every workload's full record, the legs, build flags, prose — is written to bench_detail.json (--detail-json).
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0      # same guide: what a float4 copy kernel reaches
H2D_LINK_GBS = 57.0        # what pinned copies of >= 32 MiB reach on this link (tools/h2d_probe.py; PCIe Gen5 x16)
METRIC = "Gpix/s + achieved HBM GB/s, 1080p NV12→BGRA+scale+4-layer composite, 1/2/4/8 GPU"

NV12_1080 = 3110400
BGRA_720, BGRA_1080, BGRA_2160 = 3686400, 8294400, 33177600

WORKLOADS = {
    # algorithmic bytes per tick = every distinct input byte once + every output byte once
    "pipeline": dict(desc="4 x 1920x1080 NV12 streams -> BGRA (BT.601 int) + bilinear downscale to 1280x720 + 4-layer "
                          "alpha composite (opacity 1/.75/.5/.25) onto one 720p BGRA canvas, one launch per batch of ticks",
                     short="4 x 1080p NV12 -> BGRA (BT.601 int) + bilinear scale to 720p + 4-layer alpha composite, one launch per batch",
                     kind="yuv_layers", src="nv12", sw=1920, sh=1080, dw=1280, dh=720, layers=4, frames=256,
                     bytes=4 * NV12_1080 + BGRA_720),
    "pipeline_logo": dict(desc="the pipeline tick + one ROTATED 320x180 RGBA logo (opacity .9) on top: two launches per batch — the streaming kernel "
                               "for the four videos, the strip kernel for the logo (applied per pixel, in the strips it touches)",
                          kind="yuv_layers", src="nv12", sw=1920, sh=1080, dw=1280, dh=720, layers=4, frames=128, logo=True,
                          bytes=4 * NV12_1080 + BGRA_720 + 320 * 180 * 4),
    "pipeline_y420p": dict(desc="the pipeline tick with PLANAR sources (what FFmpeg's software decoders emit, dec.video.ffmpeg.swift:187-221): 4 x 1920x1080 "
                                "y420p -> BGRA + bilinear downscale to 1280x720 + 4-layer alpha composite",
                           kind="yuv_layers", src="y420p", sw=1920, sh=1080, dw=1280, dh=720, layers=4, frames=128,
                           bytes=4 * NV12_1080 + BGRA_720),
    "pipeline_grid": dict(desc="a 2 x 2 grid: one full-canvas 1080p NV12 background + four 1080p NV12 streams drawn into the 640x360 quadrants of a "
                               "720p BGRA canvas (opacity 1 / 1 / .9 / .8 / .7): five layers of two geometries' worth of DISTINCT rectangles",
                          kind="grid", sw=1920, sh=1080, dw=1280, dh=720, layers=5, frames=128,
                          bytes=5 * NV12_1080 + BGRA_720),
    "cfg2": dict(desc="1920x1080 NV12 -> BGRA (BT.601 int) + bilinear downscale to 1280x720",
                 kind="yuv_layers", src="nv12", sw=1920, sh=1080, dw=1280, dh=720, layers=1, frames=256,
                 bytes=NV12_1080 + BGRA_720),
    "cfg2_y420p": dict(desc="cfg2 with a planar source: 1920x1080 y420p -> BGRA (BT.601 int) + bilinear downscale to 1280x720",
                       kind="yuv_layers", src="y420p", sw=1920, sh=1080, dw=1280, dh=720, layers=1, frames=256,
                       bytes=NV12_1080 + BGRA_720),
    "cfg3": dict(desc="4 x 1080p BGRA layers (opacity 1/.75/.5/.25) alpha-composited onto a 1080p BGRA canvas",
                 kind="rgb_layers", sw=1920, sh=1080, dw=1920, dh=1080, layers=4, frames=128,
                 bytes=4 * BGRA_1080 + BGRA_1080),
    "mixer_y420p": dict(desc="reference-default canvas: 1080p y420p canvas <- full-canvas 1080p y420p layer + two 640x360 BGRA "
                             "overlays (opacity .8/.6)",
                        kind="mixer420", mixer="y420p", sw=1920, sh=1080, dw=1920, dh=1080, layers=3, frames=128,
                        bytes=NV12_1080 + NV12_1080 + 2 * 921600),
    "y420p_main": dict(desc="1080p y420p canvas <- one full-canvas 1080p y420p layer (the mixer workload without its overlays)",
                       kind="mixer420", mixer="y420p", overlays=0, sw=1920, sh=1080, dw=1920, dh=1080, layers=1, frames=128,
                       bytes=NV12_1080 + NV12_1080),
    "mixer_nv12": dict(desc="1080p NV12 canvas <- full-canvas 1080p NV12 layer + two 640x360 BGRA overlays (opacity .8/.6)",
                       kind="mixer420", mixer="nv12", sw=1920, sh=1080, dw=1920, dh=1080, layers=3, frames=128,
                       bytes=NV12_1080 + NV12_1080 + 2 * 921600),
    "encode_nv12": dict(desc="the encoder side: 1080p BGRA canvas -> 1080p NV12, integer BT.601 limited-range matrix (img_bgra_nv12_int, what "
                             "PictureFilter(.nv12) issues per mixed frame in front of an H.264 encoder)",
                        kind="mixer420", mixer="nv12", main_src="bgra", main_kernel="img_bgra_nv12_int", overlays=0,
                        sw=1920, sh=1080, dw=1920, dh=1080, layers=1, frames=128, bytes=BGRA_1080 + NV12_1080),
    "cfg5": dict(desc="8 x 3840x2160 BGRA layers composited onto a 2160p canvas, then Lanczos-3 down to 1920x1080",
                 kind="rgb_layers", sw=3840, sh=2160, dw=3840, dh=2160, layers=8, frames=24,
                 bytes=8 * BGRA_2160 + BGRA_1080, lanczos=(1920, 1080)),
    "mixed": dict(desc="1080p NV12 video + 1080p y420p video (opacity .5) + two 640x360 BGRA/RGBA overlays -> 720p BGRA canvas",
                  kind="mixed", sw=1920, sh=1080, dw=1280, dh=720, layers=4, frames=128,
                  bytes=2 * NV12_1080 + 2 * 921600 + BGRA_720),
}
HEADLINE = "pipeline"
DEFAULT_SET = ["pipeline", "cfg2", "cfg3", "cfg5", "mixer_y420p", "encode_nv12", "pipeline_y420p", "pipeline_grid", "pipeline_logo"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None, help="ticks per launch per GPU (default: per workload, 256 for the headline)")
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS),
                    help="the workload `value` / `roofline` are reported on")
    ap.add_argument("--also", default=None,
                    help="comma-separated workloads timed after the headline and reported under \"workloads\" "
                         "(default: the BASELINE set when --workload is the default, none otherwise); 'none' for none")
    ap.add_argument("--group", type=int, default=0,
                    help="ticks per launch GROUP of workloads with a second stage (cfg5: composite, then Lanczos): the batch is issued as frames / G "
                         "pairs of launches (composite G ticks, resize G canvases) so that the second stage reads canvases the first one just wrote "
                         "(default: the workload's own `group`, 0 = one pair for the whole batch)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="minimum duration of the headline's timed region")
    ap.add_argument("--min-seconds-other", type=float, default=0.6, help="minimum timed region of each other workload")
    ap.add_argument("--launches-per-step", type=int, default=0, help="fixed instead of calibrated (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--full", action="store_true",
                    help="every leg on top of the default command (headline + the default workloads + live PMC + CPU sample): the upload-inclusive "
                         "and end-to-end legs, the one-tick-at-a-time legs, thread scaling, route regret, clock / power of every workload.  "
                         "They go to bench_detail.json; the last stdout line stays the compact one either way")
    ap.add_argument("--detail-json", default=None,
                    help="where rank 0 writes the full report (every workload's record, the legs, build flags, prose): default bench_detail.json "
                         "next to this script; 'none' to write nothing")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the end-to-end (H2D-inclusive) cfg2 measurement")
    ap.add_argument("--upload-group", type=int, default=8, help="frames per H2D copy in the upload-inclusive leg")
    ap.add_argument("--upload-streams", type=int, default=2, help="upload streams (contexts) in the upload-inclusive leg")
    ap.add_argument("--no-per-tick", action="store_true", help="skip the one-tick-at-a-time legs (pipeline_per_tick, pipeline_reference_sequence)")
    ap.add_argument("--per-tick", action="store_true", help="run the one-tick-at-a-time legs even with --also none")
    ap.add_argument("--route-regret", action="store_true", help="run the route-regret leg even with --also none (for the headline workload)")
    ap.add_argument("--no-route-regret", action="store_true", help="skip the route-regret leg of the default run")
    ap.add_argument("--power-probe", action="store_true", help="run the clock / power leg even with --also none (for the headline workload)")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the clock / power leg of the default run")
    ap.add_argument("--alias", default="none", choices=("none", "src", "dst", "both"),
                    help="DIAGNOSTIC (single-layer YUV workloads): every tick reads frame 0's source and/or writes frame 0's "
                         "canvas, so that side of the traffic stays in cache; the line is marked and is not a benchmark result")
    ap.add_argument("--content", default=os.environ.get("BENCH_CONTENT", "random"), choices=("random", "gradient", "natural"),
                    help="bytes of the source pictures (content_image): `random` is the benchmark's; the others are diagnostics for data-dependent "
                         "hardware paths and mark the line (config.content, data)")
    ap.add_argument("--device", type=int, default=None,
                    help="device index for every rank (default: LOCAL_RANK); lets the N>1 path be exercised on a 1-GPU box")
    ap.add_argument("--with-upload", action="store_true",
                    help="end-to-end mode only: every step also uploads its source frames from pinned host memory on a side "
                         "stream (PCIe-inclusive rate; reported for DESIGN.md, never the headline value)")
    ap.add_argument("--threads", action="store_true",
                    help="with --gpus N: ONE process, one host thread + compute context per device (the shape of the Swift host: a composer "
                         "with mixers bound to devices, composer.swift:203-224) instead of one process per GPU; the headline workload only")
    ap.add_argument("--stub-device", action="store_true",
                    help="control-plane self-test without a GPU (tests only): every launch is a short sleep, nothing is loaded or "
                         "computed; spawn, rendezvous, calibration, barriers, reductions and the report are the real ones; the line "
                         "is marked and is not a benchmark result")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two short `rocprofv3 --kernel-trace --pmc` passes of the headline "
                         "workload in child processes, after the timed regions); the committed profiles/pmc_latest.json is reported instead")
    ap.add_argument("--pmc-json", default=None,
                    help="a profiles/pmc_*.json produced by profiles/run_profile.sh for THIS command: its HBM bytes per launch "
                         "are copied into roofline.traffic with roofline.traffic_source naming the file")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# HBM traffic of the headline kernel, measured in this run
# ---------------------------------------------------------------------------------------------------------------
def live_traffic(workload, frames, kernel_substr, device):
    """FETCH_SIZE and WRITE_SIZE of the workload's kernel in two separate `rocprofv3 --kernel-trace --pmc` passes (never combined
    with another tracing domain), each a short child run of this script; bytes per launch with the gfx950 correction of
    MI355X_MICROARCH.md (FETCH_SIZE x 2, KB = 1024 B).  None on any failure — the caller falls back to the committed figure."""
    import shutil, sqlite3, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("CHV_BENCH_CHILD"):
        return None, "rocprofv3 not available"
    vals = {}
    try:
        with tempfile.TemporaryDirectory(prefix="chv_pmc_", dir="/tmp") as tmp:
            env = dict(os.environ, CHV_BENCH_CHILD="1", TMPDIR="/tmp")
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, str(ROOT / "bench.py"),
                       "--workload", workload, "--frames", str(frames), "--also", "none", "--no-cpu-baseline", "--no-verify", "--no-live-pmc",
                       "--steps", "3", "--warmup", "1", "--launches-per-step", "1", "--device", str(device)]
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180, check=True)
                dbs = list(Path(out).glob("**/*_results.db"))
                if not dbs:
                    return None, f"no rocprofv3 database for {counter}"
                con = sqlite3.connect(str(dbs[0]))
                rows = con.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? "
                                   "group by kernel_name", (counter,)).fetchall()
                con.close()
                # one kernel, or — a workload of two stages (cfg5: the composite, then the resize) — the sum over its kernels' averages
                total = 0.0
                for sub in ([kernel_substr] if isinstance(kernel_substr, str) else kernel_substr):
                    hit = [r for r in rows if sub in r[0]]
                    if not hit:
                        return None, f"kernel {sub} not in the {counter} pass"
                    total += float(max(hit, key=lambda r: r[2])[1])
                vals[counter] = total
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, None
    except Exception as e:    # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"


# ---------------------------------------------------------------------------------------------------------------
# launcher / control plane
# ---------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rank r -> device r
    unless --device pins them), pass rank 0's stdout through, fail if any rank fails."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CHV_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def dist_setup():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        # control plane only (barrier, max over ranks): gloo on the host; the pixel path has no collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist = dist_mod
    return rank, local, world, dist


class ThreadDist:
    """--threads: the control plane of one process with a host thread per device — a threading.Barrier and a shared list where the
    process-per-GPU mode has gloo"""

    def __init__(self, shared, rank):
        self.shared, self.rank = shared, rank

    def barrier(self):
        self.shared["barrier"].wait()

    def get_rank(self):
        return self.rank

    def gather(self, value):
        self.shared["slots"][self.rank] = float(value)
        self.barrier()
        out = list(self.shared["slots"])
        self.barrier()
        return out


def reduce_max(dist, value):
    """MAX over ranks of a host float (gloo all-reduce); identity when not distributed."""
    if dist is None:
        return float(value)
    if isinstance(dist, ThreadDist):
        return max(dist.gather(value))
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_floats(dist, value, world):
    """every rank's value, on every rank"""
    if dist is None:
        return [float(value)]
    if isinstance(dist, ThreadDist):
        return dist.gather(value)
    import torch
    t = torch.zeros(world, dtype=torch.float64)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t]


def whole_job_gpix(n_gpus, px_per_step_per_gpu, steps, elapsed_max):
    """Weak scaling: every rank processes the same per-GPU batch; value = all pixels / slowest rank's time."""
    return n_gpus * px_per_step_per_gpu * steps / elapsed_max / 1e9


def stream_to_device(stream_id, n_gpus):
    """configs[3]: picture bus s is bound to device s mod n_gpus."""
    return stream_id % n_gpus


def pick_device(args, local, n_visible):
    """rank -> device: --device pins every rank to one device (1-GPU boxes); otherwise LOCAL_RANK, which must exist."""
    if args.device is not None:
        return args.device
    if local >= n_visible:
        raise SystemExit(f"rank with LOCAL_RANK={local} has no device: {n_visible} visible "
                         f"(use --device D to run several ranks on one device)")
    return local


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def content_image(util, fmt, w, h, seed, content="random"):
    """A source picture's bytes.  `random` (the default and the only content `value` is ever quoted on): uniform bytes from splitmix64, SURVEY 8d.
    `gradient`: SURVEY 8d's second set (Y = (x + 2y) & 255, U = 3x & 255, V = 5y & 255; RGB pictures B / G / R the same three, alpha x ^ y).
    `natural`: low-pass-filtered noise — neighbouring samples correlated the way decoded video's are (three box blurs of radius 6 over the
    random plane, stretched back to the full code range, +-2 codes of grain).  Content changes no instruction count; it changes what
    data-dependent hardware paths see: LDS bank conflicts of a byte-indexed table, DRAM / cache compression, the power of toggling bits."""
    planes = util.alloc_image(fmt, w, h, seed=seed)
    if content == "random":
        return planes
    for k, p in enumerate(planes):
        hh, ww = p.shape[0], p.shape[1]
        comps = 1 if p.ndim == 2 else p.shape[2]
        v = p.reshape(hh, ww, comps)
        # sample coordinates in luma units (chroma planes are half size)
        sub = 2 if (fmt in ("nv12", "y420p") and k > 0) else 1
        yy, xx = np.mgrid[0:hh, 0:ww]
        xx, yy = xx * sub, yy * sub
        if content == "gradient":
            chans = {"nv12": [[(xx + 2 * yy)], [3 * xx, 5 * yy]], "y420p": [[(xx + 2 * yy)], [3 * xx], [5 * yy]]}.get(fmt)
            if chans is None:
                chans = [[xx + 2 * yy, 3 * xx, 5 * yy, xx ^ yy]]
            for c in range(comps):
                v[:, :, c] = (chans[k][c] & 255).astype(np.uint8)
        elif content == "natural":
            for c in range(comps):
                f = v[:, :, c].astype(np.float32)
                for _ in range(3):
                    for axis in (0, 1):
                        r = 6
                        pad = np.concatenate([np.repeat(f.take([0], axis=axis), r + 1, axis=axis), f, np.repeat(f.take([-1], axis=axis), r, axis=axis)], axis=axis)
                        cs = np.cumsum(pad, axis=axis, dtype=np.float64)
                        n = f.shape[axis]
                        f = ((cs.take(range(2 * r + 1, 2 * r + 1 + n), axis=axis) - cs.take(range(0, n), axis=axis)) / (2 * r + 1)).astype(np.float32)
                lo, hi = float(f.min()), float(f.max())
                f = (f - lo) / max(hi - lo, 1e-6) * 255.0
                grain = (v[:, :, c] % 5).astype(np.float32) - 2.0
                v[:, :, c] = np.clip(np.rint(f + grain), 0, 255).astype(np.uint8)
        else:
            raise ValueError(content)
    return planes


def build_workload(sv, ctx, wl, frames, seed_base, alias="none", group=0, content="random"):
    """Device-resident source frames, canvases and the batch descriptor."""
    import util
    from swiftvideo_amd import chipvideo as cv
    lib = cv.load()
    src_image = lambda fmt, w, h, seed: content_image(util, fmt, w, h, seed, content)      # noqa: E731
    sw, sh, dw, dh = wl["sw"], wl["sh"], wl["dw"], wl["dh"]
    distinct = 4
    host_src = []
    keep = []      # keep PictureSamples alive (they own the device memory)
    canvases = []
    ticks = (cv.Tick * frames)()
    layer_arrays = []
    lanczos_pairs = []
    verify = None
    up = lambda pf, size, planes: sv.uploadComputePicture(ctx, sv.pictureFromArrays(pf, size, planes), retainCpuBuffer=False)   # noqa: E731
    blank = lambda pf, size: sv.uploadComputePicture(ctx, sv.createPictureSample(size, pf), retainCpuBuffer=False)              # noqa: E731
    PF = {"nv12": sv.PixelFormat.nv12, "y420p": sv.PixelFormat.y420p, "bgra": sv.PixelFormat.BGRA, "rgba": sv.PixelFormat.RGBA}

    def finish_tick(f, dst, layers):
        arr = sv._layer_array(layers)
        layer_arrays.append(arr)
        ticks[f].target = sv._image_desc(dst)
        ticks[f].clear_first = 1
        ticks[f].n_layers = len(layers)
        ticks[f].layers = arr

    if wl["kind"] == "mixer420":
        fmt = wl["mixer"]
        sfmt, kmain = wl.get("main_src", fmt), wl.get("main_kernel", f"img_{fmt}_{fmt}")
        for i in range(distinct):
            host_src.append(src_image(sfmt, sw, sh, seed=seed_base + i))
        ov = [src_image("bgra", 640, 360, seed=seed_base + 100 + i) for i in range(2)]
        # (diagnostic: BENCH_OVERLAY_POS="x0,y0,x1,y1" moves the two overlays — how much of a mixer tick's time is strips an overlay's edge crosses)
        ovp = [int(v) for v in os.environ.get("BENCH_OVERLAY_POS", "64,64,1200,640").split(",")]
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh)),
              util.make_uniforms((dw, dh), rect=(ovp[0], ovp[1], 640, 360), opacity=0.8, in_size=(640, 360)),
              util.make_uniforms((dw, dh), rect=(ovp[2], ovp[3], 640, 360), opacity=0.6, in_size=(640, 360))]
        k_main = sv.defaultComputeKernelFromString(kmain)
        k_ov = sv.defaultComputeKernelFromString(f"img_bgra_{fmt}")
        govs = [up(sv.PixelFormat.BGRA, (640, 360), o) for o in ov]
        keep += govs
        for f in range(frames):
            src = up(PF[sfmt], (sw, sh), host_src[f % distinct])
            dst = blank(PF[fmt], (dw, dh))
            keep += [src, dst]
            if f > 0 and alias in ("src", "both"):
                src = keep[len(govs)]
            if f > 0 and alias in ("dst", "both"):
                dst = keep[len(govs) + 1]
            canvases.append(dst)
            nov = wl.get("overlays", 2)
            finish_tick(f, dst, [(k_main, src, us[0], 0)] + [(k_ov, govs[i], us[1 + i], 0) for i in range(nov)])
        verify = dict(target=fmt, layers=lambda f: [(kmain, host_src[f % distinct], us[0])] +
                                                   [(f"img_bgra_{fmt}", ov[i], us[1 + i]) for i in range(wl.get("overlays", 2))])
    elif wl["kind"] == "yuv_layers":
        sfmt, nl = wl["src"], wl["layers"]
        skernel = sv.defaultComputeKernelFromString(f"img_{sfmt}_bgra")
        for i in range(distinct):
            host_src.append(src_image(sfmt, sw, sh, seed=seed_base + i))
        ops = (1.0, 0.75, 0.5, 0.25)
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=ops[l]) for l in range(nl)]
        first_src = first_dst = None
        logo = logo_u = glogo = None
        if wl.get("logo"):
            logo = src_image("rgba", 320, 180, seed=seed_base + 200)
            logo_u = util.make_uniforms((dw, dh), rect=(820, 60, 320, 180), rotation=0.3, opacity=0.9, in_size=(320, 180))
            glogo = up(sv.PixelFormat.RGBA, (320, 180), logo)
            keep.append(glogo)
        for f in range(frames):
            layers = []
            for l in range(nl):
                src = up(PF[sfmt], (sw, sh), host_src[(f + l) % distinct])
                keep.append(src)
                if first_src is None:
                    first_src = src
                if f > 0 and alias in ("src", "both"):
                    src = first_src
                layers.append((skernel, src, us[l], cv.CSC_BT601_LIMITED))
            dst = blank(sv.PixelFormat.BGRA, (dw, dh))
            keep.append(dst)
            if first_dst is None:
                first_dst = dst
            if f > 0 and alias in ("dst", "both"):
                dst = first_dst
            canvases.append(dst)
            if glogo is not None:
                layers.append((sv.ComputeKernel.img_rgba_bgra_tx, glogo, logo_u, 0))
            finish_tick(f, dst, layers)
        verify = dict(target="bgra", layers=lambda f: [(f"img_{sfmt}_bgra", host_src[(f + l) % distinct], us[l]) for l in range(nl)] +
                                                      ([("img_rgba_bgra_tx", logo, logo_u)] if logo is not None else []))
    elif wl["kind"] == "grid":
        skernel = sv.defaultComputeKernelFromString("img_nv12_bgra")
        for i in range(distinct):
            host_src.append(src_image("nv12", sw, sh, seed=seed_base + i))
        qw, qh = dw // 2, dh // 2
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh))] + \
             [util.make_uniforms((dw, dh), rect=(qx * qw, qy * qh, qw, qh), opacity=o, in_size=(sw, sh))
              for (qx, qy), o in zip(((0, 0), (1, 0), (0, 1), (1, 1)), (1.0, 0.9, 0.8, 0.7))]
        for f in range(frames):
            layers = []
            for l in range(5):
                src = up(sv.PixelFormat.nv12, (sw, sh), host_src[(f + l) % distinct])
                keep.append(src)
                layers.append((skernel, src, us[l], cv.CSC_BT601_LIMITED))
            dst = blank(sv.PixelFormat.BGRA, (dw, dh))
            keep.append(dst)
            canvases.append(dst)
            finish_tick(f, dst, layers)
        verify = dict(target="bgra", layers=lambda f: [("img_nv12_bgra", host_src[(f + l) % distinct], us[l]) for l in range(5)])
    elif wl["kind"] == "rgb_layers":
        nl = wl["layers"]
        for i in range(distinct):
            host_src.append(src_image("bgra", sw, sh, seed=seed_base + i))
        ops = (1.0, 0.75, 0.5, 0.25) if nl <= 4 else (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3)
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=o) for o in ops[:nl]]
        for f in range(frames):
            layers = []
            for l in range(nl):
                src = up(sv.PixelFormat.BGRA, (sw, sh), host_src[(f + l) % distinct])
                keep.append(src)
                layers.append((sv.ComputeKernel.img_bgra_bgra_tx, src, us[l], 0))
            dst = blank(sv.PixelFormat.BGRA, (dw, dh))
            keep.append(dst)
            if canvases and alias in ("dst", "both"):       # (diagnostic: every tick onto frame 0's canvas — cache-resident stores)
                dst = canvases[0]
            canvases.append(dst)
            if "lanczos" in wl:
                small = blank(sv.PixelFormat.BGRA, wl["lanczos"])
                lanczos_pairs.append((small, dst))
            finish_tick(f, dst, layers)
        verify = dict(target="bgra", layers=lambda f: [("img_bgra_bgra_tx", host_src[(f + l) % distinct], us[l]) for l in range(nl)])
    elif wl["kind"] == "mixed":
        nv = [src_image("nv12", sw, sh, seed=seed_base + i) for i in range(distinct)]
        yp = [src_image("y420p", sw, sh, seed=seed_base + 50 + i) for i in range(distinct)]
        ov = [src_image("bgra", 640, 360, seed=seed_base + 100), src_image("rgba", 640, 360, seed=seed_base + 101)]
        us = [util.full_canvas_uniforms((dw, dh), (sw, sh)), util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=0.5),
              util.make_uniforms((dw, dh), rect=(48, 40, 426, 240), opacity=0.8, in_size=(640, 360)),
              util.make_uniforms((dw, dh), rect=(800, 430, 426, 240), opacity=0.6, in_size=(640, 360))]
        govs = [up(sv.PixelFormat.BGRA, (640, 360), ov[0]), up(sv.PixelFormat.RGBA, (640, 360), ov[1])]
        keep += govs
        K = sv.ComputeKernel
        for f in range(frames):
            a, b = up(sv.PixelFormat.nv12, (sw, sh), nv[f % distinct]), up(sv.PixelFormat.y420p, (sw, sh), yp[f % distinct])
            dst = blank(sv.PixelFormat.BGRA, (dw, dh))
            keep += [a, b, dst]
            canvases.append(dst)
            finish_tick(f, dst, [(K.img_nv12_bgra, a, us[0], 0), (K.img_y420p_bgra, b, us[1], 0),
                                 (K.img_bgra_bgra_tx, govs[0], us[2], 0), (K.img_rgba_bgra_tx, govs[1], us[3], 0)])
        verify = dict(target="bgra", layers=lambda f: [("img_nv12_bgra", nv[f % distinct], us[0]), ("img_y420p_bgra", yp[f % distinct], us[1]),
                                                       ("img_bgra_bgra_tx", ov[0], us[2]), ("img_rgba_bgra_tx", ov[1], us[3])])
    else:
        raise ValueError(wl["kind"])
    # one batch for all ticks — or, for two-stage workloads with a group size, one batch per group of ticks
    batches = []
    g = group if (group and lanczos_pairs and 0 < group < frames) else frames
    for first in range(0, frames, g):
        n = min(g, frames - first)
        sub = (cv.Tick * n).from_address(C.addressof(ticks) + first * C.sizeof(cv.Tick))
        b = C.c_void_p()
        cv.check(lib.chv_batch_create(ctx.handle, sub, n, C.byref(b)))
        batches.append((b, first, n))
    name = C.create_string_buffer(128)
    n_launch = C.c_int(1)
    cv.check(lib.chv_batch_describe(batches[0][0], name, 128, C.byref(n_launch)))
    return dict(batch=batches[0][0], batches=batches, keep=keep, layer_arrays=layer_arrays, ticks=ticks, kernel=name.value.decode(), verify=verify,
                lanczos=lanczos_pairs, canvases=canvases, launches_per_batch=n_launch.value)


def free_workload(w):
    from swiftvideo_amd import chipvideo as cv
    for b, _, _ in w["batches"]:
        cv.check(cv.load().chv_batch_destroy(b))
    w["keep"].clear(); w["canvases"].clear(); w["lanczos"].clear(); w["layer_arrays"].clear()
    import gc
    gc.collect()


def verify_frame(sv, ctx, wl, w, frame=0):
    """Canvas of tick `frame` == oracle: clear + the tick's layer kernels in z order (outside any timed region)."""
    import util
    from oracle import oracle as O
    v = w["verify"]
    dw, dh = wl["dw"], wl["dh"]
    fmt = v["target"]
    exp = util.alloc_image(fmt, dw, dh)
    threads = os.cpu_count() or 1
    assert O.run_kernel(f"img_clear_{fmt}", exp, threads=threads) == 0
    for kernel, src, u in v["layers"](frame):
        assert O.run_kernel(kernel, exp, src, u, threads=threads) == 0
    got = sv.downloadComputePicture(ctx, w["canvases"][frame], retainGpuBuffer=True).imageBuffer().buffers
    for g, e in zip(got, exp):
        comps = 1 if e.ndim == 2 else e.shape[2]
        view = g[: e.shape[0], : e.shape[1] * comps].reshape(e.shape)
        if not np.array_equal(view, e):
            return False
    return True


def cpu_baseline(wl, w, budget_s):
    """The oracle on this box's host cores, same kernels/uniforms, bounded sample.
    Stream-parallel like the GPU path: every worker thread composites whole ticks on its
    own canvas (ctypes releases the GIL), so no per-call thread start-up is measured."""
    import threading
    import util
    from oracle import oracle as O
    v = w["verify"]
    dw, dh = wl["dw"], wl["dh"]
    cores = os.cpu_count() or 1
    O.lib()
    counts = [0] * cores
    t_end = [0.0]
    fmt = v["target"]
    n_kernels = len(v["layers"](0))

    def worker(i):
        canvas = util.alloc_image(fmt, dw, dh)
        n = 0
        while time.perf_counter() < t_end[0]:
            O.run_kernel(f"img_clear_{fmt}", canvas, threads=1)
            for kernel, src, u in v["layers"](n):
                O.run_kernel(kernel, canvas, src, u, threads=1)
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
            "sample": f"{n} ticks of the same workload in {el:.1f} s, oracle/ref_kernels.c, {cores} threads x whole ticks (clear + {n_kernels} layer kernels)"}


class HipDevice:
    """events and synchronisation on the context's stream, through the C ABI"""

    def __init__(self, cv, lib, ctx):
        self.cv, self.lib, self.ctx = cv, lib, ctx

    def sync(self):
        self.cv.check(self.lib.chv_device_synchronize(self.ctx.handle))
        # (torch is the N > 1 control plane only; it is synchronised too when something has ALREADY imported and initialised it — never imported
        # here: the first `import torch` on a fresh box costs tens of seconds)
        torch = sys.modules.get("torch")
        try:
            if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
        except Exception:       # noqa: BLE001
            pass

    def event(self):
        e = C.c_void_p()
        self.cv.check(self.lib.chv_event_create(self.ctx.handle, C.byref(e)))
        return e

    def record(self, e):
        self.cv.check(self.lib.chv_event_record(self.ctx.handle, e))

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        self.cv.check(self.lib.chv_event_elapsed_ms(a, b, C.byref(ms)))
        return ms.value

    def destroy(self, e):
        self.cv.check(self.lib.chv_event_destroy(e))


class StubDevice:
    """--stub-device: host clock instead of stream events, nothing to synchronise"""

    def sync(self):
        pass

    def event(self):
        return [0.0]

    def record(self, e):
        e[0] = time.perf_counter()

    def elapsed_ms(self, a, b):
        return (b[0] - a[0]) * 1e3

    def destroy(self, e):
        pass


class Timer:
    """K steps of R launches each, bracketed as the contract says; per-step events on the stream the kernels run on."""

    def __init__(self, dev, dist):
        self.dev, self.dist = dev, dist

    def sync(self):
        self.dev.sync()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def calibrate(self, launch, steps, min_seconds, fixed):
        """launches per step so that `steps` steps last >= min_seconds; the same number on every rank"""
        if fixed > 0:
            return fixed
        a, b = self.dev.event(), self.dev.event()
        n = 8
        self.dev.record(a)
        for _ in range(n):
            launch()
        self.dev.record(b)
        self.sync()
        ms = max(self.dev.elapsed_ms(a, b) / n, 1e-3)
        for e in (a, b):
            self.dev.destroy(e)
        # 30 % on top: the calibration launches (clocks still ramping) run up to 15 % slower than the timed ones
        r = max(1, math.ceil(1.3 * min_seconds * 1e3 / (steps * ms)))
        return int(reduce_max(self.dist, r))

    def run(self, launch, steps, per_step):
        evs = [self.dev.event() for _ in range(steps + 1)]
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        self.dev.record(evs[0])
        for k in range(steps):
            for _ in range(per_step):
                launch()
            self.dev.record(evs[k + 1])
        self.sync()
        self.barrier()
        t1 = time.perf_counter()
        local = t1 - t0
        elapsed = reduce_max(self.dist, local)
        step_ms = [self.dev.elapsed_ms(evs[k], evs[k + 1]) for k in range(steps)]
        for e in evs:
            self.dev.destroy(e)
        return elapsed, local, float(np.mean(step_ms)) / per_step


def report_of(name, wl, args, tm, n_gpus, frames, per_step, elapsed, local, launch_ms, kernel, verified):
    locals_ = gather_floats(tm.dist, local, n_gpus)
    launch_all = gather_floats(tm.dist, launch_ms, n_gpus)
    px_per_launch = frames * wl["dw"] * wl["dh"] if "lanczos" not in wl else frames * wl["lanczos"][0] * wl["lanczos"][1]
    bytes_per_launch = frames * wl["bytes"]
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    return {
        "workload": f"{name}: {wl['desc']}", "workload_short": f"{name}: {wl.get('short', wl['desc'])}",
        "value": whole_job_gpix(n_gpus, px_per_launch * per_step, args.steps, elapsed), "unit": "Gpix/s",
        "ms_per_step": elapsed / args.steps * 1e3, "launches_per_step": per_step, "timed_seconds": elapsed,
        "frames_per_launch_per_gpu": frames, "kernel": kernel, "verified_vs_oracle": verified,
        "launch_ms": launch_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_of_copy_ceiling": achieved / HBM_COPY_GBS},
        "per_gpu_gpix": [px_per_launch * per_step * args.steps / t / 1e9 for t in locals_], "per_gpu_launch_ms": launch_all,
        # a tick belongs to one PictureSample bus: `frames` streams per GPU advance by one tick per launch
        "per_stream_ticks_per_s": 1e3 / launch_ms,
        "source_mpix_per_launch_per_gpu": frames * wl["sw"] * wl["sh"] * (wl["layers"] if wl["kind"] != "mixer420" else 1) / 1e6,
    }


def measure_stub(name, args, tm, rank, n_gpus, headline):
    """--stub-device: the control flow of measure() with sleeps for launches (rank r is (1 + r/4) x slower than rank 0)"""
    wl = WORKLOADS[name]
    frames = args.frames if (args.frames and headline) else wl["frames"]
    pause = 0.0004 * (1.0 + rank / 4.0)

    def launch():
        time.sleep(pause)

    for _ in range(max(args.warmup, 1)):
        launch()
    per_step = tm.calibrate(launch, args.steps, args.min_seconds if headline else args.min_seconds_other, args.launches_per_step)
    elapsed, local, launch_ms = tm.run(launch, args.steps, per_step)
    return report_of(name, wl, args, tm, n_gpus, frames, per_step, elapsed, local, launch_ms, "stub", None), None


def measure(name, args, sv, cv, lib, ctx, tm, rank, n_gpus, headline):
    """Build, warm up, verify, time and free one workload; returns its report (complete on rank 0)."""
    wl = WORKLOADS[name]
    frames = args.frames if (args.frames and headline) else wl["frames"]
    group = args.group or wl.get("group", 0)
    w = build_workload(sv, ctx, wl, frames, seed_base=0x5EED0000 + 16 * 2 + rank, alias=args.alias if headline else "none", group=group, content=args.content)

    # the second stage: one launch per group for the group's resizes (chv_scale_lanczos_batch)
    lzs = [sv.LanczosBatch(w["lanczos"][first:first + n]) for _, first, n in w["batches"]] if w["lanczos"] else None

    def launch():
        for i, (b, _, _) in enumerate(w["batches"]):
            cv.check(lib.chv_batch_run(ctx.handle, b))
            if lzs is not None:
                lzs[i].run(ctx)

    for _ in range(max(args.warmup, 1)):
        launch()
    tm.sync()
    verified = None
    if not args.no_verify and rank == 0 and w["verify"] is not None and (not headline or args.alias == "none"):
        verified = verify_frame(sv, ctx, wl, w, 0) and verify_frame(sv, ctx, wl, w, frames - 1)
    per_step = tm.calibrate(launch, args.steps, args.min_seconds if headline else args.min_seconds_other, args.launches_per_step)
    elapsed, local, launch_ms = tm.run(launch, args.steps, per_step)
    rep = report_of(name, wl, args, tm, n_gpus, frames, per_step, elapsed, local, launch_ms, w["kernel"], verified)
    if lzs is not None:
        rep["kernel_launches_per_batch"] = 2 * len(w["batches"])
        rep["ticks_per_group"] = w["batches"][0][2]
    elif w.get("launches_per_batch", 1) > 1:
        rep["kernel_launches_per_batch"] = w["launches_per_batch"]       # (a split batch: the videos, then the rest; `launch_ms` covers both)
    cpu = None
    if headline and rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and w["verify"] is not None:
        cpu = cpu_baseline(wl, w, args.cpu_seconds)
    free_workload(w)
    return rep, cpu


def bind_to_node(node):
    """Bind this thread (a rank's main thread, or a --threads worker) to the CPUs of NUMA node `node`; returns the previous affinity, or None
    when the platform does not say / the node has none of our CPUs."""
    if node < 0:
        return None
    try:
        cpus = set()
        for part in Path(f"/sys/devices/system/node/node{node}/cpulist").read_text().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        prev = os.sched_getaffinity(0)
        os.sched_setaffinity(0, cpus & prev or prev)
        return prev
    except (OSError, ValueError):
        return None


def bind_to_device_node(cv, lib, ctx):
    """Bind this thread to the CPUs of the NUMA node the device hangs off (chv_context_numa_node) so that the pinned upload ring
    is first-touched there; returns (node, previous affinity) — node -1 / None when the platform does not say."""
    node = C.c_int(-1)
    cv.check(lib.chv_context_numa_node(ctx.handle, C.byref(node)))
    return node.value, bind_to_node(node.value)


def run_with_upload_stub(args, tm, rank, n_gpus):
    """--stub-device --with-upload: the per-rank plumbing of the upload-inclusive mode (node binding, per-rank gathers, the report's keys)
    with sleeps for the copies and launches; rank r is (1 + r / 4) x slower.  Every rank binds to node 0 (the one every Linux box has)."""
    prev = bind_to_node(0)
    try:
        bound_cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        bound_cpus = -1
    pause = 0.0006 * (1.0 + rank / 4.0)

    def launch():
        time.sleep(pause)

    per_step = tm.calibrate(launch, args.steps, args.min_seconds_other, args.launches_per_step)
    elapsed, local, launch_ms = tm.run(launch, args.steps, per_step)
    frames, fbytes = 64, NV12_1080
    h2d = frames * fbytes * per_step * args.steps / local / 1e9
    per_rank = {"h2d_GBps": gather_floats(tm.dist, h2d, n_gpus), "launch_ms": gather_floats(tm.dist, launch_ms, n_gpus),
                "pinned_numa_node": [int(v) for v in gather_floats(tm.dist, 0.0, n_gpus)],
                "cpus_bound": [int(v) for v in gather_floats(tm.dist, float(bound_cpus), n_gpus)]}
    if prev is not None:
        os.sched_setaffinity(0, prev)
    return {"workload": "cfg2_upload (STUB: sleeps)", "value": whole_job_gpix(n_gpus, frames * 1280 * 720 * per_step, args.steps, elapsed), "unit": "Gpix/s",
            "ms_per_step": elapsed / args.steps * 1e3, "launches_per_step": per_step, "timed_seconds": elapsed, "frames_per_launch_per_gpu": frames,
            "kernel": "stub", "h2d_GBps_per_gpu": h2d, "h2d_frac_of_link": h2d / H2D_LINK_GBS, "h2d_link_GBps": H2D_LINK_GBS, "upload_copy_MB": 8 * fbytes / 1e6,
            "upload_streams": args.upload_streams, "pinned_numa_node": 0, "per_stream_ticks_per_s": 1e3 / launch_ms, "verified_vs_oracle": None, "per_rank": per_rank}


def run_with_upload(args, sv, cv, lib, ctx, tm, rank, n_gpus, frames=64, group=8, streams=2):
    """PCIe-inclusive pipeline for cfg2: two frame sets; while set A is converted on the compute context's stream, set B's
    decoded frames go up from a pinned host ring on `streams` sharing contexts' streams (one copy engine each).  Frames live
    `group` to a device slab (PictureSlab) and `group` to a run of the host ring, so one LINEAR hipMemcpyAsync moves `group`
    frames (8 x 3.1 MB = 25 MB: the link gives 55-57 GB/s at that size, 48 for one 3.1 MB frame — tools/h2d_probe.py).
    Ordering: per-slab upload events (the kernel waits for its inputs on its own stream) and a per-set 'batch done' event (the
    next upload into the set waits for the kernel that read it); no host waits inside the loop."""
    import util
    from oracle import oracle as O
    wl = WORKLOADS["cfg2"]
    sw, sh, dw, dh = wl["sw"], wl["sh"], wl["dw"], wl["dh"]
    fbytes = sw * sh * 3 // 2
    distinct = 4
    ups = [sv.createComputeContext(sharing=ctx) for _ in range(streams)]
    node, prev_aff = bind_to_device_node(cv, lib, ctx)
    pinned = C.c_void_p()
    cv.check(lib.chv_host_alloc(ups[0].handle, frames * fbytes, C.byref(pinned)))
    host = np.ctypeslib.as_array((C.c_uint8 * (frames * fbytes)).from_address(pinned.value))
    imgs = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 32 + i) for i in range(distinct)]
    for f in range(frames):                 # the host ring: frame f of a set = distinct frame f % 4, packed Y then UV (first touch here)
        img = imgs[f % distinct]
        host[f * fbytes: f * fbytes + sw * sh] = img[0].reshape(-1)
        host[f * fbytes + sw * sh: (f + 1) * fbytes] = img[1].reshape(-1)
    u = util.full_canvas_uniforms((dw, dh), (sw, sh))
    k = sv.defaultComputeKernelFromString("img_nv12_bgra")
    sets = []
    for _ in range(2):
        slabs = [sv.PictureSlab(ctx, (sw, sh), sv.PixelFormat.nv12, group) for _ in range(frames // group)]
        canvases = [sv.uploadComputePicture(ctx, sv.createPictureSample((dw, dh), sv.PixelFormat.BGRA), retainCpuBuffer=False) for _ in range(frames)]
        ticks = [(canvases[f], True, [(k, slabs[f // group].pictures[f % group], u, cv.CSC_BT601_LIMITED)]) for f in range(frames)]
        sets.append(dict(slabs=slabs, canvases=canvases, batch=sv.TickBatch(ctx, ticks)))
    done = []
    for _ in sets:
        e = C.c_void_p()
        cv.check(lib.chv_event_create(ctx.handle, C.byref(e)))
        cv.check(lib.chv_event_record(ctx.handle, e))
        done.append(e)

    def upload_set(s):
        for upc in ups:
            cv.check(lib.chv_event_wait(upc.handle, done[s]))       # the kernel that last read this set is finished
        for g, slab in enumerate(sets[s]["slabs"]):
            slab.upload(ups[g % streams], 0, group, pinned.value + g * group * fbytes, mode=2)

    def convert_set(s):
        sets[s]["batch"].run(ctx)                                   # waits for the slabs' upload events on its stream
        cv.check(lib.chv_event_record(ctx.handle, done[s]))

    state = [0]

    def launch():
        s = state[0] & 1
        state[0] += 1
        upload_set(s)
        convert_set(s)

    for _ in range(max(args.warmup, 2)):
        launch()
    tm.sync()
    # frame 5 of the set converted last == oracle on the host ring's bytes (uploaded, not pre-resident)
    verified = None
    if not args.no_verify and rank == 0:
        f = 5
        exp = util.alloc_image("bgra", dw, dh)
        assert O.run_kernel("img_clear_bgra", exp, threads=os.cpu_count() or 1) == 0
        assert O.run_kernel("img_nv12_bgra", exp, imgs[f % distinct], u, threads=os.cpu_count() or 1) == 0
        got = sv.downloadComputePicture(ctx, sets[(state[0] - 1) & 1]["canvases"][f], retainGpuBuffer=True).imageBuffer().buffers[0]
        verified = bool(np.array_equal(got[:dh, : dw * 4].reshape(dh, dw, 4), exp[0]))
    per_step = tm.calibrate(launch, args.steps, args.min_seconds_other, args.launches_per_step)
    elapsed, local, launch_ms = tm.run(launch, args.steps, per_step)
    px = frames * dw * dh
    h2d = frames * fbytes * per_step * args.steps / local / 1e9
    # per rank (whoever runs the 8-GPU line sees WHICH rank bent the curve, and whether its pinned ring sat on the device's own NUMA node)
    try:
        bound_cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        bound_cpus = -1
    per_rank = {"h2d_GBps": gather_floats(tm.dist, h2d, n_gpus), "launch_ms": gather_floats(tm.dist, launch_ms, n_gpus),
                "pinned_numa_node": [int(v) for v in gather_floats(tm.dist, float(node), n_gpus)],
                "cpus_bound": [int(v) for v in gather_floats(tm.dist, float(bound_cpus), n_gpus)]}
    rep = {
        "workload": f"cfg2_upload: cfg2 END-TO-END incl. H2D upload of every 1080p NV12 source frame from a pinned host ring, {group} frames "
                    f"({group * fbytes / 1e6:.1f} MB) per copy on {streams} side streams (PCIe-bound; never the headline value)",
        "value": whole_job_gpix(n_gpus, px * per_step, args.steps, elapsed), "unit": "Gpix/s",
        "ms_per_step": elapsed / args.steps * 1e3, "launches_per_step": per_step, "timed_seconds": elapsed,
        "frames_per_launch_per_gpu": frames, "kernel": sets[0]["batch"].kernelName,
        "h2d_GBps_per_gpu": h2d, "h2d_frac_of_link": h2d / H2D_LINK_GBS, "h2d_link_GBps": H2D_LINK_GBS,
        "upload_copy_MB": group * fbytes / 1e6, "upload_streams": streams, "pinned_numa_node": node,
        "per_stream_ticks_per_s": 1e3 / launch_ms, "verified_vs_oracle": verified, "per_rank": per_rank,
    }
    for e in done:
        cv.check(lib.chv_event_destroy(e))
    for st in sets:
        st["batch"].destroy()
        st["slabs"].clear(); st["canvases"].clear()
    cv.check(lib.chv_host_free(ups[0].handle, pinned))
    for upc in ups:
        sv.destroyComputeContext(upc)
    if prev_aff is not None:
        os.sched_setaffinity(0, prev_aff)
    return rep


def run_e2e(args, sv, cv, lib, ctx, tm, rank, n_gpus, frames=32, group=8, streams=2):
    """The whole picture path END TO END, both directions over PCIe (never the headline value): per tick four decoded 1080p NV12 frames go
    up from a pinned host ring (slabs of `group` frames per linear copy, `streams` upload contexts), the fused pipeline tick composes them
    onto a 720p BGRA canvas, img_bgra_nv12_int converts the canvas for an encoder, and the NV12 result comes back into a pinned ring through
    chv_download_async on a download context of its own.  Two sets of `frames` ticks are in flight; order comes from events only (upload ->
    kernel by the per-buffer upload events, kernel -> next upload / kernel -> copy / copy -> next kernel by chv_event_wait), no host waits in
    the loop.  Reports ticks/s and both link rates."""
    import util
    from oracle import oracle as O
    wl = WORKLOADS["pipeline"]
    sw, sh, dw, dh, nl = wl["sw"], wl["sh"], wl["dw"], wl["dh"], wl["layers"]
    fbytes, obytes = sw * sh * 3 // 2, dw * dh * 3 // 2
    distinct = 4
    ups = [sv.createComputeContext(sharing=ctx) for _ in range(streams)]
    dl = sv.createComputeContext(sharing=ctx)
    node, prev_aff = bind_to_device_node(cv, lib, ctx)
    n_in = frames * nl
    pin_in, pin_out = C.c_void_p(), C.c_void_p()
    cv.check(lib.chv_host_alloc(ups[0].handle, n_in * fbytes, C.byref(pin_in)))
    cv.check(lib.chv_host_alloc(dl.handle, 2 * frames * obytes, C.byref(pin_out)))
    host_in = np.ctypeslib.as_array((C.c_uint8 * (n_in * fbytes)).from_address(pin_in.value))
    host_out = np.ctypeslib.as_array((C.c_uint8 * (2 * frames * obytes)).from_address(pin_out.value))
    host_out[:] = 0
    imgs = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 96 + i) for i in range(distinct)]
    for f in range(n_in):                    # tick t, layer l reads ring frame t * nl + l = distinct frame (t + l) % 4
        img = imgs[(f // nl + f % nl) % distinct]
        host_in[f * fbytes: f * fbytes + sw * sh] = img[0].reshape(-1)
        host_in[f * fbytes + sw * sh: (f + 1) * fbytes] = img[1].reshape(-1)
    ops = (1.0, 0.75, 0.5, 0.25)
    us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=ops[l]) for l in range(nl)]
    u_enc = util.full_canvas_uniforms((dw, dh), (dw, dh))
    k_in, k_enc = sv.defaultComputeKernelFromString("img_nv12_bgra"), sv.defaultComputeKernelFromString("img_bgra_nv12_int")
    sets = []
    for _ in range(2):
        slabs = [sv.PictureSlab(ctx, (sw, sh), sv.PixelFormat.nv12, group) for _ in range(n_in // group)]
        outs = [sv.PictureSlab(ctx, (dw, dh), sv.PixelFormat.nv12, group) for _ in range(frames // group)]
        canvases = [sv.uploadComputePicture(ctx, sv.createPictureSample((dw, dh), sv.PixelFormat.BGRA), retainCpuBuffer=False) for _ in range(frames)]
        pic = lambda i: slabs[i // group].pictures[i % group]      # noqa: E731
        compose = [(canvases[t], True, [(k_in, pic(t * nl + l), us[l], cv.CSC_BT601_LIMITED) for l in range(nl)]) for t in range(frames)]
        encode = [(outs[t // group].pictures[t % group], True, [(k_enc, canvases[t], u_enc, cv.CSC_BT601_LIMITED)]) for t in range(frames)]
        sets.append(dict(slabs=slabs, outs=outs, canvases=canvases, compose=sv.TickBatch(ctx, compose), encode=sv.TickBatch(ctx, encode)))

    def event(c):
        e = C.c_void_p()
        cv.check(lib.chv_event_create(c.handle, C.byref(e)))
        cv.check(lib.chv_event_record(c.handle, e))
        return e
    composed = [event(ctx) for _ in sets]       # the kernel that read the set's source slabs has run
    encoded = [event(ctx) for _ in sets]        # the set's output slabs hold the encoder frames
    copied = [event(dl) for _ in sets]          # the copies that read the set's output slabs have run
    state = [0]

    def launch():
        s = state[0] & 1
        state[0] += 1
        for upc in ups:
            cv.check(lib.chv_event_wait(upc.handle, composed[s]))
        for g, slab in enumerate(sets[s]["slabs"]):
            slab.upload(ups[g % streams], 0, group, pin_in.value + g * group * fbytes, mode=2)
        sets[s]["compose"].run(ctx)                                  # waits for the slabs' upload events on its stream
        cv.check(lib.chv_event_record(ctx.handle, composed[s]))
        cv.check(lib.chv_event_wait(ctx.handle, copied[s]))          # the previous read-back of these output slabs is through
        sets[s]["encode"].run(ctx)
        cv.check(lib.chv_event_record(ctx.handle, encoded[s]))
        cv.check(lib.chv_event_wait(dl.handle, encoded[s]))
        for g, out in enumerate(sets[s]["outs"]):
            out.download(dl, 0, group, pin_out.value + (s * frames + g * group) * obytes)
        cv.check(lib.chv_event_record(dl.handle, copied[s]))

    for _ in range(max(args.warmup, 2)):
        launch()
    tm.sync()
    sv.endComputePass(dl, True)
    verified = None
    if not args.no_verify and rank == 0:
        # tick 5 of the set processed last, as it arrived in the pinned output ring == oracle (clear + 4 layers, then the integer matrix)
        t, s = 5, (state[0] - 1) & 1
        threads = os.cpu_count() or 1
        cvs = util.alloc_image("bgra", dw, dh)
        assert O.run_kernel("img_clear_bgra", cvs, threads=threads) == 0
        for l in range(nl):
            assert O.run_kernel("img_nv12_bgra", cvs, imgs[(t + l) % distinct], us[l], threads=threads) == 0
        exp = util.alloc_image("nv12", dw, dh)
        assert O.run_kernel("img_clear_nv12", exp, threads=threads) == 0
        assert O.run_kernel("img_bgra_nv12_int", exp, cvs, u_enc, threads=threads) == 0
        got = host_out[(s * frames + t) * obytes: (s * frames + t + 1) * obytes]
        verified = bool(np.array_equal(got[:dw * dh].reshape(dh, dw), exp[0]) and np.array_equal(got[dw * dh:].reshape(dh // 2, dw // 2, 2), exp[1]))
    per_step = tm.calibrate(launch, args.steps, args.min_seconds_other, args.launches_per_step)
    elapsed, local, launch_ms = tm.run(launch, args.steps, per_step)
    sv.endComputePass(dl, True)
    n_sets = per_step * args.steps
    h2d = n_in * fbytes * n_sets / local / 1e9
    d2h = frames * obytes * n_sets / local / 1e9
    rep = {
        "workload": f"pipeline_e2e: the pipeline tick END TO END, both directions over PCIe: H2D of 4 x 1080p NV12 per tick ({group} frames = "
                    f"{group * fbytes / 1e6:.1f} MB per copy, {streams} upload streams) -> fused 4-layer tick -> img_bgra_nv12_int -> D2H of the 720p "
                    f"NV12 frame ({group} frames = {group * obytes / 1e6:.1f} MB per copy, chv_download_async on its own context); two sets of "
                    f"{frames} ticks in flight, events only (PCIe-bound; never the headline value)",
        "value": whole_job_gpix(n_gpus, frames * dw * dh * per_step, args.steps, elapsed), "unit": "Gpix/s",
        "ticks_per_s": frames * n_sets / local, "ms_per_step": elapsed / args.steps * 1e3, "launches_per_step": per_step, "timed_seconds": elapsed,
        "frames_per_launch_per_gpu": frames, "kernel": sets[0]["compose"].kernelName + " + " + sets[0]["encode"].kernelName,
        "h2d_GBps_per_gpu": h2d, "h2d_frac_of_link": h2d / H2D_LINK_GBS, "d2h_GBps_per_gpu": d2h, "d2h_frac_of_link": d2h / H2D_LINK_GBS,
        "link_GBps": H2D_LINK_GBS, "pinned_numa_node": node, "verified_vs_oracle": verified,
    }
    for e in composed + encoded + copied:
        cv.check(lib.chv_event_destroy(e))
    for st in sets:
        st["compose"].destroy(); st["encode"].destroy()
        st["slabs"].clear(); st["outs"].clear(); st["canvases"].clear()
    cv.check(lib.chv_host_free(ups[0].handle, pin_in))
    cv.check(lib.chv_host_free(dl.handle, pin_out))
    for c in ups + [dl]:
        sv.destroyComputeContext(c)
    if prev_aff is not None:
        os.sched_setaffinity(0, prev_aff)
    return rep


def run_per_tick(args, sv, cv, lib, ctx, seconds=0.35, ring=10, distinct=4, mixers=8):
    """The path a Swift VideoMixer takes: ONE tick at a time with the reference's host wait after it (usingContext,
    compute.swift:131-134).  Two ways of issuing the headline tick (4 x 1080p NV12 -> 720p BGRA canvas from a ring of 10):
      fused     — one chv_composite + chv_pass_end(wait): the optional `#if GPGPU_HIP` hunk of mix.video.swift
      sequence  — img_clear_bgra + 4 x chv_run_kernel(img_nv12_bgra, blends) + chv_pass_end(wait): an UNCHANGED
                  mix.video.swift:116-124
    each from one host thread, and from `mixers` threads with their own context, sources and canvas ring (many mixers per
    device, composer.swift:203-224).  Host: this process (Python, ctypes: ~2-3 us per call on top of the library)."""
    import threading
    import util
    wl = WORKLOADS["pipeline"]
    sw, sh, dw, dh = wl["sw"], wl["sh"], wl["dw"], wl["dh"]
    ops = (1.0, 0.75, 0.5, 0.25)
    us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=o) for o in ops]
    host = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 48 + i) for i in range(distinct)]
    k_layer, k_clear = sv.ComputeKernel.img_nv12_bgra, sv.ComputeKernel.img_clear_bgra

    class Mixer:
        def __init__(self, c):
            self.ctx = c
            self.src = [sv.uploadComputePicture(c, sv.pictureFromArrays(sv.PixelFormat.nv12, (sw, sh), h), retainCpuBuffer=False) for h in host]
            self.canvas = [sv.uploadComputePicture(c, sv.createPictureSample((dw, dh), sv.PixelFormat.BGRA), retainCpuBuffer=False) for _ in range(ring)]
            self.tdesc = [sv._image_desc(cn) for cn in self.canvas]
            # the four layers of tick t: sources rotate like the batch workload's
            self.layer_arr = [sv._layer_array([(k_layer, self.src[(t + l) % distinct], us[l], cv.CSC_BT601_LIMITED) for l in range(4)]) for t in range(distinct)]
            self.sdesc = [sv._image_desc(x) for x in self.src]
            self.uni = [(cv.Uniforms).from_buffer_copy(np.asarray(u, dtype=np.float32).tobytes()) for u in us]
            self.opts = cv.KernelOpts(cv.CSC_BT601_LIMITED)
            self.n = 0

        def tick_fused(self):
            t = self.n; self.n += 1
            h = self.ctx.handle
            lib.chv_pass_begin(h)
            rc = lib.chv_composite(h, C.byref(self.tdesc[t % ring]), 1, self.layer_arr[t % distinct], 4)
            rc |= lib.chv_pass_end(h, 1)
            if rc:
                cv.check(rc)

        def tick_sequence(self):
            t = self.n; self.n += 1
            h = self.ctx.handle
            td = C.byref(self.tdesc[t % ring])
            lib.chv_pass_begin(h)
            rc = lib.chv_run_kernel(h, int(k_clear), td, None, 0, None, 0, 0, None)
            for l in range(4):
                rc |= lib.chv_run_kernel(h, int(k_layer), td, C.byref(self.sdesc[(t + l) % distinct]), 1, C.byref(self.uni[l]), 236, 1, C.byref(self.opts))
            rc |= lib.chv_pass_end(h, 1)
            if rc:
                cv.check(rc)

    def timed(fn, secs):
        for _ in range(20):
            fn()
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(20):
                fn()
            n += 20
            el = time.perf_counter() - t0
            if el >= secs:
                return el / n * 1e6, n

    out = {}
    m0 = Mixer(ctx)
    for mode in ("fused", "sequence"):
        us_tick, n = timed(getattr(m0, "tick_" + mode), seconds)
        out[mode] = {"us_per_tick": us_tick, "ticks": n}
    # oracle check of the last canvas written by each mode's final tick is covered by the batch workloads' verification of the same
    # kernels; here: fused == sequence on one tick (same sources, two canvases), byte for byte
    a, b = Mixer(ctx), Mixer(ctx)
    a.tick_fused(); b.tick_sequence()
    ga = sv.downloadComputePicture(ctx, a.canvas[0], retainGpuBuffer=True).imageBuffer().buffers[0]
    gb = sv.downloadComputePicture(ctx, b.canvas[0], retainGpuBuffer=True).imageBuffer().buffers[0]
    same = bool(np.array_equal(ga, gb))
    # `mixers` threads, one context + ring each
    ctxs = [sv.createComputeContext(sharing=ctx) for _ in range(mixers)]
    ms = [Mixer(c) for c in ctxs]
    for mode in ("fused", "sequence"):
        counts = [0] * mixers
        stop = [0.0]

        def worker(i, mode=mode):
            fn = getattr(ms[i], "tick_" + mode)
            n = 0
            while time.perf_counter() < stop[0]:
                fn(); n += 1
            counts[i] = n

        for m in ms:
            getattr(m, "tick_" + mode)()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(mixers)]
        t0 = time.perf_counter(); stop[0] = t0 + seconds
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        out[mode].update({f"us_per_tick_per_mixer_{mixers}_threads": el / (sum(counts) / mixers) * 1e6, f"ticks_per_s_{mixers}_mixers": sum(counts) / el})
    for c in ctxs:
        sv.destroyComputeContext(c)
    desc = ("the headline tick issued ONE AT A TIME with a host wait after every tick (what a Swift VideoMixer does): {} — {}; "
            "Python host (ctypes), canvas ring of %d" % ring)
    return {
        "pipeline_per_tick": dict(out["fused"], workload=desc.format("one chv_composite + chv_pass_end(wait)", "the optional GPGPU_HIP hunk of mix.video.swift"),
                                  launches_per_tick=1, fused_equals_sequence=same, gpix_per_s=dw * dh / out["fused"]["us_per_tick"] / 1e3),
        "pipeline_reference_sequence": dict(out["sequence"], workload=desc.format("img_clear_bgra + 4 x chv_run_kernel(img_nv12_bgra) + chv_pass_end(wait)",
                                                                                   "an unchanged mix.video.swift:116-124"),
                                            launches_per_tick=5, fused_equals_sequence=same, gpix_per_s=dw * dh / out["sequence"]["us_per_tick"] / 1e3),
    }


def run_group_tick_fresh(cv, lib, ctx):
    """What a host that GROUPS its mixers pays per group tick when it builds the batch of every tick afresh (new pictures every tick), runs it once
    and frees it — against running one prebuilt batch again, which is what the headline times: native host over the C ABI
    (tools/batch_create_probe.cpp), groups of 8 / 64 / 256 headline ticks, microseconds."""
    out = {"workload": "group_tick_built_fresh: chv_batch_create + chv_batch_run + chv_pass_end(wait) + chv_batch_destroy per GROUP tick of 8 / 64 / 256 "
                       "headline ticks (what a VideoMixerGroup does every tick) against the same batch run again; native host, us"}
    exe = ROOT / "tools" / "batch_create_probe.bin"
    if not exe.exists():
        out["error"] = "tools/batch_create_probe.bin not built (__graft_entry__.build())"
        return {"group_tick_built_fresh": out}
    dev = C.c_int(0)
    cv.check(lib.chv_context_device(ctx.handle, C.byref(dev)))
    try:
        p = subprocess.run([str(exe), "--json", str(dev.value)], capture_output=True, text=True, timeout=120)
        if p.returncode == 0:
            out["groups"] = json.loads(p.stdout)
            for g, r in out["groups"].items():
                r["fresh_over_prebuilt"] = round(r["built_fresh_us"] / r["run_again_us"], 3) if r.get("run_again_us") else None
        else:
            out["error"] = p.stderr[-300:]
    except Exception as e:    # noqa: BLE001
        out["error"] = repr(e)
    return {"group_tick_built_fresh": out}


def run_thread_scaling(args, sv, cv, lib, ctx, seconds=0.25):
    """How the one-tick-at-a-time path scales with host threads (one mixer = one thread + context, all on this device), from BOTH hosts: Python
    threads over ctypes (this process) and native threads over the C ABI (tools/tick_threads.cpp, what a Swift composer's mixer queues would
    be).  1 / 2 / 4 / 8 mixers, fused tick and the unchanged 1 + 4 launch sequence; ticks per second of all mixers together.  Where the native
    host scales and the Python host does not, the interpreter lock is the limit; where neither does, the HIP runtime (or the chip) is."""
    import subprocess
    import tempfile
    import threading
    import util
    wl = WORKLOADS["pipeline"]
    sw, sh, dw, dh = wl["sw"], wl["sh"], wl["dw"], wl["dh"]
    us = [util.full_canvas_uniforms((dw, dh), (sw, sh), opacity=o) for o in (1.0, 0.75, 0.5, 0.25)]
    out = {"workload": "per_tick_thread_scaling: the headline tick one at a time with a host wait after each, N mixers = N host threads with a context "
                       "each on one device; ticks/s of all mixers together at N = 1, 2, 4, 8; Python host (ctypes) and native host (tools/tick_threads.cpp)"}
    # ---- native host
    exe = ROOT / "tools" / "tick_threads.bin"
    if exe.exists():
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            for u in us:
                f.write(np.asarray(u, dtype=np.float32).tobytes())
            upath = f.name
        dev = C.c_int(0)
        cv.check(lib.chv_context_device(ctx.handle, C.byref(dev)))
        try:
            p = subprocess.run([str(exe), upath, str(seconds), str(dev.value)], capture_output=True, text=True, timeout=120)
            out["native"] = json.loads(p.stdout) if p.returncode == 0 else {"error": p.stderr[-300:]}
        except Exception as e:    # noqa: BLE001
            out["native"] = {"error": repr(e)}
        finally:
            os.unlink(upath)
    else:
        out["native"] = {"error": "tools/tick_threads.bin not built (__graft_entry__.build())"}
    # ---- Python host
    host = [util.alloc_image("nv12", sw, sh, seed=0x5EED0000 + 112 + i) for i in range(4)]
    k_layer, k_clear = sv.ComputeKernel.img_nv12_bgra, sv.ComputeKernel.img_clear_bgra

    class Mixer:
        def __init__(self, c):
            self.ctx = c
            self.src = [sv.uploadComputePicture(c, sv.pictureFromArrays(sv.PixelFormat.nv12, (sw, sh), h), retainCpuBuffer=False) for h in host]
            self.canvas = [sv.uploadComputePicture(c, sv.createPictureSample((dw, dh), sv.PixelFormat.BGRA), retainCpuBuffer=False) for _ in range(10)]
            self.tdesc = [sv._image_desc(cn) for cn in self.canvas]
            self.layer_arr = [sv._layer_array([(k_layer, self.src[(t + l) % 4], us[l], cv.CSC_BT601_LIMITED) for l in range(4)]) for t in range(4)]
            self.sdesc = [sv._image_desc(x) for x in self.src]
            self.uni = [(cv.Uniforms).from_buffer_copy(np.asarray(u, dtype=np.float32).tobytes()) for u in us]
            self.opts = cv.KernelOpts(cv.CSC_BT601_LIMITED)
            self.n = 0

        def tick_fused(self):
            t = self.n; self.n += 1
            h = self.ctx.handle
            lib.chv_pass_begin(h)
            lib.chv_composite(h, C.byref(self.tdesc[t % 10]), 1, self.layer_arr[t % 4], 4)
            lib.chv_pass_end(h, 1)

        def tick_sequence(self):
            t = self.n; self.n += 1
            h = self.ctx.handle
            td = C.byref(self.tdesc[t % 10])
            lib.chv_pass_begin(h)
            lib.chv_run_kernel(h, int(k_clear), td, None, 0, None, 0, 0, None)
            for l in range(4):
                lib.chv_run_kernel(h, int(k_layer), td, C.byref(self.sdesc[(t + l) % 4]), 1, C.byref(self.uni[l]), 236, 1, C.byref(self.opts))
            lib.chv_pass_end(h, 1)

    ctxs = [sv.createComputeContext(sharing=ctx) for _ in range(8)]
    ms = [Mixer(c) for c in ctxs]
    py = {}
    for mode in ("fused", "sequence"):
        py[mode] = {}
        for n in (1, 2, 4, 8):
            counts = [0] * n
            stop = [0.0]

            def worker(i, mode=mode):
                fn = getattr(ms[i], "tick_" + mode)
                k = 0
                while time.perf_counter() < stop[0]:
                    fn(); k += 1
                counts[i] = k
            for m in ms[:n]:
                for _ in range(10):
                    getattr(m, "tick_" + mode)()
            th = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
            t0 = time.perf_counter(); stop[0] = t0 + seconds
            for t in th:
                t.start()
            for t in th:
                t.join()
            py[mode][str(n)] = sum(counts) / (time.perf_counter() - t0)
    out["python"] = py
    for c in ctxs:
        sv.destroyComputeContext(c)
    for name in ("native", "python"):
        r = out.get(name, {})
        for mode in ("fused", "sequence"):
            if isinstance(r.get(mode), dict) and r[mode].get("1"):
                out[f"{name}_{mode}_speedup_8_threads"] = r[mode]["8"] / r[mode]["1"]
    return {"per_tick_thread_scaling": out}


def run_per_tick_mixer420(args, sv, cv, lib, ctx, fmt="y420p", seconds=0.3, ring=10, distinct=4):
    """The reference-default mixer tick (1080p 4:2:0 canvas <- full-canvas video + two 640x360 BGRA overlays: the `mixer_<fmt>` workload) issued
    ONE AT A TIME with the reference's host wait after it: fused (one chv_composite) and as the unchanged 1 + 3 launch sequence
    (img_clear_<fmt> + 3 x chv_run_kernel, mix.video.swift:116-124)."""
    import util
    dw, dh = 1920, 1080
    us = [util.full_canvas_uniforms((dw, dh), (dw, dh)),
          util.make_uniforms((dw, dh), rect=(64, 64, 640, 360), opacity=0.8, in_size=(640, 360)),
          util.make_uniforms((dw, dh), rect=(1200, 640, 640, 360), opacity=0.6, in_size=(640, 360))]
    PF = {"nv12": sv.PixelFormat.nv12, "y420p": sv.PixelFormat.y420p}[fmt]
    host = [util.alloc_image(fmt, dw, dh, seed=0x5EED0000 + 64 + i) for i in range(distinct)]
    ov = [util.alloc_image("bgra", 640, 360, seed=0x5EED0000 + 80 + i) for i in range(2)]
    K = sv.defaultComputeKernelFromString
    k_main, k_ov, k_clear = K(f"img_{fmt}_{fmt}"), K(f"img_bgra_{fmt}"), K(f"img_clear_{fmt}")
    src = [sv.uploadComputePicture(ctx, sv.pictureFromArrays(PF, (dw, dh), h), retainCpuBuffer=False) for h in host]
    gov = [sv.uploadComputePicture(ctx, sv.pictureFromArrays(sv.PixelFormat.BGRA, (640, 360), o), retainCpuBuffer=False) for o in ov]
    canvas = [sv.uploadComputePicture(ctx, sv.createPictureSample((dw, dh), PF), retainCpuBuffer=False) for _ in range(ring)]
    tdesc = [sv._image_desc(cn) for cn in canvas]
    layer_arr = [sv._layer_array([(k_main, src[t], us[0], 0), (k_ov, gov[0], us[1], 0), (k_ov, gov[1], us[2], 0)]) for t in range(distinct)]
    sdesc = [sv._image_desc(x) for x in src] + [sv._image_desc(x) for x in gov]
    uni = [(cv.Uniforms).from_buffer_copy(np.asarray(u, dtype=np.float32).tobytes()) for u in us]
    opts = cv.KernelOpts(0)
    h = ctx.handle
    state = {"n": 0}

    def tick_fused():
        t = state["n"]; state["n"] += 1
        lib.chv_pass_begin(h)
        rc = lib.chv_composite(h, C.byref(tdesc[t % ring]), 1, layer_arr[t % distinct], 3)
        rc |= lib.chv_pass_end(h, 1)
        if rc:
            cv.check(rc)

    def tick_sequence():
        t = state["n"]; state["n"] += 1
        td = C.byref(tdesc[t % ring])
        lib.chv_pass_begin(h)
        rc = lib.chv_run_kernel(h, int(k_clear), td, None, 0, None, 0, 0, None)
        rc |= lib.chv_run_kernel(h, int(k_main), td, C.byref(sdesc[t % distinct]), 1, C.byref(uni[0]), 236, 1, C.byref(opts))
        for i in (0, 1):
            rc |= lib.chv_run_kernel(h, int(k_ov), td, C.byref(sdesc[distinct + i]), 1, C.byref(uni[1 + i]), 236, 1, C.byref(opts))
        rc |= lib.chv_pass_end(h, 1)
        if rc:
            cv.check(rc)

    def timed(fn, secs):
        for _ in range(20):
            fn()
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(20):
                fn()
            n += 20
            el = time.perf_counter() - t0
            if el >= secs:
                return el / n * 1e6, n

    out = {}
    for mode, fn in (("fused", tick_fused), ("sequence", tick_sequence)):
        us_tick, n = timed(fn, seconds)
        out[mode] = {"us_per_tick": us_tick, "ticks": n}
    # fused == sequence on one tick, byte for byte (both against the same sources)
    state["n"] = 0; tick_fused()
    ga = [b.copy() for b in sv.downloadComputePicture(ctx, canvas[0], retainGpuBuffer=True).imageBuffer().buffers]
    state["n"] = 0; tick_sequence()
    gb = sv.downloadComputePicture(ctx, canvas[0], retainGpuBuffer=True).imageBuffer().buffers
    same = all(bool(np.array_equal(a, b)) for a, b in zip(ga, gb))
    desc = ("the reference-default mixer tick (1080p %s canvas <- 1080p %s video + two 640x360 BGRA overlays) issued ONE AT A TIME with a host "
            "wait after every tick: {}; Python host (ctypes), canvas ring of %d" % (fmt, fmt, ring))
    return {
        f"mixer_{fmt}_per_tick": dict(out["fused"], workload=desc.format("one chv_composite + chv_pass_end(wait)"), launches_per_tick=1,
                                      fused_equals_sequence=same, gpix_per_s=dw * dh / out["fused"]["us_per_tick"] / 1e3),
        f"mixer_{fmt}_reference_sequence": dict(out["sequence"], workload=desc.format(f"img_clear_{fmt} + 3 x chv_run_kernel + chv_pass_end(wait): an unchanged "
                                                                                      "mix.video.swift:116-124"),
                                                launches_per_tick=4, fused_equals_sequence=same, gpix_per_s=dw * dh / out["sequence"]["us_per_tick"] / 1e3),
    }


# Routes a batch can take, as switch settings (chv_debug_set_switch); every one gives the oracle's bytes (tests/test_gpu_fuzz.py forces each),
# so the only question is which is fastest — and whether the library's own choice (no switch set) is that one.
ROUTES_BGRA = [("stream_off", {"CHV_STREAM": "0"}), ("stream_forced", {"CHV_BGRA_PATH": "stream"}), ("tiled", {"CHV_BGRA_PATH": "tiled"}),
               ("strip_8_rows", {"CHV_BGRA_PATH": "wave", "CHV_WAVE_ROWS": "8"}), ("strip_16_rows", {"CHV_BGRA_PATH": "wave", "CHV_WAVE_ROWS": "16"})]
ROUTES_420 = [("yuv_stream_off_8_rows", {"CHV_YUV_STREAM": "0", "CHV_WAVE_ROWS": "8"}), ("yuv_stream_off_16_rows", {"CHV_YUV_STREAM": "0", "CHV_WAVE_ROWS": "16"}),
              ("yuv_stream_forced", {"CHV_YUV_STREAM": "force"})]


def run_route_regret(args, sv, cv, lib, ctx, names, seconds=0.12, rounds=2):
    """For each workload: the batch as the library routes it, and the same ticks (same device buffers, a new chv_batch_create under
    switches) through every other route that accepts them; ms per launch between stream events, routes interleaved, best of `rounds`.
    regret = t(chosen) / t(best) - 1.  The general kernels (the fallback of everything) are not a choice and are left out."""
    dev = HipDevice(cv, lib, ctx)
    out = {}
    for name in names:
        wl = WORKLOADS[name]
        w = build_workload(sv, ctx, wl, wl["frames"], seed_base=0x5EED0000 + 16 * 3)
        n = len(w["ticks"])
        routes = [("chosen", {}, w["batch"], w["kernel"], w.get("launches_per_batch", 1))]
        seen = {(w["kernel"], w.get("launches_per_batch", 1), None)}
        for label, sw in (ROUTES_420 if wl["kind"] == "mixer420" else ROUTES_BGRA):
            for k, v in sw.items():
                cv.set_switch(k, v)
            b = C.c_void_p()
            rc = lib.chv_batch_create(ctx.handle, w["ticks"], n, C.byref(b))
            for k in sw:
                cv.set_switch(k, None)
            if rc != 0:
                continue
            kname = C.create_string_buffer(128)
            nl = C.c_int(1)
            cv.check(lib.chv_batch_describe(b, kname, 128, C.byref(nl)))
            kn = kname.value.decode()
            rows = sw.get("CHV_WAVE_ROWS") if "wave" in kn else None           # (strip height only matters where a strip kernel runs)
            if "general" in kn or (kn, nl.value, rows) in seen or (rows is None and any(s[0] == kn and s[1] == nl.value for s in seen)):
                cv.check(lib.chv_batch_destroy(b))
                continue
            seen.add((kn, nl.value, rows))
            routes.append((label, sw, b, kn, nl.value))
        times = {r[0]: float("inf") for r in routes}
        e0, e1 = dev.event(), dev.event()
        for r in routes:                                                        # warm up every route once
            for k, v in r[1].items():
                cv.set_switch(k, v)
            cv.check(lib.chv_batch_run(ctx.handle, r[2]))
            for k in r[1]:
                cv.set_switch(k, None)
        dev.sync()
        for _ in range(rounds):
            for label, sw, b, kn, nl in routes:
                for k, v in sw.items():                                         # (strip and tile heights are picked per LAUNCH: the route's switches stay set while it runs)
                    cv.set_switch(k, v)
                reps, el = 2, 0.0
                while True:
                    dev.record(e0)
                    for _ in range(reps):
                        cv.check(lib.chv_batch_run(ctx.handle, b))
                    dev.record(e1)
                    dev.sync()
                    el = dev.elapsed_ms(e0, e1)
                    if el >= seconds * 1e3 or reps >= 4096:
                        break
                    reps = max(reps * 2, int(reps * seconds * 1e3 / max(el, 1e-3)) + 1)
                times[label] = min(times[label], el / reps)
                for k in sw:
                    cv.set_switch(k, None)
        dev.destroy(e0); dev.destroy(e1)
        for r in routes[1:]:
            cv.check(lib.chv_batch_destroy(r[2]))
        best = min(times, key=times.get)
        out[name] = {"chosen": routes[0][3], "chosen_ms": round(times["chosen"], 4), "best": best if best != "chosen" else "chosen",
                     "best_kernel": next(r[3] for r in routes if r[0] == best), "regret": round(times["chosen"] / times[best] - 1.0, 4),
                     "routes_ms": {r[0]: [round(times[r[0]], 4), r[3]] for r in routes}}
        free_workload(w)
    return out


def smi_parse(text):
    """(shader clock MHz, socket power W) of every JSON object rocm-smi printed into `text`"""
    got = []
    for line in text.splitlines():
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            c = next(iter(json.loads(line).values()))
            sclk = next((v for k, v in c.items() if k.startswith("sclk clock speed")), None)
            power = next((v for k, v in c.items() if "Package Power" in k), None)
            if sclk and power:
                got.append((float(sclk.strip("()").lower().replace("mhz", "")), float(power)))
        except Exception:       # noqa: BLE001
            pass
    return got


def smi_power_cap(device):
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout[r.stdout.index("{"):])
        return float(next(v for k, v in next(iter(j.values())).items() if "Max Graphics Package Power" in k))
    except Exception:       # noqa: BLE001
        return None


def run_power_probe(args, sv, cv, lib, ctx, names, device, samples=3, limit=8.0):
    """Shader clock and socket power WHILE each workload's batch runs back to back (outside every timed region; the sampler is one child
    process per workload: a short wait for the governor to settle, then `samples` rocm-smi readings).  On this part the path runs at the
    socket's power cap: a plain copy moves 5.5 TB/s at ~860 W and 2400 MHz, the composite kernels sit at 1400 W with the clock pulled
    down to 1.96-2.35 GHz (profiles/r05_notes.md section 10) — the roofline fraction of a kernel that is at neither its instruction floor
    nor its memory floor is set by the energy its instructions cost."""
    import shutil
    if shutil.which("rocm-smi") is None:
        return None
    dev = HipDevice(cv, lib, ctx)
    cap = smi_power_cap(device)
    cmd = "sleep 0.8; " + "; sleep 0.1; ".join([f"rocm-smi -d {int(device)} --showclocks --showpower --json"] * samples)
    out = {}
    for name in names:
        wl = WORKLOADS[name]
        w = build_workload(sv, ctx, wl, wl["frames"], seed_base=0x5EED0000 + 16 * 5)
        cv.check(lib.chv_batch_run(ctx.handle, w["batch"]))
        dev.sync()
        t0 = time.time()
        child = subprocess.Popen(["bash", "-c", cmd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        while child.poll() is None and time.time() - t0 < limit:
            for _ in range(4):
                cv.check(lib.chv_batch_run(ctx.handle, w["batch"]))
            dev.sync()
        if child.poll() is None:
            child.kill()
        text = child.communicate()[0]
        free_workload(w)
        got = smi_parse(text)
        if got:
            sclk, power = sum(g[0] for g in got) / len(got), sum(g[1] for g in got) / len(got)
            out[name] = {"sclk_mhz": round(sclk), "power_w": round(power), "power_cap_w": cap, "at_power_cap": bool(cap and power >= 0.97 * cap),
                         "kernel": w["kernel"], "probe_seconds": round(time.time() - t0, 2)}
    return out or None


def run_threads(args):
    """--gpus N --threads: N host threads in THIS process, thread r with its own compute context on device r (or --device); the threads meet at
    a barrier around the timed region exactly as the ranks of the process-per-GPU mode do, rank 0 prints the line"""
    import threading
    shared = {"barrier": threading.Barrier(args.gpus), "slots": [0.0] * args.gpus}
    errors = []

    def worker(r):
        try:
            run_rank(args, r, r, args.gpus, ThreadDist(shared, r))
        except BaseException as e:    # noqa: BLE001
            errors.append((r, e))
            shared["barrier"].abort()

    th = [threading.Thread(target=worker, args=(r,), name=f"device-{r}") for r in range(args.gpus)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
    if real:
        raise SystemExit(f"thread of device {real[0][0]} failed: {real[0][1]!r}")


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.threads and args.gpus > 1:
        args.also = "none"
        return run_threads(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    rank, local, world, dist = dist_setup()
    run_rank(args, rank, local, world, dist)


def run_rank(args, rank, local, world, dist):
    if world != args.gpus:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    n_gpus = max(world, 1)

    if args.stub_device:
        sv = cv = lib = ctx = None
        tm = Timer(StubDevice(), dist)
    else:
        from swiftvideo_amd import chipvideo as cv
        from swiftvideo_amd import compute as sv
        lib = cv.load()
        dev = pick_device(args, local, cv.device_count())
        ctx = sv.makeComputeContext(forType="GPU", index=dev)
        tm = Timer(HipDevice(cv, lib, ctx), dist)
    do = (lambda name, headline: measure_stub(name, args, tm, rank, n_gpus, headline)) if args.stub_device else \
         (lambda name, headline: measure(name, args, sv, cv, lib, ctx, tm, rank, n_gpus, headline))

    if args.with_upload:
        rep = run_with_upload_stub(args, tm, rank, n_gpus) if args.stub_device else \
            run_with_upload(args, sv, cv, lib, ctx, tm, rank, n_gpus, group=args.upload_group, streams=args.upload_streams)
        if rank == 0:
            print(compact_line({"metric": METRIC, "value": rep["value"], "unit": "Gpix/s", "n_gpus": n_gpus, "steps": args.steps,
                                "warmup": args.warmup, "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                                "config": {"workload": rep["workload"], "mode": "END-TO-END (not the headline mode)",
                                           "launches_per_step": rep["launches_per_step"], "h2d_GBps_per_gpu": rep["h2d_GBps_per_gpu"],
                                           "h2d_frac_of_link": rep["h2d_frac_of_link"], "upload_copy_MB": rep["upload_copy_MB"], "upload_streams": rep["upload_streams"],
                                           "pinned_numa_node": rep["pinned_numa_node"], "verified_vs_oracle": rep["verified_vs_oracle"],
                                           "per_rank": rep["per_rank"], "frames_per_launch_per_gpu": rep["frames_per_launch_per_gpu"],
                                           "frames_per_step_per_gpu": rep["frames_per_launch_per_gpu"] * rep["launches_per_step"],
                                           "kernel": rep["kernel"]}}), flush=True)
        tm.barrier()
        if dist is not None and not isinstance(dist, ThreadDist):
            dist.destroy_process_group()
        return

    head, cpu = do(args.workload, True)
    if args.also is None:
        others = [n for n in DEFAULT_SET if n != args.workload] if args.workload == HEADLINE else []
    else:
        others = [n for n in args.also.split(",") if n and n != "none"]
    reports = {args.workload: head}
    for name in others:
        reports[name], _ = do(name, False)
    # ---- the legs: behind --full (or their own switch); none of them feeds `value`
    real = not args.stub_device
    if args.full and others and real and not args.no_upload_leg:
        reports["cfg2_upload"] = run_with_upload(args, sv, cv, lib, ctx, tm, rank, n_gpus, group=args.upload_group, streams=args.upload_streams)
        reports["pipeline_e2e"] = run_e2e(args, sv, cv, lib, ctx, tm, rank, n_gpus, group=args.upload_group, streams=args.upload_streams)
    if ((args.full and others) or args.per_tick) and real and not args.no_per_tick and n_gpus == 1:
        reports.update(run_per_tick(args, sv, cv, lib, ctx))
        reports.update(run_per_tick_mixer420(args, sv, cv, lib, ctx, fmt="y420p"))
        reports.update(run_thread_scaling(args, sv, cv, lib, ctx))
        reports.update(run_group_tick_fresh(cv, lib, ctx))
    regret = None
    if ((args.full and others) or args.route_regret) and real and not args.no_route_regret and n_gpus == 1:
        regret = run_route_regret(args, sv, cv, lib, ctx, [n for n in ([args.workload] + others) if n in WORKLOADS])
    power = None
    if real and not args.no_power_probe and n_gpus == 1 and (others or args.power_probe):
        # the default command samples the headline's clock / power only (1.5 s); --full every workload's
        probed = [n for n in ([args.workload] + (others if (args.full or args.power_probe) else [])) if n in WORKLOADS]
        power = run_power_probe(args, sv, cv, lib, ctx, probed, args.device if args.device is not None else 0)

    if rank == 0:
        detail = build_detail(args, dist, n_gpus, head, cpu, reports, regret, power, None if args.stub_device else cv.build_flags())
        line = compact_line(detail)
        path = None if args.detail_json == "none" else Path(args.detail_json or ROOT / "bench_detail.json")
        if path is not None:
            try:
                path.write_text(json.dumps(detail, indent=1) + "\n")
                print(f"bench: full report (every workload's record, legs, prose) -> {path}", file=sys.stderr, flush=True)
            except OSError as e:
                print(f"bench: could not write {path}: {e}", file=sys.stderr, flush=True)
        print(line, flush=True)
    tm.barrier()
    if dist is not None and not isinstance(dist, ThreadDist):
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# the report: bench_detail.json (everything) and the ONE compact stdout line (what the driver parses)
# ---------------------------------------------------------------------------------------------------------------
# What bounds each workload's kernel — named by the probes, not by the roofline's name (SURVEY 8d fixes the roofline at HBM):
#   valu_issue  the kernel's arithmetic alone takes >= 90 % of the shipped launch at full clock (CHV_ST_ABL / CHV_ABL halves:
#               profiles/r05_notes.md 10.4; mixer_y420p: section 4 + r06_notes.md — 128 VALU per overlay row, LDS pipe idle, below the cap)
#   power_cap   at the socket's 1400 W with the shader clock pulled >= 5 % below its 2.4 GHz maximum (r05_notes.md section 10)
#   latency     below the cap at full clock, staging alone 1.2-1.7x its bytes' copy time (r05_notes.md 10.5)
# A live clock / power sample (the headline in the default run, every workload under --full) overrides the table where it shows the cap.
LIMITER_TABLE = {"pipeline": "valu_issue", "pipeline_y420p": "valu_issue", "pipeline_logo": "valu_issue", "cfg2": "valu_issue", "cfg2_y420p": "valu_issue",
                 "cfg3": "power_cap", "cfg5": "power_cap", "mixer_y420p": "valu_issue", "mixer_nv12": "valu_issue", "y420p_main": "valu_issue",
                 "encode_nv12": "valu_issue", "pipeline_grid": "latency", "mixed": "latency"}
SCLK_MAX_MHZ = 2400.0
# the headline kernel's additive issue model (profiles/r03_notes.md section 1, r05_notes.md 10.4, r06_notes.md section 11): per 64-pixel row of a
# 4-layer tick one wave issues 54 full-rate + 95 half-rate vector instructions + 50 LDS instructions (83 + 95 + 48 until round 6 took 29
# full-rate ones out) at 1.1 / 1.8 / 1.05 ns; 256 CUs x 4 SIMDs issue in parallel
ISSUE_MODEL = {"tick_bgra_stream": {"layers": 4, "fast": 54, "slow": 95, "lds": 50, "ns": (1.1, 1.8, 1.05), "simds": 1024}}


def limiter_of(name, probe):
    lim = LIMITER_TABLE.get(name)
    if probe and probe.get("at_power_cap") and probe.get("sclk_mhz", SCLK_MAX_MHZ) <= 0.95 * SCLK_MAX_MHZ and lim != "valu_issue":
        lim = "power_cap"
    return lim


def issue_model_ms(kernel, wl, frames):
    m = ISSUE_MODEL.get(kernel)
    if m is None or wl.get("layers") != m["layers"] or wl["kind"] != "yuv_layers" or wl.get("logo"):
        return None
    row_ns = m["fast"] * m["ns"][0] + m["slow"] * m["ns"][1] + m["lds"] * m["ns"][2]
    wave_rows = frames * wl["dh"] * math.ceil(wl["dw"] / 64)
    return wave_rows / m["simds"] * row_ns * 1e-6


def build_detail(args, dist, n_gpus, head, cpu, reports, regret, power, build_flags):
    """Everything the run measured, as one dict (bench_detail.json)."""
    wl = WORKLOADS[args.workload]
    roof = dict(head["roofline"])
    probe = (power or {}).get(args.workload)
    if probe:
        # (the clock the kernel actually ran at and the power the socket drew meanwhile — sampled in a leg of its own after the timed regions)
        roof.update({k: probe[k] for k in ("sclk_mhz", "power_w", "power_cap_w", "at_power_cap")})
    roof.update({"limiter": limiter_of(args.workload, probe), "traffic": None, "traffic_ratio": None, "traffic_source": None, "kernel": head["kernel"],
                 "launch_ms": head["launch_ms"], "issue_model_ms": issue_model_ms(head["kernel"], wl, head["frames_per_launch_per_gpu"]),
                 "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"]})
    # HBM traffic cannot be counted inside this process (PMC needs rocprofv3 around it, in passes of their own): two short child passes of
    # this script after the timed regions; where rocprofv3 is missing, the committed measurement of the same workload — and the field says which
    pmc_path = Path(args.pmc_json) if args.pmc_json else ROOT / "profiles" / "pmc_latest.json"
    live_err = None
    # (the default run only — what the driver times; `--also …` runs are the builder's A/Bs and profiling passes)
    if n_gpus == 1 and args.also is None and not args.no_live_pmc and not args.stub_device and args.alias == "none" and not args.pmc_json:
        kname = head["kernel"].split("<")[0]
        t, live_err = live_traffic(args.workload, head["frames_per_launch_per_gpu"], kname, args.device if args.device is not None else 0)
        if t is not None:
            roof["traffic"] = t
            roof["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in two separate child passes of "
                                      f"`bench.py --workload {args.workload}` (3 launches each), FETCH_SIZE x 2 (gfx950), KB = 1024 B, kernel {kname}")
    if roof["traffic"] is None and pmc_path.exists():
        try:
            j = json.loads(pmc_path.read_text())
            if j.get("workload") == args.workload and j.get("frames") == head["frames_per_launch_per_gpu"]:
                roof["traffic"] = j.get("hbm_bytes_per_launch")
                roof["traffic_source"] = (f"NOT measured in this run: {pmc_path.name}, rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE "
                                          f"in separate passes of `bench.py --workload {args.workload}` (profiles/collect_round.sh), kernel {j.get('kernel')}")
                if live_err:
                    roof["traffic_source"] += f" (live measurement unavailable: {live_err})"
        except Exception as e:    # noqa: BLE001
            roof["traffic_source"] = f"unreadable {pmc_path}: {e}"
    if roof["traffic"]:
        roof["traffic_ratio"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
    # --full: the counted traffic of cfg5 as well — the one workload that moves more than its algorithmic bytes (the 2160p canvases travel out
    # and back between the composite and the resize): both kernels of the pair, per batch
    if args.full and "cfg5" in reports and n_gpus == 1 and not args.no_live_pmc and not args.stub_device and args.workload != "cfg5":
        t5, err5 = live_traffic("cfg5", reports["cfg5"]["frames_per_launch_per_gpu"], ["tick_bgra_wave", "lanczos3"], args.device if args.device is not None else 0)
        r5 = reports["cfg5"]["roofline"]
        r5["traffic"], r5["traffic_error"] = t5, err5
        r5["traffic_ratio"] = None if t5 is None else t5 / reports["cfg5"]["algorithmic_bytes_per_launch"]
    for k, v in reports.items():          # every workload's own limiter beside its fraction
        if "roofline" in v:
            v["roofline"]["limiter"] = limiter_of(k, (power or {}).get(k))
    out = {
        "metric": METRIC, "value": head["value"], "unit": "Gpix/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": head["workload"], "workload_short": head["workload_short"], "launches_per_step": head["launches_per_step"],
                   "frames_per_launch_per_gpu": head["frames_per_launch_per_gpu"],
                   "frames_per_step_per_gpu": head["frames_per_launch_per_gpu"] * head["launches_per_step"],
                   "timed_seconds": head["timed_seconds"], "gpix_counts": "target pixels written",
                   "source_mpix_per_launch_per_gpu": head["source_mpix_per_launch_per_gpu"],
                   "parallelism": (f"ONE process, {n_gpus} host threads, a compute context per device" if isinstance(dist, ThreadDist) else
                                   f"{n_gpus} process(es), one per GPU") + f"; {head['frames_per_launch_per_gpu']} independent picture "
                                  f"buses per device, bus s -> device s mod {n_gpus}; no collective",
                   "per_gpu_gpix": head["per_gpu_gpix"], "per_gpu_launch_ms": head["per_gpu_launch_ms"], "per_stream_ticks_per_s": head["per_stream_ticks_per_s"],
                   "kernel": head["kernel"], "verified_vs_oracle": head["verified_vs_oracle"], "build_flags": build_flags, "full": bool(args.full),
                   # every workload of the run: name -> [fraction of the 8 TB/s HBM peak, ms per launch, kernel]
                   "workload_fracs": {k: [round(v["roofline"]["frac"], 4), round(v["launch_ms"], 4), v["kernel"]] for k, v in reports.items() if "roofline" in v},
                   "workload_limiters": {k: v["roofline"]["limiter"] for k, v in reports.items() if "roofline" in v},
                   "legs": {**{k + "_us_per_tick": round(v["us_per_tick"], 2) for k, v in reports.items() if "us_per_tick" in v},
                            **({"pipeline_e2e_ticks_per_s": round(reports["pipeline_e2e"]["ticks_per_s"], 1),
                                "pipeline_e2e_h2d_GBps": round(reports["pipeline_e2e"]["h2d_GBps_per_gpu"], 2)} if "pipeline_e2e" in reports else {}),
                            **({"cfg2_upload_h2d_GBps": round(reports["cfg2_upload"]["h2d_GBps_per_gpu"], 2),
                                "pinned_numa_node": reports["cfg2_upload"]["pinned_numa_node"]} if "cfg2_upload" in reports else {}),
                            **({"native_fused_speedup_8_threads": round(reports["per_tick_thread_scaling"]["native_fused_speedup_8_threads"], 2)}
                               if "per_tick_thread_scaling" in reports and "native_fused_speedup_8_threads" in reports["per_tick_thread_scaling"] else {})},
                   # is the kernel the library picks for each workload the fastest of the routes that accept it?  regret = t(chosen) / t(best) - 1
                   "route_regret": None if regret is None else {k: {"chosen": v["chosen"], "best": v["best"], "regret": v["regret"]} for k, v in regret.items()},
                   "route_regret_max": None if not regret else max(v["regret"] for v in regret.values()),
                   # name -> [shader clock MHz, socket power W] while the workload's batch runs back to back (cap: roofline.power_cap_w)
                   "workload_power": None if not power else {k: [v["sclk_mhz"], v["power_w"]] for k, v in power.items()}},
        "roofline": roof,
        "workloads": {k: {kk: vv for kk, vv in v.items() if kk not in ("source_mpix_per_launch_per_gpu",)} for k, v in reports.items()},
    }
    if regret is not None:
        out["workloads"]["route_regret"] = regret
    if power:
        out["workloads"]["power_probe"] = power
    if args.stub_device:
        out["data"] = "STUB --stub-device: launches are sleeps; control-plane self-test, not a benchmark result"
        out["roofline"]["frac"] = None
    out["config"]["content"] = args.content
    if args.content != "random":
        out["data"] = f"synthetic, DIAGNOSTIC --content {args.content} (the benchmark's content is uniform random bytes)"
    if args.alias != "none":
        out["data"] = f"DIAGNOSTIC --alias {args.alias}: ticks share frame 0's buffers, cache-resident traffic; not a benchmark result"
        out["roofline"]["frac"] = None
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


COMPACT_LIMIT = 4096      # bytes: the driver keeps a bounded tail of stdout and parses its last line (BENCH_r05.json: a 20 KB line did not parse)
STRING_LIMIT = 120


def _short(text, limit=STRING_LIMIT):
    text = str(text)
    return text if len(text) <= limit else text[: limit - 1].rstrip() + "…"


def _num(v, digits=5):
    """a float with `digits` significant digits (the full precision stays in bench_detail.json)"""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None                     # never NaN / Infinity in the line: they are not JSON
        return float(f"{v:.{digits}g}")
    if isinstance(v, (list, tuple)):
        return [_num(x, digits) for x in v]
    return v


def compact_line(d):
    """The ONE line the driver parses: the contract's keys + roofline + cpu_baseline + numeric per-workload fractions, < COMPACT_LIMIT bytes,
    no string longer than STRING_LIMIT, no NaN / Infinity.  Everything else lives in bench_detail.json."""
    cfg, roof = d["config"], d.get("roofline")
    c = {"workload": _short(cfg.get("workload_short") or cfg["workload"]), "launches_per_step": cfg.get("launches_per_step"),
         "frames_per_launch_per_gpu": cfg.get("frames_per_launch_per_gpu"), "kernel": _short(cfg.get("kernel")),
         "verified_vs_oracle": cfg.get("verified_vs_oracle"), "parallelism": _short(cfg.get("parallelism", "")),
         "per_gpu_gpix": _num(cfg.get("per_gpu_gpix")), "per_gpu_launch_ms": _num(cfg.get("per_gpu_launch_ms"))}
    for k in ("mode", "h2d_GBps_per_gpu", "h2d_frac_of_link", "upload_copy_MB", "upload_streams", "pinned_numa_node", "frames_per_step_per_gpu"):
        if k in cfg and cfg.get("mode"):            # (--with-upload lines)
            c[k] = _num(cfg[k]) if not isinstance(cfg[k], str) else _short(cfg[k])
    if cfg.get("per_rank"):
        c["per_rank"] = {k: _num(v) for k, v in cfg["per_rank"].items()}
    if cfg.get("workload_fracs"):
        c["workload_fracs"] = {k: [_num(v[0], 4), _num(v[1], 4)] for k, v in cfg["workload_fracs"].items()}       # name -> [frac of 8 TB/s, ms per launch]
        c["workload_limiters"] = cfg.get("workload_limiters")
    c["full"] = cfg.get("full", False)
    if cfg.get("content", "random") != "random":
        c["content"] = cfg["content"]
    r = None
    if roof is not None:
        keys = ("bound", "limiter", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "kernel", "launch_ms", "issue_model_ms",
                "algorithmic_bytes_per_launch", "sclk_mhz", "power_w", "power_cap_w")
        r = {k: (_short(roof[k]) if isinstance(roof.get(k), str) else _num(roof.get(k))) for k in keys if k in roof}
        if roof.get("traffic_source"):
            r["traffic_measured_in_this_run"] = roof["traffic_source"].startswith("measured in this run")
    out = {k: (_num(d[k]) if not isinstance(d[k], str) else d[k]) for k in
           ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype")}
    out["data"] = _short(d["data"])
    out["config"] = c
    if r is not None:
        out["roofline"] = r
    if "cpu_baseline" in d:
        cb = d["cpu_baseline"]
        out["cpu_baseline"] = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb["sample"])}
    line = json.dumps(out, separators=(",", ":"), allow_nan=False)
    # should a future key push the line over the limit, optional keys go first — the contract's keys, roofline and cpu_baseline never
    for victim in ("workload_limiters", "per_rank", "workload_fracs", "parallelism"):
        if len(line.encode()) < COMPACT_LIMIT:
            break
        out["config"].pop(victim, None)
        line = json.dumps(out, separators=(",", ":"), allow_nan=False)
    assert len(line.encode()) < COMPACT_LIMIT, len(line.encode())
    return line


if __name__ == "__main__":
    main()
