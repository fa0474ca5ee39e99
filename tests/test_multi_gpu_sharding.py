"""The N>1 control path of bench.py on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.
The data path has no collective (streams are independent), so what needs covering is the
barrier + max-over-ranks timing reduction and the stream -> device binding."""
import os
import socket
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dist.barrier()
    elapsed = 0.5 if rank == 0 else 0.8            # rank 1 is the slow one
    mx = bench.reduce_max(dist, elapsed)
    dist.barrier()
    q.put((rank, mx, [s for s in range(8) if bench.stream_to_device(s, world) == rank]))
    dist.destroy_process_group()


def test_two_rank_timing_reduction_and_stream_binding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0.8, 0.8]                       # every rank sees the slowest rank's time
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5, 7]  # stream s -> device s mod N, disjoint cover
    import bench
    # whole-job throughput: all ranks' pixels over the max time
    assert bench.whole_job_gpix(2, 1e9, 4, 0.8) == pytest.approx(2 * 1e9 * 4 / 0.8 / 1e9)
    assert bench.reduce_max(None, 1.25) == 1.25


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _strings(obj):
    if isinstance(obj, str):
        yield obj
    elif isinstance(obj, dict):
        for k, v in obj.items():
            yield k
            yield from _strings(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _strings(v)


def check_compact_line(raw):
    """The contract of the LAST stdout line (a 20 KB line did not reach the driver's parser): one JSON object, under 4 KB,
    the contract's keys, no prose string longer than 120 characters (the metric string is BASELINE.json's), no NaN / Infinity, no per-route or
    per-workload record tables."""
    import json
    assert len(raw.encode()) < 4096, len(raw.encode())
    assert "NaN" not in raw and "Infinity" not in raw
    d = json.loads(raw)
    assert all(k in d for k in REQUIRED), [k for k in REQUIRED if k not in d]
    assert d["unit"] == "Gpix/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workloads" not in d and "build_flags" not in d["config"] and "route_regret" not in d["config"]
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    long_ones = [s for s in _strings({k: v for k, v in d.items() if k != "metric"}) if len(s) > 120]
    assert not long_ones, long_ones
    return d


def _run_bench(argv, timeout=600, detail=False):
    """runs bench.py; returns the parsed compact line (and, with detail=True, the bench_detail.json of the run as a second value)"""
    import json
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    with tempfile.TemporaryDirectory(prefix="chv_bench_") as tmp:
        path = os.path.join(tmp, "detail.json")
        p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--detail-json", path] + argv, env=env, capture_output=True, text=True, timeout=timeout)
        assert p.returncode == 0, p.stderr[-2000:]
        out = p.stdout.rstrip("\n").splitlines()
        lines = [l for l in out if l.startswith("{")]
        assert len(lines) == 1 and out[-1] == lines[0], f"exactly one JSON line, the last one, expected; got {len(lines)}: {p.stdout[-500:]}"
        d = check_compact_line(lines[0])
        if not detail:
            return d
        return d, json.loads(Path(path).read_text())


def test_self_launch_spawns_one_rank_per_gpu_and_reports_the_slowest():
    """`python bench.py --gpus 2` with no launcher environment: the script starts its two ranks itself (the real spawn,
    rendezvous, calibration, barrier and reduction code), here with --stub-device (launches are sleeps; rank 1 is 25 % slower)."""
    d = _run_bench(["--gpus", "2", "--stub-device", "--steps", "4", "--warmup", "1", "--min-seconds", "0.3",
                    "--min-seconds-other", "0.05", "--also", "cfg2"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["data"].startswith("STUB") and d["roofline"]["frac"] is None            # never mistaken for a measurement
    cfg = d["config"]
    assert len(cfg["per_gpu_gpix"]) == 2
    assert cfg["per_gpu_gpix"][0] >= cfg["per_gpu_gpix"][1] * 0.95                    # rank 1 sleeps longer per launch
    # whole job = both ranks' pixels over the slowest rank's time: not more than twice the slower rank's own rate
    assert d["value"] <= 2.0 * min(cfg["per_gpu_gpix"]) * 1.02
    assert d["value"] >= 2.0 * min(cfg["per_gpu_gpix"]) * 0.7
    assert d["ms_per_step"] * d["steps"] >= 300 * 0.9                                 # the calibrated region lasts >= --min-seconds
    assert set(cfg["workload_fracs"]) == {"pipeline", "cfg2"} and all(len(v) == 2 for v in cfg["workload_fracs"].values())
    assert "cpu_baseline" not in d                                                    # N = 1 only


def test_one_process_with_a_host_thread_per_device():
    """`--gpus 2 --threads`: ONE process, two host threads, a context per device — the shape of the Swift host (a composer with mixers bound to
    devices, composer.swift:203-224, SURVEY section 8e "one host feeder thread per device"); the same barrier + slowest-thread reduction as the
    process-per-GPU mode, through threading primitives (here with --stub-device: thread 1 sleeps 25 % longer per launch)"""
    d, full = _run_bench(["--gpus", "2", "--threads", "--stub-device", "--steps", "4", "--warmup", "1", "--min-seconds", "0.3"], detail=True)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["data"].startswith("STUB")
    cfg = d["config"]
    assert cfg["parallelism"].startswith("ONE process, 2 host threads")
    assert len(cfg["per_gpu_gpix"]) == 2 and cfg["per_gpu_gpix"][0] >= cfg["per_gpu_gpix"][1] * 0.95
    assert 2.0 * min(cfg["per_gpu_gpix"]) * 0.7 <= d["value"] <= 2.0 * min(cfg["per_gpu_gpix"]) * 1.02
    assert set(full["workloads"]) == {"pipeline"} and set(cfg["workload_fracs"]) == {"pipeline"}
    import bench
    import threading
    shared = {"barrier": threading.Barrier(1), "slots": [0.0]}
    assert bench.reduce_max(bench.ThreadDist(shared, 0), 0.75) == 0.75 and bench.gather_floats(bench.ThreadDist(shared, 0), 0.5, 1) == [0.5]


def test_rank_without_a_device_is_an_error_not_a_silent_single_rank_run():
    import bench
    args = bench.parse_args(["--gpus", "2"])
    assert bench.pick_device(args, 0, 1) == 0
    with pytest.raises(SystemExit):
        bench.pick_device(args, 1, 1)                      # LOCAL_RANK 1 on a 1-GPU box
    args = bench.parse_args(["--gpus", "2", "--device", "0"])
    assert bench.pick_device(args, 1, 1) == 0              # --device pins every rank (1-GPU boxes)


@pytest.mark.gpu
def test_two_ranks_on_one_device_end_to_end():
    """The N > 1 data path for real, on the one GPU of the test box: `--gpus 2 --device 0` self-launches two ranks, each with
    its own context, frames and canvases on device 0; every rank's output is verified against the oracle by rank 0's twin
    logic (verified_vs_oracle) and the line reports both ranks."""
    d = _run_bench(["--gpus", "2", "--device", "0", "--steps", "3", "--warmup", "1", "--min-seconds", "0.2", "--frames", "32",
                    "--also", "none", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and len(d["config"]["per_gpu_gpix"]) == 2
    assert d["config"]["verified_vs_oracle"] is True
    assert d["config"]["kernel"] == "tick_bgra_stream"       # (32 four-layer 720p ticks per rank: a launch that fills the chip)
    assert 0 < d["roofline"]["frac"] < 1
    assert d["value"] <= sum(d["config"]["per_gpu_gpix"]) * 1.01


@pytest.mark.gpu
def test_two_host_threads_in_one_process_on_one_device():
    """`--gpus 2 --device 0 --threads`: two contexts of ONE process (each chv_context_create makes its own stream and descriptor rings) driven
    from two host threads — every thread's canvases verified against the oracle, both reported"""
    d = _run_bench(["--gpus", "2", "--device", "0", "--threads", "--steps", "3", "--warmup", "1", "--min-seconds", "0.2", "--frames", "32",
                    "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and len(d["config"]["per_gpu_gpix"]) == 2
    assert d["config"]["parallelism"].startswith("ONE process, 2 host threads")
    assert d["config"]["verified_vs_oracle"] is True and d["config"]["kernel"] == "tick_bgra_stream"
    assert 0 < d["roofline"]["frac"] < 1 and d["value"] <= sum(d["config"]["per_gpu_gpix"]) * 1.01


@pytest.mark.gpu
def test_two_ranks_with_uploads_on_one_device():
    """the end-to-end (H2D-inclusive) mode at N = 2: per-rank pinned frames, side-stream uploads, per-buffer events"""
    d = _run_bench(["--gpus", "2", "--device", "0", "--with-upload", "--steps", "3", "--warmup", "2", "--min-seconds-other", "0.2"])
    assert d["n_gpus"] == 2 and d["config"]["h2d_GBps_per_gpu"] > 1.0
    assert "END-TO-END" in d["config"]["mode"]
    pr = d["config"]["per_rank"]            # which rank bent the curve, and where its pinned ring sat
    assert len(pr["h2d_GBps"]) == len(pr["launch_ms"]) == len(pr["pinned_numa_node"]) == len(pr["cpus_bound"]) == 2
    assert all(v > 1.0 for v in pr["h2d_GBps"]) and all(v > 0 for v in pr["launch_ms"]) and all(c >= 1 for c in pr["cpus_bound"])


DEFAULT_NAMES = {"pipeline", "cfg2", "cfg3", "cfg5", "mixer_y420p", "encode_nv12", "pipeline_y420p", "pipeline_grid", "pipeline_logo"}
LIMITERS = {"valu_issue", "power_cap", "latency"}


def test_the_last_stdout_line_of_the_default_command_is_compact():
    """The driver's command shape without a GPU (--stub-device): the last stdout line parses, is under 4 KB, carries the contract's keys, the
    roofline object with the limiter the probes name, and every default workload's [fraction, ms]; the tables live in bench_detail.json."""
    d, full = _run_bench(["--stub-device", "--steps", "2", "--warmup", "1", "--min-seconds", "0.1", "--min-seconds-other", "0.02"], detail=True)
    cfg, r = d["config"], d["roofline"]
    assert set(cfg["workload_fracs"]) == DEFAULT_NAMES and all(len(v) == 2 and all(isinstance(x, float) for x in v) for v in cfg["workload_fracs"].values())
    assert set(cfg["workload_limiters"]) == DEFAULT_NAMES and set(cfg["workload_limiters"].values()) <= LIMITERS
    assert cfg["full"] is False and len(cfg["per_gpu_gpix"]) == len(cfg["per_gpu_launch_ms"]) == 1
    for k in ("bound", "limiter", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "kernel", "launch_ms", "issue_model_ms", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["limiter"] == "valu_issue" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    # the detail file: the same headline, full precision, every workload's record, prose allowed
    assert full["value"] == pytest.approx(d["value"], rel=1e-4) and set(full["workloads"]) == DEFAULT_NAMES
    assert full["config"]["workload"].startswith("pipeline: 4 x 1920x1080 NV12 streams")
    assert all("roofline" in v and v["roofline"]["limiter"] in LIMITERS for v in full["workloads"].values())


def test_the_compact_line_at_eight_devices_keeps_its_per_gpu_arrays():
    """what a SCALE record will carry: --gpus 8 (threads on stub devices here) -> eight per-GPU rates and launch times in a line still < 4 KB"""
    d = _run_bench(["--gpus", "8", "--threads", "--stub-device", "--steps", "2", "--warmup", "1", "--min-seconds", "0.1"])
    assert d["n_gpus"] == 8 and len(d["config"]["per_gpu_gpix"]) == len(d["config"]["per_gpu_launch_ms"]) == 8
    d = _run_bench(["--gpus", "2", "--device", "0", "--stub-device", "--steps", "2", "--warmup", "1", "--min-seconds", "0.1", "--min-seconds-other", "0.02"])
    assert d["n_gpus"] == 2 and len(d["config"]["per_gpu_gpix"]) == len(d["config"]["per_gpu_launch_ms"]) == 2
    assert set(d["config"]["workload_fracs"]) == DEFAULT_NAMES


def test_the_compact_line_never_carries_nan_and_sheds_optional_keys_first():
    import json
    import bench
    d = {"metric": bench.METRIC, "value": float("nan"), "unit": "Gpix/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": float("inf"),
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
         "config": {"workload": "w" * 500, "kernel": "k", "per_gpu_gpix": [1.0], "per_gpu_launch_ms": [1.0], "parallelism": "p" * 300,
                    "workload_fracs": {f"workload_{i}": [0.123456789, 1.23456789, "kernel"] for i in range(200)},
                    "workload_limiters": {f"workload_{i}": "valu_issue" for i in range(200)}},
         "roofline": {"bound": "hbm", "limiter": "valu_issue", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.000125, "traffic": None},
         "cpu_baseline": {"value": 0.07, "unit": "Gpix/s", "cores": 8, "kind": "port", "sample": "s" * 400}}
    out = check_compact_line(bench.compact_line(d))
    assert out["value"] is None and out["ms_per_step"] is None                       # not JSON otherwise
    assert "workload_fracs" not in out["config"] and out["roofline"]["frac"] == 0.000125 and out["cpu_baseline"]["cores"] == 8
    assert json.dumps(out)


@pytest.mark.gpu
def test_default_run_measures_its_hbm_traffic():
    """The driver's command shape (no --also, no --full): every workload of the default set timed and verified, roofline.traffic measured in
    the run itself by two rocprofv3 --pmc child passes (or, where rocprofv3 is missing, the committed figure — and the line says which), the
    headline's clock and power sampled; the last stdout line compact (check_compact_line), everything else in bench_detail.json."""
    import shutil
    d, full = _run_bench(["--steps", "2", "--warmup", "1", "--min-seconds", "0.1", "--min-seconds-other", "0.05", "--cpu-seconds", "1.5"], detail=True)
    assert set(full["workloads"]) - {"power_probe"} == DEFAULT_NAMES                  # no legs in the default command
    assert all(v["verified_vs_oracle"] is True for k, v in full["workloads"].items() if k in DEFAULT_NAMES)
    assert full["config"]["build_flags"].startswith("arch=gfx950;") and "abl=0" in full["config"]["build_flags"]
    cfg = d["config"]
    assert cfg["verified_vs_oracle"] is True and cfg["kernel"] == "tick_bgra_stream" and cfg["full"] is False
    assert set(cfg["workload_fracs"]) == DEFAULT_NAMES and all(0 < f[0] < 1 and f[1] > 0 for f in cfg["workload_fracs"].values())
    assert set(cfg["workload_limiters"]) == DEFAULT_NAMES and set(cfg["workload_limiters"].values()) <= LIMITERS
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["limiter"] in LIMITERS and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert 0.5 * r["launch_ms"] < r["issue_model_ms"] < 1.5 * r["launch_ms"]          # the additive issue model of the headline kernel, beside what was measured
    if shutil.which("rocm-smi") and "sclk_mhz" in r:
        assert 500 <= r["sclk_mhz"] <= 3000 and 100 <= r["power_w"] <= 2000
        assert r["power_cap_w"] is None or r["power_w"] <= 1.05 * r["power_cap_w"]
    if shutil.which("rocprofv3"):
        assert r["traffic_measured_in_this_run"] is True, full["roofline"]["traffic_source"]
        assert 0.97 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.10 * r["algorithmic_bytes_per_launch"], r
        assert r["traffic_ratio"] == pytest.approx(r["traffic"] / r["algorithmic_bytes_per_launch"], rel=1e-3)
    else:
        assert r["traffic"] is None or r["traffic_measured_in_this_run"] is False
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Gpix/s"


@pytest.mark.gpu
def test_full_run_carries_every_leg_in_the_detail_file():
    """`--full`: the upload-inclusive and end-to-end legs, the one-tick-at-a-time legs, thread scaling, route regret and every workload's clock /
    power — all in bench_detail.json; the stdout line stays the compact one."""
    import shutil
    d, full = _run_bench(["--full", "--steps", "2", "--warmup", "1", "--min-seconds", "0.1", "--min-seconds-other", "0.05", "--no-cpu-baseline", "--no-live-pmc"],
                         detail=True)
    assert d["config"]["full"] is True
    w = full["workloads"]
    assert set(w) >= DEFAULT_NAMES
    e2e = w["pipeline_e2e"]
    assert e2e["verified_vs_oracle"] is True and e2e["ticks_per_s"] > 100 and 0 < e2e["d2h_frac_of_link"] < e2e["h2d_frac_of_link"] < 1.1
    ts = w["per_tick_thread_scaling"]
    assert set(ts["python"]["fused"]) == {"1", "2", "4", "8"} and "error" not in ts["native"], ts.get("native")
    assert ts["native"]["fused"]["8"] > ts["native"]["fused"]["1"] * 0.8
    # the path a Swift VideoMixer takes — one tick at a time with a host wait — fused and as the unchanged 5-launch sequence
    pt, seq = w["pipeline_per_tick"], w["pipeline_reference_sequence"]
    assert pt["fused_equals_sequence"] is True and pt["launches_per_tick"] == 1 and seq["launches_per_tick"] == 5
    assert 5 < pt["us_per_tick"] < 2000 and 5 < seq["us_per_tick"] < 2000
    # the same for the reference-default 4:2:0 mixer tick (video + two overlays; 1 + 3 launches unchanged)
    mt, mseq = w["mixer_y420p_per_tick"], w["mixer_y420p_reference_sequence"]
    assert mt["fused_equals_sequence"] is True and mt["launches_per_tick"] == 1 and mseq["launches_per_tick"] == 4
    assert 5 < mt["us_per_tick"] < 2000 and 5 < mseq["us_per_tick"] < 2000
    cfg = full["config"]
    assert cfg["legs"]["pipeline_e2e_ticks_per_s"] > 100 and cfg["legs"]["pipeline_per_tick_us_per_tick"] > 5
    assert set(cfg["route_regret"]) == set(cfg["workload_fracs"]) and cfg["route_regret_max"] < 0.25, cfg["route_regret"]
    # the clock the kernels ran at and the socket power meanwhile (the path runs at the power cap: profiles/r05_notes.md section 10)
    if shutil.which("rocm-smi") and cfg["workload_power"]:
        assert set(cfg["workload_power"]) == set(cfg["workload_fracs"]), cfg["workload_power"]
        assert all(500 <= v[0] <= 3000 and 100 <= v[1] <= 2000 for v in cfg["workload_power"].values()), cfg["workload_power"]


@pytest.mark.parametrize("mode", [[], ["--threads"]])
def test_upload_mode_per_rank_plumbing_on_stub_devices(mode):
    """--with-upload at N = 2 without a GPU: every rank (process or thread) binds to a NUMA node's CPUs and reports its own H2D rate, launch
    time, node and CPU count under config.per_rank — what whoever runs the 8-GPU line reads to see which rank bent the curve."""
    d = _run_bench(["--gpus", "2", "--stub-device", "--with-upload", "--steps", "3", "--warmup", "1", "--min-seconds-other", "0.2"] + mode)
    pr = d["config"]["per_rank"]
    assert d["n_gpus"] == 2 and len(pr["h2d_GBps"]) == 2 and len(pr["launch_ms"]) == 2
    assert pr["launch_ms"][1] > pr["launch_ms"][0] * 1.1             # the stub's rank 1 is 25 % slower: the per-rank numbers are per rank
    assert pr["pinned_numa_node"] == [0, 0] and all(c >= 1 for c in pr["cpus_bound"])


def test_clock_and_power_readings_are_parsed_from_what_rocm_smi_prints():
    """bench.py's clock / power leg reads `rocm-smi --showclocks --showpower --json` (one JSON object per call, warnings in between): the shader
    clock in "(1965Mhz)" form and the socket power as a string; a call that printed something else contributes nothing."""
    import bench
    text = "\n".join([
        "WARNING: AMD GPU device(s) is/are in a low-power state. Check power control/runtime_status",
        '{"card0": {"fclk clock speed:": "(1250Mhz)", "fclk clock level:": "0", "mclk clock speed:": "(2000Mhz)", "mclk clock level:": "0", '
        '"sclk clock speed:": "(1965Mhz)", "sclk clock level:": "1", "socclk clock speed:": "(50Mhz)", "socclk clock level:": "S", '
        '"Current Socket Graphics Package Power (W)": "1400.0"}}',
        "/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory",
        '{"card0": {"sclk clock speed:": "(1971Mhz)", "Current Socket Graphics Package Power (W)": "1398.0"}}',
        '{"card0": {"mclk clock speed:": "(2000Mhz)"}}',
        "{not json",
    ])
    assert bench.smi_parse(text) == [(1965.0, 1400.0), (1971.0, 1398.0)]
    assert bench.smi_parse("") == []
