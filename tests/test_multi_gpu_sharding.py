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


def _run_bench(argv, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}: {p.stdout[-500:]}"
    return json.loads(lines[0])


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
    assert set(d["workloads"]) == {"pipeline", "cfg2"}
    assert d["workloads"]["cfg2"]["launches_per_step"] >= 1
    assert "cpu_baseline" not in d                                                    # N = 1 only


def test_one_process_with_a_host_thread_per_device():
    """`--gpus 2 --threads`: ONE process, two host threads, a context per device — the shape of the Swift host (a composer with mixers bound to
    devices, composer.swift:203-224, SURVEY section 8e "one host feeder thread per device"); the same barrier + slowest-thread reduction as the
    process-per-GPU mode, through threading primitives (here with --stub-device: thread 1 sleeps 25 % longer per launch)"""
    d = _run_bench(["--gpus", "2", "--threads", "--stub-device", "--steps", "4", "--warmup", "1", "--min-seconds", "0.3"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["data"].startswith("STUB")
    cfg = d["config"]
    assert cfg["parallelism"].startswith("ONE process, 2 host threads")
    assert len(cfg["per_gpu_gpix"]) == 2 and cfg["per_gpu_gpix"][0] >= cfg["per_gpu_gpix"][1] * 0.95
    assert 2.0 * min(cfg["per_gpu_gpix"]) * 0.7 <= d["value"] <= 2.0 * min(cfg["per_gpu_gpix"]) * 1.02
    assert set(d["workloads"]) == {"pipeline"}
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


@pytest.mark.gpu
def test_default_run_measures_its_hbm_traffic():
    """The driver's command shape (no --also): every workload of the default set timed and verified, and roofline.traffic measured
    in the run itself by two rocprofv3 --pmc child passes (or, where rocprofv3 is missing, the committed figure — and the line says
    which)."""
    import shutil
    d = _run_bench(["--steps", "2", "--warmup", "1", "--min-seconds", "0.1", "--min-seconds-other", "0.05", "--no-cpu-baseline"])
    assert set(d["workloads"]) >= {"pipeline", "cfg2", "cfg3", "cfg5", "mixer_y420p", "encode_nv12", "pipeline_y420p", "pipeline_grid", "pipeline_logo"}
    e2e = d["workloads"]["pipeline_e2e"]
    assert e2e["verified_vs_oracle"] is True and e2e["ticks_per_s"] > 100 and 0 < e2e["d2h_frac_of_link"] < e2e["h2d_frac_of_link"] < 1.1
    legs = ("cfg2_upload", "pipeline_per_tick", "pipeline_reference_sequence", "mixer_y420p_per_tick", "mixer_y420p_reference_sequence", "pipeline_e2e", "per_tick_thread_scaling",
            "route_regret", "power_probe")
    ts = d["workloads"]["per_tick_thread_scaling"]
    assert set(ts["python"]["fused"]) == {"1", "2", "4", "8"} and "error" not in ts["native"], ts.get("native")
    assert ts["native"]["fused"]["8"] > ts["native"]["fused"]["1"] * 0.8
    assert all(v["verified_vs_oracle"] is True for k, v in d["workloads"].items() if k not in legs)
    # the path a Swift VideoMixer takes — one tick at a time with a host wait — fused and as the unchanged 5-launch sequence
    pt, seq = d["workloads"]["pipeline_per_tick"], d["workloads"]["pipeline_reference_sequence"]
    assert pt["fused_equals_sequence"] is True and pt["launches_per_tick"] == 1 and seq["launches_per_tick"] == 5
    assert 5 < pt["us_per_tick"] < seq["us_per_tick"] < 2000
    # the same for the reference-default 4:2:0 mixer tick (video + two overlays; 1 + 3 launches unchanged)
    mt, mseq = d["workloads"]["mixer_y420p_per_tick"], d["workloads"]["mixer_y420p_reference_sequence"]
    assert mt["fused_equals_sequence"] is True and mt["launches_per_tick"] == 1 and mseq["launches_per_tick"] == 4
    assert 5 < mt["us_per_tick"] < mseq["us_per_tick"] < 2000
    assert d["config"]["build_flags"].startswith("arch=gfx950;") and "abl=0" in d["config"]["build_flags"]
    # every workload's fraction and the route-regret leg where the driver's parser keeps them (keys under `config`)
    cfg = d["config"]
    assert set(cfg["workload_fracs"]) >= {"pipeline", "cfg2", "cfg3", "cfg5", "mixer_y420p", "encode_nv12", "pipeline_y420p", "pipeline_grid", "pipeline_logo"}
    assert all(0 < f[0] < 1 and f[1] > 0 and isinstance(f[2], str) for f in cfg["workload_fracs"].values())
    assert cfg["legs"]["pipeline_e2e_ticks_per_s"] > 100 and cfg["legs"]["pipeline_per_tick_us_per_tick"] > 5
    assert set(cfg["route_regret"]) == set(cfg["workload_fracs"]) and cfg["route_regret_max"] < 0.25, cfg["route_regret"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    # the clock the kernels ran at and the socket power meanwhile (the path runs at the power cap: profiles/r05_notes.md section 10)
    if shutil.which("rocm-smi") and cfg["workload_power"]:
        assert set(cfg["workload_power"]) == set(cfg["workload_fracs"]), cfg["workload_power"]
        assert all(500 <= v[0] <= 3000 and 100 <= v[1] <= 2000 for v in cfg["workload_power"].values()), cfg["workload_power"]
        assert r["power_cap_w"] is None or r["power_w"] <= 1.05 * r["power_cap_w"]
    if shutil.which("rocprofv3"):
        assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
        assert 0.97 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.10 * r["algorithmic_bytes_per_launch"], r
    else:
        assert r["traffic"] is None or "NOT measured in this run" in r["traffic_source"]


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
