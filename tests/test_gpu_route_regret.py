"""Is the kernel the library picks the fast one?  `select_fast_path` (kernels_fast.hip.cpp) is a decision tree with thresholds measured on one
box; every route gives the oracle's bytes (tests/test_gpu_fuzz.py forces each), so a regression in one kernel — or a box where the thresholds
are off — would silently leave ticks on the slower path.  bench.py's route_regret leg times every route that accepts a default workload's batch
in one process; here the same leg over the workloads whose routes compete, failing beyond 10 % (the leg's own target, reported by the default
bench run under config.route_regret, is 3 %)."""
import argparse

import pytest

pytestmark = pytest.mark.gpu

WORKLOADS = ["pipeline", "cfg2", "pipeline_y420p", "pipeline_grid", "mixer_y420p", "encode_nv12"]


def test_chosen_route_is_within_ten_percent_of_the_best(ctx):
    import bench
    from swiftvideo_amd import chipvideo as cv
    from swiftvideo_amd import compute as sv
    res = bench.run_route_regret(argparse.Namespace(), sv, cv, cv.load(), ctx, WORKLOADS, seconds=0.08, rounds=2)
    report = {k: (v["chosen"], v["chosen_ms"], v["best"], v["regret"], {r: t[0] for r, t in v["routes_ms"].items()}) for k, v in res.items()}
    bad = {k: v for k, v in report.items() if v[3] > 0.10}
    assert not bad, f"route regret > 10 %: {bad}\nall: {report}"
    for k, v in res.items():
        assert len(v["routes_ms"]) >= 2, f"{k}: no alternative route was eligible — the leg measured nothing: {v}"
