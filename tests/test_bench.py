"""bench.py contract checks: `python bench.py --gpus N` without a launcher starts its own N ranks (the driver's N = 1
command and a future multi-GPU run go through the same entry), and the N > 1 line carries every object of the N = 1 line
the driver's rules ask for.  The 2-rank run shares the box's one GPU (HYD_BENCH_ONE_DEVICE=1) and all-reduces on gloo
(RCCL refuses two ranks on one device); sharding follows /root/reference/hydragen/tp.py:90-124."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def test_respawn_command_line(monkeypatch):
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench._respawn(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _run_bench(args, extra_env=None, timeout=600):
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env,
                       cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher():
    res = _run_bench(["--gpus", "2", "--steps", "8", "--warmup", "2", "--trials", "1", "--cpu-seconds", "2", "--no-xgmi"],
                     {"HYD_BENCH_BACKEND": "gloo", "HYD_BENCH_ONE_DEVICE": "1"})
    assert res["n_gpus"] == 2 and res["steps"] == 8 and res["scaling"] == "strong"
    for key in ("roofline", "roofline_other", "cpu_baseline", "allreduce_us", "rccl_ranks", "trials", "accuracy"):
        assert key in res, key
    assert res["rccl_ranks"] == 2 and res["value"] > 0
    assert res["roofline"]["bound"] in ("hbm", "mfma") and 0 < res["roofline"]["frac"] < 1
    assert res["cpu_baseline"]["kind"] == "port" and res["cpu_baseline"]["value"] > 0
    assert "TP2" in res["config"]["parallelism"].upper()


@pytest.mark.gpu
def test_bench_preflight_runs_every_stage_on_two_ranks():
    """`--gpus 2` pre-flight (gloo, both ranks on cuda:0): devices, process group + checked 1 KiB all-reduce, peer access, one
    hyd_allreduce_sum of 1 MiB over hipIpc checked against the group's result, agreement; the report rides on the line."""
    env = dict(os.environ, HYD_BENCH_BACKEND="gloo", HYD_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--trials", "0",
                        "--no-cpu-baseline", "--no-accuracy"], capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-3000:]
    stages = [json.loads(ln[len("[preflight] "):]) for ln in r.stdout.splitlines() if ln.startswith("[preflight] ")]
    assert [s_["stage"] for s_ in stages] == ["devices", "rccl_init", "peer_access", "xgmi_allreduce", "graph_collective", "agreement"]
    assert all(s_["ok"] for s_ in stages), stages
    assert stages[3]["bytes"] == 1 << 20 and stages[3]["max_abs_diff_vs_group"] < 0.1
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert set(res["preflight"]) == {s_["stage"] for s_ in stages} and res["value"] > 0
    assert res["allreduce_xgmi"]["max_abs_diff_vs_rccl"] < 0.5  # the 8 MiB leg ran: the pre-flight left it switched on


@pytest.mark.gpu
@pytest.mark.parametrize("stage,rc", [("peer_access", 3), ("xgmi_allreduce", 0)])
def test_bench_preflight_names_the_failing_stage(stage, rc):
    """A failing hard stage ends the job with an error line naming it and exit code 3; a failing xGMI stage only switches
    that leg off and the headline still runs."""
    env = dict(os.environ, HYD_BENCH_BACKEND="gloo", HYD_BENCH_ONE_DEVICE="1", HYD_BENCH_FAIL_STAGE=stage)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--trials", "0",
                        "--no-cpu-baseline", "--no-accuracy"], capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    if rc == 3:
        assert r.returncode != 0 and lines, r.stderr[-2000:]
        assert all(ln["preflight_stage"] == stage and stage in ln["error"] and ln["value"] is None for ln in lines)
    else:
        assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
        res = lines[0]
        assert res["preflight"][stage]["ok"] is False and "allreduce_xgmi" not in res and res["value"] > 0


@pytest.mark.gpu
def test_bench_c5_workload_on_two_ranks():
    """BASELINE config 5 as north_star states it (batch 2048, prefix 4096, 64 q / 8 kv heads, suffix 1..256), sharded over
    heads on two ranks (32 q / 4 kv heads each, tp.py:90-124) with the 32 MiB [2048, 1, 8192] all-reduce per step."""
    res = _run_bench(["--workload", "c5", "--gpus", "2", "--steps", "8", "--warmup", "2", "--trials", "1", "--cpu-seconds", "2",
                      "--no-xgmi", "--no-accuracy"], {"HYD_BENCH_BACKEND": "gloo", "HYD_BENCH_ONE_DEVICE": "1"}, timeout=900)
    cfg = res["config"]
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["value"] > 0
    assert cfg["preset"] == "c5" and cfg["batch"] == 2048 and cfg["prefix_len"] == 4096 and cfg["qheads"] == 64 and cfg["kvheads"] == 8
    assert max(cfg["suffix_lens"]) <= 256 and "C5" in cfg["workload"] and "TP2" in cfg["parallelism"].upper()
    assert res["allreduce_bytes"] == 2048 * 8192 * 2
    for key in ("roofline", "roofline_other", "cpu_baseline"):
        assert key in res, key
    assert res["roofline"]["bound"] in ("hbm", "mfma") and 0 < res["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_c5_workload_on_eight_ranks():
    """What `--gpus 8 --workload c5` executes on a real node (the reference's sweeps run --nproc_per_node=8,
    docs/sweeps_from_paper.md:25-150; sharding tp.py:115-132), rehearsed with eight processes on the one device: the
    Hkv / N = 1 rank shape (8 q / 1 kv head per rank), an 8-process pre-flight -- every stage, incl. one `hyd_allreduce_sum`
    over eight hipIpc-mapped blocks -- and the xGMI leg's 8-slice exchange of the step's [B, 1, 8192] block output.  Batch
    reduced to 256 so that eight shards and eight contexts share one GPU comfortably; everything else is the preset's."""
    env = dict(os.environ, HYD_BENCH_BACKEND="gloo", HYD_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--workload", "c5", "--gpus", "8", "--batch", "256", "--steps", "8",
                        "--warmup", "2", "--trials", "0", "--no-cpu-baseline", "--no-accuracy"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-3000:]
    stages = [json.loads(ln[len("[preflight] "):]) for ln in r.stdout.splitlines() if ln.startswith("[preflight] ")]
    assert [s_["stage"] for s_ in stages] == ["devices", "rccl_init", "peer_access", "xgmi_allreduce", "graph_collective", "agreement"]
    assert all(s_["ok"] for s_ in stages), stages
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    cfg = res["config"]
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8 and res["value"] > 0
    assert cfg["preset"] == "c5" and cfg["batch"] == 256 and cfg["qheads"] == 64 and cfg["kvheads"] == 8
    assert "TP8" in cfg["parallelism"].upper()
    assert res["allreduce_bytes"] == 256 * 8192 * 2
    assert res["allreduce_xgmi"]["max_abs_diff_vs_rccl"] < 0.5  # eight ranks' hyd_allreduce_sum agrees with the group's


def test_workload_presets_and_overrides(monkeypatch):
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.workload, a.batch, a.prefix, a.max_suffix, a.qheads, a.kvheads) == ("c2", 1024, 2048, 128, 32, 32)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "c5", "--batch", "512"])
    a = bench.parse()
    assert (a.batch, a.prefix, a.max_suffix, a.qheads, a.kvheads) == (512, 4096, 256, 64, 8)


def test_region_watchdog_prints_an_error_line_and_exits():
    """A rank whose timed region never finishes (a peer that never arrives) leaves with an error line and a non-zero
    exit code instead of hanging the job."""
    code = ("import bench, sys, time, types\n"
            "a = types.SimpleNamespace(steps=8, warmup=2, region_timeout=0.3)\n"
            "bench._region_watchdog(a, 1, 2)\n"
            "time.sleep(30)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=str(REPO))
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and "rank 1" in line["error"]


@pytest.mark.gpu
def test_bench_single_gpu_line_is_complete():
    res = _run_bench(["--gpus", "1", "--steps", "8", "--warmup", "2", "--trials", "2", "--cpu-seconds", "2", "--no-model",
                      "--no-protocol"])
    assert res["n_gpus"] == 1 and res["metric"] == "decode_attention_tokens_per_sec" and res["unit"] == "tokens/s"
    assert res["vs_baseline"] is None and res["dtype"] == "bf16" and res["data"] == "synthetic"
    assert len(res["trials"]["repeat_us_per_step"]) == 2
    # timed steps are eager C calls; the same schedule as graph replays is timed beside them
    assert res["step_forms"]["form"] == "eager" and res["trials"]["other_form"] == "graph" and res["trials"]["other_form_us_per_step"] > 0
    assert abs(res["trials"]["trial0_us_per_step"] / (res["ms_per_step"] * 1e3) - 1) < 1e-4  # (nested figures carry 5 digits)
    assert len(json.dumps(res)) < 8000 and res["detail_file"] == "gpurun_out/bench_detail.json"  # the driver keeps 8 KB
    detail = json.loads((REPO / res["detail_file"]).read_text())
    assert detail["value"] == res["value"] and "trials" in detail and "roofline_other" in detail
    assert set(res["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(res["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "c1_full"}
    # HBM bytes per launch are counted in the run itself (rocprofv3 --pmc child passes), for both kernels
    for key in ("roofline", "roofline_other"):
        roof = res[key]
        assert roof["traffic"] and roof["traffic_source"].startswith("counted in this run"), roof.get("traffic_source")
        assert 0.95 < roof["traffic_over_algorithmic"] < 1.25, roof


def test_launch_schedule_is_what_a_profiler_sees():
    """profiles/traffic_latest.json divides the PMC totals by bench.launch_schedule(): 3 eager warm-ups per captured suffix
    length, the warm-up steps, the timed steps (and the repeats when there are any) -- with the mean suffix of the timed
    schedule, so per-launch averages of a profile compare with the bench line."""
    import bench

    ls = bench.launch_schedule(20, 5, 0, 128)
    assert len(ls) == 3 * 25 + 5 + 20 and abs(sum(ls) / len(ls) - 64.6) < 0.05
    assert len(bench.launch_schedule(20, 5, 3, 128)) == len(ls) + 60
    sched = bench.suffix_schedule(20, 128)
    ev = sorted({i for i in range(10) if i % 5 == 1} | {19 - i for i in range(10) if i % 5 == 1})
    assert abs(sum(sched[i] for i in ev) / len(ev) - sum(sched) / 20) < 0.2  # event steps: the schedule's mean suffix


def test_two_stream_policy_is_shapes_only():
    """'auto' never looks at device data: capture state + shapes decide (hydragen_amd/attention.py::_want_two_stream)."""
    import torch

    from hydragen_amd import attention as A

    q = torch.empty(1024, 1, 32, 128, dtype=torch.bfloat16, device="meta")
    k = torch.empty(1024, 128, 32, 128, dtype=torch.bfloat16, device="meta")
    sk = torch.empty(1, 2048, 32, 128, dtype=torch.bfloat16, device="meta")
    args = (q, k, [sk], [None], [False])
    assert A._two_stream_mode == "off" and not A._want_two_stream(*args, capturing=True)
    prev = A.set_two_stream("auto")
    try:
        assert A._want_two_stream(*args, capturing=True) and not A._want_two_stream(*args, capturing=False)
        assert not A._want_two_stream(q, k[:, :8], [sk], [None], [False], capturing=True)     # a cache too short to hide a prefix pass
        assert not A._want_two_stream(q[:4], k[:4], [sk[:, :64]], [None], [False], capturing=True)  # C1-sized: nothing worth hiding
        assert not A._want_two_stream(q, k, [], [], [], capturing=True)
        A.set_two_stream("on")
        assert A._want_two_stream(*args, capturing=False)
    finally:
        A.set_two_stream(prev)


def test_compact_line_fits_the_drivers_record(tmp_path, monkeypatch):
    """The driver keeps the last 8 KB of bench.py's line: the line carries means at five digits, the per-point statistics
    and the prose go to the detail file (round 3's 17 KB line lost roofline_other / trials / prefix_us)."""
    import bench

    monkeypatch.setattr(bench, "DETAIL_FILE", tmp_path / "detail.json")
    monkeypatch.setattr(bench, "REPO", tmp_path)
    stat = lambda m: {"mean_us": m * 1.000001234, "std_us": 1.23456789, "rstd": 0.0123456789, "n": 20}
    cols = ("hydragen_flushed", "hydragen_flushed_clean", "hydragen_back_to_back", "two_stream_flushed_clean",
            "two_stream_back_to_back", "nosharing_flushed")
    res = {
        "metric": "decode_attention_tokens_per_sec", "value": 4604443.873278889, "unit": "tokens/s", "ms_per_step": 0.2223938499810174,
        "config": {"workload": "w" * 250, "suffix_lens": list(range(20))},
        "roofline": {"frac": 0.7682519848229409, "traffic_source": "counted in this run: " + "x" * 300},
        "roofline_other": {"frac": 0.3338571052407633, "traffic_source": "counted in this run: " + "y" * 300},
        "trials": {"repeat_us_per_step": [219.81835016049445, 221.13814920885488], "headline": "h" * 60, "repeat_note": "n" * 90},
        "reference_protocol": {"protocol": "p" * 700, "iters": 20,
                               "by_suffix_len": {str(s): dict({c: stat(100.0 + s) for c in cols}, speedup=25.123456789) for s in range(16, 129, 16)}},
        "paper_sweep": {"rows": [{"batch": 512 * (1 + i // 3), "prefix": 1024, "suffix": 128 * (i % 3), "hydragen_us": 28.146833181381226,
                                  "hydragen_rstd": 0.0255, "nosharing_us": 60.14766792456309, "speedup": 2.136924873109704} for i in range(12)]},
        "events": {"steps": list(range(64)), "why": "e" * 170},
        "cpu_baseline": {"value": 1030.123456, "sample": "s" * 207},
        "accuracy": {f"k{i}": 0.001234567891 * i for i in range(30)},
    }
    line = bench.compact_line(res)
    text = json.dumps(line)
    assert len(text) < 8000, len(text)
    assert line["value"] == res["value"] and line["ms_per_step"] == res["ms_per_step"]
    assert line["roofline"]["frac"] == 0.76825 and line["trials"]["repeat_us_per_step"] == [219.82, 221.14]
    by = line["reference_protocol"]["by_suffix_len"]
    assert by["columns"][-1] == "speedup" and len(by["mean_us"]["16"]) == len(cols) + 1 and by["mean_us"]["16"][-1] == 25.123
    assert line["paper_sweep"]["rows"]["columns"][:3] == ["batch", "prefix", "suffix"] and len(line["paper_sweep"]["rows"]["values"]) == 12
    full = json.loads((tmp_path / "detail.json").read_text())
    assert full["reference_protocol"]["by_suffix_len"]["16"]["hydragen_flushed"]["n"] == 20 and full["roofline"]["frac"] == res["roofline"]["frac"]
