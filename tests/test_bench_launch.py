"""bench.py's own launcher: `python bench.py --gpus N` with N > 1 and no launcher environment
must start N ranks itself and must never fall through to a one-GPU measurement.  CPU-only: the ranks rehearse the
rendezvous and the collectives of the timed region on gloo (`--rendezvous-only`), no GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus2_without_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rendezvous-only"],
                       capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout  # stdout = rank 0's one line, nothing else (gloo's
    out = json.loads(lines[0])                                       # connection chatter and library banners go to stderr)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["rank_sum"] == 3.0  # ranks 0 and 1 both took part in the all-reduce


def _stub(n, extra=()):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--backend", "gloo", "--stub-renderer", "--steps", "6",
                        "--warmup", "2", "--spinup-ms", "120"] + list(extra),
                       capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    return json.loads(lines[0])


def test_timing_protocol_issues_equal_collectives_on_every_rank_2_ranks():
    """the clock spin-up ran for a per-rank wall-clock time while every step issues an all-reduce, so
    ranks with skewed clocks issued different numbers of collectives.  `--stub-renderer` runs bench.py's REAL protocol
    (run_protocol / spin_up / _timed_region / reduce_over_ranks -- the functions the GPU run calls) around a stand-in
    step on gloo, with rank-dependent delays in front of the spin-up; the count of step collectives must be equal on
    all ranks, the job must finish (a mismatch hangs or aborts), and the line must carry both timed regions."""
    one = _stub(1)
    out = _stub(2)
    assert out["stub"] is True and out["value"] is None  # never mistakable for a measurement
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["step_collectives_rank_min"] == out["step_collectives_rank_max"] > 0
    # warm-up 2 + cold region 6 + agreed spin-up + timed region 6
    assert out["step_collectives_rank_min"] == 2 + 6 + out["spinup_steps"] + 6
    assert out["spinup_steps"] >= 20 and out["spinup_steps"] % 10 == 0
    assert out["config"]["pairs_per_step"] == 2 * one["config"]["pairs_per_step"]
    assert out["ms_per_step_rank_max"] >= out["ms_per_step_rank_min"] > 0
    assert out["ms_per_step_cold"] > 0 and out["ms_per_step"] == out["ms_per_step_rank_max"]
    assert one["rccl_ranks"] == 1 and one["ms_per_step_cold"] > 0


def test_fake_8_rank_launch_prints_one_line_with_8_ranks():
    """what the driver runs first on an 8-GPU node, rehearsed with 8 gloo ranks on the CPU: exactly
    one JSON line, rccl_ranks 8, pairs summed over the ranks = 8 x the one-rank value, max >= min over ranks."""
    out = _stub(8)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8
    assert out["config"]["pairs_per_step"] == 8 * out["config"]["pairs_per_step_per_rank"]
    assert out["ms_per_step_rank_max"] >= out["ms_per_step_rank_min"] > 0
    assert out["step_collectives_rank_min"] == out["step_collectives_rank_max"] == 2 + 6 + out["spinup_steps"] + 6


def test_gpu_run_and_stub_share_the_protocol_functions():
    """The stub is only evidence for the GPU run if both go through the same code: main() must call run_protocol and
    reduce_over_ranks, and no second spin-up loop may exist."""
    src = open(BENCH).read()
    main_src = src[src.index("def main("):]
    assert "run_protocol(step, torch.cuda.synchronize, dist" in main_src and "reduce_over_ranks(dist" in main_src
    assert src.count("perf_counter() - t_spin") == 1 and "t_spin" not in main_src


def test_too_few_devices_is_an_error_not_a_one_gpu_run():
    """No GPU here: --gpus 2 on the RCCL backend must exit non-zero with a message and print no result line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_rejected():
    env = _clean_env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_parent_imports_nothing_gpu_related_before_launching():
    """The launcher branch runs before torch / fresnel_amd are imported in the parent (a process that has initialised
    the GPU must not be the one that starts the ranks)."""
    src = open(BENCH).read()
    head = src[:src.index("def _import_compute")]
    assert "import torch" not in head.replace("import torch as _torch", "") and "fresnel_amd" not in head.split('"""')[2]
    main_src = src[src.index("def main("):]
    assert main_src.index("launch_ranks(args, argv)") < main_src.index("_import_compute()")


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["config3", "config4", "config5"])
def test_bench_line_on_the_gpu(workload):
    """The line the driver records, from a short real run: one JSON object on stdout with the contract's fields, both timed
    regions, the roofline object of the dominant kernel (hipEvent-timed in the library) and the workload's name."""
    r = subprocess.run([sys.executable, BENCH, "--workload", workload, "--steps", "3", "--warmup", "2", "--spinup-ms", "60",
                        "--no-cpu-baseline"] + (["--images-per-gpu", "2"] if workload == "config3" else []),
                       capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "value_cold", "spinup_steps",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 1e8 and d["ms_per_step"] > 0 and d["ms_per_step_cold"] > 0 and d["spinup_steps"] >= 20
    assert workload in d["config"]["workload"] and d["config"]["pairs_per_step"] > 0
    rf = d["roofline"]
    assert rf["bound"] in ("valu", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["avg_launch_ms"] > 0
    assert set(rf["stage_avg_ms"]) >= {"project", "project_bwd"} and rf["stage_avg_ms_cold"]
    assert "{" not in (rf.get("traffic_source") or "")  # (an unformatted "{run_key}" once sat in the committed lines)
    assert rf["kernel"].startswith({"config3": "k_composite_bwd", "config4": "k_phase_bwd", "config5": ""}[workload])


@pytest.mark.gpu
def test_per_image_loop_line_on_the_gpu():
    """`--per-image-loop`: the reference's literal call pattern (B sequential (N,.) module calls + torch.stack, TGD:1209-1226) gives
    the same unit count as the batched call on the same synthetic batch, and the line says which pattern it timed and what the host
    side of a step costs (round 5)."""
    out = {}
    for flag in ([], ["--per-image-loop"]):
        r = subprocess.run([sys.executable, BENCH, "--workload", "config1", "--steps", "3", "--warmup", "2", "--spinup-ms", "0"] + flag,
                           capture_output=True, text=True, env=_clean_env(), timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
        assert d["host_enqueue_us_per_step"] > 0 and d["allreduce_us"] is None and d["cpu_baseline"] is None
        out[bool(flag)] = d
    assert out[False]["call_pattern"].startswith("batched") and out[True]["call_pattern"].startswith("per-image loop: 32 sequential")
    assert out[False]["config"]["pairs_per_step"] == out[True]["config"]["pairs_per_step"] > 0
    assert out[True]["ms_per_step"] > out[False]["ms_per_step"]  # 32 calls against one
