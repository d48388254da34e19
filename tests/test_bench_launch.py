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
