#!/usr/bin/env python3
"""bench.py -- throughput of the HIP rasterizer hot path on synthetic N-Gaussian x HxW batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one forward + backward pass of the rasterizer (project, depth sort, tile binning,
composite, composite backward, projection backward) over one batch of synthetic images that is
already resident in HBM, called through the same nn.Module (TileBasedRenderer / ASMWaveFieldRenderer)
the training harness calls, plus -- for N > 1 -- the RCCL all-reduce of the decoder-gradient
bucket that the image-wise data-parallel training step performs.

Default workload = BASELINE.json configs[2] per GPU (the configuration the metric "512^2 render,
1/2/4/8 MI355X" is quoted on): 32 768 Gaussians, 512x512, 8 images per GPU (64 over 8 GPUs),
create_dummy_saag distribution (reference scripts/training/train_gaussian_decoder.py:760-778).
Weak scaling: per-GPU work is fixed as N grows.  --workload config2|config4|config5 and
--distribution decoder_like (SURVEY §8d's second distribution) select the other measured cases.

Prints ONE JSON line (rank 0).  value = composited Gaussian-pixels per second, whole job.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# numpy / torch are imported by _import_compute(), AFTER the launcher decision in main(): the parent of a --gpus N > 1
# run must start its N ranks before anything in this process can have touched a GPU.
np = torch = None


def _import_compute():
    global np, torch
    import numpy
    import torch as _torch
    np, torch = numpy, _torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (packed / dual-issue rate)
DECODER_GRAD_FLOATS = 673_537  # DirectPatchDecoder gradient bucket (measured)
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_summary.json") for r in (5, 4)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r05_pmc_summary.json"))  # this round's counters; the previous round's until they exist

WORKLOADS = {
    # name: (N gaussians, resolution, images per GPU)   -- BASELINE.json configs[1..4]
    "config3": (32768, 512, 8),    # the metric's configuration (default)
    "config2": (8192, 256, 16),
    "config4": (8192, 256, 16),    # config-2 shapes + 8 depth zones + scalar phases, phase blending
    "config5": (8192, 512, 1),     # ASMWaveFieldRenderer, per-channel wavelengths, hipFFT
    "config1": (256, 128, 32),     # configs[0]'s shape (the reference's CPU plumbing case) on the GPU: the launch-bound end
}

# ALGORITHMIC work per stage.  P = composited Gaussian-pixels, D = tile
# duplicates, N = Gaussians, B = images, HW = pixels per image, U = depth-segment units beyond each tile's first.
FLOPS_PER_PAIR = {"composite_fwd": 23.0, "composite_bwd": 60.0,   # SURVEY §8d "Which roofline"
                  "splat_fwd": 12.0, "splat_bwd": 36.0}            # 6 MACs per pair; adjoint 3x
# Phase path (config 4), round 5: its OWN per-pair count (until round 4 it was scored with the blend path's 23 / 60).
# Counted from the recurrence DR:629-667 and its adjoint, FMA = 2, every exp / cos / sin / divide = 1:
#   forward 38 = dx, dy (2) + quadratic form (8) + exp, x opacity (2) + |phi - Phi|, min(pd, 1 - pd) (3) + 2 pi pd, cos, (1 - amp) + amp cos,
#                x alpha (5) + clamp (1) + w = alpha (1 - A) (2) + C, D, A += (9) + pc = w / max(A, 1e-6) (2) + Phi update (4);
#   backward 71 = the same per-pair quantities recomputed (dx, dy, quadratic form, exp, phase difference, cos, raw alpha: 23) + the adjoint
#                sweep (w-bar incl. gI.c + gD d: 11, colour / depth sums 4, A-bar / Phi-bar updates 9, clamp / min / sign selects 0, sin and
#                the phase chain 8, dL/dopacity 1, moments of dL/dm 9, pc-bar chain 6).  The re-run of the forward between checkpoints
#                (~45 flop per pair) is the implementation's price for not storing (A, Phi) per pair and is NOT credited.
FLOPS_PER_PAIR_PHASE = {"composite_fwd": 38.0, "composite_bwd": 71.0}
PHASE_CKPT_BYTES_PER_DUP = 115  # (A, Phi) of 64 lanes = 512 B per group of <= 8 touched entries of a sub-tile wave: ~3.6 groups per 64-entry
                                # scan block and wave at config 4 (~40 % of a tile's entries touch a given sub-tile) x 4 waves / 64 entries
ASM_BYTES_PER_IMAGE = 0.36e9  # SURVEY §8d "ASM": 51 FFTs of 512^2 with the linearity trick, per direction


def synth_batch(n_img, N, seed0, device, distribution="saag"):
    """saag: create_dummy_saag (TGD:760-778): pos~N(0,0.5^2), z-=2, scale .05, identity quat, colour~U(0,1),
    opacity .8.  decoder_like (what an untrained DirectPatchDecoder emits, GDM:740-948): grid x,y in
    linspace(-1,1,s), s = floor(sqrt(N)), z = -2-2U, scale~U(.13,.16), random unit quats, opacity~U(.4,.6).
    Generator seeded per image."""
    pos, col, scl, quat, opa = [], [], [], [], []
    for i in range(n_img):
        g = torch.Generator().manual_seed(seed0 + i)
        if distribution == "decoder_like":
            s = int(np.floor(np.sqrt(N)))
            lin = torch.linspace(-1.0, 1.0, s)
            gy, gx = torch.meshgrid(lin, lin, indexing="ij")
            z = -2.0 - 2.0 * torch.rand(s * s, generator=g)
            pos.append(torch.stack([gx.reshape(-1), gy.reshape(-1), z], 1))
            scl.append(0.13 + 0.03 * torch.rand(s * s, 3, generator=g))
            q = torch.randn(s * s, 4, generator=g)
            quat.append(q / q.norm(dim=1, keepdim=True))
            col.append(torch.rand(s * s, 3, generator=g))
            opa.append(0.4 + 0.2 * torch.rand(s * s, generator=g))
        else:
            p = torch.randn(N, 3, generator=g) * 0.5
            p[:, 2] -= 2
            pos.append(p)
            col.append(torch.rand(N, 3, generator=g))
            scl.append(torch.full((N, 3), 0.05))
            q = torch.zeros(N, 4)
            q[:, 0] = 1
            quat.append(q)
            opa.append(torch.full((N,), 0.8))
    return [torch.stack(t).to(device) for t in (pos, scl, quat, col, opa)]


def pmc_field_traffic(run_key):
    """config 5: HBM bytes per step of both field stages together (rocFFT + spectral kernels), same source."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    run = json.load(open(PMC_SUMMARY)).get("runs", {}).get(run_key) or {}
    v = run.get("field_stages_hbm_bytes_per_step")
    return int(v) if v else None


def pmc_traffic(run_key, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r0N_pmc_summary.json of this round: separate FETCH_SIZE / WRITE_SIZE passes, 2*FETCH + WRITE per the gfx950
    correction of MI355X_MICROARCH.md).  None when that run was not profiled."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    run = json.load(open(PMC_SUMMARY)).get("runs", {}).get(run_key)
    for row in (run or {}).get("kernels", []):
        if row["kernel"] == kernel and row.get("hbm_bytes_corrected"):
            return int(row["hbm_bytes_corrected"])
    return None


def host_cores():
    """CPU cores this process may use: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU
    box exposes all of the host's cores but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_leg(leaves, gI, gD, N, S, budget_s=8.0):
    """The reference's pure-PyTorch per-Gaussian-loop rasterizer cannot travel to this box, so the CPU baseline is
    this repo's restatement of it (oracle/torch_loop.py, same algorithm / same per-iteration tensor work, checked
    against the C oracle in tests/) on a BOUNDED sample of image 0: its first n Gaussians, n <= 8192 chosen from a
    64-Gaussian calibration run so that forward + autograd backward take about `budget_s` seconds -- once with all
    host cores, once with one thread; plus the scalar C oracle on two full images."""
    from oracle import fgs_oracle as orc
    from oracle import torch_loop as tl
    cores = host_cores()
    view = np.eye(4, dtype=np.float32)
    gi, gd = gI[0].cpu().numpy(), gD[0].cpu().numpy()
    fx = 0.8 * S

    def run(threads):
        sub = lambda n: [t[0, :n].detach().cpu().numpy() for t in leaves]
        _, t_cal = tl.timed_fwd_bwd(sub(min(64, N)), view, fx, fx, S / 2, S / 2, S, S, gi, gd, threads=threads)
        n = int(max(64, min(8192, N, budget_s / max(t_cal / min(64, N), 1e-6))))
        p, t = tl.timed_fwd_bwd(sub(n), view, fx, fx, S / 2, S / 2, S, S, gi, gd, threads=threads)
        return n, p, t

    n_all, p_all, t_all = run(cores)
    n_one, p_one, t_one = run(1)
    ocam = orc.make_camera(view, fx, fx, S / 2, S / 2, S, S)
    n_c = min(2, leaves[0].shape[0])
    Pc, tc = 0, 0.0
    for i in range(n_c):
        full = [t[i].detach().cpu().numpy() for t in leaves]
        t0 = time.perf_counter()
        Pi, _, _ = orc.render_fwd_bwd_timed(*full, ocam, gI[i].cpu().numpy(), gD[i].cpu().numpy())
        tc += time.perf_counter() - t0
        Pc += Pi
    return {"value": round(p_all / t_all, 1), "unit": "Gaussian-pixels/s", "cores": cores, "kind": "port",
            "sample": f"first {n_all} Gaussians of image 0 @ {S}x{S} ({p_all} pairs), forward + autograd backward of the "
                      f"pure-PyTorch per-Gaussian loop (oracle/torch_loop.py, restatement of DR:582-667) in {t_all:.2f} s "
                      f"with torch.set_num_threads({cores})",
            "one_thread": {"value": round(p_one / t_one, 1), "seconds": round(t_one, 2), "cores": 1,
                           "sample": f"first {n_one} Gaussians ({p_one} pairs)"},
            "c_oracle": {"value": round(Pc / tc, 1), "cores": 1, "seconds": round(tc, 2),
                         "sample": f"images 0..{n_c - 1} at full size ({Pc} pairs), scalar C restatement (oracle/fgs_oracle.c), fwd+bwd"},
            "host_cpus_visible": os.cpu_count(), "note": "baseline, not the target (see roofline.frac)"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _visible_gpus():
    """Number of GPUs a rank would see, counted in a CHILD process so that this one stays clean of any GPU state."""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                       capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves -- one process
    per GPU under torch.distributed.run, rendezvous on 127.0.0.1 -- relay their output (rank 0 prints the JSON line)
    and exit with their status.  Nothing in THIS process has imported torch or touched a GPU.  Never falls back to
    fewer ranks: too few devices is an error, not an n_gpus: 1 measurement."""
    if args.backend == "nccl" and not args.rendezvous_only and not args.stub_renderer:
        have = _visible_gpus()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to measure "
                             f"fewer ranks than asked for\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank job failed (exit status {rc}); no result line\n")
    return rc


def rendezvous_only(args, world, rank, result_fd):
    """Launcher / rendezvous rehearsal without a GPU (tests/test_bench_launch.py): the ranks form the process group,
    run the same collectives the timed region uses on CPU tensors, rank 0 prints what it saw."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    ranks = dist.get_world_size()
    if rank == 0:
        os.write(result_fd, (json.dumps({"rendezvous_only": True, "n_gpus": world, "rccl_ranks": ranks, "backend": "gloo",
                                         "rank_sum": float(t.item())}) + "\n").encode())
    dist.destroy_process_group()


class _NoStageTimers:
    """--stub-renderer: there is no library and no GPU; the protocol's stage-timer calls become no-ops."""
    @staticmethod
    def stage_timing_enable(on, stages=None):
        pass

    @staticmethod
    def stage_timing_read():
        return {}


def _timed_region(step, sync, dist, steps):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides (the contract of the driver)."""
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    return time.perf_counter() - t0


def spin_up(step, sync, dist, spinup_ms, flag):
    """Untimed steps until `spinup_ms` milliseconds of load have passed ON EVERY RANK.  Every step issues a collective
    (the bucket all-reduce), so all ranks must run the same number of steps: after each batch of ten every rank
    contributes "my clock says keep going" to a MAX all-reduce of `flag` (a one-element tensor on the collective's
    device) and all of them continue while any of them wants to -- the count is agreed, never a per-rank wall-clock
    decision (a few ms of skew at the boundary left ranks with unequal numbers of all-reduces).  Returns the
    agreed number of steps."""
    n = 0
    if spinup_ms <= 0:
        return n
    sync()
    t_spin = time.perf_counter()
    while True:
        for _ in range(10):
            step()
        sync()
        n += 10
        more = (time.perf_counter() - t_spin) * 1e3 < spinup_ms
        if dist is not None:
            flag.fill_(1.0 if more else 0.0)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item() > 0.0)
        if not more:
            return n


def run_protocol(step, sync, dist, args, timers, flag, compute_stages=(), after_warmup=None):
    """The measurement protocol, shared by the GPU run and the CPU stub (--stub-renderer):
      1. W warm-up steps with every stage bracketed by events (stage split at COLD clocks);
      2. `after_warmup()`: the caller's bookkeeping (the device-side pair count);
      3. COLD timed region: K steps straight after the warm-up -- how rounds 1 and 2 measured (ms_per_step_cold);
      4. clock spin-up with a collectively agreed step count (spin_up), then ten bracketed steps (stage split at
         SUSTAINED clocks);
      5. the timed region proper: EXACTLY K steps, barrier + synchronize on both sides -> ms_per_step / value.
    With --spinup-ms 0 steps 3 and 4 are skipped: the one timed region then IS the cold protocol.
    During the timed regions only the dominant kernel carries an event pair (each pair costs stream time: all stages
    together slowed the step by 2.5 %)."""
    def avgs(raw):
        return {k: (v[0] / v[1] if v[1] else 0.0) for k, v in raw.items()}

    def dominant(avg):
        return max(reversed(compute_stages), key=lambda k: avg.get(k, 0.0)) if compute_stages else None  # --warmup 0: the backward

    def timed(dom):
        timers.stage_timing_enable(True, stages=[dom] if dom else None)
        timers.stage_timing_read()
        t = _timed_region(step, sync, dist, args.steps)
        raw = timers.stage_timing_read()
        timers.stage_timing_enable(False)
        ms = raw[dom][0] / raw[dom][1] if dom and raw.get(dom, (0, 0))[1] else None
        return t, ms

    timers.stage_timing_enable(True)
    for i in range(args.warmup):
        if i == min(1, args.warmup - 1):  # the first step carries one-time costs (allocator, plan caches): drop it
            sync()
            timers.stage_timing_read()
        step()
    sync()
    cold_avg = avgs(timers.stage_timing_read())
    timers.stage_timing_enable(False)
    out = {"extra": after_warmup() if after_warmup is not None else None, "cold_avg": cold_avg, "sustained_avg": None,
           "elapsed_cold": None, "dom_ms_cold": None, "spinup_steps": 0}
    dom = dominant(cold_avg)
    if args.spinup_ms > 0:
        out["dom_stage_cold"] = dom
        out["elapsed_cold"], out["dom_ms_cold"] = timed(dom)
        out["spinup_steps"] = spin_up(step, sync, dist, args.spinup_ms, flag) + 10
        timers.stage_timing_enable(True)
        timers.stage_timing_read()
        for _ in range(10):
            step()
        sync()
        out["sustained_avg"] = avgs(timers.stage_timing_read())
        timers.stage_timing_enable(False)
        dom = dominant(out["sustained_avg"])
    out["dom_stage"] = dom
    out["elapsed"], out["dom_ms"] = timed(dom)
    return out


def reduce_over_ranks(dist, device, elapsed, elapsed_cold, pairs_local, steps):
    """MAX / MIN over ranks of each rank's own clock, SUM of the pairs.  Returns (elapsed_max, elapsed_cold_max,
    pairs_all, [rank_min_ms, rank_max_ms])."""
    rank_ms = [elapsed / steps * 1e3] * 2
    if dist is None:
        return elapsed, elapsed_cold, float(pairs_local), rank_ms
    cold = -1.0 if elapsed_cold is None else elapsed_cold
    tmax = torch.tensor([elapsed, cold], dtype=torch.float64, device=device)
    tmin = torch.tensor([elapsed], dtype=torch.float64, device=device)
    psum = torch.tensor([float(pairs_local)], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(psum, op=dist.ReduceOp.SUM)
    rank_ms = [float(tmin[0].item()) / steps * 1e3, float(tmax[0].item()) / steps * 1e3]
    cold_max = float(tmax[1].item())
    return float(tmax[0].item()), (None if elapsed_cold is None else cold_max), float(psum.item()), rank_ms


def stub_run(args, world, rank, result_fd):
    """--stub-renderer: the protocol above on CPU tensors under gloo.  The step is a stand-in (no rasterizer, no GPU):
    the line says so in `metric`, carries no roofline, and `value` is null -- it can not be mistaken for a measurement."""
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    device = torch.device("cpu")
    bucket = torch.zeros(4096) if dist is not None else None
    work = torch.zeros(256, 256)
    n_collectives = [0]

    def step():
        work.add_(1.0)
        if rank % 2:  # odd ranks are slower: the spin-up boundary falls in different batches on different ranks
            time.sleep(0.002)
        if bucket is not None:
            dist.all_reduce(bucket)
            n_collectives[0] += 1

    # after_warmup: unsynchronised, rank-dependent host work (the real run's pair count + .item() calls): the ranks reach
    # the spin-up with skewed clocks, the situation in which a per-rank wall-clock loop issued unequal collective counts
    res = run_protocol(step, lambda: None, dist, args, _NoStageTimers, torch.zeros(1),
                       after_warmup=lambda: time.sleep(0.013 * rank))
    elapsed = res["elapsed"]
    pairs_local = 1_000_000  # per rank and step: a constant, so that the SUM over ranks is checkable
    elapsed, elapsed_cold, pairs_all, rank_ms = reduce_over_ranks(dist, device, elapsed, res["elapsed_cold"], pairs_local, args.steps)
    ncoll = torch.tensor([float(n_collectives[0])] * 2, dtype=torch.float64)
    if dist is not None:
        lo, hi = ncoll[:1].clone(), ncoll[1:].clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ncoll = torch.cat([lo, hi])
    if rank == 0:
        line = {"metric": "STUB: bench.py timing protocol on CPU tensors (gloo), no rasterizer, no GPU -- not a measurement",
                "value": None, "stub": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "spinup_steps": res["spinup_steps"], "ms_per_step": elapsed / args.steps * 1e3,
                "ms_per_step_cold": None if elapsed_cold is None else elapsed_cold / args.steps * 1e3,
                "rccl_ranks": dist.get_world_size() if dist is not None else 1, "backend": "gloo" if dist is not None else None,
                "ms_per_step_rank_min": rank_ms[0], "ms_per_step_rank_max": rank_ms[1],
                "step_collectives_rank_min": int(ncoll[0].item()), "step_collectives_rank_max": int(ncoll[1].item()),
                "config": {"pairs_per_step": int(pairs_all), "pairs_per_step_per_rank": pairs_local}}
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--distribution", default="saag", choices=["saag", "decoder_like"],
                    help="synthetic Gaussian distribution (SURVEY 8d): create_dummy_saag or the ~3x heavier decoder-like grid")
    ap.add_argument("--images-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-image-loop", action="store_true",
                    help="the LITERAL drop-in call pattern of the reference's train_epoch (TGD:1209-1226): B sequential (N,.) calls "
                         "of the module from a Python loop, torch.stack of the results, one backward through the stack -- what "
                         "INTEGRATION.md route A (swap the class, keep the loop) costs; the default is the batched (B,N,.) call")
    ap.add_argument("--saturation-skip", action="store_true",
                    help="NOT the default and not the headline: FgsDims.saturation_skip=1 (stop compositing sub-tiles whose "
                         "accumulated alpha reached 1.0f); value still counts the reference's pairs, see DESIGN.md")
    ap.add_argument("--tuning", default="",
                    help="experiments only, e.g. seg_len=64,tile_w=16: FgsDims work-split overrides (results are the same; "
                         "the default leaves every choice to the library)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse the N>1 path on a 1-GPU box")
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="after the W warm-up steps, keep running UNTIMED steps until this many milliseconds have passed, so that "
                         "the timed K steps see the GPU at its sustained clocks (a cold MI355X runs the same kernels ~7 %% slower "
                         "for its first ~100 ms of load); the count is reported as spinup_steps.  0 = off")
    ap.add_argument("--force-dist", action="store_true",
                    help="form the process group, barrier and all-reduce even with ONE rank: rehearses the RCCL path "
                         "(communicator creation, collectives on the compute stream) on a one-GPU box")
    ap.add_argument("--stub-renderer", action="store_true",
                    help="NO GPU, NOT a measurement: the whole timing protocol of this file (warm-up, cold region, collectively "
                         "agreed clock spin-up, timed region, MAX/MIN/SUM reductions, the one JSON line) around a stand-in step "
                         "(a small CPU tensor op + the bucket all-reduce on gloo).  tests/test_bench_launch.py runs it with 2 and "
                         "8 ranks: every rank must issue the same collectives in the same order")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="rehearse launcher + rendezvous + collectives on CPU tensors (gloo), no GPU work, no bench line")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- launcher decision: BEFORE torch / fresnel_amd are imported or any GPU call is made ----
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1 or args.force_dist:
            sys.exit(launch_ranks(args, argv))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # stdout carries ONE thing: rank 0's JSON line.  Libraries print there too (RCCL's version banner at communicator
    # creation goes to stdout), so file descriptor 1 points at stderr for the whole run and the line is written to the saved one.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    _import_compute()
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank, result_fd)
    if args.stub_renderer:
        return stub_run(args, world, rank, result_fd)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP rasterizer has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and n_dev < world:
        raise SystemExit(f"rank {rank}: {world} RCCL ranks but only {n_dev} GPU(s) visible (one process per GPU; "
                         f"--backend gloo rehearses several ranks on one GPU)")
    local_rank = local_rank % n_dev  # (gloo rehearsal only: several ranks may share one GPU)
    torch.cuda.set_device(local_rank)  # pin the rank to its GPU before the process group exists
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # nccl IS RCCL on ROCm
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from fresnel_amd import _binding as B
    from fresnel_amd import renderer as R

    N, S, per_gpu = WORKLOADS[args.workload]
    if args.images_per_gpu:
        per_gpu = args.images_per_gpu
    cfg_id = int(args.workload[-1])
    if args.workload == "config1":
        args.no_cpu_baseline = True  # (its CPU figure is BASELINE.md's reference timing; the bounded-sample leg is sized for configs 2-3)
    # image-wise shard: rank r owns images [r*per_gpu, (r+1)*per_gpu)
    pos, scale, quat, col, opa = synth_batch(per_gpu, N, 1000 * cfg_id + rank * per_gpu, device, args.distribution)
    N = pos.shape[1]  # decoder_like: floor(sqrt(N))^2
    phases = None
    g = torch.Generator().manual_seed(977 + rank)
    if args.workload == "config4":  # z snapped to 8 zone centres, edge-aware scale factor, phases U(0,1)
        zone = torch.randint(0, 8, (per_gpu, N), generator=g).float().to(device)
        pos[..., 2] = -2.0 - 2.0 * (zone + 0.5) / 8.0
        scale = scale * (0.5 + 0.5 * torch.rand(per_gpu, N, 1, generator=g).to(device))
        phases = torch.rand(per_gpu, N, generator=g).to(device).requires_grad_(True)
    if args.workload == "config5":
        phases = (torch.rand(per_gpu, N, generator=g) * 2 * np.pi).to(device).requires_grad_(True)
    leaves = [t.requires_grad_(True) for t in (pos, scale, quat, col, opa)]
    cam = R.Camera(0.8 * S, 0.8 * S, S / 2, S / 2, S, S)
    is_asm = args.workload == "config5"
    if is_asm:
        ren = R.ASMWaveFieldRenderer(S, S).to(device)
        wl = torch.tensor([0.0635, 0.05, 0.041], device=device, requires_grad=True)
    else:
        ren = R.TileBasedRenderer(S, S, use_phase_blending=args.workload == "config4", phase_amplitude=0.25,
                                  saturation_skip=args.saturation_skip).to(device)
        if args.tuning:
            ren.tuning = {k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))}
        elif args.workload == "config4":
            ren.tuning = dict(sort_mode=1)  # --use_fresnel_zones 8: depths snapped to zone centres -> zone-key depth sort (fgs_sort.hip)
    g = torch.Generator().manual_seed(4242 + rank)
    gI = torch.randn(per_gpu, 3, S, S, generator=g).to(device)
    gD = (torch.randn(per_gpu, S, S, generator=g) * 0.1).to(device)
    bucket = torch.zeros(DECODER_GRAD_FLOATS, device=device) if dist is not None else None
    grad_leaves = leaves + ([phases] if phases is not None else []) + ([wl] if is_asm else [])

    def step_loop():
        # reference call pattern, TGD:1209-1226: `for b in range(B): rendered, depth = renderer(out['positions'][b], ...,
        # camera, return_depth=True)`; `torch.stack(rendered_images)`; the loss backward runs through the stack
        for t in grad_leaves:
            t.grad = None
        imgs, deps = [], []
        for b in range(per_gpu):
            if is_asm:
                imgs.append(ren(pos[b], scale[b], quat[b], col[b], opa[b], cam, phases=phases[b], wavelengths_rgb=wl))
            else:
                im, dp_ = ren(pos[b], scale[b], quat[b], col[b], opa[b], cam, return_depth=True,
                              phases=None if phases is None else phases[b])
                imgs.append(im)
                deps.append(dp_)
        if is_asm:
            torch.autograd.backward([torch.stack(imgs)], [gI])
        else:
            torch.autograd.backward([torch.stack(imgs), torch.stack(deps)], [gI, gD])
        if bucket is not None:
            dist.all_reduce(bucket)

    def step():
        # the nn.Module call the training harness makes (fresnel_amd/train.py train_step): batched tensors, one Camera
        if args.per_image_loop:
            return step_loop()
        for t in grad_leaves:  # (every leaf: a gradient left in place makes autograd launch an accumulation kernel per step)
            t.grad = None
        if is_asm:
            img = ren(*leaves, cam, phases=phases, wavelengths_rgb=wl)
            torch.autograd.backward([img], [gI])
        else:
            img, dep = ren(*leaves, cam, return_depth=True, phases=phases)
            torch.autograd.backward([img, dep], [gI, gD])
        if bucket is not None:
            dist.all_reduce(bucket)  # decoder-gradient bucket of the DP training step

    # unit of work: composited Gaussian-pixels of this rank's batch (device-side count over the reference bboxes,
    # which are the same for every renderer: DR:594-597 / DR:1240-1247)
    def count_pairs():
        cfg0 = R._Cfg(S, S, (0.0, 0.0, 0.0), 64, False, 0.25,
                      tuning=dict(tile_w=16) if args.workload in ("config4", "config5") or args.saturation_skip else
                      ({k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))} if args.tuning else None))
        _, _, saved, dims, _ = R.forward_raw(*[t.detach() for t in leaves], None, R.pack_cameras(cam, device), cfg0)
        pairs_dev = torch.zeros(1, dtype=torch.int64, device=device)
        B.check(B.load().fgs_count_pairs(ctypes.byref(dims), ctypes.c_void_p(saved.data_ptr()),
                                         ctypes.c_void_p(pairs_dev.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fgs_count_pairs")
        st = R.inspect_saved(saved, dims)
        return (int(st["counters"][0].item()), int(st["counters"][2].item()),  # D; backward work units = depth segments
                int(st["layout"].tile_w), int(pairs_dev.item()))

    # warm-up -> pair count -> cold timed region -> collectively agreed clock spin-up -> timed region: run_protocol
    compute_stages = ("splat_fwd", "field_fwd", "field_bwd", "splat_bwd") if is_asm else ("composite_fwd", "composite_bwd")
    res = run_protocol(step, torch.cuda.synchronize, dist, args, B, torch.zeros(1, device=device), compute_stages, count_pairs)
    D_local, U_local, tile_w, pairs_local = res["extra"]
    dom_stage, spinup_steps = res["dom_stage"], res["spinup_steps"]
    warm_avg = res["sustained_avg"] if res["sustained_avg"] is not None else res["cold_avg"]
    elapsed, elapsed_cold = res["elapsed"], res["elapsed_cold"]

    elapsed, elapsed_cold, pairs_all, rank_ms = reduce_over_ranks(dist, device, elapsed, elapsed_cold, pairs_local, args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    value = pairs_all / (elapsed / args.steps)

    # ---- after the measurement: what the host and the collective cost by themselves (not part of `value`) ----
    # host_enqueue_us: wall time of the Python side of one step (module call(s), ctypes, autograd, allocator) with the GPU
    # running behind -- no synchronisation inside; where it exceeds ms_per_step the step is host-bound.
    torch.cuda.synchronize()
    t_host = []
    for _ in range(5):
        t0 = time.perf_counter()
        step()
        t_host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    host_enqueue_us = sorted(t_host)[len(t_host) // 2] * 1e6
    allreduce_us = None
    if dist is not None:  # the step's one collective alone: K back-to-back all-reduces of the bucket, barrier + sync around them
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(bucket)
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - t0) / 50 * 1e6

    if rank == 0:
        # ---- roofline (rank 0's launches, hipEvent-timed in the library on the stream the kernels run on) ----
        HW = S * S
        tiles = per_gpu * ((S + tile_w - 1) // tile_w) * ((S + 15) // 16)
        extra_units = max(U_local - tiles, 0)  # checkpoints: 5 floats per tile pixel per depth segment after a tile's first
        ckpt_bytes = 5 * 4 * 16 * tile_w
        row_bytes = 4 * 48 if args.workload == "config4" else 40
        alg_bytes = {  # ALGORITHMIC HBM bytes per launch of each stage, B images per launch
            "project": per_gpu * N * (56 + 48 + 8),
            "depth_sort": per_gpu * N * 8 * (1 + 2 * 4),
            "list_building": per_gpu * N * 8 + 4 * D_local,  # dup_emit + tile_ranges + tile_sort stages together
            "composite_fwd": per_gpu * ((40 if args.workload == "config4" else 36) * HW) + 52 * D_local + ckpt_bytes * extra_units  # state planes + rgb + depth out
                             + (PHASE_CKPT_BYTES_PER_DUP * D_local if args.workload == "config4" else 0),  # phase path: (A, Phi) checkpoints written ...
            "composite_bwd": per_gpu * (36 * HW) + (52 + row_bytes) * D_local + ckpt_bytes * extra_units
                             + (PHASE_CKPT_BYTES_PER_DUP * D_local if args.workload == "config4" else 0),  # ... and read back
            "project_bwd": row_bytes * D_local + per_gpu * N * 2 * 56,
            "field_fwd": ASM_BYTES_PER_IMAGE * per_gpu, "field_bwd": ASM_BYTES_PER_IMAGE * per_gpu,
        }
        # per-stage average launch time: the dominant kernel's comes from the timed region (`stage`); the split of
        # the other stages was taken during the warm-up steps of this same run (`warm_avg`)
        avg_ms = dict(warm_avg)
        if res["dom_ms"] is not None:
            avg_ms[dom_stage] = res["dom_ms"]
        avg_ms["list_building"] = sum(avg_ms.get(k, 0.0) for k in ("dup_emit", "tile_sort", "tile_ranges"))

        def stage_roofline(name):
            dur = avg_ms.get(name, 0.0) * 1e-3
            if dur <= 0:
                return None
            if name in FLOPS_PER_PAIR:  # per-pair blend / splat arithmetic on the vector ALU (no MFMA: gather/blend)
                fpp = FLOPS_PER_PAIR_PHASE.get(name, FLOPS_PER_PAIR[name]) if args.workload == "config4" else FLOPS_PER_PAIR[name]
                tfl = fpp * pairs_local / dur / 1e12
                return {"bound": "valu", "achieved": round(tfl, 3), "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(tfl / FP32_VECTOR_PEAK_TF, 5), "avg_launch_ms": round(avg_ms[name], 4),
                        "algorithmic_flops_per_launch": int(fpp * pairs_local), "flops_per_pair": fpp,
                        "algorithmic_bytes_per_launch": int(alg_bytes[name]) if name in alg_bytes else None}
            gbs = alg_bytes[name] / dur / 1e9
            return {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "avg_launch_ms": round(avg_ms[name], 4),
                    "algorithmic_bytes_per_launch": int(alg_bytes[name])}

        kernel_names = {"composite_fwd": "k_phase_fwd" if args.workload == "config4" else ("k_composite_fwd" if args.saturation_skip else "k_blend_fwd_parts"),
                        "composite_bwd": "k_phase_bwd" if args.workload == "config4" else "k_composite_bwd",
                        "splat_fwd": "k_asm_splat<false>", "splat_bwd": "k_asm_splat<true>",
                        "field_fwd": "rocFFT rows + k_colfft_fwd (column FFT x transfer function, plane sum) + k_asm_transfer/max/output",
                        "field_bwd": "k_asm_output_bwd + k_colfft_bwd (gAcc conj(H), inverse column FFT) + rocFFT rows"}
        run_key = f"{args.workload}_{args.distribution}_b{per_gpu}"
        roofline = {"kernel": kernel_names[dom_stage], "stage": dom_stage}
        roofline.update(stage_roofline(dom_stage))
        traffic = pmc_traffic(run_key, kernel_names[dom_stage].split("<")[0]) if not args.saturation_skip else None
        roofline["traffic"] = traffic
        roofline["traffic_source"] = (f"profiles/" + os.path.basename(PMC_SUMMARY) + f" run {run_key}: separate rocprofv3 --pmc FETCH_SIZE / "
                                      "WRITE_SIZE passes of this command, 2*FETCH_SIZE + WRITE_SIZE per launch"
                                      if traffic is not None else None)
        if dom_stage in ("field_fwd", "field_bwd") and pmc_field_traffic(run_key) is not None:
            # a stage of many kernels (rocFFT's + ours): the counters are summed over BOTH field stages of a step, to
            # be compared with the algorithmic bytes of both (2 x 0.36 GB per image)
            roofline["traffic"] = pmc_field_traffic(run_key)
            roofline["traffic_source"] = (f"profiles/" + os.path.basename(PMC_SUMMARY) + f" run {run_key}: 2*FETCH_SIZE + WRITE_SIZE summed over all "
                                          "kernels of field_fwd AND field_bwd per step; algorithmic counterpart = "
                                          f"{int(2 * ASM_BYTES_PER_IMAGE * per_gpu)} bytes")
        roofline["flop_model"] = ("SURVEY 8d: 23 flop per Gaussian-pixel forward, 60 backward; phase path (config 4): its own count, 38 / 71 "
                                  "(recurrence DR:629-667 and its adjoint, FMA = 2, exp / cos / sin / divide = 1; the checkpoint re-run is not "
                                  "credited; bytes include the (A, Phi) checkpoint stream); splat 12 / 36; peak = fp32 vector peak (plain, "
                                  "non-packed VALU code tops out at about half of it, DESIGN.md section 4)")
        stages = {}
        for name in ("project", "depth_sort", "list_building", "composite_fwd", "composite_bwd", "splat_fwd", "field_fwd",
                     "field_bwd", "splat_bwd", "project_bwd"):
            r = stage_roofline(name)
            if r is not None:
                stages[name] = r
        roofline["stages"] = stages
        roofline["stage_avg_ms"] = {k: round(v, 4) for k, v in avg_ms.items() if v > 0}
        roofline["stage_avg_ms_cold"] = {k: round(v, 4) for k, v in res["cold_avg"].items() if v > 0}
        if res["dom_ms_cold"] is not None:
            roofline["stage_avg_ms_cold"][res["dom_stage_cold"]] = round(res["dom_ms_cold"], 4)
        roofline["stage_avg_ms_note"] = ("dominant kernel: timed region; other stages: ten untimed steps of this run with every stage "
                                         "bracketed, taken after the clock spin-up (the W warm-up steps when --spinup-ms 0); "
                                         "stage_avg_ms_cold: the W warm-up steps + the dominant kernel in the cold timed region")
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = cpu_baseline_leg(leaves, gI, gD, N, S)
        dist_name = {"saag": "create_dummy_saag distribution", "decoder_like": "decoder-like distribution (SURVEY 8d)"}[args.distribution]
        line = {
            "metric": "composited Gaussian-pixels/sec + train-step ms, 512^2 render",
            "value": value, "unit": "Gaussian-pixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "spinup_steps": spinup_steps, "ms_per_step": ms_per_step,
            # the same K steps timed straight after the W warm-up steps, BEFORE the clock spin-up: the protocol of rounds
            # 1 and 2, so that rounds stay comparable from the driver record alone (max over ranks, like ms_per_step)
            "ms_per_step_cold": (elapsed_cold / args.steps * 1e3) if elapsed_cold is not None else ms_per_step,
            "value_cold": pairs_all / ((elapsed_cold if elapsed_cold is not None else elapsed) / args.steps),
            "higher_is_better": True,
            "rccl_ranks": dist.get_world_size() if dist is not None else 1, "backend": args.backend if dist is not None else None,
            "ms_per_step_rank_min": rank_ms[0], "ms_per_step_rank_max": rank_ms[1],
            # rank 0's own figures, measured after the timed region: the decoder-gradient all-reduce alone (50 back to back), and
            # the wall time of the Python side of one step with no synchronisation inside (host-bound where it exceeds ms_per_step)
            "allreduce_us": None if allreduce_us is None else round(allreduce_us, 1),
            "host_enqueue_us_per_step": round(host_enqueue_us, 1),
            "call_pattern": ("per-image loop: %d sequential (N,.) module calls + torch.stack, as TGD:1209-1226" % per_gpu)
                            if args.per_image_loop else "batched (B,N,.) module call",
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE {args.workload}: {N} Gaussians, {S}x{S}, {per_gpu} images/GPU "
                                   f"({per_gpu * world} global), {dist_name}, rasterizer fwd+bwd through the nn.Module call"
                                   + (", + all-reduce of the 2.7 MB decoder-grad bucket (" + args.backend + ")" if dist is not None else ""),
                       "gaussians": N, "resolution": S, "images_per_gpu": per_gpu, "global_batch": per_gpu * world,
                       "distribution": args.distribution, "pairs_per_step": int(pairs_all),
                       "tile_w": tile_w, "tile_duplicates_rank0": D_local, "depth_segments_rank0": U_local,
                       "saturation_skip": bool(args.saturation_skip), "parallelism": f"image-wise dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
