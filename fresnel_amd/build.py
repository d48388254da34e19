"""In-tree build of libfgs_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.

    python -m fresnel_amd.build [--force] [--verbose]

The shared library lands in fresnel_amd/_lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
OBJ_DIR = os.path.join(HERE, "_lib", "obj")
LIB = os.path.join(OUT_DIR, "libfgs_hip.so")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"]
# -fno-slp-vectorize on the compositing / splat translation units: packed fp32 instructions (v_pk_fma_f32 ...) issue at
# two plain instructions' cost on gfx950 and forming their register pairs costs v_movs on top:
# -9 % on the whole config-5 step when fgs_asm.hip got the flag.  (Not on fgs_project.hip: its row sums / double-precision
# adjoint are 10 % FASTER with the vectoriser on.)
NO_SLP = ["-fno-slp-vectorize"]
# Machine-scheduler strategy of the compositing unit: "max-memory-clause" orders the list loops so that the LDS record reads
# sit together ahead of the arithmetic; same-box A/B over 4 x 100 steps at config 3 (round 3): step 1.856-1.864 -> 1.832-1.838 ms
# (-1.35 %; forward -2 %, backward -1 %), "max-ilp" -0.7 %, wave-priority / metric-bias: nothing.
# (FGS_BUILD_SCHED=<strategy|none>: experiment builds with another strategy; counts as an experiment define)
_SCHED_ENV = os.environ.get("FGS_BUILD_SCHED", "")
SCHED = ([] if _SCHED_ENV == "none" else ["-mllvm", "-amdgpu-sched-strategy=" + (_SCHED_ENV or "max-memory-clause")])
# per-file extra flags.  fgs_project.hip carries the "canonical fp32" contract: no FMA
# contraction, IEEE divide/sqrt, so integer decisions match the CPU oracle bit for bit.
SOURCES = {
    "fgs_plan.cpp": ["-x", "c++"],  # host-only plan / layout arithmetic (also built by g++ under ASan/UBSan in tests/)
    "fgs_api.hip": [],
    "fgs_project.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"],
    "fgs_sort.hip": [],
    "fgs_bin.hip": [],
    "fgs_composite.hip": ["-ffast-math", "-fno-finite-math-only"] + NO_SLP + SCHED,
    # no fast-math here: the transfer function needs the accurate sincosf.  No FMA contraction either (round 4): with kz^2 =
    # 1/l^2 - fx^2 - fy^2 contracted, dL/dlambda -- which weights the near-evanescent frequencies by 1 / kz -- came out 2e-4 from
    # the reference on the K5 fixture where the reference's own fp32 run is 4e-6 from its fp64 run, and every other K5 tensor 2-3x
    # further out than the reference's fp32 run.  kz^2 itself is protected by fgs_kz2 (opaque values: hipcc's __fmul_rn do NOT stop
    # contraction, and `#pragma clang fp contract(off)` does nothing under the global setting); the flag makes the rest of the field
    # chain independent of the compiler's fusion choices as well (profiles/r04_dlambda_probe.txt, r04_referee_table.txt).  The two
    # VALU-bound loops have their FMAs WRITTEN OUT (k_asm_splat's pass, fgs_colfft.h's butterflies), so the flag costs nothing:
    # config 5 at 8 images 1.845 ms contracted / 1.958 with the flag alone (splat kernels +10 %) / 1.852 with the explicit FMAs;
    # 0.437 / 0.462 / 0.438 at one image (profiles/r04_ab_config5_contract_off.txt).
    "fgs_asm.hip": NO_SLP + ["-ffp-contract=off"],
    "fgs_gather.hip": [],
    "fgs_fft.hip": NO_SLP,
    "fgs_spectral.hip": [],
}
LINK_LIBS = ["-lhipfft"]


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=False, defines=(), suffix=""):
    """`defines` / `suffix`: experiment builds (`python -m fresnel_amd.build --define FGS_X=1 --suffix _x` writes
    _lib/libfgs_hip_x.so next to the product library; select it with FGS_LIB=... for same-box A/B runs)."""
    global OBJ_DIR, LIB
    if suffix:
        OBJ_DIR = os.path.join(OUT_DIR, "obj" + suffix)
        LIB = os.path.join(OUT_DIR, f"libfgs_hip{suffix}.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "fgs.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    objs, rebuilt = [], False
    for name, extra in SOURCES.items():
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ_DIR, os.path.splitext(name)[0] + ".o")
        # experiment builds: FGS_BUILD_EXTRA_<FILE STEM> = extra compiler flags for one file (e.g. scheduler options)
        more = os.environ.get("FGS_BUILD_EXTRA_" + name.split(".")[0].upper(), "").split()
        common = COMMON if name.endswith(".hip") else [c for c in COMMON if "offload" not in c and "gpu-rdc" not in c]
        # experiment builds: every unit sees FGS_EXPERIMENT_BUILD (timing-only switches #error without it), and fgs_version() lists
        # the defines (fgs_api.hip), so a library selected through FGS_LIB says what it is
        exp = []
        if defines or more or _SCHED_ENV:
            exp = ["-DFGS_EXPERIMENT_BUILD=1"]
            if name == "fgs_api.hip":
                exp.append('-DFGS_BUILD_DEFINES="' + " ".join(list(defines) + more + (["sched=" + _SCHED_ENV] if _SCHED_ENV else [])).replace('"', "'") + '"')
        cmd = [hipcc] + common + extra + more + ["-D" + d for d in defines] + exp + ["-c", src, "-o", obj]
        # an object is stale when its source or a header is newer OR when it was compiled with another command line (the
        # flags live in this file: editing them must rebuild)
        stamp = obj + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or _newer(src, obj) or os.path.getmtime(obj) < hdr_time:
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(" ".join(cmd))
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + LINK_LIBS
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    _defs = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--define"]
    _suf = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--suffix"]
    print(build(force="--force" in sys.argv or bool(_defs), verbose="--verbose" in sys.argv or "-v" in sys.argv,
                defines=_defs, suffix=_suf[0] if _suf else ""))
