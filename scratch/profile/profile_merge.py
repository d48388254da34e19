"""Merge gpurun_out/prof_<tag>/*/pmc_run.json into profiles/<tag>_pmc_summary.json and copy bench lines / kernel stats.
usage: profile_merge.py <tag>"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dst = f"profiles/{tag}_pmc_summary.json"
out = json.load(open(dst)) if os.path.exists(dst) else {
    "note": "rocprofv3 --pmc passes (FETCH_SIZE+GRBM_GUI_ACTIVE | WRITE_SIZE | TCC_HIT_sum+TCC_MISS_sum (from r03 on) | two SQ groups, separate runs of `bench.py <run args> --steps 3 "
            "--warmup 1 --no-cpu-baseline`, scratch/profile/profile_run.sh); values are per launch; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE) KB: "
            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); gather-style reads are uncalibrated, "
            "treat the read side as an upper bound.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles.  avg_us_kernel_trace comes "
            "from the --kernel-trace --stats run of the same command (steps 20).", "runs": {}}
for d in sorted(glob.glob(f"gpurun_out/prof_{tag}/*/")):
    key = os.path.basename(d.rstrip('/'))
    if not os.path.exists(d + "pmc_run.json"):
        continue
    out["runs"][key] = json.load(open(d + "pmc_run.json"))
    shutil.copy(d + "bench.json", f"profiles/{tag}_bench_{key}.json")
    if os.path.exists(d + "kernel_stats.csv"):
        shutil.copy(d + "kernel_stats.csv", f"profiles/{tag}_kernel_stats_{key}.csv")
json.dump(out, open(dst, "w"), indent=1)
print("runs:", sorted(out["runs"]))
