"""Collect one run of scratch/profile/profile_run.sh into <out>/pmc_run.json (merged into profiles/<tag>_pmc_summary.json
by scratch/profile/profile_merge.py)."""
import sys, glob, csv, collections, json
root, key = sys.argv[1], sys.argv[2]
KEYS = ('k_colfft_fwd', 'k_colfft_bwd', 'k_phase_fwd', 'k_phase_bwd', 'k_composite_bwd_phase', 'k_composite_bwd', 'k_blend_fwd_parts', 'k_composite_fwd', 'k_asm_splat', 'k_asm_accumulate_bwd',
        'k_asm_accumulate', 'k_asm_transfer', 'k_project_bwd', 'k_project', 'k_sort_image', 'k_radix_downsweep', 'k_radix_upsweep',
        'k_mask_build', 'k_mask_count', 'k_mask_emit', 'k_row_sum', 'k_tile_pre', 'k_tile_post', 'k_dup_emit')


def match(name):
    for k in KEYS:  # longest names first in KEYS where one is a prefix of another
        if name.startswith(k) or ('::' + k) in name or (' ' + k) in name:
            rest = name[name.index(k) + len(k):][:1]
            if rest in ('', '(', '<', ' '):
                if k == 'k_asm_splat':
                    return k + ('<true>' if '<true' in name else '<false>')
                return k
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(set))
# field stages of the ASM renderer (config 5): rocFFT's kernels + the spectral elementwise kernels, both directions
FIELD = ('fft_rtc', 'k_asm_transfer', 'k_asm_accumulate', 'k_asm_max', 'k_asm_output', 'k_colfft', 'k_asm_wavelength_grad', 'k_fft_twiddles')
field = collections.defaultdict(float)
for f in glob.glob(root + '/g*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = match(row['Kernel_Name'])
        if k:
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            calls[k][row['Counter_Name']].add(row['Dispatch_Id'])
        if any(t in row['Kernel_Name'] for t in FIELD) and row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            field[row['Counter_Name']] += float(row['Counter_Value'])
stats = {}
for sf in glob.glob(root + '/stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(sf)):
        k = match(row['Name'])
        if k:
            stats[k] = {'calls': int(row['Calls']), 'avg_us': float(row['AverageNs']) / 1e3}
bench = json.loads(open(root + '/bench.json').read().strip().splitlines()[-1])
out = {"command": "bench.py " + " ".join(f"--{a} {b}" for a, b in (("workload", key.split('_')[0]),)) + f"  (run key {key})",
       "ms_per_step": bench["ms_per_step"], "value": bench["value"], "roofline_kernel": bench["roofline"]["kernel"],
       "roofline_frac": bench["roofline"]["frac"], "kernels": []}
for k in sorted(set(list(acc) + list(stats))):
    row = {"kernel": k}
    if k in stats:
        row["avg_us_kernel_trace"] = round(stats[k]['avg_us'], 2); row["calls_kernel_trace"] = stats[k]['calls']
    for c, v in sorted(acc.get(k, {}).items()):
        row[c if c not in ('FETCH_SIZE', 'WRITE_SIZE') else c + '_KB'] = v / max(len(calls[k][c]), 1)
    if 'FETCH_SIZE_KB' in row and 'WRITE_SIZE_KB' in row:
        row['hbm_bytes_corrected'] = (2 * row['FETCH_SIZE_KB'] + row['WRITE_SIZE_KB']) * 1024
    if row.get('TCC_HIT_sum') is not None and row.get('TCC_MISS_sum') is not None and row['TCC_HIT_sum'] + row['TCC_MISS_sum'] > 0:
        row['l2_hit_rate'] = round(row['TCC_HIT_sum'] / (row['TCC_HIT_sum'] + row['TCC_MISS_sum']), 4)
    if row.get('SQ_LDS_IDX_ACTIVE'):
        row['lds_conflict_share'] = round(row.get('SQ_LDS_BANK_CONFLICT', 0.0) / row['SQ_LDS_IDX_ACTIVE'], 4)
    out['kernels'].append(row)
if field:
    # steps of a PMC pass = launches of a once-per-step kernel of this path (`--steps 3 --warmup 1` + the five untimed steps of bench.py's
    # host-enqueue measurement since round 5: nine; a hard-coded four inflated the figure 2.25x in the first r05 summaries)
    once = calls.get('k_asm_splat<true>', {}).get('FETCH_SIZE') or calls.get('k_colfft_bwd', {}).get('FETCH_SIZE') or ()
    nsteps = max(len(once), 1) if once else 4
    out["field_stages_hbm_bytes_per_step"] = (2 * field['FETCH_SIZE'] + field['WRITE_SIZE']) * 1024 / nsteps
    out["field_stages_note"] = ("(2*FETCH_SIZE + WRITE_SIZE) summed over rocFFT's kernels and k_asm_transfer / accumulate[_bwd] / max / "
                                f"output[_bwd] of field_fwd AND field_bwd, per step ({nsteps} steps in the PMC pass, counted from the launches of a "
                                "once-per-step kernel)")
json.dump(out, open(root + '/pmc_run.json', 'w'), indent=1)
for r in out['kernels']:
    print(r['kernel'], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in
                        ('avg_us_kernel_trace', 'hbm_bytes_corrected', 'lds_conflict_share', 'SQ_INSTS_VALU', 'WRITE_SIZE_KB', 'FETCH_SIZE_KB', 'l2_hit_rate')})
print(json.dumps(bench)[:600])
