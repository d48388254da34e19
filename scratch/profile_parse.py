"""Collect the round profile (scratch/profile_round.sh) into the files committed under profiles/."""
import sys, glob, csv, collections, json, shutil, os
root = sys.argv[1]
KEYS = ('k_composite_bwd', 'k_blend_fwd_parts', 'k_composite_fwd', 'k_project_bwd', 'k_radix_downsweep', 'k_radix_upsweep', 'k_dup_emit', 'k_tile_order')
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(list)
for f in glob.glob(root + '/g*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        for key in KEYS:
            if key + '(' in row['Kernel_Name'] or key + '<' in row['Kernel_Name']:
                acc[key][row['Counter_Name']] += float(row['Counter_Value'])
                calls[key][row['Counter_Name']].add(row['Dispatch_Id'])
bench = json.loads(open(root + '/bench.json').read().strip().splitlines()[-1])
stats = {}
sf = glob.glob(root + '/stats/**/*kernel_stats.csv', recursive=True)
if sf:
    for row in csv.DictReader(open(sf[0])):
        for key in KEYS:
            if key + '(' in row['Name'] or key + '<' in row['Name']:
                stats[key] = {'calls': int(row['Calls']), 'avg_us': float(row['AverageNs']) / 1e3}
    shutil.copy(sf[0], root + '/kernel_stats.csv')
out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE+GRBM_GUI_ACTIVE | WRITE_SIZE | two SQ groups, separate runs), bench.py --steps 3 --warmup 1, "
               "config 3, 8 images; values are per launch; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); gather-style reads are uncalibrated, treat the read side as an "
               "upper bound.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles.", "kernels": []}
alg = {'k_blend_fwd_parts': bench['roofline'].get('algorithmic_bytes_fwd'), 'k_composite_fwd': bench['roofline'].get('algorithmic_bytes_fwd'), 'k_composite_bwd': bench['roofline'].get('algorithmic_bytes_bwd')}
for k in KEYS:
    if k not in acc: continue
    row = {"kernel": k}
    if k in stats: row["avg_us_kernel_trace"] = round(stats[k]['avg_us'], 2)
    for c, v in sorted(acc[k].items()):
        row[c if c not in ('FETCH_SIZE', 'WRITE_SIZE') else c + '_KB'] = v / max(len(calls[k][c]), 1)
    if 'FETCH_SIZE_KB' in row and 'WRITE_SIZE_KB' in row:
        row['hbm_bytes_corrected'] = (2 * row['FETCH_SIZE_KB'] + row['WRITE_SIZE_KB']) * 1024
    if alg.get(k): row['algorithmic_bytes'] = alg[k]
    out['kernels'].append(row)
json.dump(out, open(root + '/pmc_summary.json', 'w'), indent=1)
for r in out['kernels']:
    print(r['kernel'], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k != 'kernel'})
print(json.dumps(bench)[:400])
