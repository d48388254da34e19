"""Sum rocprofv3 counter_collection.csv values per (kernel, counter) and print per-launch averages."""
import sys, glob, csv, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + '/g*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        for key in ('k_composite_bwd', 'k_composite_fwd', 'k_project_bwd', 'k_radix_downsweep', 'k_dup_emit'):
            if key in k:
                acc[key][row['Counter_Name']] += float(row['Counter_Value'])
                calls[key][row['Counter_Name']].add(row['Dispatch_Id'])
out = {}
for k, d in acc.items():
    out[k] = {c: v / max(len(calls[k][c]), 1) for c, v in d.items()}
    print(k)
    for c in sorted(out[k]): print('   %-28s %.4g' % (c, out[k][c]))
json.dump(out, open(root + '/summary.json', 'w'), indent=1)
