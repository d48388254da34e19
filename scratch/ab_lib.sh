#!/bin/bash
# same-box A/B of two builds of the library: prints ms_per_step, fwd, bwd, project_bwd for each, twice
for rep in 1 2; do
for lib in libfgs_hip_prev.so libfgs_hip.so; do
  FGS_LIB=$PWD/fresnel_amd/_lib/$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$lib.json 2>gpurun_out/ab_$lib.err || exit 1
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
st=d['roofline'].get('stage_avg_ms',{}) if 'stage_avg_ms' in d.get('roofline',{}) else d.get('stage_avg_ms',{})
print(sys.argv[1], d['ms_per_step'], ' '.join('%s=%.3f' % (k[:9], v) for k, v in st.items()))
PY
done; done
