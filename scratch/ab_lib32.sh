#!/bin/bash
# same-box A/B at 32 images per GPU (throughput regime, no tail)
for rep in 1 2; do
for lib in libfgs_hip_prev.so libfgs_hip.so; do
  FGS_LIB=$PWD/fresnel_amd/_lib/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --images-per-gpu 32 > gpurun_out/ab32_$lib.json 2>gpurun_out/ab32_$lib.err || exit 1
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab32_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
st=d['roofline'].get('stage_avg_ms',{})
print('B32', sys.argv[1], d['ms_per_step'], ' '.join('%s=%.3f' % (k[:9], v) for k, v in st.items()))
PY
done; done
