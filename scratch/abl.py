import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import scratch.ab as ab
for nimg in (8, 32):
    step = ab.setup(32768, 512, nimg)
    print('B', nimg, 'fwd ablations 0=full 1=no passes 2=no gather (median,min ms):', ab.ab(step, 'FGS_ABL', [0, 1, 2], 'composite_fwd'))
