"""Per-tile timeline of the backward compositing kernel (scratch instrumentation, FGS_DBG_TS)."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import scratch.ab as ab
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
step = ab.setup(32768, 512, nimg)
for _ in range(3): step()
torch.cuda.synchronize()
path = 'gpurun_out/ts_%d.bin' % nimg
os.environ['FGS_DBG_TS'] = path
step(); torch.cuda.synchronize()
del os.environ['FGS_DBG_TS']
d = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
d = d[d[:, 1] > 0]
t0 = d[:, 0].astype(np.int64); t1 = d[:, 1].astype(np.int64); hw = d[:, 2]; n = d[:, 3].astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base
tick = 1e-5  # wall_clock64: 100 MHz -> 10 ns = 1e-5 ms
print('blocks', len(d), 'kernel span ms', t1.max() * tick, 'list len mean/min/max', n.mean(), n.min(), n.max())
print('sum list entries', n.sum())
dur = (t1 - t0) * tick
print('tile duration ms: mean %.4f max %.4f; ns per entry (mean) %.1f' % (dur.mean(), dur.max(), 1e6 * dur.sum() / n.sum()))
# active waves over time
T = t1.max(); bins = 40
edges = np.linspace(0, T, bins + 1)
act = np.zeros(bins)
for i in range(bins):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None)
    act[i] = ov.sum() / (hi - lo)
print('avg resident waves per 2.5%% of the span:', ' '.join('%d' % a for a in act))
hwid = (hw & 0xFFFFFFFF).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
cu = (hwid >> 8) & 0xF; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1; simd = (hwid >> 4) & 3
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
u = np.unique(key)
print('distinct CUs seen', len(u))
per_cu_end = np.array([t1[key == k].max() for k in u]) * tick
per_cu_work = np.array([n[key == k].sum() for k in u])
print('per-CU finish time ms: min %.3f median %.3f max %.3f' % (per_cu_end.min(), np.median(per_cu_end), per_cu_end.max()))
print('per-CU entries: min %d median %d max %d (mean %.0f)' % (per_cu_work.min(), np.median(per_cu_work), per_cu_work.max(), per_cu_work.mean()))
# start-time distribution: how many blocks start in the first 5% of the span
print('blocks started within first 5%% of span: %d' % (t0 < 0.05 * T).sum())
order = np.argsort(t0)
print('first-started blocks list len (first 10):', n[order[:10]], ' last-started:', n[order[-10:]])
