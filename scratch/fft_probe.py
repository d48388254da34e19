"""One-off: rocFFT 2-D C2C vs two 1-D row passes + a transpose on the config-5 batch (384 fields of 512 x 512)."""
import torch, time
dev = torch.device('cuda:0')
x = torch.randn(384, 512, 512, dtype=torch.complex64, device=dev)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('fft2           %.3f ms' % t(lambda: torch.fft.fft2(x)))
print('fft rows       %.3f ms' % t(lambda: torch.fft.fft(x, dim=-1)))
print('fft cols       %.3f ms' % t(lambda: torch.fft.fft(x, dim=-2)))
print('transpose copy %.3f ms' % t(lambda: x.transpose(-1, -2).contiguous()))
y = torch.empty_like(x)
print('plain copy     %.3f ms' % t(lambda: y.copy_(x)))
