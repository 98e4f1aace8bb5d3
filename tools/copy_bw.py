"""Device copy / fill bandwidth reference points for the roofline discussion."""
import torch, time
x = torch.empty(2*1024**3//8, dtype=torch.complex64, device='cuda'); y = torch.empty_like(x)
def T(f, n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
t = T(lambda: y.copy_(x)); print("copy 2 GiB -> 2 GiB: %.3f ms, %.2f TB/s (read+write)" % (t*1e3, 2*x.numel()*8/t/1e12))
t = T(lambda: y.zero_()); print("fill 2 GiB: %.3f ms, %.2f TB/s" % (t*1e3, x.numel()*8/t/1e12))
t = T(lambda: x.real.sum()); print("read-reduce 1 GiB (strided real): %.3f ms" % (t*1e3))
xf = torch.empty(2*1024**3//4, dtype=torch.float32, device='cuda')
t = T(lambda: xf.sum()); print("read 2 GiB (sum): %.3f ms, %.2f TB/s" % (t*1e3, xf.numel()*4/t/1e12))
