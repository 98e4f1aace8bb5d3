import torch, time
def T(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
nb=4096
for dt in (torch.complex128, torch.complex64):
    x = torch.randn(nb,256,256,dtype=dt,device='cuda'); A = torch.randn(150,256,dtype=dt,device='cuda')
    t1 = T(lambda: torch.matmul(x, A.T))
    tmp = torch.matmul(x, A.T)
    t2 = T(lambda: torch.matmul(A, tmp))
    print(dt, "mm1 ms", t1, "mm2 ms", t2)
    del x, tmp
