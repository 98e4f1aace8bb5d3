"""Multi-mode fused CG: torch-side GPU time and idle gaps (torch profiler, 6 iterations)."""
import sys, time, json, collections; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
M = 4
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan = D(p['psi']),D(p['scan'])
prb = D(syn.hermite_modes(256, M))
data = torch.zeros((1,4096,256,256),dtype=torch.float32,device='cuda')
for k in range(M): data += torch.abs(slv.fwd(psi,scan,prb[:,k].contiguous()))**2
slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=8); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=6); torch.cuda.synchronize()
prof.export_chrome_trace("/tmp/cgm_trace.json")
tr = json.load(open("/tmp/cgm_trace.json"))["traceEvents"]
ks = sorted([e for e in tr if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e], key=lambda e: e["ts"])
byname = collections.defaultdict(lambda: [0.0, 0])
for e in ks:
    byname[e["name"][:70]][0] += e["dur"]; byname[e["name"][:70]][1] += 1
busy = sum(e["dur"] for e in ks); span = ks[-1]["ts"] + ks[-1]["dur"] - ks[0]["ts"]
print("span ms/iter %.2f busy %.2f idle %.2f" % (span/6e3, busy/6e3, (span-busy)/6e3))
for k,(d,n) in sorted(byname.items(), key=lambda kv:-kv[1][0])[:16]:
    print("%8.3f ms/iter n/iter=%5.1f  %s" % (d/6e3, n/6, k))
