"""Where the non-kernel time of a CG iteration goes: torch profiler over 10 steady iterations."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=12); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
t=time.perf_counter()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=12); torch.cuda.synchronize()
print("wall ms/iter under profiler", (time.perf_counter()-t)/12*1e3)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
# ---- idle gaps on the GPU timeline --------------------------------------------------------
import json, os, collections
prof.export_chrome_trace("/tmp/cg_trace.json")
tr = json.load(open("/tmp/cg_trace.json"))["traceEvents"]
ks = sorted([e for e in tr if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e], key=lambda e: e["ts"])
gaps = collections.defaultdict(lambda: [0.0, 0])
tot_gap = 0.0
for a, b in zip(ks, ks[1:]):
    g = b["ts"] - (a["ts"] + a["dur"])
    if g > 0:
        tot_gap += g
        key = (a["name"][:40] + " -> " + b["name"][:40])
        gaps[key][0] += g; gaps[key][1] += 1
print("GPU span ms", (ks[-1]["ts"] + ks[-1]["dur"] - ks[0]["ts"]) / 1e3, "busy ms", sum(e["dur"] for e in ks) / 1e3, "idle ms", tot_gap / 1e3)
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{g/1e3:8.3f} ms  n={n:3d}  avg {g/n:7.1f} us   {k}")
