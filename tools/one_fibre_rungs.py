import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from proxtv_amd import _lib, device
lib=_lib.require_device()
r=np.random.default_rng(3)
for kind in ("walk+noise","noise"):
    for n in (1<<22, 1<<20):
        x = (np.cumsum(r.standard_normal(n))*0.05 if kind=="walk+noise" else 0) + r.standard_normal(n)
        sig=torch.from_numpy(x).cuda(); out=torch.empty_like(sig)
        for lam in (0.7,1.0,1.2,1.5,2.0,3.0):
            row=[]
            for mode in (-1,1,2,3):
                lib.proxtv_set_option(b"chunk_mode",mode)
                run=lambda: device.tv1_fibres(sig.reshape(-1,1),lam,0,out=out.reshape(-1,1))
                run(); best=1e9
                for _ in range(2):
                    torch.cuda.synchronize(); t0=time.perf_counter(); run(); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
                row.append(f"mode {mode}: {best*1e3:8.2f} ms (ran {lib.proxtv_chunk_mode()}, fixups {lib.proxtv_last_fixups()})")
            print(kind,n,"lambda",lam," | ".join(row),flush=True)
lib.proxtv_set_option(b"chunk_mode",-1)
