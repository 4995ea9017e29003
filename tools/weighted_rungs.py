import sys, time, subprocess, os
ROOT='/root/repo' if os.path.exists('/root/repo/proxtv_amd') else os.getcwd()
if len(sys.argv)>1 and sys.argv[1]=='child':
    sys.path.insert(0,ROOT)
    import numpy as np, torch
    from proxtv_amd import _lib, device
    lib=_lib.require_device()
    sc=float(sys.argv[2]); mode=int(sys.argv[3])
    n=4096
    dev=lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    X=dev(np.random.default_rng(0).standard_normal((n,n))); out=device.colmajor_empty((n,n))
    r=np.random.default_rng(1)
    W1,W2=dev(r.uniform(0.5,1.5,(n-1,n))*sc*0.1),dev(r.uniform(0.5,1.5,(n,n-1))*sc*0.1)
    lib.proxtv_set_option(b"chunk_mode",mode)
    run=lambda: device.tv1w_2d(X,W1,W2,out=out)
    run(); best=1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter(); run(); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    print(f"RESULT scale {sc} mode {mode}: {best*1e3:.2f} ms fixups {lib.proxtv_last_fixups()} ran {lib.proxtv_chunk_mode()}",flush=True)
    if mode<0:
        lib.proxtv_set_option(b"verbose",1); run()
    sys.exit(0)
for sc in (5,6,7,8,9):
    for mode in (-1,1,3):
        r=subprocess.run([sys.executable,__file__,'child',str(sc),str(mode)],capture_output=True,text=True)
        for line in (r.stdout+r.stderr).splitlines():
            if line.startswith('RESULT') or ('certain fraction' in line and 'sweep 0 ' in line): print(line.replace('[proxtv_amd] policy: ','    '),flush=True)
