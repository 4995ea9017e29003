#!/usr/bin/env python3
"""HOST-ONLY design study (no GPU, no oracle): how much does the SET OF KNOTS of the two fibre sweeps change from one DR iteration
to the next?  (Prices "verify the previous iteration's structure, walk only where it fails": DESIGN.md 9.)

    python tools/study/structure_study.py [N [lambda ...]]

Per sampled iteration: the fraction of edges whose knot status (none / up / down) differs from the previous iteration's, and the
fraction of 17-sample chunks that contain such an edge -- columns | rows."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import links_study as L
def knots(x, axis):
    d = np.diff(x, axis=axis)
    return np.sign(d).astype(np.int8)
def run(N, lam, iters=35):
    rng = np.random.default_rng(5)
    U = rng.standard_normal((N,N))
    t = np.full(U.shape, 2*U.mean())
    pc=pr=None
    for it in range(iters):
        xc = L.prox_along(t, lam, 0)
        kc = knots(xc,0)
        s = t - xc; sp = 2*s - t; v = U - sp
        xr = L.prox_along(v, lam, 1)
        kr = knots(xr,1)
        t = 0.5*(t+(sp+2*xr))
        if pc is not None and it in (1,2,3,5,8,12,16,20,25,30,34):
            dc = (kc!=pc); dr=(kr!=pr)
            # per 17-chunk along fibre
            def chunkfrac(d,axis):
                d = np.moveaxis(d,axis,-1)
                n = (d.shape[1]//17)*17
                c = d[:,:n].reshape(d.shape[0],-1,17).any(2)
                return c.mean()
            print(f"lam {lam} it {it:2d}: col edges changed {dc.mean():.4f} chunks {chunkfrac(dc,0):.3f} | row edges changed {dr.mean():.4f} chunks {chunkfrac(dr,1):.3f}")
        pc,pr=kc,kr
if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    for lam in ([float(v) for v in sys.argv[2:]] or [0.1, 0.5, 1.0]):
        run(N, lam)
