/*
 * tv1d_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's exact 1-D TV-L1 proximity solvers:
 *     prox(y) = argmin_x  1/2 ||x - y||^2 + sum_i lambda_i |x_i - x_{i+1}|
 * See tv_oracle.h for the parity status (pinned against oracle/_ref and
 * tests/golden).  The arithmetic (operation order, tolerances) follows the
 * cited reference lines so that results agree to the last few ulps; the code
 * organisation is this repository's own.
 */
#include "tv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/*  Linearized taut string ("tube walker")                                    */
/* ------------------------------------------------------------------------- */

/*
 * State of the walk along the tube (reference: src/TVL1opt_hybridtautstring.cpp:57-67).
 * Two candidate straight pieces leave the last knot k0: a low one (slope lo)
 * hugging the tube floor and a high one (slope hi) hugging the ceiling.
 * hlo/hhi are their heights relative to the tube centre at the current sample,
 * klo/khi the last samples where they touched their own wall.
 */
typedef struct {
    double lo, hi;
    double hlo, hhi;
    int klo, khi, k0;
} tube_t;

static inline void paint(double *x, int from, int to, double v)
{
    for (int j = from; j <= to; j++) x[j] = v;
}

/*
 * Walks the tube.  `w == NULL`  -> uniform half-width `lam`
 *                                  (src/TVL1opt_hybridtautstring.cpp:56-235, == src/TVL1opt.cpp:359-564
 *                                   when the step budget is infinite);
 *                  `w != NULL`  -> per-edge half-widths w[0..n-2]
 *                                  (src/TVL1Wopt.cpp:364-567).
 * `budget`: number of walked samples after which, at the next knot, the walk
 * gives up (hybrid switch, src/TVL1opt_hybridtautstring.cpp:31-35).  Returns n
 * when the whole prox was written, otherwise the index at which the caller
 * must resume with another solver; *resume_offset receives the height of the
 * string above the tube centre at that index.
 */
static int tube_walk(const double *y, const double *w, double lam, int n, double *x,
                     double budget, double *resume_offset)
{
    tube_t s;
    int i = 0;
    int walked = 0;
    const int last = n - 1;

    /* src/TVL1opt_hybridtautstring.cpp:75-80 ; src/TVL1Wopt.cpp:386-391 */
    const double r0 = w ? w[0] : lam;
    s.hlo = s.hhi = 0;
    s.lo = -r0 + y[0];
    s.hi = r0 + y[0];
    s.k0 = -1;
    s.klo = s.khi = 0;

    while (i < n) {
        /* interior samples: tube half-width r */
        while (i < last) {
            const double r = w ? w[i] : lam;

            /* low piece against the ceiling (hybrid :89-111 ; weighted :401-423) */
            s.hlo += s.lo - y[i];
            if (r < s.hlo) {
                i = s.klo + 1;
                paint(x, s.k0 + 1, s.klo, s.lo);
                if (walked > budget && i + 1 < n) { *resume_offset = -lam; return i; }
                s.k0 = s.klo;
                if (w) {
                    s.lo = y[i] + w[i - 1] - w[i];
                    s.hi = y[i] + w[i - 1] + w[i];
                    s.hhi = w[i];
                    s.hlo = -w[i];
                } else {
                    s.lo = y[i];
                    s.hi = 2 * lam + y[i];
                    s.hhi = lam;
                    s.hlo = -lam;
                }
                s.klo = s.khi = i;
                i++; walked++;
                continue;
            }

            /* high piece against the floor (hybrid :115-137 ; weighted :427-449) */
            s.hhi += s.hi - y[i];
            if (-r > s.hhi) {
                i = s.khi + 1;
                paint(x, s.k0 + 1, s.khi, s.hi);
                if (walked > budget && i + 1 < n) { *resume_offset = lam; return i; }
                s.k0 = s.khi;
                if (w) {
                    s.hi = y[i] - w[i - 1] + w[i];
                    s.lo = y[i] - w[i - 1] - w[i];
                    s.hhi = w[i];
                    s.hlo = -w[i];
                } else {
                    s.hi = y[i];
                    s.lo = 2 * (-lam) + y[i];
                    s.hhi = lam;
                    s.hlo = -lam;
                }
                s.klo = s.khi = i;
                i++; walked++;
                continue;
            }

            /* no violation: pull the pieces back inside the tube (hybrid :142-160 ; weighted :454-472) */
            if (s.hhi >= r) {
                s.hi += (r - s.hhi) / (i - s.k0);
                s.hhi = r;
                s.khi = i;
            }
            if (s.hlo <= -r) {
                s.lo += (-r - s.hlo) / (i - s.k0);
                s.hlo = -r;
                s.klo = i;
            }
            i++; walked++;
        }

        /* Only reachable with a negative lambda (bend at sample 0, restart steps to i == n when n == 2): the
           reference falls through and reads y[n] (hybrid :172 after :110) -- undefined.  Stop instead. */
        if (i > last) break;

        /* last sample: tube collapses to its centre, tested with the absolute
           tolerance of src/general.h:64-67 (hybrid :172-230 ; weighted :488-553) */
        s.hlo += s.lo - y[i];
        if (s.hlo > ORC_EPSILON) {
            i = s.klo + 1;
            paint(x, s.k0 + 1, s.klo, s.lo);
            if (walked > budget && i + 1 < n) { *resume_offset = -lam; return i; }
            s.k0 = s.klo;
            if (w) {
                const double rn = (i == last) ? 0 : w[i];
                s.lo = y[i] + w[i - 1] - rn;
                s.hi = y[i] + w[i - 1] + rn;
                s.hhi = s.hlo = -w[i - 1];
            } else {
                s.lo = y[i];
                s.hi = 2 * lam + y[i];
                s.hhi = s.hlo = -lam;
            }
            s.klo = s.khi = i;
            continue;
        }
        s.hhi += s.hi - y[i];
        if (s.hhi < -ORC_EPSILON) {
            i = s.khi + 1;
            paint(x, s.k0 + 1, s.khi, s.hi);
            if (walked > budget && i + 1 < n) { *resume_offset = lam; return i; }
            s.k0 = s.khi;
            if (w) {
                const double rn = (i == last) ? 0 : w[i];
                s.hi = y[i] - w[i - 1] + rn;
                s.lo = y[i] - w[i - 1] - rn;
                s.hhi = s.hlo = w[i - 1];
            } else {
                s.hi = y[i];
                s.lo = 2 * (-lam) + y[i];
                s.hhi = s.hlo = lam;
            }
            s.klo = s.khi = i;
            continue;
        }
        if (s.hlo <= 0)
            s.lo += (-s.hlo) / (i - s.k0);
        i++;
    }

    /* closing piece (hybrid :233-235) */
    paint(x, s.k0 + 1, last, s.lo);
    return n;
}

int orc_linearizedTautString_TV1(const double *y, double lambda, double *x, int n)
{
    double off;
    if (n <= 0) return 1;
    tube_walk(y, NULL, lambda, n, x, DBL_MAX, &off);
    return 1;
}

int orc_tautString_TV1_Weighted(const double *y, const double *lambda, double *x, int n)
{
    double off;
    if (n <= 0) return 1;
    /* the reference reads lambda[0] even for n == 1 (out of bounds, src/TVL1Wopt.cpp:388);
       the only meaningful answer there is x = y */
    if (n == 1) { x[0] = y[0]; return 1; }
    tube_walk(y, lambda, 0.0, n, x, DBL_MAX, &off);
    return 1;
}

void orc_hybridTautString_TV1_custom(const double *y, int n, double lambda, double *x, double backtracksexp)
{
    double off = 0;
    if (n <= 0) return;
    /* src/TVL1opt_hybridtautstring.cpp:73 */
    const double budget = pow((double)n, backtracksexp);
    const int at = tube_walk(y, NULL, lambda, n, x, budget, &off);
    if (at < n)
        orc_classicTautString_TV1_offset(y + at, n - at, lambda, x + at, off);
}

void orc_hybridTautString_TV1(const double *y, int n, double lambda, double *x)
{
    /* BACKTRACKSEXP, src/TVL1opt_hybridtautstring.cpp:11 */
    orc_hybridTautString_TV1_custom(y, n, lambda, x, 1.05);
}

/* ------------------------------------------------------------------------- */
/*  Classic taut string (convex/concave hulls of the tube walls)              */
/* ------------------------------------------------------------------------- */

/* one straight piece of a hull: spans dx samples and dy in cumulative height
   (src/TVL1opt_tautstring.cpp:24-28) */
typedef struct { int dx; double dy; double slope; } piece_t;

/* double-ended run of pieces inside a flat array (src/TVL1opt_tautstring.cpp:40-44) */
typedef struct { piece_t *buf; int head, tail; } hull_t;   /* valid pieces: buf[head..tail] */

static inline int hull_count(const hull_t *h) { return h->tail - h->head + 1; }
static inline void hull_reset(hull_t *h) { h->head = 0; h->tail = -1; }
static inline void hull_push(hull_t *h, piece_t p) { h->buf[++h->tail] = p; }

/* append to the concave majorant of the floor (upper == 0) or to the convex
   minorant of the ceiling (upper == 1), merging pieces that would break
   concavity / convexity (src/TVL1opt_tautstring.cpp:149-181) */
static inline void hull_append(hull_t *h, piece_t p, int upper)
{
    const piece_t *lastp = &h->buf[h->tail];
    const int bends_wrong = upper ? (p.slope < lastp->slope) : (p.slope > lastp->slope);
    if (bends_wrong) {
        int left = hull_count(h);
        for (;;) {
            const piece_t *q = &h->buf[h->tail--];
            p.dx += q->dx;
            p.dy += q->dy;
            left--;
            if (left < 1) break;
            const double reach = p.dx * h->buf[h->tail].slope;
            if (upper ? !(p.dy < reach) : !(p.dy > reach)) break;
        }
        p.slope = p.dy / p.dx;
    }
    hull_push(h, p);
}

int orc_classicTautString_TV1_offset(const double *signal, int n, double lam, double *prox, double offset)
{
    /* degenerate inputs: src/TVL1opt_tautstring.cpp:258-263 */
    if (n <= 0) return 1;
    if (lam <= 0 || n == 1) { memcpy(prox, signal, (size_t)n * sizeof(double)); return 1; }

    hull_t lowh, upph;                      /* majorant of floor / minorant of ceiling */
    lowh.buf = (piece_t *)malloc(sizeof(piece_t) * (size_t)n);
    upph.buf = (piece_t *)malloc(sizeof(piece_t) * (size_t)n);
    if (!lowh.buf || !upph.buf) { free(lowh.buf); free(upph.buf); return 0; }
    hull_reset(&lowh);
    hull_reset(&upph);

    /* first sample: src/TVL1opt_tautstring.cpp:271-278 */
    piece_t p;
    p.dx = 1; p.slope = p.dy = signal[0] - offset - lam; hull_push(&lowh, p);
    p.dx = 1; p.slope = p.dy = signal[0] - offset + lam; hull_push(&upph, p);

    /* current knot and running tube centre: :281-291 */
    int    knot_x = 0;        double knot_y = offset;
    int    seen_x = 1;        double seen_y = signal[0];
    double *out = prox;

    for (int i = 1; i < n - 1; i++) {
        p.dx = 1; p.slope = p.dy = signal[i]; hull_append(&lowh, p, 0);
        p.dx = 1; p.slope = p.dy = signal[i]; hull_append(&upph, p, 1);
        seen_x++;
        seen_y += signal[i];

        /* leading slopes crossed -> the shorter leading piece is part of the string (:308-313, 187-223) */
        while (upph.buf[upph.head].slope < lowh.buf[lowh.head].slope) {
            const piece_t up = upph.buf[upph.head];
            const piece_t lo = lowh.buf[lowh.head];
            piece_t fixed, span;
            if (up.dx < lo.dx) {
                fixed = up;
                upph.head++;
                span.dx = seen_x - knot_x - up.dx;
                span.dy = seen_y - lam - knot_y - up.dy;
                span.slope = span.dy / span.dx;
                hull_reset(&lowh);
                hull_push(&lowh, span);
            } else {
                fixed = lo;
                lowh.head++;
                span.dx = seen_x - knot_x - lo.dx;
                span.dy = seen_y + lam - knot_y - lo.dy;
                span.slope = span.dy / span.dx;
                hull_reset(&upph);
                hull_push(&upph, span);
            }
            knot_x += fixed.dx;
            knot_y += fixed.dy;
            for (int j = 0; j < fixed.dx; j++) out[j] = fixed.slope;
            out += fixed.dx;
        }
    }

    /* last sample closes the tube: both walls end at the centre (:317-324) */
    p.dx = 1; p.slope = p.dy = signal[n - 1] + lam; hull_append(&lowh, p, 0);
    p.dx = 1; p.slope = p.dy = signal[n - 1] - lam; hull_append(&upph, p, 1);

    /* the hull with more pieces is the remaining string (:330-335) */
    const hull_t *rest = (hull_count(&lowh) > hull_count(&upph)) ? &lowh : &upph;
    for (int k = rest->head; k <= rest->tail; k++) {
        for (int j = 0; j < rest->buf[k].dx; j++) out[j] = rest->buf[k].slope;
        out += rest->buf[k].dx;
    }

    free(lowh.buf);
    free(upph.buf);
    return 1;
}

int orc_classicTautString_TV1(const double *signal, int n, double lam, double *prox)
{
    return orc_classicTautString_TV1_offset(signal, n, lam, prox, 0);
}

/* ------------------------------------------------------------------------- */
/*  Condat's direct algorithm                                                 */
/* ------------------------------------------------------------------------- */

/*
 * L. Condat, "A direct algorithm for 1D total variation denoising", IEEE SPL 2013;
 * restated after the reference's call-compatible implementation
 * (src/condat_fast_tv.cpp:78-121).  Dual bounds ulo/uhi, candidate segment
 * values vlo/vhi, last saturation points klo/khi, segment start k0.
 */
void orc_TV1D_denoise(const double *input, double *output, int width, double lambda)
{
    if (!(width > 0 && lambda >= 0)) return;                 /* :79 */

    int k = 0, k0 = 0, khi = 0, klo = 0;
    double ulo = lambda, uhi = -lambda;
    double vlo = input[0] - lambda, vhi = input[0] + lambda;
    const double two = 2.0 * lambda, neg = -lambda;

    for (;;) {
        if (k == width - 1) {
            /* right boundary: dual must return to zero (:87-99) */
            if (ulo < 0.0) {
                do output[k0++] = vlo; while (k0 <= klo);
                klo = k = k0;
                vlo = input[k0];
                ulo = lambda;
                uhi = vlo + ulo - vhi;
            } else if (uhi > 0.0) {
                do output[k0++] = vhi; while (k0 <= khi);
                khi = k = k0;
                vhi = input[k0];
                uhi = neg;
                ulo = vhi + uhi - vlo;
            } else {
                vlo += ulo / (k - k0 + 1);
                do output[k0++] = vlo; while (k0 <= k);
                return;
            }
            continue;
        }
        ulo += input[k + 1] - vlo;
        if (ulo < neg) {                                     /* negative jump (:100-103) */
            do output[k0++] = vlo; while (k0 <= klo);
            khi = klo = k = k0;
            vlo = input[k0];
            vhi = vlo + two;
            ulo = lambda; uhi = neg;
            continue;
        }
        uhi += input[k + 1] - vhi;
        if (uhi > lambda) {                                  /* positive jump (:104-107) */
            do output[k0++] = vhi; while (k0 <= khi);
            khi = klo = k = k0;
            vhi = input[k0];
            vlo = vhi - two;
            ulo = lambda; uhi = neg;
            continue;
        }
        k++;                                                 /* no jump (:108-118) */
        if (ulo >= lambda) {
            klo = k;
            vlo += (ulo - lambda) / (klo - k0 + 1);
            ulo = lambda;
        }
        if (uhi <= neg) {
            khi = k;
            vhi += (uhi + lambda) / (khi - k0 + 1);
            uhi = neg;
        }
    }
}

/* ------------------------------------------------------------------------- */
/*  TV-L2 prox  min_x 1/2 ||x - y||^2 + lambda ||Dx||_2 , solved to convergence */
/*  Same method as the reference's morePG_TV2 (src/TVL2opt.cpp:190-445): the    */
/*  dual  min_u 1/2 ||D'u - y||^2, ||u||_2 <= lambda  is a trust-region problem  */
/*  on the tridiagonal T = DD' = tridiag(-1, 2, -1); either u = T^-1 Dy is       */
/*  feasible, or u = (T + mu I)^-1 Dy with ||u|| = lambda, and mu is found by    */
/*  More-Sorensen's Newton iteration on 1/||u(mu)|| (:340-385: factor T + mu I,  */
/*  solve for p, solve the Cholesky system for q, mu += (|p|^2/|q|^2)(|p| -      */
/*  lambda)/lambda).  What is NOT restated: the reference's projected-gradient   */
/*  prelude (:250-307), its stopping rule (duality gap 1e-5: its x is ~1e-3      */
/*  from the minimiser) and its warm start from the previous fibre of the same   */
/*  thread -- this solver starts every fibre at mu = 0 and iterates until        */
/*  | ||u|| - lambda | <= 1e-14 lambda (or mu stops increasing), so its result    */
/*  depends on (y, lambda) only.  tests/ check it against the compiled           */
/*  reference within the reference's own guarantee (||dx||_2 <= sqrt(2 STOP_MS)). */
/*  info: iterations of the mu search, final | ||u|| - lambda |, RC_OK.           */
/* ------------------------------------------------------------------------- */
static void tri_solve(int nn, double a, const double *rhs, double *d, double *z, double *u)
{
    /* (tridiag(-1, a, -1)) u = rhs  by LDL' (what dpttrf_/dpttrs_ do at :352-357) */
    d[0] = a;
    z[0] = rhs[0];
    for (int i = 1; i < nn; i++) {
        d[i] = a - 1.0 / d[i - 1];
        z[i] = rhs[i] + z[i - 1] / d[i - 1];
    }
    u[nn - 1] = z[nn - 1] / d[nn - 1];
    for (int i = nn - 2; i >= 0; i--) u[i] = (z[i] + u[i + 1]) / d[i];
}

int orc_TV2_exact(const double *y, double lambda, double *x, double *info, int n)
{
    const int nn = n - 1;
    int iters = 0;
    double dist = 0;
    if (n <= 0) return 1;
    if (nn == 0 || !(lambda > 0)) {
        memcpy(x, y, sizeof(double) * (size_t)n);
    } else {
        double *b = (double *)malloc(sizeof(double) * (size_t)nn * 5);
        if (!b) { if (info) info[ORC_INFO_RC] = ORC_RC_ERROR; return 0; }
        double *d = b + nn, *z = d + nn, *u = z + nn, *v = u + nn;
        for (int i = 0; i < nn; i++) b[i] = y[i + 1] - y[i];
        double mu = 0, nu2 = 0;
        tri_solve(nn, 2.0 + mu, b, d, z, u);
        for (int i = 0; i < nn; i++) nu2 += u[i] * u[i];
        double nu = sqrt(nu2);
        if (nu > lambda) {
            while (iters < 200) {
                tri_solve(nn, 2.0 + mu, u, d, z, v);
                double q2 = 0;
                for (int i = 0; i < nn; i++) q2 += u[i] * v[i];
                const double next = mu + (nu2 / q2) * (nu - lambda) / lambda;
                iters++;
                if (!(next > mu)) break;
                mu = next;
                tri_solve(nn, 2.0 + mu, b, d, z, u);
                nu2 = 0;
                for (int i = 0; i < nn; i++) nu2 += u[i] * u[i];
                nu = sqrt(nu2);
                if (fabs(nu - lambda) <= 1e-14 * lambda) break;
            }
            dist = fabs(nu - lambda);
        }
        /* x = y + D'u  (DUAL2PRIMAL, src/TVmacros.h:10-14) */
        x[0] = y[0] + u[0];
        for (int i = 1; i < nn; i++) x[i] = y[i] - u[i - 1] + u[i];
        x[nn] = y[nn] - u[nn - 1];
        free(b);
    }
    if (info) {
        info[ORC_INFO_ITERS] = iters;
        info[ORC_INFO_GAP] = dist;
        info[ORC_INFO_RC] = ORC_RC_OK;
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/*  TV() dispatcher (src/TVgenopt.cpp:30-57): p == 1 -> hybrid taut string    */
/*  (bit-identical to the reference), p == 2 -> the exact TV-L2 prox above.    */
/* ------------------------------------------------------------------------- */
int orc_TV(const double *y, double lambda, double *x, double *info, int n, double p)
{
    if (p == 2) return orc_TV2_exact(y, lambda, x, info, n);
    if (p != 1) {
        if (info) info[ORC_INFO_RC] = ORC_RC_ERROR;
        return 0;
    }
    orc_hybridTautString_TV1(y, n, lambda, x);
    if (info) {
        info[ORC_INFO_RC] = ORC_RC_OK;
        info[ORC_INFO_ITERS] = 0;
        info[ORC_INFO_GAP] = 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/*  Johnson's dynamic programme for the 1-D fused lasso (tv1_1d method 'dp')  */
/*  src/johnsonRyanTV.cpp:9-116                                               */
/*  Forward pass: the derivative of the k-th "message" (a convex piecewise    */
/*  quadratic) is piecewise linear; it is kept as a deque of knots with the   */
/*  increments (da, db) of slope and intercept that become active to the right */
/*  (left end) / left (right end) of each knot.  Clipping it to [-lambda,      */
/*  lambda] yields the two back-pointer knots tm[k] <= tp[k].  Backward pass:  */
/*  x[k] = clamp(x[k+1], tm[k], tp[k]).  Same arithmetic order as the          */
/*  reference, so results are bit-identical.                                   */
/* ------------------------------------------------------------------------- */
void orc_dp(int n, const double *y, double lam, double *beta)
{
    if (n == 0) return;
    if (n == 1 || lam == 0) {                                   /* :12-15 */
        for (int i = 0; i < n; i++) beta[i] = y[i];
        return;
    }
    /* deque storage: positions n-1 and n are the first two knots, it grows outwards by one per side and step */
    double *knot = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *da = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *db = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tm = (double *)malloc(sizeof(double) * (size_t)(n - 1));
    double *tp = (double *)malloc(sizeof(double) * (size_t)(n - 1));
    int head = n - 1, tail = n;
    tm[0] = -lam + y[0];                                        /* first message, by hand (:30-43) */
    tp[0] = lam + y[0];
    knot[head] = tm[0]; da[head] = 1;  db[head] = -y[0] + lam;
    knot[tail] = tp[0]; da[tail] = -1; db[tail] = y[0] + lam;
    double a_left = 1, b_left = -lam - y[1];                    /* derivative left of every knot ... */
    double a_right = -1, b_right = -lam + y[1];                 /* ... and (negated) right of every knot */

    for (int k = 1; k < n - 1; k++) {                           /* :48-87 */
        /* walk in from the left until the derivative exceeds -lambda */
        double a = a_left, b = b_left;
        int pos;
        for (pos = head; pos <= tail; pos++) {
            if (a * knot[pos] + b > -lam) break;
            a += da[pos];
            b += db[pos];
        }
        tm[k] = (-lam - b) / a;
        head = pos - 1;
        knot[head] = tm[k];
        /* walk in from the right until it drops below lambda */
        double ar = a_right, br = b_right;
        for (pos = tail; pos >= head; pos--) {
            if (-ar * knot[pos] - br < lam) break;
            ar += da[pos];
            br += db[pos];
        }
        tp[k] = (lam + br) / (-ar);
        tail = pos + 1;
        knot[tail] = tp[k];
        da[head] = a;  db[head] = b + lam;
        da[tail] = ar; db[tail] = br + lam;
        a_left = 1;   b_left = -lam - y[k + 1];
        a_right = -1; b_right = -lam + y[k + 1];
    }
    /* last coefficient: the zero of the derivative (:92-99) */
    {
        double a = a_left, b = b_left;
        for (int pos = head; pos <= tail; pos++) {
            if (a * knot[pos] + b > 0) break;
            a += da[pos];
            b += db[pos];
        }
        beta[n - 1] = -b / a;
    }
    for (int k = n - 2; k >= 0; k--) {                          /* back-pointers (:103-107) */
        if (beta[k + 1] > tp[k]) beta[k] = tp[k];
        else if (beta[k + 1] < tm[k]) beta[k] = tm[k];
        else beta[k] = beta[k + 1];
    }
    free(knot); free(da); free(db); free(tm); free(tp);
}

