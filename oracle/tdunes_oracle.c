/*
 * tdunes_oracle.c -- CPU restatement of the tdunes hot path.  TEST INFRASTRUCTURE ONLY
 * (see tdunes_oracle.h for the rules and the pinning statement).
 *
 * Reference files restated (paths relative to the reference repo root):
 *   treeqp/utils/tree.c, treeqp/src/dual_Newton_tree.c, treeqp/src/dual_Newton_tree_clipping.c,
 *   treeqp/src/dual_Newton_common.c, treeqp/src/tree_qp_common.c (LTI filler + KKT residual).
 * BLAS-level semantics restated from the public BLASFEO reference API (column major here):
 *   dgemv_n/t  z = beta*y + alpha*op(A)*x          dsyrk_ln  D = beta*C + alpha*A*B' (lower)
 *   dgemm_nt   D = beta*C + alpha*A*B'             dgemm_nd  D = beta*C + alpha*A*diag(b)
 *   dpotrf_l   lower Cholesky, pivot <= 0 -> column of zeros (inverse diagonal set to 0)
 *   dtrsv_lnn / dtrsv_ltn  triangular solves through the reciprocal diagonal kept by dpotrf_l
 *              (lnn: row-oriented, k ascending; ltn: column-oriented, k descending)
 *   dtrsm_rltn D = alpha*B*A^-T                    dveccl_mask inclusive clip with +-1 mask
 * Summation order everywhere: k ascending, accumulate from 0, then apply alpha/beta.
 */
#include "tdunes_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OMAX(a, b) ((a) > (b) ? (a) : (b))

/* ======================================================================================= */
/* integer tree logic                                                                      */
/* ======================================================================================= */

int oracle_ipow(int base, int exp) {           /* utils.c:34-47 (square-and-multiply) */
    int result = 1;
    while (exp) {
        if (exp & 1) result *= base;
        exp >>= 1;
        base *= base;
    }
    return result;
}

int oracle_calculate_number_of_nodes(int md, int Nr, int Nh) {   /* tree.c:36-48 */
    if (md == 1) return Nh + 1;
    return (Nh - Nr) * oracle_ipow(md, Nr) + (oracle_ipow(md, Nr + 1) - 1) / (md - 1);
}

int oracle_number_of_nodes_from_nkids(const int *nk) {           /* tree.c:105-126 */
    int indx = 0, in_stage = 1;
    for (;;) {
        int in_next = 0;
        for (int ii = 0; ii < in_stage; ii++) {
            if (nk[indx + ii] < 0) return -1;
            if (nk[indx + ii] == 0) break;      /* reached a leaf: uniform leaf depth */
            in_next += nk[indx + ii];
        }
        indx += in_stage;
        if (in_next == 0) break;
        if (in_next < in_stage) return -1;
        in_stage = in_next;
    }
    return indx;
}

void oracle_setup_multistage_tree(int md, int Nr, int Nh, int *nk) {  /* tree.c:247-280 */
    int in_stage = 1, idx = 0;
    for (int kk = 0; kk < Nh; kk++) {
        int in_next = 0;
        for (int ii = 0; ii < in_stage; ii++) {
            nk[idx + ii] = (kk < Nr) ? md : 1;
            in_next += nk[idx + ii];
        }
        idx += in_stage;
        in_stage = in_next;
    }
    for (int ii = 0; ii < in_stage; ii++) nk[idx + ii] = 0;
}

int oracle_tree_create(int Nn, const int *nk, int *dad, int *stage, int *real, int *idxkid,
                       int *kid0) {            /* tree.c:171-243 */
    for (int ii = 0; ii < Nn; ii++) { stage[ii] = -1; real[ii] = -1; kid0[ii] = -1; }
    dad[0] = -1; stage[0] = 0; idxkid[0] = 0;
    int Np = 0;
    int next_free = 1;                          /* children are the next unassigned indices */
    for (int ii = 0; ii < Nn; ii++) {
        if (nk[ii] > 0) { Np++; kid0[ii] = next_free; }
        int realization = 0;
        for (int jj = next_free; jj < next_free + nk[ii]; jj++) {
            dad[jj] = ii;
            stage[jj] = stage[ii] + 1;
            idxkid[jj] = jj - next_free;
            if (nk[ii] > 1) real[jj] = realization++;     /* tree.c:224-227 */
            else real[jj] = (ii > 0) ? real[ii] : 0;      /* tree.c:228-238 */
        }
        next_free += nk[ii];
    }
    return Np;                                  /* tree.c:52-61 */
}

void oracle_setup_idxpos(int Nn, const int *dad, const int *idxkid, const int *kid0,
                         const int *nx, int *idxpos) {    /* dual_Newton_tree.c:177-194 */
    for (int kk = 0; kk < Nn; kk++) {
        idxpos[kk] = 0;
        if (kk == 0) continue;
        int first = kid0[dad[kk]];
        for (int ii = 0; ii < idxkid[kk]; ii++) idxpos[kk] += nx[first + ii];
    }
}

void oracle_setup_npar(int Nn, const int *stage, int Nh, int *npar) {  /* :166-173 */
    for (int kk = 0; kk <= Nh; kk++) npar[kk] = 0;
    for (int kk = 0; kk < Nn; kk++) npar[stage[kk]]++;
}

void oracle_opts_set_default(oracle_opts_t *o) {          /* dual_Newton_tree.c:92-120 */
    o->maxIter = 100;
    o->termCondition = ORC_INFNORM;
    o->stationarityTolerance = 1.0e-8;
    o->checkLastActiveSet = 1;
    o->lineSearchMaxIter = 50;
    o->lineSearchGamma = 0.1;
    o->lineSearchBeta = 0.6;
    o->lineSearchRestartTrigger = -1;
    o->regType = ORC_ON_THE_FLY_LM;
    o->regTol = 1.0e-6;
    o->regValue = 1.0e-6;
    o->num_threads = 1;
}

/* ======================================================================================= */
/* LTI filler (tree_qp_common.c:1837-1949)                                                 */
/* ======================================================================================= */

void oracle_fill_lti_diag(int Nn, const int *nk, int nx, int nu,
                          const double *A, const double *B, const double *b,
                          const double *Qd, const double *q, const double *Pd, const double *p,
                          const double *Rd, const double *r,
                          const double *xmin, const double *xmax,
                          const double *umin, const double *umax, const double *x0,
                          double *oA, double *oB, double *ob,
                          double *oQd, double *oRd, double *oq, double *orr,
                          double *oxmin, double *oxmax, double *oumin, double *oumax) {
    int *dad = malloc(Nn * sizeof(int)), *stage = malloc(Nn * sizeof(int));
    int *real = malloc(Nn * sizeof(int)), *idxkid = malloc(Nn * sizeof(int));
    int *kid0 = malloc(Nn * sizeof(int));
    oracle_tree_create(Nn, nk, dad, stage, real, idxkid, kid0);

    int numberOfLeaves = 1;                                  /* :1872-1883 */
    for (int ii = Nn - 1; ii > 0; ii--) {
        if (stage[ii] == stage[ii - 1]) numberOfLeaves++; else break;
    }
    int *xo = malloc((Nn + 1) * sizeof(int)), *uo = malloc((Nn + 1) * sizeof(int));
    xo[0] = uo[0] = 0;
    for (int ii = 0; ii < Nn; ii++) {
        xo[ii + 1] = xo[ii] + nx;
        uo[ii + 1] = uo[ii] + (nk[ii] > 0 ? nu : 0);
    }
    int currentStage = 0, nodesInStage = 0;
    for (int ii = 0; ii < Nn; ii++) {
        int nui = nk[ii] > 0 ? nu : 0;
        if (ii > 0) {                                        /* :1891-1898 */
            int re = real[ii];
            memcpy(oA + (size_t)(ii - 1) * nx * nx, A + (size_t)re * nx * nx, sizeof(double) * nx * nx);
            memcpy(oB + (size_t)(ii - 1) * nx * nu, B + (size_t)re * nx * nu, sizeof(double) * nx * nu);
            memcpy(ob + (size_t)(ii - 1) * nx, b + (size_t)re * nx, sizeof(double) * nx);
        }
        for (int j = 0; j < nx; j++) {                       /* :1900-1907 */
            oQd[xo[ii] + j] = nk[ii] > 0 ? Qd[j] : Pd[j];
            oq[xo[ii] + j] = nk[ii] > 0 ? q[j] : p[j];
        }
        for (int j = 0; j < nui; j++) { oRd[uo[ii] + j] = Rd[j]; orr[uo[ii] + j] = r[j]; }

        if (stage[ii] > currentStage) {                      /* :1909-1928 */
            double scalingFactor = numberOfLeaves / nodesInStage;   /* integer division! :1911 */
            for (int jj = 1; jj <= nodesInStage; jj++) {
                int n = ii - jj;
                for (int j = xo[n]; j < xo[n + 1]; j++) { oQd[j] *= scalingFactor; oq[j] *= scalingFactor; }
                for (int j = uo[n]; j < uo[n + 1]; j++) { oRd[j] *= scalingFactor; orr[j] *= scalingFactor; }
            }
            currentStage = stage[ii];
            nodesInStage = 1;
        } else {
            nodesInStage++;
        }
        for (int j = 0; j < nx; j++) {                       /* :1930-1937 */
            oxmin[xo[ii] + j] = (ii == 0) ? x0[j] : xmin[j];
            oxmax[xo[ii] + j] = (ii == 0) ? x0[j] : xmax[j];
        }
        for (int j = 0; j < nui; j++) { oumin[uo[ii] + j] = umin[j]; oumax[uo[ii] + j] = umax[j]; }
    }
    free(dad); free(stage); free(real); free(idxkid); free(kid0); free(xo); free(uo);
}

/* ======================================================================================= */
/* workspace                                                                               */
/* ======================================================================================= */

typedef struct {
    int Nn, Np, Nh;
    const int *nk, *nx, *nu;
    int *dad, *stage, *real, *idxkid, *kid0, *pos, *npar;
    int *xoff, *uoff, *zoff, *aoff, *boff;
    int *bdim, *woff, *utoff, *wdoff, *poff;
    const double *A, *B, *b, *Qd, *Rd, *q, *r, *xmin, *xmax, *umin, *umax;
    int dense;
    const double *Q, *R, *S;
    double *P;                 /* dense: per node (nx+nu)^2 elimination matrix H^-1            */
    double *Hc;                /* dense: Cholesky factor of H                                   */
    double *Qinv, *Rinv, *QinvCal, *RinvCal;
    double *qmod, *rmod, *x, *u, *xUnc, *uUnc, *xas, *uas, *xasPrev, *uasPrev;
    double *lam, *dlam, *res, *resMod;     /* node indexed at xoff[k]; root slot unused */
    double *W, *CholW, *Ut, *CholUt, *Wdiag;
    double *invd;              /* reciprocal Cholesky diagonals, block p at xoff[kid0[p]] */
    double *Hinv;              /* dense: reciprocal diagonal of Hc, node k at zoff[k]          */
    double *fval, *cmod;
    int *xasChanged, *uasChanged, *blockChanged;
    int maxM;                  /* scratch M size per node */
    int lineSearchRestartCounter, lsIter;
    int n_regularized;
} ws_t;

static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "[oracle] out of memory\n"); exit(1); }
    return p;
}

static void ws_setup(ws_t *w, int Nn, const int *nk, const int *nx, const int *nu) {
    memset(w, 0, sizeof(*w));
    w->Nn = Nn; w->nk = nk; w->nx = nx; w->nu = nu;
    w->dad = xcalloc(Nn, sizeof(int)); w->stage = xcalloc(Nn, sizeof(int));
    w->real = xcalloc(Nn, sizeof(int)); w->idxkid = xcalloc(Nn, sizeof(int));
    w->kid0 = xcalloc(Nn, sizeof(int)); w->pos = xcalloc(Nn, sizeof(int));
    w->Np = oracle_tree_create(Nn, nk, w->dad, w->stage, w->real, w->idxkid, w->kid0);
    w->Nh = w->stage[Nn - 1];                               /* dual_Newton_tree.c:647 */
    w->npar = xcalloc(w->Nh + 1, sizeof(int));
    oracle_setup_npar(Nn, w->stage, w->Nh, w->npar);
    oracle_setup_idxpos(Nn, w->dad, w->idxkid, w->kid0, nx, w->pos);
    /* the reference assumes parents are exactly nodes 0..Np-1 (uniform leaf depth) */
    for (int k = 0; k < Nn; k++) {
        if ((nk[k] > 0) != (k < w->Np)) { fprintf(stderr, "[oracle] non-uniform leaf depth\n"); exit(1); }
    }
    w->xoff = xcalloc(Nn + 1, sizeof(int)); w->uoff = xcalloc(Nn + 1, sizeof(int));
    w->zoff = xcalloc(Nn + 1, sizeof(int));
    w->aoff = xcalloc(Nn + 1, sizeof(int)); w->boff = xcalloc(Nn + 1, sizeof(int));
    w->bdim = xcalloc(Nn, sizeof(int)); w->woff = xcalloc(Nn + 1, sizeof(int));
    w->utoff = xcalloc(Nn + 1, sizeof(int)); w->wdoff = xcalloc(Nn + 1, sizeof(int));
    w->poff = xcalloc(Nn + 1, sizeof(int));
    for (int k = 0; k < Nn; k++) {
        w->xoff[k + 1] = w->xoff[k] + nx[k];
        w->uoff[k + 1] = w->uoff[k] + nu[k];
        w->zoff[k + 1] = w->zoff[k] + nx[k] + nu[k];
        w->wdoff[k + 1] = w->wdoff[k] + nx[k] * nx[k];
        w->poff[k + 1] = w->poff[k] + (nx[k] + nu[k]) * (nx[k] + nu[k]);
        if (k > 0) {
            w->aoff[k + 1] = w->aoff[k] + nx[k] * nx[w->dad[k]];
            w->boff[k + 1] = w->boff[k] + nx[k] * nu[w->dad[k]];
        } else { w->aoff[1] = 0; w->boff[1] = 0; }
        int d = 0;
        for (int j = 0; j < nk[k]; j++) d += nx[w->kid0[k] + j];
        w->bdim[k] = d;
        w->woff[k + 1] = w->woff[k] + d * d;
        w->utoff[k + 1] = w->utoff[k] + (k > 0 ? nx[k] * d : 0);
        int mrows = 0;
        if (k > 0) {
            int p = w->dad[k];
            for (int j = 0; j < nk[p]; j++) mrows = OMAX(mrows, nx[w->kid0[p] + j]);
            w->maxM = OMAX(w->maxM, mrows * (nx[p] + nu[p]));
        }
    }
    int sx = w->xoff[Nn], su = w->uoff[Nn];
    w->Qinv = xcalloc(sx, 8); w->QinvCal = xcalloc(sx, 8); w->qmod = xcalloc(sx, 8);
    w->x = xcalloc(sx, 8); w->xUnc = xcalloc(sx, 8); w->xas = xcalloc(sx, 8); w->xasPrev = xcalloc(sx, 8);
    w->Rinv = xcalloc(su, 8); w->RinvCal = xcalloc(su, 8); w->rmod = xcalloc(su, 8);
    w->u = xcalloc(su, 8); w->uUnc = xcalloc(su, 8); w->uas = xcalloc(su, 8); w->uasPrev = xcalloc(su, 8);
    w->lam = xcalloc(sx, 8); w->dlam = xcalloc(sx, 8); w->res = xcalloc(sx, 8); w->resMod = xcalloc(sx, 8);
    w->W = xcalloc(w->woff[Nn], 8); w->CholW = xcalloc(w->woff[Nn], 8);
    w->Ut = xcalloc(w->utoff[Nn], 8); w->CholUt = xcalloc(w->utoff[Nn], 8);
    w->Wdiag = xcalloc(w->wdoff[Nn], 8);
    w->invd = xcalloc(sx, 8);
    w->fval = xcalloc(Nn, 8); w->cmod = xcalloc(Nn, 8);
    w->xasChanged = xcalloc(Nn, sizeof(int)); w->uasChanged = xcalloc(Nn, sizeof(int));
    w->blockChanged = xcalloc(Nn, sizeof(int));
}

static void ws_free(ws_t *w) {
    free(w->dad); free(w->stage); free(w->real); free(w->idxkid); free(w->kid0); free(w->pos);
    free(w->npar); free(w->xoff); free(w->uoff); free(w->zoff); free(w->aoff); free(w->boff);
    free(w->bdim); free(w->woff); free(w->utoff); free(w->wdoff); free(w->poff);
    free(w->Qinv); free(w->QinvCal); free(w->qmod); free(w->x); free(w->xUnc); free(w->xas);
    free(w->xasPrev); free(w->Rinv); free(w->RinvCal); free(w->rmod); free(w->u); free(w->uUnc);
    free(w->uas); free(w->uasPrev); free(w->lam); free(w->dlam); free(w->res); free(w->resMod);
    free(w->W); free(w->CholW); free(w->Ut); free(w->CholUt); free(w->Wdiag); free(w->fval);
    free(w->cmod); free(w->xasChanged); free(w->uasChanged); free(w->blockChanged);
    free(w->P); free(w->Hc); free(w->invd); free(w->Hinv);
}

/* ======================================================================================= */
/* small dense kernels (column major, explicit leading dimension)                          */
/* ======================================================================================= */

/* z += alpha * A' * x,  A is m x n (ld), x length m, z length n   (blasfeo_dgemv_t, beta=1) */
static void gemv_t_acc(int m, int n, double alpha, const double *A, int ld, const double *x, double *z) {
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < m; i++) acc += A[i + j * ld] * x[i];
        z[j] += alpha * acc;
    }
}
/* z += alpha * A * x,  A is m x n (ld), x length n, z length m   (blasfeo_dgemv_n, beta=1) */
static void gemv_n_acc(int m, int n, double alpha, const double *A, int ld, const double *x, double *z) {
    for (int i = 0; i < m; i++) {
        double acc = 0.0;
        for (int j = 0; j < n; j++) acc += A[i + j * ld] * x[j];
        z[i] += alpha * acc;
    }
}
static double dot(int n, const double *a, const double *b) {
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc += a[i] * b[i];
    return acc;
}

/* blasfeo_dpotrf_l semantics: L lower, pivot <= 0 -> zero column.  D may alias C.  Like BLASFEO's
 * reference implementation the reciprocal diagonal computed during the factorization is kept
 * (`inv`, BLASFEO's dA) and re-used by the triangular solves on the same factor. */
static void potrf_l(int n, const double *C, int ldc, double *D, int ldd, double *inv) {
    for (int j = 0; j < n; j++) {
        double c = C[j + j * ldc];
        for (int k = 0; k < j; k++) c -= D[j + k * ldd] * D[j + k * ldd];
        double finv = (c > 0.0) ? 1.0 / sqrt(c) : 0.0;
        inv[j] = finv;
        D[j + j * ldd] = c * finv;
        for (int i = j + 1; i < n; i++) {
            double s = C[i + j * ldc];
            for (int k = 0; k < j; k++) s -= D[i + k * ldd] * D[j + k * ldd];
            D[i + j * ldd] = s * finv;
        }
    }
}
/* z = L^-1 x  (blasfeo_dtrsv_lnn), k ascending ; z may alias x */
static void trsv_lnn(int n, const double *L, int ld, const double *inv, const double *x, double *z) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[i + k * ld] * z[k];
        z[i] = s * inv[i];
    }
}
/* z = L^-T x  (blasfeo_dtrsv_ltn), column-oriented: k descending ; z may alias x */
static void trsv_ltn(int n, const double *L, int ld, const double *inv, const double *x, double *z) {
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = n - 1; k > i; k--) s -= L[k + i * ld] * z[k];
        z[i] = s * inv[i];
    }
}
/* D = B * L^-T, B is m x n, L is n x n lower  (blasfeo_dtrsm_rltn, alpha = 1) */
static void trsm_rltn(int m, int n, const double *L, int ldl, const double *inv, const double *B, int ldb, double *D, int ldd) {
    for (int j = 0; j < n; j++) {
        for (int i = 0; i < m; i++) {
            double s = B[i + j * ldb];
            for (int k = 0; k < j; k++) s -= D[i + k * ldd] * L[j + k * ldl];
            D[i + j * ldd] = s * inv[j];
        }
    }
}

/* ======================================================================================= */
/* dense stage data (extension used only to pin phases G/H/F on the reference's goldens)   */
/* ======================================================================================= */

static void dense_init(ws_t *w) {
    int Nn = w->Nn;
    w->P = xcalloc(w->poff[Nn], 8);
    w->Hc = xcalloc(w->poff[Nn], 8);
    w->Hinv = xcalloc(w->zoff[Nn], 8);
    for (int k = 0; k < Nn; k++) {
        int nx = w->nx[k], nu = w->nu[k], nz = nx + nu;
        double *H = w->Hc + w->poff[k], *P = w->P + w->poff[k];
        const double *Q = w->Q + w->wdoff[k];
        int roff = 0, soff = 0;
        for (int j = 0; j < k; j++) { roff += w->nu[j] * w->nu[j]; soff += w->nu[j] * w->nx[j]; }
        const double *R = w->R + roff, *S = w->S + soff;
        for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) H[i + j * nz] = Q[i + j * nx];
        for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) H[nx + i + (nx + j) * nz] = R[i + j * nu];
        for (int j = 0; j < nx; j++) for (int i = 0; i < nu; i++) {      /* S is nu x nx */
            H[nx + i + j * nz] = S[i + j * nu];
            H[j + (nx + i) * nz] = S[i + j * nu];
        }
        double *hi = w->Hinv + w->zoff[k];
        potrf_l(nz, H, nz, H, nz, hi);
        /* P = H^-1 = L^-T L^-1 : solve column by column */
        double *e = xcalloc(nz, 8);
        for (int j = 0; j < nz; j++) {
            memset(e, 0, nz * 8); e[j] = 1.0;
            trsv_lnn(nz, H, nz, hi, e, e);
            trsv_ltn(nz, H, nz, hi, e, e);
            for (int i = 0; i < nz; i++) P[i + j * nz] = e[i];
        }
        free(e);
    }
}

/* ======================================================================================= */
/* Phase S / L-eval: solve_stage_problems (dual_Newton_tree.c:218-330) and                 */
/* evaluate_dual_function (:823-918) share one sweep                                       */
/* ======================================================================================= */

static void stage_sweep(ws_t *w, int extended, int eval, int nthreads) {
    const int Nn = w->Nn;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1)
#endif
    for (int kk = 0; kk < Nn; kk++) {
        const int nx = w->nx[kk], nu = w->nu[kk];
        double *qmod = w->qmod + w->xoff[kk], *rmod = w->rmod + w->uoff[kk];
        const double *q = w->q + w->xoff[kk], *r = w->r + w->uoff[kk];
        /* qmod = -q + lambda_k  (:266-275); lambda_0 = 0 */
        for (int j = 0; j < nx; j++) {
            double l = (kk == 0) ? 0.0 : w->lam[w->xoff[kk] + j];
            qmod[j] = l + (-1.0) * q[j];
        }
        for (int j = 0; j < nu; j++) rmod[j] = -1.0 * r[j];            /* :278 */
        double cmod = 0.0;
        for (int ii = 0; ii < w->nk[kk]; ii++) {                      /* :280-292 */
            int kid = w->kid0[kk] + ii;
            const double *lk = w->lam + w->xoff[kid];
            if (eval) cmod += dot(w->nx[kid], w->b + (w->xoff[kid] - w->nx[0]), lk);   /* :892 */
            gemv_t_acc(w->nx[kid], nx, -1.0, w->A + w->aoff[kid], w->nx[kid], lk, qmod);
            gemv_t_acc(w->nx[kid], nu, -1.0, w->B + w->boff[kid], w->nx[kid], lk, rmod);
        }
        double *x = w->x + w->xoff[kk], *u = w->u + w->uoff[kk];
        if (w->dense) {
            /* z = H^-1 [qmod; rmod] */
            int nz = nx + nu;
            double *z = malloc(sizeof(double) * (nz ? nz : 1));
            for (int j = 0; j < nx; j++) z[j] = qmod[j];
            for (int j = 0; j < nu; j++) z[nx + j] = rmod[j];
            trsv_lnn(nz, w->Hc + w->poff[kk], nz, w->Hinv + w->zoff[kk], z, z);
            trsv_ltn(nz, w->Hc + w->poff[kk], nz, w->Hinv + w->zoff[kk], z, z);
            for (int j = 0; j < nx; j++) x[j] = z[j];
            for (int j = 0; j < nu; j++) u[j] = z[nx + j];
            free(z);
            if (extended) {
                for (int j = 0; j < nx; j++) w->xas[w->xoff[kk] + j] = 0.0;
                for (int j = 0; j < nu; j++) w->uas[w->uoff[kk] + j] = 0.0;
            }
        } else if (extended) {
            /* clipping solve_extended (dual_Newton_tree_clipping.c:188-227) */
            const double *Qinv = w->Qinv + w->xoff[kk], *Rinv = w->Rinv + w->uoff[kk];
            double *xUnc = w->xUnc + w->xoff[kk], *uUnc = w->uUnc + w->uoff[kk];
            double *xas = w->xas + w->xoff[kk], *uas = w->uas + w->uoff[kk];
            for (int j = 0; j < nx; j++) {
                xUnc[j] = Qinv[j] * qmod[j];                          /* dvecmuldot :209 */
                double lb = w->xmin[w->xoff[kk] + j], ub = w->xmax[w->xoff[kk] + j];
                if (xUnc[j] >= ub) { x[j] = ub; xas[j] = 1.0; }       /* dveccl_mask :212 */
                else if (xUnc[j] <= lb) { x[j] = lb; xas[j] = -1.0; }
                else { x[j] = xUnc[j]; xas[j] = 0.0; }
                w->QinvCal[w->xoff[kk] + j] = (xas[j] == 0.0) ? Qinv[j] : 0.0;   /* dvecze :221 */
            }
            for (int j = 0; j < nu; j++) {
                uUnc[j] = Rinv[j] * rmod[j];                          /* :215 */
                double lb = w->umin[w->uoff[kk] + j], ub = w->umax[w->uoff[kk] + j];
                if (uUnc[j] >= ub) { u[j] = ub; uas[j] = 1.0; }       /* :218 */
                else if (uUnc[j] <= lb) { u[j] = lb; uas[j] = -1.0; }
                else { u[j] = uUnc[j]; uas[j] = 0.0; }
                w->RinvCal[w->uoff[kk] + j] = (uas[j] == 0.0) ? Rinv[j] : 0.0;   /* :224 */
            }
        } else {
            /* clipping solve (dual_Newton_tree_clipping.c:231-260): dveccl is inclusive too */
            const double *Qinv = w->Qinv + w->xoff[kk], *Rinv = w->Rinv + w->uoff[kk];
            for (int j = 0; j < nx; j++) {
                double v = Qinv[j] * qmod[j];
                double lb = w->xmin[w->xoff[kk] + j], ub = w->xmax[w->xoff[kk] + j];
                x[j] = (v >= ub) ? ub : ((v <= lb) ? lb : v);
            }
            for (int j = 0; j < nu; j++) {
                double v = Rinv[j] * rmod[j];
                double lb = w->umin[w->uoff[kk] + j], ub = w->umax[w->uoff[kk] + j];
                u[j] = (v >= ub) ? ub : ((v <= lb) ? lb : v);
            }
        }
        if (eval) {
            /* eval_dual_term (dual_Newton_tree_clipping.c:359-382; xas/uas used as scratch) */
            w->cmod[kk] = cmod;
            double f;
            if (w->dense) {
                int nz = nx + nu;
                /* f = -1/2 z'Hz + hmod'z - cmod with the full H (incl. S cross term) */
                const double *Q = w->Q + w->wdoff[kk];
                int roff = 0, soff = 0;
                for (int j = 0; j < kk; j++) { roff += w->nu[j] * w->nu[j]; soff += w->nu[j] * w->nx[j]; }
                const double *R = w->R + roff, *S = w->S + soff;
                double quad = 0.0;
                for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) quad += x[i] * Q[i + j * nx] * x[j];
                for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) quad += u[i] * R[i + j * nu] * u[j];
                for (int j = 0; j < nx; j++) for (int i = 0; i < nu; i++) quad += 2.0 * u[i] * S[i + j * nu] * x[j];
                (void)nz;
                f = -0.5 * quad - cmod + dot(nx, qmod, x) + dot(nu, rmod, u);
            } else {
                double *xas = w->xas + w->xoff[kk], *uas = w->uas + w->uoff[kk];
                for (int j = 0; j < nx; j++) xas[j] = w->Qd[w->xoff[kk] + j] * x[j];   /* :374 */
                f = -0.5 * dot(nx, xas, x) - cmod;                                     /* :375 */
                f += dot(nx, qmod, x);                                                 /* :376 */
                for (int j = 0; j < nu; j++) uas[j] = w->Rd[w->uoff[kk] + j] * u[j];   /* :379 */
                f -= 0.5 * dot(nu, uas, u);                                            /* :380 */
                f += dot(nu, rmod, u);                                                 /* :381 */
            }
            w->fval[kk] = f;
        }
    }
}

/* ======================================================================================= */
/* Phase G+H: build_dual_problem (dual_Newton_tree.c:446-637)                              */
/* ======================================================================================= */

static void compare_with_previous_active_set(ws_t *w, int isLeaf, int k) {   /* :334-368 */
    const int nx = w->nx[k], nu = w->nu[k];
    double *xas = w->xas + w->xoff[k], *xp = w->xasPrev + w->xoff[k];
    w->xasChanged[k] = 0;
    for (int i = 0; i < nx; i++) if (xas[i] != xp[i]) { w->xasChanged[k] = 1; break; }   /* NaN != x */
    memcpy(xp, xas, sizeof(double) * nx);
    if (!isLeaf) {
        double *uas = w->uas + w->uoff[k], *up = w->uasPrev + w->uoff[k];
        w->uasChanged[k] = 0;
        for (int i = 0; i < nu; i++) if (uas[i] != up[i]) { w->uasChanged[k] = 1; break; }
        memcpy(up, uas, sizeof(double) * nu);
    }
}

static int find_starting_point_of_factorization(ws_t *w) {                   /* :371-405 */
    int Np = w->Np, start = Np;
    for (int k = 0; k < Np; k++) w->blockChanged[k] = 0;
    for (int k = w->Nn - 1; k > 0; k--) {
        int d = w->dad[k];
        int asDadChanged = w->xasChanged[d] | w->uasChanged[d];
        if (asDadChanged || w->xasChanged[k]) w->blockChanged[d] = 1;
    }
    for (int k = Np - 1; k >= 0; k--) {
        if (!w->blockChanged[k]) start--; else break;
    }
    return start;
}

static double calculate_error_in_residuals(const ws_t *w, int cond) {         /* :412-442 */
    double err = 0.0;
    const int n0 = w->xoff[1], n1 = w->xoff[w->Nn];
    if (cond == ORC_SUMSQUAREDERRORS || cond == ORC_TWONORM) {
        /* reference: sum over blocks of ddot(block) */
        for (int p = 0; p < w->Np; p++) {
            const double *r = w->res + w->xoff[w->kid0[p]];
            err += dot(w->bdim[p], r, r);
        }
        if (cond == ORC_TWONORM) err = sqrt(err);
    } else {
        for (int i = n0; i < n1; i++) { double a = fabs(w->res[i]); if (a > err) err = a; }
    }
    return err;
}

/* M = [A_s*diag(Qcal_p) | B_s*diag(Rcal_p)]  (clipping.c:285,292,342,349) or C_s*P_p (dense) */
static void build_M(const ws_t *w, int s, int p, double *M) {
    const int nxs = w->nx[s], nxp = w->nx[p], nup = w->nu[p];
    const double *A = w->A + w->aoff[s], *B = w->B + w->boff[s];
    if (!w->dense) {
        const double *Qc = w->QinvCal + w->xoff[p], *Rc = w->RinvCal + w->uoff[p];
        for (int j = 0; j < nxp; j++) for (int i = 0; i < nxs; i++) M[i + j * nxs] = A[i + j * nxs] * Qc[j];
        for (int j = 0; j < nup; j++) for (int i = 0; i < nxs; i++) M[i + (nxp + j) * nxs] = B[i + j * nxs] * Rc[j];
    } else {
        const int nz = nxp + nup;
        const double *P = w->P + w->poff[p];
        for (int j = 0; j < nz; j++) for (int i = 0; i < nxs; i++) {
            double acc = 0.0;
            for (int k = 0; k < nxp; k++) acc += A[i + k * nxs] * P[k + j * nz];
            for (int k = 0; k < nup; k++) acc += B[i + k * nxs] * P[nxp + k + j * nz];
            M[i + j * nxs] = acc;
        }
    }
}

static int build_dual_problem(ws_t *w, const oracle_opts_t *o, int *idxFactorStart, double *err_out, int nthreads) {
    const int Nn = w->Nn;
    *idxFactorStart = -1;
    (void)nthreads;
    if (o->checkLastActiveSet) {                                                /* :501-512 */
        for (int k = Nn - 1; k >= 0; k--) compare_with_previous_active_set(w, w->nk[k] > 0 ? 0 : 1, k);
        *idxFactorStart = find_starting_point_of_factorization(w);
    }
    /* dual gradient (:519-539) */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1)
#endif
    for (int k = Nn - 1; k > 0; k--) {
        const int p = w->dad[k], nx = w->nx[k];
        double *res = w->res + w->xoff[k];
        const double *b = w->b + (w->xoff[k] - w->nx[0]);
        for (int i = 0; i < nx; i++) res[i] = b[i] + (-1.0) * w->x[w->xoff[k] + i];        /* :527 */
        gemv_n_acc(nx, w->nx[p], 1.0, w->A + w->aoff[k], nx, w->x + w->xoff[p], res);      /* :530 */
        gemv_n_acc(nx, w->nu[p], 1.0, w->B + w->boff[k], nx, w->u + w->uoff[p], res);      /* :534 */
        memcpy(w->resMod + w->xoff[k], res, sizeof(double) * nx);                          /* :538 */
    }
    double err = calculate_error_in_residuals(w, o->termCondition);                        /* :542 */
    *err_out = err;
    if (err < o->stationarityTolerance) return ORC_OPTIMAL;                                /* :543 */

    /* dual Hessian (:551-615) */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1)
#endif
    for (int k = Nn - 1; k > 0; k--) {
        const int p = w->dad[k], pos = w->pos[k], nx = w->nx[k];
        const int nxp = w->nx[p], nup = w->nu[p], d = w->bdim[p];
        double *Wp = w->W + w->woff[p];
        int asDadChanged = 0;
        if (o->checkLastActiveSet) asDadChanged = w->xasChanged[p] | w->uasChanged[p];
        if (o->checkLastActiveSet == 0 || asDadChanged || w->xasChanged[k]) {              /* :563 */
            double *M = malloc(sizeof(double) * (w->maxM ? w->maxM : 1));
            const double *A = w->A + w->aoff[k], *B = w->B + w->boff[k];
            /* set_CmPnCmT (clipping.c:264-297): lower part of the nx x nx diagonal sub-block */
            build_M(w, k, p, M);
            for (int j = 0; j < nx; j++) for (int i = j; i < nx; i++) {
                double acc = 0.0;
                for (int c = 0; c < nxp; c++) acc += A[i + c * nx] * M[j + c * nx];
                double acc2 = 0.0;
                for (int c = 0; c < nup; c++) acc2 += B[i + c * nx] * M[j + (nxp + c) * nx];
                Wp[(pos + i) + (pos + j) * d] = acc + acc2;      /* syrk beta=0 then syrk beta=1 */
            }
            /* add_EPmE (clipping.c:301-314 / qpoases.c add_EPmE) */
            if (!w->dense) {
                for (int i = 0; i < nx; i++) Wp[(pos + i) + (pos + i) * d] += w->QinvCal[w->xoff[k] + i];
            } else {
                const int nz = nx + w->nu[k];
                const double *Pk = w->P + w->poff[k];
                for (int j = 0; j < nx; j++) for (int i = j; i < nx; i++) Wp[(pos + i) + (pos + j) * d] += Pk[i + j * nz];
            }
            if (o->checkLastActiveSet) {                                                   /* :573-577 */
                double *Wd = w->Wdiag + w->wdoff[k];
                for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) Wd[i + j * nx] = Wp[(pos + i) + (pos + j) * d];
            }
            /* parent coupling Ut (:581-589): Ut_p[:, pos..] = -(M[:, 0:nxp])' */
            if (w->dad[p] >= 0) {
                if (o->checkLastActiveSet == 0 || asDadChanged) {
                    double *Ut = w->Ut + w->utoff[p];
                    for (int j = 0; j < nx; j++) for (int i = 0; i < nxp; i++) Ut[i + (pos + j) * nxp] = -1.0 * M[j + i * nx];
                }
            }
            /* preceding siblings (:593-608), add_CmPnCkT (clipping.c:318-355) */
            if (o->checkLastActiveSet == 0 || asDadChanged) {
                int col = 0;
                for (int ii = 0; ii < w->nk[p] - 1; ii++) {
                    int s = w->kid0[p] + ii;
                    if (s == k) break;
                    const int nxs = w->nx[s];
                    build_M(w, s, p, M);
                    for (int j = 0; j < nxs; j++) for (int i = 0; i < nx; i++) {
                        double acc = 0.0;
                        for (int c = 0; c < nxp; c++) acc += A[i + c * nx] * M[j + c * nxs];
                        double acc2 = 0.0;
                        for (int c = 0; c < nup; c++) acc2 += B[i + c * nx] * M[j + (nxp + c) * nxs];
                        Wp[(pos + i) + (col + j) * d] = acc + acc2;
                    }
                    col += nxs;
                }
            }
            free(M);
        } else {
            const double *Wd = w->Wdiag + w->wdoff[k];                                     /* :613 */
            for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) Wp[(pos + i) + (pos + j) * d] = Wd[i + j * nx];
        }
    }
    return -1;   /* TREEQP_OK: continue */
}

/* ======================================================================================= */
/* Phase F: calculate_delta_lambda (dual_Newton_tree.c:641-805) with                       */
/* treeqp_dpotrf_l_with_reg_opts (dual_Newton_common.c:36-78)                              */
/* ======================================================================================= */

static void potrf_with_reg(ws_t *w, int p, const oracle_opts_t *o) {
    const int d = w->bdim[p];
    double *M = w->W + w->woff[p], *L = w->CholW + w->woff[p];
    double *inv = w->invd + w->xoff[w->kid0[p]];
    if (o->regType == ORC_NO_REG) {
        potrf_l(d, M, d, L, d, inv);
    } else if (o->regType == ORC_ALWAYS_LM) {
        for (int i = 0; i < d; i++) M[i + i * d] += o->regValue;                 /* ddiare :49 */
        potrf_l(d, M, d, L, d, inv);
#ifdef _OPENMP
#pragma omp atomic
#endif
        w->n_regularized++;
    } else {
        potrf_l(d, M, d, L, d, inv);
        for (int j = 0; j < d; j++) {
            if (L[j + j * d] <= o->regTol) {                                     /* :62 */
                for (int i = 0; i < d; i++) M[i + i * d] += o->regValue;         /* :65 */
                potrf_l(d, M, d, L, d, inv);                                          /* :68 */
#ifdef _OPENMP
#pragma omp atomic
#endif
                w->n_regularized++;
                break;
            }
        }
    }
}

static void calculate_delta_lambda(ws_t *w, const oracle_opts_t *o, int idxFactorStart, int nthreads) {
    const int Nh = w->Nh, Np = w->Np;
    int icur = Np - 1;
    (void)nthreads;
    for (int kk = Nh - 1; kk > 0; kk--) {                                        /* :668 */
        const int lo = icur - w->npar[kk];
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1)
#endif
        for (int ii = icur; ii > lo; ii--) {                                     /* :675 */
            const int d = w->bdim[ii], nxi = w->nx[ii];
            double *L = w->CholW + w->woff[ii];
            double *rm = w->resMod + w->xoff[w->kid0[ii]];
            double *dl = w->dlam + w->xoff[w->kid0[ii]];
            if (o->checkLastActiveSet == 0 || ii < idxFactorStart) potrf_with_reg(w, ii, o);   /* :703-707 */
            const double *inv = w->invd + w->xoff[w->kid0[ii]];
            trsv_lnn(d, L, d, inv, rm, dl);                                      /* :712 */
            double *Ut = w->Ut + w->utoff[ii], *CUt = w->CholUt + w->utoff[ii];
            trsm_rltn(nxi, d, L, d, inv, Ut, nxi, CUt, nxi);                          /* :718 */
            const int dd = w->dad[ii], pos = w->pos[ii], ddim = w->bdim[dd];
            double *Wd = w->W + w->woff[dd];
            for (int j = 0; j < nxi; j++) for (int i = j; i < nxi; i++) {        /* dsyrk_ln :726 */
                double acc = 0.0;
                for (int c = 0; c < d; c++) acc += CUt[i + c * nxi] * CUt[j + c * nxi];
                Wd[(pos + i) + (pos + j) * ddim] = Wd[(pos + i) + (pos + j) * ddim] + (-1.0) * acc;
            }
            gemv_n_acc(nxi, d, -1.0, CUt, nxi, dl, w->resMod + w->xoff[ii]);     /* :731 */
        }
        icur -= w->npar[kk];
    }
    {   /* root (:746-758) */
        const int d = w->bdim[0];
        double *L = w->CholW + w->woff[0];
        double *rm = w->resMod + w->xoff[w->kid0[0]], *dl = w->dlam + w->xoff[w->kid0[0]];
        /* NOTE: the reference always refactorizes the root (:748) */
        potrf_with_reg(w, 0, o);
        const double *inv = w->invd + w->xoff[w->kid0[0]];
        trsv_lnn(d, L, d, inv, rm, dl);
        trsv_ltn(d, L, d, inv, dl, dl);
    }
    icur = 1;
    for (int kk = 1; kk < Nh; kk++) {                                            /* :760-775 */
        const int hi = icur + w->npar[kk];
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1)
#endif
        for (int ii = icur; ii < hi; ii++) {
            const int d = w->bdim[ii], nxi = w->nx[ii];
            double *dl = w->dlam + w->xoff[w->kid0[ii]];
            const double *CUt = w->CholUt + w->utoff[ii];
            gemv_t_acc(nxi, d, -1.0, CUt, nxi, w->dlam + w->xoff[ii], dl);       /* :768 */
            trsv_ltn(d, w->CholW + w->woff[ii], d, w->invd + w->xoff[w->kid0[ii]], dl, dl);                      /* :771 */
        }
        icur += w->npar[kk];
    }
}

/* ======================================================================================= */
/* Phase L: line_search (dual_Newton_tree.c:922-1019)                                      */
/* ======================================================================================= */

static double gradient_trans_times_direction(const ws_t *w) {                   /* :808-820 */
    double ans = 0.0;
    for (int p = 0; p < w->Np; p++) {
        int o = w->xoff[w->kid0[p]];
        ans += dot(w->bdim[p], w->res + o, w->dlam + o);
    }
    return -ans;
}

static double evaluate_dual_function(ws_t *w, int nthreads) {                   /* :823-918 */
    stage_sweep(w, 0, 1, nthreads);
    double f = 0.0;
    for (int k = 0; k < w->Nn; k++) f += w->fval[k];                            /* :915 */
    return f;
}

static int line_search(ws_t *w, const oracle_opts_t *o, double *fval_out, int nthreads) {
    double tau = 1.0, tauPrev = 0.0, fval = 0.0;
    double dot_product = gradient_trans_times_direction(w);                     /* :944 */
    double fval0 = evaluate_dual_function(w, nthreads);                         /* :945 */
    if (dot_product > 1e-10 || !((dot_product > 1e-10) || (dot_product < 1e-10)))   /* :951 */
        return ORC_NOT_DESCENT;
    const int n0 = w->xoff[1], n1 = w->xoff[w->Nn];
    int lsIter;
    for (lsIter = 1; lsIter <= o->lineSearchMaxIter; lsIter++) {                /* :958 */
        const double step = tau - tauPrev;
        for (int i = n0; i < n1; i++) w->lam[i] = w->lam[i] + step * w->dlam[i];   /* daxpy :966 */
        fval = evaluate_dual_function(w, nthreads);                             /* :970 */
        if (w->lineSearchRestartCounter == o->lineSearchRestartTrigger) break;  /* :973 */
        if (fval <= fval0 + o->lineSearchGamma * tau * dot_product) break;      /* :982 */
        tauPrev = tau;
        tau = o->lineSearchBeta * tauPrev;
    }
    if (lsIter >= o->lineSearchMaxIter) w->lineSearchRestartCounter++;          /* :993-1000 */
    else w->lineSearchRestartCounter = 0;
    w->lsIter = lsIter;
    *fval_out = fval;
    return -1;
}

/* ======================================================================================= */
/* driver: treeqp_tdunes_solve (dual_Newton_tree.c:1104-1263)                              */
/* ======================================================================================= */

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int validate_opts(const oracle_opts_t *o) {                              /* :1078-1100 */
    if (o->termCondition != ORC_SUMSQUAREDERRORS && o->termCondition != ORC_TWONORM &&
        o->termCondition != ORC_INFNORM) return 0;
    if (o->regType != ORC_NO_REG && o->regType != ORC_ALWAYS_LM && o->regType != ORC_ON_THE_FLY_LM) return 0;
    if (o->regValue < 0) return 0;
    return 1;
}

static int newton_loop(ws_t *w, const oracle_opts_t *o, oracle_info_t *info) {
    const int nthreads = o->num_threads > 1 ? o->num_threads : 1;
    int status = ORC_OPTIMAL, NewtonIter, idxFactorStart;
    info->ls_total = 0;
    w->lineSearchRestartCounter = 0;                                            /* :1137 */
    double t0 = now_sec();
    for (NewtonIter = 0; NewtonIter < o->maxIter; NewtonIter++) {               /* :1166 */
        stage_sweep(w, 1, 0, nthreads);                                         /* :1176 */
        double err = 0.0;
        int st = build_dual_problem(w, o, &idxFactorStart, &err, nthreads);     /* :1187 */
        if (info->trace_err) info->trace_err[NewtonIter] = err;
        if (st == ORC_OPTIMAL) break;                                           /* :1191-1197 */
        calculate_delta_lambda(w, o, idxFactorStart, nthreads);                 /* :1203 */
        double fv = 0.0;
        st = line_search(w, o, &fv, nthreads);                                  /* :1214 */
        if (st == ORC_NOT_DESCENT) { info->solver_time = now_sec() - t0; info->iter = NewtonIter; return ORC_NOT_DESCENT; }
        if (info->trace_fval) info->trace_fval[NewtonIter] = fv;
        if (info->trace_ls) info->trace_ls[NewtonIter] = w->lsIter;
        info->ls_total += w->lsIter;
    }
    info->solver_time = now_sec() - t0;
    info->iter = NewtonIter;                                                    /* :1248 */
    if (NewtonIter == o->maxIter) status = ORC_MAXITER;                         /* :1252 */
    return status;
}

static void load_lambda0(ws_t *w, const double *lambda0) {                      /* :1654-1663 */
    const int n0 = w->xoff[1], n1 = w->xoff[w->Nn];
    for (int i = n0; i < n1; i++) w->lam[i] = lambda0 ? lambda0[i - n0] : 0.0;
}

int oracle_tdunes_solve(int Nn, const int *nk, const int *nx, const int *nu,
                        const double *A, const double *B, const double *b,
                        const double *Qd, const double *Rd, const double *q, const double *r,
                        const double *xmin, const double *xmax,
                        const double *umin, const double *umax,
                        const oracle_opts_t *opts, const double *lambda0,
                        double *x, double *u, double *lam, double *mu_x, double *mu_u,
                        oracle_info_t *info) {
    oracle_info_t local; memset(&local, 0, sizeof(local));
    if (!info) info = &local;
    if (!validate_opts(opts)) { info->status = ORC_INVALID_OPTION; return ORC_INVALID_OPTION; }
    ws_t w;
    ws_setup(&w, Nn, nk, nx, nu);
    w.A = A; w.B = B; w.b = b; w.Qd = Qd; w.Rd = Rd; w.q = q; w.r = r;
    w.xmin = xmin; w.xmax = xmax; w.umin = umin; w.umax = umax;
    /* stage_qp_clipping_init (clipping.c:149-184) + NaN previous active set (:1155-1159) */
    for (int i = 0; i < w.xoff[Nn]; i++) { w.Qinv[i] = 1.0 / Qd[i]; w.xasPrev[i] = NAN; }
    for (int i = 0; i < w.uoff[Nn]; i++) { w.Rinv[i] = 1.0 / Rd[i]; w.uasPrev[i] = NAN; }
    load_lambda0(&w, lambda0);

    int status = newton_loop(&w, opts, info);
    info->status = status;
    info->n_regularized = w.n_regularized;
    if (status != ORC_NOT_DESCENT) {
        /* export (:1235-1247) and export_mu (clipping.c:386-399) */
        const int n0 = w.xoff[1];
        memcpy(x, w.x, sizeof(double) * w.xoff[Nn]);
        memcpy(u, w.u, sizeof(double) * w.uoff[Nn]);
        memcpy(lam, w.lam + n0, sizeof(double) * (w.xoff[Nn] - n0));
        int nact = 0;
        for (int i = 0; i < w.xoff[Nn]; i++) {
            mu_x[i] = Qd[i] * (w.xUnc[i] + (-1.0) * w.x[i]);
            if (w.QinvCal[i] == 0.0) nact++;
        }
        for (int i = 0; i < w.uoff[Nn]; i++) {
            mu_u[i] = Rd[i] * (w.uUnc[i] + (-1.0) * w.u[i]);
            if (w.RinvCal[i] == 0.0) nact++;
        }
        info->n_active = nact;
    }
    ws_free(&w);
    return status;
}

int oracle_tdunes_solve_dense(int Nn, const int *nk, const int *nx, const int *nu,
                              const double *A, const double *B, const double *b,
                              const double *Q, const double *R, const double *S,
                              const double *q, const double *r,
                              const oracle_opts_t *opts, const double *lambda0,
                              double *x, double *u, double *lam, oracle_info_t *info) {
    oracle_info_t local; memset(&local, 0, sizeof(local));
    if (!info) info = &local;
    if (!validate_opts(opts)) { info->status = ORC_INVALID_OPTION; return ORC_INVALID_OPTION; }
    ws_t w;
    ws_setup(&w, Nn, nk, nx, nu);
    w.A = A; w.B = B; w.b = b; w.q = q; w.r = r;
    w.dense = 1; w.Q = Q; w.R = R; w.S = S;
    dense_init(&w);
    for (int i = 0; i < w.xoff[Nn]; i++) w.xasPrev[i] = NAN;
    for (int i = 0; i < w.uoff[Nn]; i++) w.uasPrev[i] = NAN;
    load_lambda0(&w, lambda0);
    int status = newton_loop(&w, opts, info);
    info->status = status;
    if (status != ORC_NOT_DESCENT) {
        const int n0 = w.xoff[1];
        memcpy(x, w.x, sizeof(double) * w.xoff[Nn]);
        memcpy(u, w.u, sizeof(double) * w.uoff[Nn]);
        memcpy(lam, w.lam + n0, sizeof(double) * (w.xoff[Nn] - n0));
    }
    ws_free(&w);
    return status;
}

/* ======================================================================================= */
/* KKT residual (tree_qp_common.c:540-788)                                                 */
/* ======================================================================================= */

double oracle_max_kkt(int Nn, const int *nk, const int *nx, const int *nu,
                      const double *A, const double *B, const double *b,
                      const double *Qd, const double *Rd,
                      const double *Q, const double *R, const double *S,
                      const double *q, const double *r,
                      const double *xmin, const double *xmax,
                      const double *umin, const double *umax,
                      const double *x, const double *u, const double *lam,
                      const double *mu_x, const double *mu_u) {
    ws_t w;
    ws_setup(&w, Nn, nk, nx, nu);
    double err = 0.0;
    int maxn = 1;
    for (int k = 0; k < Nn; k++) maxn = OMAX(maxn, OMAX(nx[k], nu[k]));
    double *tx = xcalloc(maxn, 8), *tu = xcalloc(maxn, 8);
    int roff = 0, soff = 0;
    const int n0 = nx[0];
#define UPD(v) do { double a_ = fabs(v); if (a_ > err || a_ != a_) err = a_; } while (0)
    for (int k = 0; k < Nn; k++) {
        const int nxk = nx[k], nuk = nu[k];
        const double *xk = x + w.xoff[k], *uk = u + w.uoff[k];
        /* stationarity (:589-625) */
        for (int i = 0; i < nxk; i++) {
            double acc = 0.0;
            if (Q) { const double *Qk = Q + w.wdoff[k]; for (int j = 0; j < nxk; j++) acc += Qk[i + j * nxk] * xk[j]; }
            else acc = Qd[w.xoff[k] + i] * xk[i];
            tx[i] = q[w.xoff[k] + i] + acc;
            if (S) { const double *Sk = S + soff; double a2 = 0.0; for (int j = 0; j < nuk; j++) a2 += Sk[j + i * nuk] * uk[j]; tx[i] += a2; }
            if (mu_x) tx[i] += mu_x[w.xoff[k] + i];
            if (k > 0) tx[i] += -1.0 * lam[w.xoff[k] - n0 + i];
        }
        for (int i = 0; i < nuk; i++) {
            double acc = 0.0;
            if (R) { const double *Rk = R + roff; for (int j = 0; j < nuk; j++) acc += Rk[i + j * nuk] * uk[j]; }
            else acc = Rd[w.uoff[k] + i] * uk[i];
            tu[i] = r[w.uoff[k] + i] + acc;
            if (S) { const double *Sk = S + soff; double a2 = 0.0; for (int j = 0; j < nxk; j++) a2 += Sk[i + j * nuk] * xk[j]; tu[i] += a2; }
            if (mu_u) tu[i] += mu_u[w.uoff[k] + i];
        }
        for (int jj = 0; jj < nk[k]; jj++) {
            int kid = w.kid0[k] + jj;
            const double *lk = lam + (w.xoff[kid] - n0);
            gemv_t_acc(nx[kid], nxk, 1.0, A + w.aoff[kid], nx[kid], lk, tx);
            gemv_t_acc(nx[kid], nuk, 1.0, B + w.boff[kid], nx[kid], lk, tu);
        }
        for (int i = 0; i < nxk; i++) UPD(tx[i]);
        for (int i = 0; i < nuk; i++) UPD(tu[i]);
        /* dynamics (:629-646) */
        if (k > 0) {
            const int p = w.dad[k];
            for (int i = 0; i < nxk; i++) tx[i] = b[w.xoff[k] - n0 + i];
            gemv_n_acc(nxk, nx[p], 1.0, A + w.aoff[k], nxk, x + w.xoff[p], tx);
            gemv_n_acc(nxk, nu[p], 1.0, B + w.boff[k], nxk, u + w.uoff[p], tx);
            for (int i = 0; i < nxk; i++) UPD(tx[i] - xk[i]);
        }
        /* bounds feasibility (:651-683) and complementarity (:688-714) */
        for (int i = 0; i < nxk; i++) {
            double lb = xmin ? xmin[w.xoff[k] + i] : -1e12, ub = xmax ? xmax[w.xoff[k] + i] : 1e12;
            if (xk[i] > ub) UPD(xk[i] - ub); else if (xk[i] < lb) UPD(lb - xk[i]);
            double mu = mu_x ? mu_x[w.xoff[k] + i] : 0.0;
            if (mu > 0) UPD(mu * (xk[i] - ub)); else UPD(mu * (-xk[i] + lb));
        }
        for (int i = 0; i < nuk; i++) {
            double lb = umin ? umin[w.uoff[k] + i] : -1e12, ub = umax ? umax[w.uoff[k] + i] : 1e12;
            if (uk[i] > ub) UPD(uk[i] - ub); else if (uk[i] < lb) UPD(lb - uk[i]);
            double mu = mu_u ? mu_u[w.uoff[k] + i] : 0.0;
            if (mu > 0) UPD(mu * (uk[i] - ub)); else UPD(mu * (-uk[i] + lb));
        }
        roff += nuk * nuk; soff += nuk * nxk;
    }
#undef UPD
    free(tx); free(tu);
    ws_free(&w);
    return err;
}
