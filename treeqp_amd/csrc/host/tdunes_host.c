/*
 * tdunes_host.c -- treeqp_tdunes_* front end of the treeqp_amd build (host C).
 *
 * Keeps the reference's API (treeqp/src/dual_Newton_tree.h:154-172) and its observable
 * contract: caller-owned buffers, `return_t` by value, qp_out untouched on an
 * early error return, host-visible sx/su/slambda/sDeltalambda after a solve, warm start from
 * whatever slambda holds.  Everything numerical is delegated to the device C-ABI
 * (include/treeqp_amd.h): this file only validates, stages flat regions and copies results.
 * There is deliberately no CPU solve path: if the HIP device path is unavailable,
 * treeqp_tdunes_create reports the device error and exits (the reference's convention for
 * fatal configuration errors, dual_Newton_tree_clipping.c:70-74).
 */
#include "treeqp/src/dual_Newton_tree.h"
#include "treeqp/src/tree_qp_common.h"
#include "treeqp/utils/blasfeo.h"
#include "treeqp/utils/memory.h"
#include "treeqp/utils/profiling.h"
#include "treeqp/utils/timing.h"
#include "treeqp/utils/tree.h"
#include "treeqp/utils/utils.h"
#include "treeqp_amd.h"

#include <blasfeo_d_aux.h>

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------- */
/* options (dual_Newton_tree.c:71-120)                                                   */
/* ------------------------------------------------------------------------------------- */

int treeqp_tdunes_opts_calculate_size(int Nn) { return Nn * (int)sizeof(stage_qp_t); }

void treeqp_tdunes_opts_create(int Nn, treeqp_tdunes_opts_t *opts, void *ptr)
{
    (void)Nn;
    opts->qp_solver = (stage_qp_t *)ptr;
}

void treeqp_tdunes_opts_set_default(int Nn, treeqp_tdunes_opts_t *opts)
{
    opts->maxIter = 100;
    opts->termCondition = TREEQP_INFNORM;
    opts->stationarityTolerance = 1.0e-8;
    opts->checkLastActiveSet = 1;
    for (int k = 0; k < Nn; k++) opts->qp_solver[k] = TREEQP_CLIPPING_SOLVER;
    opts->lineSearchMaxIter = 50;
    opts->lineSearchGamma = 0.1;
    opts->lineSearchBeta = 0.6;
    opts->lineSearchRestartTrigger = -1;
    opts->regType = TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT;
    opts->regTol = 1.0e-6;
    opts->regValue = 1.0e-6;
}

/* dual_Newton_tree.c:1078-1100 */
static return_t validate_opts(const treeqp_tdunes_opts_t *opts)
{
    if (opts->termCondition != TREEQP_SUMSQUAREDERRORS && opts->termCondition != TREEQP_TWONORM &&
        opts->termCondition != TREEQP_INFNORM) return TREEQP_INVALID_OPTION;
    if (opts->regType != TREEQP_NO_REGULARIZATION && opts->regType != TREEQP_ALWAYS_LEVENBERG_MARQUARDT &&
        opts->regType != TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT) return TREEQP_INVALID_OPTION;
    if (opts->regValue < 0) return TREEQP_INVALID_OPTION;
    return TREEQP_OK;
}

/* ------------------------------------------------------------------------------------- */
/* live device mirrors: released at exit because the reference API has no destroy call   */
/* ------------------------------------------------------------------------------------- */

enum { MAX_LIVE = 4096 };
static tqgpu_solver *g_live[MAX_LIVE];
static int g_nlive = 0, g_atexit_registered = 0;

static void release_all(void)
{
    for (int i = 0; i < g_nlive; i++) if (g_live[i]) { tqgpu_destroy(g_live[i]); g_live[i] = NULL; }
    g_nlive = 0;
}
static void track(tqgpu_solver *s)
{
    if (!g_atexit_registered) { atexit(release_all); g_atexit_registered = 1; }
    for (int i = 0; i < g_nlive; i++) if (!g_live[i]) { g_live[i] = s; return; }
    if (g_nlive < MAX_LIVE) g_live[g_nlive++] = s;
}
static void untrack(tqgpu_solver *s)
{
    for (int i = 0; i < g_nlive; i++) if (g_live[i] == s) g_live[i] = NULL;
}

/* ------------------------------------------------------------------------------------- */
/* sizing / creation (dual_Newton_tree.c:1291-1648)                                      */
/* ------------------------------------------------------------------------------------- */

static int block_dim(const tree_qp_in *qp_in, int p)
{
    int d = 0;
    for (int c = 0; c < qp_in->tree[p].nkids; c++) d += qp_in->nx[qp_in->tree[p].kids[c]];
    return d;
}

/* doubles in the host staging slab: Qd,Rd + room to gather every flat region if the
 * container's views turn out not to be contiguous */
static int staging_doubles(const tree_qp_in *qp_in)
{
    const int Nn = qp_in->N;
    size_t d = 0;
    for (int k = 0; k < Nn; k++) {
        d += 4 * (size_t)qp_in->nx[k] + 4 * (size_t)qp_in->nu[k];
        /* dense stage solver: Q, R, S flattened */
        d += (size_t)(qp_in->nx[k] + qp_in->nu[k]) * (qp_in->nx[k] + qp_in->nu[k]);
        if (k > 0) {
            const int p = qp_in->tree[k].dad;
            d += (size_t)qp_in->nx[k] * (qp_in->nx[p] + qp_in->nu[p] + 1);
        }
    }
    return (int)d;
}

int treeqp_tdunes_calculate_size(const tree_qp_in *qp_in, const treeqp_tdunes_opts_t *opts)
{
    const int Nn = qp_in->N;
    const int Nh = qp_in->tree[Nn - 1].stage;
    const int Np = get_number_of_parent_nodes(Nn, qp_in->tree);
    size_t bytes = 0;
    bytes += (size_t)(Nh + 1 + Nn) * sizeof(int);                         /* npar, idxpos */
    bytes += (size_t)(2 * Nn + 2 * Np) * sizeof(struct blasfeo_dvec);     /* sx, su, slambda, sDeltalambda */
    size_t dbl = 0;
    for (int k = 0; k < Nn; k++) dbl += (size_t)qp_in->nx[k] + qp_in->nu[k];
    for (int p = 0; p < Np; p++) dbl += 2 * (size_t)block_dim(qp_in, p);
    dbl += (size_t)staging_doubles(qp_in);
    bytes += dbl * sizeof(double);
    bytes += (size_t)timers_calculate_size(opts->maxIter);
    int ib = (int)bytes;
    make_int_multiple_of(64, &ib);
    return ib + 3 * 64;
}

static void fatal(const char *what, const char *detail)
{
    printf("[TREEQP]: Error! %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
    exit(1);
}

/* stage_qp_clipping_is_applicable (dual_Newton_tree_clipping.c:45-77) */
static void require_clipping_applicable(const tree_qp_in *qp_in, int k)
{
    if (is_strmat_diagonal(&qp_in->Q[k]) == NO || is_strmat_diagonal(&qp_in->R[k]) == NO ||
        is_strmat_zero(&qp_in->S[k]) == NO || qp_in->nc[k] > 0)
        fatal("Specified stage QP solver (clipping) not applicable.", NULL);
}

/* The dense stage solver of this build covers unconstrained nodes only (bounds at +-inf as written by
 * tree_qp_in_set_inf_bounds, no general constraints); constrained dense stage QPs need an active-set QP
 * solver (qpOASES in the reference), which is out of scope. */
static int is_inf_bound(double lo, double hi) { return lo <= -1e12 && hi >= 1e12; }
static void require_dense_unconstrained(const tree_qp_in *qp_in, int k)
{
    int ok = qp_in->nc[k] == 0;
    for (int j = 0; ok && j < qp_in->nx[k]; j++) ok = is_inf_bound(BLASFEO_DVECEL(&qp_in->xmin[k], j), BLASFEO_DVECEL(&qp_in->xmax[k], j));
    for (int j = 0; ok && j < qp_in->nu[k]; j++) ok = is_inf_bound(BLASFEO_DVECEL(&qp_in->umin[k], j), BLASFEO_DVECEL(&qp_in->umax[k], j));
    if (!ok)
        fatal("TREEQP_QPOASES_SOLVER is available for unconstrained nodes only in the MI355X build (dense stage solver; qpOASES itself is out of scope).", NULL);
}

void treeqp_tdunes_create(const tree_qp_in *qp_in, const treeqp_tdunes_opts_t *opts,
    treeqp_tdunes_workspace *work, void *ptr)
{
    const int Nn = qp_in->N;
    const struct node *tree = qp_in->tree;
    const int Nh = tree[Nn - 1].stage;
    const int Np = get_number_of_parent_nodes(Nn, tree);
    char *c_ptr = (char *)ptr;

    memset(work, 0, sizeof(*work));
    work->Nn = Nn;
    work->Np = Np;
    work->maxIterAtCreate = opts->maxIter;

    /* Stage solvers (dual_Newton_tree.c:1399-1425 dispatches per node): clipping, or -- under the reference's
     * TREEQP_QPOASES_SOLVER selector -- the dense stage solver for UNCONSTRAINED nodes (the part of that backend
     * that needs no active-set QP solver: z = H^-1 h, P = H^-1).  One kind for the whole tree. */
    int n_dense = 0;
    for (int k = 0; k < Nn; k++) {
        /* the solver (like the reference, dual_Newton_tree.c:675,1377) needs parents == nodes 0..Np-1 */
        if ((tree[k].nkids > 0) != (k < Np)) fatal("tdunes needs all leaves at the same depth.", NULL);
        if (opts->qp_solver[k] == TREEQP_CLIPPING_SOLVER) require_clipping_applicable(qp_in, k);
        else if (opts->qp_solver[k] == TREEQP_QPOASES_SOLVER) { require_dense_unconstrained(qp_in, k); n_dense++; }
        else fatal("Unknown stage QP solver.", NULL);
    }
    /* any mix of the two kinds across nodes is fine (the reference binds the vtable per node, dual_Newton_tree.c:124-162) */
    work->denseStageSolver = n_dense > 0;

    /* integer tables: dual_Newton_tree.c:166-194 */
    work->npar = (int *)c_ptr; c_ptr += (size_t)(Nh + 1) * sizeof(int);
    for (int s = 0; s <= Nh; s++) work->npar[s] = 0;
    for (int k = 0; k < Nn; k++) work->npar[tree[k].stage]++;
    work->idxpos = (int *)c_ptr; c_ptr += (size_t)Nn * sizeof(int);
    for (int k = 0; k < Nn; k++) {
        work->idxpos[k] = 0;
        for (int c = 0; c < tree[k].idxkid; c++) work->idxpos[k] += qp_in->nx[tree[tree[k].dad].kids[c]];
    }

    align_char_to(8, &c_ptr);
    work->sx = (struct blasfeo_dvec *)c_ptr; c_ptr += (size_t)Nn * sizeof(struct blasfeo_dvec);
    work->su = (struct blasfeo_dvec *)c_ptr; c_ptr += (size_t)Nn * sizeof(struct blasfeo_dvec);
    work->slambda = (struct blasfeo_dvec *)c_ptr; c_ptr += (size_t)Np * sizeof(struct blasfeo_dvec);
    work->sDeltalambda = (struct blasfeo_dvec *)c_ptr; c_ptr += (size_t)Np * sizeof(struct blasfeo_dvec);
    align_char_to(64, &c_ptr);

    /* flat, region-contiguous mirrors: x | u | lambda | Deltalambda */
    for (int k = 0; k < Nn; k++) init_strvec(qp_in->nx[k], &work->sx[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(qp_in->nu[k], &work->su[k], &c_ptr);
    for (int p = 0; p < Np; p++) init_strvec(block_dim(qp_in, p), &work->slambda[p], &c_ptr);
    for (int p = 0; p < Np; p++) init_strvec(block_dim(qp_in, p), &work->sDeltalambda[p], &c_ptr);

    work->stage_doubles = staging_doubles(qp_in);
    work->stage = (double *)c_ptr; c_ptr += (size_t)work->stage_doubles * sizeof(double);

    timers_create(opts->maxIter, &work->timings, c_ptr);
    c_ptr += timers_calculate_size(opts->maxIter);
    timers_initialize(&work->timings);

    assert((char *)ptr + treeqp_tdunes_calculate_size(qp_in, opts) >= c_ptr);

    /* device mirror */
    int *nk = malloc(sizeof(int) * (size_t)Nn);
    for (int k = 0; k < Nn; k++) nk[k] = tree[k].nkids;
    int device = -1;                                   /* -1: keep the process' current device */
    const char *env = getenv("TREEQP_AMD_DEVICE");
    if (env && *env) device = atoi(env);
    int rc = tqgpu_create(&work->device, device, Nn, nk, qp_in->nx, qp_in->nu);
    free(nk);
    if (rc != TQGPU_OK) fatal("cannot create the MI355X device mirror for tdunes (no CPU fallback exists)", tqgpu_last_error());
    /* this front end fetches the solution after every solve and times the solve with the host clock: the download goes out behind the
     * solve's launch, and no HIP event pair is recorded around it (TREEQP_AMD_DROPIN_PLAIN=1: neither, for comparison) */
    if (!getenv("TREEQP_AMD_DROPIN_PLAIN")) { (void)tqgpu_set_export_ahead(work->device, 1); (void)tqgpu_set_event_timing(work->device, 0); }
    track(work->device);
}

void treeqp_tdunes_destroy(treeqp_tdunes_workspace *work)
{
    if (work && work->device) {
        untrack(work->device);
        tqgpu_destroy(work->device);
        work->device = NULL;
    }
}

/* dual_Newton_tree.c:1654-1663: flat lambda in block order (== edge order) */
void treeqp_tdunes_set_dual_initialization(const double *lambda, treeqp_tdunes_workspace *work)
{
    int at = 0;
    for (int p = 0; p < work->Np; p++) {
        blasfeo_pack_dvec(work->slambda[p].m, (double *)&lambda[at], &work->slambda[p], 0);
        at += work->slambda[p].m;
    }
}

/* ------------------------------------------------------------------------------------- */
/* flat regions of the container                                                         */
/* ------------------------------------------------------------------------------------- */

/* Return a pointer to the concatenation of `n` column-major pieces.  The treeqp_amd container
 * stores each kind contiguously, in which case this is zero-copy; foreign/non-contiguous views
 * are gathered into the staging slab. */
static const double *flat_of_mats(const struct blasfeo_dmat *M, int n, double **stage)
{
    const double *first = NULL, *expect = NULL;
    int contiguous = 1;
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        const size_t cnt = (size_t)M[i].m * (size_t)M[i].n;
        if (cnt == 0 || !M[i].pA) continue;
        if (!first) first = M[i].pA; else if (M[i].pA != expect) contiguous = 0;
        expect = M[i].pA + cnt;
        total += cnt;
    }
    if (contiguous) return first ? first : *stage;
    double *out = *stage, *w = out;
    for (int i = 0; i < n; i++) {
        const size_t cnt = (size_t)M[i].m * (size_t)M[i].n;
        if (cnt == 0 || !M[i].pA) continue;
        memcpy(w, M[i].pA, cnt * sizeof(double)); w += cnt;
    }
    *stage += total;
    return out;
}
static const double *flat_of_vecs(const struct blasfeo_dvec *v, int n, double **stage)
{
    const double *first = NULL, *expect = NULL;
    int contiguous = 1;
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (v[i].m == 0 || !v[i].pa) continue;
        if (!first) first = v[i].pa; else if (v[i].pa != expect) contiguous = 0;
        expect = v[i].pa + v[i].m;
        total += (size_t)v[i].m;
    }
    if (contiguous) return first ? first : *stage;
    double *out = *stage, *w = out;
    for (int i = 0; i < n; i++) {
        if (v[i].m == 0 || !v[i].pa) continue;
        memcpy(w, v[i].pa, (size_t)v[i].m * sizeof(double)); w += v[i].m;
    }
    *stage += total;
    return out;
}
/* writable variant for outputs: returns the region start if the views the solver writes (dims[i] > 0
 * entries each; dims == NULL: the views' own sizes) are contiguous and exactly that long, else NULL.
 * The solver's dimensions rule: after tree_qp_in_eliminate_x0 the caller's qp_out may still carry an
 * x[0] of the original size (solve_qp_json.cpp never shrinks it), which then is simply not written,
 * as in the reference's per-node blasfeo_dveccp (dual_Newton_tree.c:1235-1247). */
static double *flat_out(struct blasfeo_dvec *v, int n, const int *dims)
{
    double *first = NULL, *expect = NULL;
    for (int i = 0; i < n; i++) {
        const int m = dims ? dims[i] : v[i].m;
        if (m == 0 || !v[i].pa) continue;
        if (v[i].m != m) return NULL;
        if (!first) first = v[i].pa; else if (v[i].pa != expect) return NULL;
        expect = v[i].pa + m;
    }
    return first;
}
static void scatter(const double *flat, struct blasfeo_dvec *v, int n, const int *dims)
{
    for (int i = 0; i < n; i++) {
        const int m = dims ? dims[i] : v[i].m;
        if (m == 0 || !v[i].pa) continue;
        memcpy(v[i].pa, flat, (size_t)m * sizeof(double)); flat += m;
    }
}

/* ------------------------------------------------------------------------------------- */
/* solve (dual_Newton_tree.c:1104-1263)                                                  */
/* ------------------------------------------------------------------------------------- */

#define DEV_CALL(expr) do { int rc_ = (expr); if (rc_ != TQGPU_OK) fatal("device call failed in treeqp_tdunes_solve", tqgpu_last_error()); } while (0)

return_t treeqp_tdunes_solve(const tree_qp_in *qp_in, tree_qp_out *qp_out,
    const treeqp_tdunes_opts_t *opts, treeqp_tdunes_workspace *work)
{
    treeqp_timer total_tmr, interface_tmr, solver_tmr;
    const int Nn = work->Nn, Np = work->Np;
    treeqp_profiling_t *timings = &work->timings;

    treeqp_tic(&total_tmr);
    treeqp_tic(&interface_tmr);
    /* the reference asserts here ("Number of iterations cannot be increased after initializing solver"); an option the
     * workspace was not sized for is reported like any other invalid option instead of aborting the caller */
    if (timings->num_iter < opts->maxIter) return TREEQP_INVALID_OPTION;

    work->lineSearchRestartCounter = 0;
    return_t status = validate_opts(opts);
    if (status != TREEQP_OK) return status;
    if (!work->device) fatal("treeqp_tdunes_solve called without a device mirror", NULL);

    /* --- init (stage_qp_clipping_init, clipping.c:149-184: the diagonals are re-read at every
     * solve) + staging of the values that may have changed since the last solve */
    double *stage = work->stage;
    double *Qd = stage; for (int k = 0; k < Nn; k++) for (int j = 0; j < qp_in->nx[k]; j++) *stage++ = BLASFEO_DMATEL(&qp_in->Q[k], j, j);
    double *Rd = stage; for (int k = 0; k < Nn; k++) for (int j = 0; j < qp_in->nu[k]; j++) *stage++ = BLASFEO_DMATEL(&qp_in->R[k], j, j);
    const double *A = flat_of_mats(qp_in->A, Nn - 1, &stage);
    const double *B = flat_of_mats(qp_in->B, Nn - 1, &stage);
    const double *b = flat_of_vecs(qp_in->b, Nn - 1, &stage);
    const double *q = flat_of_vecs(qp_in->q, Nn, &stage);
    const double *r = flat_of_vecs(qp_in->r, Nn, &stage);
    const double *xmin = flat_of_vecs(qp_in->xmin, Nn, &stage);
    const double *xmax = flat_of_vecs(qp_in->xmax, Nn, &stage);
    const double *umin = flat_of_vecs(qp_in->umin, Nn, &stage);
    const double *umax = flat_of_vecs(qp_in->umax, Nn, &stage);
    assert(stage <= work->stage + work->stage_doubles);

    if (work->denseStageSolver) {
        DEV_CALL(tqgpu_set_dynamics(work->device, A, B, b));
        /* flat Q, R, S in the order of tree_qp_in_set_ltv_objective_colmajor; bounds are re-checked, not uploaded */
        double *Qf = stage; for (int k = 0; k < Nn; k++) for (int j = 0; j < qp_in->nx[k]; j++) for (int i = 0; i < qp_in->nx[k]; i++) *stage++ = BLASFEO_DMATEL(&qp_in->Q[k], i, j);
        double *Rf = stage; for (int k = 0; k < Nn; k++) for (int j = 0; j < qp_in->nu[k]; j++) for (int i = 0; i < qp_in->nu[k]; i++) *stage++ = BLASFEO_DMATEL(&qp_in->R[k], i, j);
        double *Sf = stage; for (int k = 0; k < Nn; k++) for (int j = 0; j < qp_in->nx[k]; j++) for (int i = 0; i < qp_in->nu[k]; i++) *stage++ = BLASFEO_DMATEL(&qp_in->S[k], i, j);
        assert(stage <= work->stage + work->stage_doubles);
        int *kind = malloc(sizeof(int) * (size_t)Nn);
        for (int k = 0; k < Nn; k++) {
            kind[k] = opts->qp_solver[k] == TREEQP_QPOASES_SOLVER;
            if (kind[k]) require_dense_unconstrained(qp_in, k); else require_clipping_applicable(qp_in, k);
        }
        int rc_obj = tqgpu_set_objective_mixed(work->device, kind, Qf, Rf, Sf, q, r);
        free(kind);
        if (rc_obj != TQGPU_OK) fatal("device call failed in treeqp_tdunes_solve", tqgpu_last_error());
        DEV_CALL(tqgpu_set_bounds(work->device, xmin, xmax, umin, umax));
        /* warm start from whatever slambda holds (set_dual_initialization or the previous solve) */
        DEV_CALL(tqgpu_set_lambda(work->device, flat_of_vecs(work->slambda, Np, &stage)));
    } else {
        /* one call: only what changed since the last solve is uploaded (no synchronisation) */
        DEV_CALL(tqgpu_set_problem(work->device, A, B, b, Qd, Rd, q, r, xmin, xmax, umin, umax, flat_of_vecs(work->slambda, Np, &stage)));
    }

    tqgpu_opts dopts;
    dopts.maxIter = opts->maxIter;
    dopts.termCondition = (int)opts->termCondition;
    dopts.stationarityTolerance = opts->stationarityTolerance;
    dopts.regType = (int)opts->regType;
    dopts.regTol = opts->regTol;
    dopts.regValue = opts->regValue;
    dopts.lineSearchMaxIter = opts->lineSearchMaxIter;
    dopts.lineSearchGamma = opts->lineSearchGamma;
    dopts.lineSearchBeta = opts->lineSearchBeta;
    dopts.lineSearchRestartTrigger = opts->lineSearchRestartTrigger;
    dopts.checkLastActiveSet = opts->checkLastActiveSet;
    const char *penv = getenv("TREEQP_AMD_PROFILE");
    dopts.profile = (penv && atoi(penv) > 0) ? atoi(penv) : 0;      /* 1, 2: per-iteration times; 3: per-phase times as well (profiling.h levels) */

    double interface_time = treeqp_toc(&interface_tmr);
    const double t_upload = interface_time;

    /* --- Newton loop on the device */
    treeqp_tic(&solver_tmr);
    tqgpu_result res;
    if (tqgpu_solve(work->device, &dopts, &res) != TQGPU_OK) {
        /* a failure of the device at run time (a lost device, a launch that cannot be placed) is the solver's failure, not a
         * configuration error of the caller: reported, qp_out untouched -- the caller decides (an MPC loop may hold its last input) */
        fprintf(stderr, "[TREEQP] treeqp_tdunes_solve: device solve failed: %s\n", tqgpu_last_error());
        return TREEQP_UNKNOWN_ERROR;
    }
    const double solver_time = treeqp_toc(&solver_tmr);

    work->lsIter = res.ls_last;
    work->lsTotal = res.ls_total;
    status = (return_t)res.status;
    /* like the reference, an early error return leaves qp_out untouched (:1177,1215) */
    if (status == TREEQP_DN_NOT_DESCENT_DIRECTION) return status;

    /* --- export (:1235-1247) */
    treeqp_tic(&interface_tmr);
    double *ox = flat_out(qp_out->x, Nn, qp_in->nx), *ou = flat_out(qp_out->u, Nn, qp_in->nu), *ol = flat_out(qp_out->lam, Nn - 1, NULL);
    double *omx = flat_out(qp_out->mu_x, Nn, qp_in->nx), *omu = flat_out(qp_out->mu_u, Nn, qp_in->nu);
    double *wx = flat_out(work->sx, Nn, NULL), *wl = flat_out(work->slambda, Np, NULL), *wd = flat_out(work->sDeltalambda, Np, NULL);
    int sum_nx = 0, sum_nu = 0, sum_lam = 0;
    tqgpu_dims(work->device, &sum_nx, &sum_nu, &sum_lam, NULL, NULL);
    /* download once into the workspace mirrors (always contiguous), then fan out */
    double *wu = flat_out(work->su, Nn, NULL);
    double *tmp_mx = work->stage, *tmp_mu = work->stage + sum_nx;
    DEV_CALL(tqgpu_get_solution(work->device, wx, wu, wl, omx ? omx : tmp_mx, omu ? omu : tmp_mu, wd));
    if (ox) { if (sum_nx) memcpy(ox, wx, sizeof(double) * (size_t)sum_nx); } else scatter(wx, qp_out->x, Nn, qp_in->nx);
    if (ou) { if (sum_nu) memcpy(ou, wu, sizeof(double) * (size_t)sum_nu); } else scatter(wu, qp_out->u, Nn, qp_in->nu);
    if (ol) { if (sum_lam) memcpy(ol, wl, sizeof(double) * (size_t)sum_lam); } else scatter(wl, qp_out->lam, Nn - 1, NULL);
    if (!omx) scatter(tmp_mx, qp_out->mu_x, Nn, qp_in->nx);
    if (!omu) scatter(tmp_mu, qp_out->mu_u, Nn, qp_in->nu);

    qp_out->info.iter = res.iter;
    qp_out->info.solver_time = solver_time;
    qp_out->info.interface_time = interface_time + treeqp_toc(&interface_tmr);
    if (getenv("TREEQP_AMD_HOSTPROF"))
        fprintf(stderr, "[treeqp_amd] solve: staging+upload %.1f us, device solve %.1f us, download+export %.1f us\n",
                1e6 * t_upload, 1e6 * solver_time, 1e6 * (qp_out->info.interface_time - t_upload));
    if (res.iter == opts->maxIter) status = TREEQP_MAXIMUM_ITERATIONS_REACHED;
    qp_out->info.total_time = treeqp_toc(&total_tmr);

    /* profiling record (profiling.h): per-iteration data comes from the device log */
    timings->total_time = qp_out->info.total_time;
    for (int i = 0; i < timings->num_iter; i++) { timings->ls_iters[i] = 0; timings->iter_times[i] = NAN; }
    tqgpu_get_iteration_log(work->device, timings->ls_iters, timings->iter_times, timings->num_iter);
    for (int i = 0; i < timings->num_iter; i++)
        timings->stage_qps_times[i] = timings->build_dual_times[i] = timings->newton_direction_times[i] = timings->line_search_times[i] = NAN;
    if (dopts.profile >= 3)
        tqgpu_get_phase_log(work->device, timings->stage_qps_times, timings->build_dual_times, timings->newton_direction_times,
                            timings->line_search_times, timings->num_iter);
    timers_update(timings);

    return status;
}

/* dual_Newton_tree.c:1023-1073: dumps next to the spring-mass example data, same file names
 * (including the reference's quirk of writing lambda into deltalambda_opt.txt) */
void write_solution_to_txt(const tree_qp_in *qp_in, int Np, int iter, struct node *tree,
    treeqp_tdunes_workspace *work)
{
    (void)tree;
    const int Nn = qp_in->N;
    const int dimx = total_number_of_states(qp_in), dimu = total_number_of_controls(qp_in);
    const int dimlam = dimx - qp_in->nx[0];
    double *buf = malloc(sizeof(double) * (size_t)(dimx + dimu + dimlam + 1));
    convert_strvecs_to_single_vec(Nn, work->sx, buf);
    write_double_vector_to_txt(buf, dimx, "examples/spring_mass_utils/x_opt.txt");
    convert_strvecs_to_single_vec(Nn, work->su, buf);
    write_double_vector_to_txt(buf, dimu, "examples/spring_mass_utils/u_opt.txt");
    convert_strvecs_to_single_vec(Np, work->slambda, buf);
    write_double_vector_to_txt(buf, dimlam, "examples/spring_mass_utils/deltalambda_opt.txt");
    write_double_vector_to_txt(buf, dimlam, "examples/spring_mass_utils/lambda_opt.txt");
    write_int_vector_to_txt(&iter, 1, "examples/spring_mass_utils/iter.txt");
    free(buf);
}

/* struct sizes, so that foreign-function bindings (treeqp_amd/capi.py) can verify their
 * declarations against the compiled library */
int treeqp_amd_sizeof(int which)
{
    switch (which) {
        case 0: return (int)sizeof(struct blasfeo_dmat);
        case 1: return (int)sizeof(struct blasfeo_dvec);
        case 2: return (int)sizeof(struct node);
        case 3: return (int)sizeof(tree_qp_in);
        case 4: return (int)sizeof(tree_qp_out);
        case 5: return (int)sizeof(treeqp_tdunes_opts_t);
        case 6: return (int)sizeof(treeqp_tdunes_workspace);
        case 7: return (int)sizeof(treeqp_profiling_t);
        case 8: return (int)sizeof(qp_internal_t);
        default: return -1;
    }
}
