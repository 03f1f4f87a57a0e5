/*
 * qp_container.c -- tree_qp_in / tree_qp_out of the treeQP C API for the treeqp_amd build.
 *
 * API and observable behaviour follow the reference's treeqp/src/tree_qp_common.c (function by
 * function citations below).  The storage is different by design: every *kind* of datum lives
 * in its own contiguous region of the caller's buffer, concatenated in node / edge order
 *
 *     [A_1 A_2 ...][B_1 B_2 ...][b_1 b_2 ...][Q_0 Q_1 ...] ... [xmin_0 xmin_1 ...] ...
 *
 * and the per-node `struct blasfeo_dmat/dvec` entries are views into those regions.  This is
 * exactly the flat "ltv" layout the device C-ABI (include/treeqp_amd.h) consumes, so a solve
 * stages the dynamics, linear terms and bounds to HBM straight from the container without
 * re-packing (see tdunes_host.c).  Eliminating x0 only shortens the views of node 0 / the root
 * edges; the regions themselves never move.
 */
#include "treeqp/src/tree_qp_common.h"
#include "treeqp/utils/blasfeo.h"
#include "treeqp/utils/memory.h"
#include "treeqp/utils/tree.h"
#include "treeqp/utils/utils.h"

#include <blasfeo_d_aux.h>
#include <blasfeo_d_aux_ext_dep.h>
#include <blasfeo_d_blas.h>

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* parent of node idx from the children counts alone (tree not built yet) */
static int parent_from_nk(int idx, const int *nk)
{
    if (idx == 0) return -1;
    int last_child = 0;
    for (int p = 0; p < idx; p++) {
        last_child += nk[p];
        if (last_child >= idx) return p;
    }
    return -1;
}

static int nc_of(const int *nc, int k) { return nc ? nc[k] : 0; }

/* ------------------------------------------------------------------------------------- */
/* sizes                                                                                 */
/* ------------------------------------------------------------------------------------- */

#define SUM_OVER_NODES(expr) do { int acc_ = 0; for (int k = 0; k < qp_in->N; k++) acc_ += (expr); return acc_; } while (0)
#define MAX_OVER_NODES(expr) do { int acc_ = 0; for (int k = 0; k < qp_in->N; k++) if ((expr) > acc_) acc_ = (expr); return acc_; } while (0)

int total_number_of_states(const tree_qp_in *const qp_in) { SUM_OVER_NODES(qp_in->nx[k]); }
int max_number_of_states(const tree_qp_in *const qp_in) { MAX_OVER_NODES(qp_in->nx[k]); }
int total_number_of_controls(const tree_qp_in *const qp_in) { SUM_OVER_NODES(qp_in->nu[k]); }
int max_number_of_controls(const tree_qp_in *const qp_in) { MAX_OVER_NODES(qp_in->nu[k]); }
int total_number_of_general_constraints(const tree_qp_in *const qp_in) { SUM_OVER_NODES(qp_in->nc[k]); }
int max_number_of_general_constraints(const tree_qp_in *const qp_in) { MAX_OVER_NODES(qp_in->nc[k]); }
int total_number_of_primal_variables(const tree_qp_in *const qp_in) { SUM_OVER_NODES(qp_in->nx[k] + qp_in->nu[k]); }
int total_number_of_dynamic_constraints(const tree_qp_in *const qp_in) { SUM_OVER_NODES(k > 0 ? qp_in->nx[k] : 0); }

/* doubles needed by the value regions + the x0-elimination copies */
static size_t qp_in_value_doubles(int Nn, const int *nx, const int *nu, const int *nc, const int *nk)
{
    size_t d = 0;
    for (int k = 0; k < Nn; k++) {
        const int c = nc_of(nc, k);
        if (k > 0) {
            const int p = parent_from_nk(k, nk);
            d += (size_t)nx[k] * nx[p] + (size_t)nx[k] * nu[p] + nx[k];          /* A, B, b */
            if (k <= nk[0]) d += (size_t)nx[k] * nx[p] + nx[k];                   /* A0, b0 */
        }
        d += (size_t)nx[k] * nx[k] + (size_t)nu[k] * nu[k] + (size_t)nu[k] * nx[k];   /* Q, R, S */
        d += 3 * (size_t)nx[k] + 3 * (size_t)nu[k];                               /* q,xmin,xmax / r,umin,umax */
        d += (size_t)c * nx[k] + (size_t)c * nu[k] + 2 * (size_t)c;               /* C, D, dmin, dmax */
    }
    const int c0 = nc_of(nc, 0);
    d += nx[0] + (size_t)c0 * nx[0] + 2 * (size_t)c0 + (size_t)nu[0] * nx[0] + nu[0];   /* x0, C0, dmin0, dmax0, S0, r0 */
    return d;
}

/* tree_qp_common.c:60-144 */
int tree_qp_in_calculate_size(int Nn, const int *nx, const int *nu, const int *nc, const int *nk)
{
    size_t bytes = 0;
    bytes += (size_t)Nn * sizeof(struct node) + (size_t)tree_calculate_size(nk);
    bytes += 3 * (size_t)Nn * sizeof(int) + 2 * (size_t)nk[0] * sizeof(int);
    bytes += (size_t)(2 * (Nn - 1) + 5 * Nn + nk[0]) * sizeof(struct blasfeo_dmat);   /* A,B | Q,R,S,C,D | A0 */
    bytes += (size_t)((Nn - 1) + 8 * Nn + nk[0]) * sizeof(struct blasfeo_dvec);       /* b | q,r,4 bounds,dmin,dmax | b0 */
    bytes += qp_in_value_doubles(Nn, nx, nu, nc, nk) * sizeof(double);
    int ib = (int)bytes;
    make_int_multiple_of(64, &ib);
    return ib + 2 * 64;
}

/* carve `count` view structs */
#define TAKE(type, count) ((type *)take_bytes(&c_ptr, (size_t)(count) * sizeof(type)))
static void *take_bytes(char **c_ptr, size_t n) { void *p = *c_ptr; *c_ptr += n; return p; }

/* tree_qp_common.c:148-306 */
void tree_qp_in_create(int Nn, const int *nx, const int *nu, const int *nc, const int *nk,
    tree_qp_in *qp_in, void *ptr)
{
    char *c_ptr = (char *)ptr;
    qp_internal_t *im = &qp_in->internal_memory;

    qp_in->N = Nn;
    qp_in->tree = TAKE(struct node, Nn);
    tree_create(nk, qp_in->tree, c_ptr);
    c_ptr += tree_calculate_size(nk);
    assert(Nn == number_of_nodes_from_tree(qp_in->tree) && "Detected number of nodes different than given one");

    const int nk0 = qp_in->tree[0].nkids;
    qp_in->nx = TAKE(int, Nn); qp_in->nu = TAKE(int, Nn); qp_in->nc = TAKE(int, Nn);
    im->is_A_initialized = TAKE(int, nk0); im->is_b_initialized = TAKE(int, nk0);
    for (int k = 0; k < Nn; k++) { qp_in->nx[k] = nx[k]; qp_in->nu[k] = nu[k]; qp_in->nc[k] = nc_of(nc, k); }
    for (int k = 0; k < nk0; k++) im->is_A_initialized[k] = im->is_b_initialized[k] = 0;
    im->is_C_initialized = im->is_dmin_initialized = im->is_dmax_initialized = 0;
    im->is_S_initialized = im->is_r_initialized = 0;

    align_char_to(8, &c_ptr);
    qp_in->A = TAKE(struct blasfeo_dmat, Nn - 1); qp_in->B = TAKE(struct blasfeo_dmat, Nn - 1);
    qp_in->Q = TAKE(struct blasfeo_dmat, Nn); qp_in->R = TAKE(struct blasfeo_dmat, Nn);
    qp_in->S = TAKE(struct blasfeo_dmat, Nn); qp_in->C = TAKE(struct blasfeo_dmat, Nn);
    qp_in->D = TAKE(struct blasfeo_dmat, Nn); im->A0 = TAKE(struct blasfeo_dmat, nk0);
    qp_in->b = TAKE(struct blasfeo_dvec, Nn - 1);
    qp_in->q = TAKE(struct blasfeo_dvec, Nn); qp_in->r = TAKE(struct blasfeo_dvec, Nn);
    qp_in->xmin = TAKE(struct blasfeo_dvec, Nn); qp_in->xmax = TAKE(struct blasfeo_dvec, Nn);
    qp_in->umin = TAKE(struct blasfeo_dvec, Nn); qp_in->umax = TAKE(struct blasfeo_dvec, Nn);
    qp_in->dmin = TAKE(struct blasfeo_dvec, Nn); qp_in->dmax = TAKE(struct blasfeo_dvec, Nn);
    im->b0 = TAKE(struct blasfeo_dvec, nk0);

    align_char_to(64, &c_ptr);

    /* value regions, one kind after the other (each region is node/edge-order contiguous) */
    const struct node *tree = qp_in->tree;
    const int *ncq = qp_in->nc;
    for (int k = 1; k < Nn; k++) init_strmat(nx[k], nx[tree[k].dad], &qp_in->A[k - 1], &c_ptr);
    for (int k = 1; k < Nn; k++) init_strmat(nx[k], nu[tree[k].dad], &qp_in->B[k - 1], &c_ptr);
    for (int k = 1; k < Nn; k++) init_strvec(nx[k], &qp_in->b[k - 1], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strmat(nx[k], nx[k], &qp_in->Q[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strmat(nu[k], nu[k], &qp_in->R[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strmat(nu[k], nx[k], &qp_in->S[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nx[k], &qp_in->q[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nu[k], &qp_in->r[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nx[k], &qp_in->xmin[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nx[k], &qp_in->xmax[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nu[k], &qp_in->umin[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nu[k], &qp_in->umax[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strmat(ncq[k], nx[k], &qp_in->C[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strmat(ncq[k], nu[k], &qp_in->D[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(ncq[k], &qp_in->dmin[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(ncq[k], &qp_in->dmax[k], &c_ptr);
    /* copies used to re-derive the root-coupled data when x0 changes after elimination */
    for (int k = 1; k <= nk0; k++) init_strmat(nx[k], nx[0], &im->A0[k - 1], &c_ptr);
    for (int k = 1; k <= nk0; k++) init_strvec(nx[k], &im->b0[k - 1], &c_ptr);
    init_strvec(nx[0], &im->x0, &c_ptr);
    init_strmat(ncq[0], nx[0], &im->C0, &c_ptr);
    init_strvec(ncq[0], &im->dmin0, &c_ptr);
    init_strvec(ncq[0], &im->dmax0, &c_ptr);
    init_strmat(nu[0], nx[0], &im->S0, &c_ptr);
    init_strvec(nu[0], &im->r0, &c_ptr);

    tree_qp_in_set_inf_bounds(qp_in);

    assert((char *)ptr + tree_qp_in_calculate_size(Nn, nx, nu, nc, nk) >= c_ptr);
}

/* tree_qp_common.c:310-344 */
int tree_qp_out_calculate_size(const int Nn, const int *const nx, const int *const nu, const int *const nc)
{
    size_t bytes = (size_t)(6 * Nn - 1) * sizeof(struct blasfeo_dvec);
    for (int k = 0; k < Nn; k++)
        bytes += sizeof(double) * (size_t)(2 * nx[k] + 2 * nu[k] + nc_of(nc, k) + (k > 0 ? nx[k] : 0));
    int ib = (int)bytes;
    make_int_multiple_of(64, &ib);
    return ib + 2 * 64;
}

/* tree_qp_common.c:348-400; regions: x | u | lam | mu_x | mu_u | mu_d */
void tree_qp_out_create(const int Nn, const int *const nx, const int *const nu, const int *const nc,
    tree_qp_out *const qp_out, void *ptr)
{
    char *c_ptr = (char *)ptr;
    qp_out->x = TAKE(struct blasfeo_dvec, Nn); qp_out->u = TAKE(struct blasfeo_dvec, Nn);
    qp_out->mu_x = TAKE(struct blasfeo_dvec, Nn); qp_out->mu_u = TAKE(struct blasfeo_dvec, Nn);
    qp_out->mu_d = TAKE(struct blasfeo_dvec, Nn); qp_out->lam = TAKE(struct blasfeo_dvec, Nn - 1);
    align_char_to(64, &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nx[k], &qp_out->x[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nu[k], &qp_out->u[k], &c_ptr);
    for (int k = 1; k < Nn; k++) init_strvec(nx[k], &qp_out->lam[k - 1], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nx[k], &qp_out->mu_x[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nu[k], &qp_out->mu_u[k], &c_ptr);
    for (int k = 0; k < Nn; k++) init_strvec(nc_of(nc, k), &qp_out->mu_d[k], &c_ptr);
    qp_out->info.Nn = Nn;
    qp_out->info.iter = 0;
    qp_out->info.total_time = qp_out->info.solver_time = qp_out->info.interface_time = 0.0;
    assert((char *)ptr + tree_qp_out_calculate_size(Nn, nx, nu, nc) >= c_ptr);
}
#undef TAKE

/* ------------------------------------------------------------------------------------- */
/* x0 elimination (tree_qp_common.c:404-536, 2154-2235)                                  */
/* ------------------------------------------------------------------------------------- */

static void empty_mat(struct blasfeo_dmat *M, int keep_rows) { M->pA = NULL; M->n = 0; if (!keep_rows) M->m = 0; }
static void empty_vec(struct blasfeo_dvec *v) { v->pa = NULL; v->m = 0; }

void tree_qp_in_eliminate_x0(tree_qp_in *const qp_in)
{
    if (qp_in->nx[0] == 0) return;
    qp_internal_t *im = &qp_in->internal_memory;
    const int nc0 = qp_in->nc[0];
    const int nk0 = qp_in->tree[0].nkids;

    /* keep originals of everything that multiplies x0 */
    if (nc0 > 0 && !im->is_C_initialized) {
        blasfeo_dgecp(qp_in->C[0].m, qp_in->C[0].n, &qp_in->C[0], 0, 0, &im->C0, 0, 0);
        im->is_C_initialized = 1;
    }
    if (nc0 > 0 && !im->is_dmin_initialized) blasfeo_dveccp(qp_in->dmin[0].m, &qp_in->dmin[0], 0, &im->dmin0, 0);
    if (nc0 > 0 && !im->is_dmax_initialized) blasfeo_dveccp(qp_in->dmax[0].m, &qp_in->dmax[0], 0, &im->dmax0, 0);
    if (!im->is_S_initialized) {
        blasfeo_dgecp(qp_in->S[0].m, qp_in->S[0].n, &qp_in->S[0], 0, 0, &im->S0, 0, 0);
        im->is_S_initialized = 1;
    }
    if (!im->is_r_initialized) {
        blasfeo_dveccp(qp_in->r[0].m, &qp_in->r[0], 0, &im->r0, 0);
        im->is_r_initialized = 1;
    }
    for (int e = 0; e < nk0; e++) {
        if (!im->is_A_initialized[e]) {
            blasfeo_dgecp(qp_in->A[e].m, qp_in->A[e].n, &qp_in->A[e], 0, 0, &im->A0[e], 0, 0);
            im->is_A_initialized[e] = 1;
        }
        if (!im->is_b_initialized[e]) {
            blasfeo_dveccp(qp_in->b[e].m, &qp_in->b[e], 0, &im->b0[e], 0);
            im->is_b_initialized[e] = 1;
        }
        empty_mat(&qp_in->A[e], 1);
    }
    empty_mat(&qp_in->C[0], 1);
    empty_mat(&qp_in->S[0], 1);

    /* x0 must be pinned by equal bounds */
    assert(check_error_strvec(&qp_in->xmin[0], &qp_in->xmax[0]) < 1e-10);

    qp_in->nx[0] = 0;
    tree_qp_in_set_x0_strvec(qp_in, &qp_in->xmin[0]);     /* folds x0 into b, r, dmin, dmax */

    empty_mat(&qp_in->Q[0], 0);
    empty_vec(&qp_in->q[0]);
    empty_vec(&qp_in->xmin[0]);
    empty_vec(&qp_in->xmax[0]);
}

void tree_qp_out_eliminate_x0(tree_qp_out *const qp_out)
{
    empty_vec(&qp_out->x[0]);
    empty_vec(&qp_out->mu_x[0]);
}

void tree_qp_in_set_x0_strvec(tree_qp_in *qp_in, struct blasfeo_dvec *sx0)
{
    qp_internal_t *im = &qp_in->internal_memory;
    if (qp_in->nx[0] > 0) {             /* x0 still a variable: pin it through its bounds */
        blasfeo_dveccp(sx0->m, sx0, 0, &qp_in->xmin[0], 0);
        blasfeo_dveccp(sx0->m, sx0, 0, &qp_in->xmax[0], 0);
        return;
    }
    const int nx0 = sx0->m, nc0 = im->C0.m, nu0 = im->S0.m;
    assert(im->x0.m == nx0 && qp_in->nu[0] == nu0 && qp_in->nc[0] == nc0);
    if (im->x0.pa != sx0->pa) blasfeo_dveccp(nx0, sx0, 0, &im->x0, 0);
    for (int e = 0; e < qp_in->tree[0].nkids; e++) {
        assert(im->is_A_initialized[e] == 1 && im->is_b_initialized[e] == 1);
        /* b_e = b0_e + A0_e x0 */
        blasfeo_dgemv_n(im->A0[e].m, nx0, 1.0, &im->A0[e], 0, 0, sx0, 0, 1.0, &im->b0[e], 0, &qp_in->b[e], 0);
    }
    if (nc0 > 0) {
        assert(im->is_C_initialized == 1);
        blasfeo_dgemv_n(nc0, nx0, -1.0, &im->C0, 0, 0, sx0, 0, 1.0, &im->dmin0, 0, &qp_in->dmin[0], 0);
        blasfeo_dgemv_n(nc0, nx0, -1.0, &im->C0, 0, 0, sx0, 0, 1.0, &im->dmax0, 0, &qp_in->dmax[0], 0);
    }
    assert(im->is_S_initialized == 1 && im->is_r_initialized == 1);
    /* r_0 = r0 + S0 x0 */
    blasfeo_dgemv_n(nu0, nx0, 1.0, &im->S0, 0, 0, sx0, 0, 1.0, &im->r0, 0, &qp_in->r[0], 0);
}

void tree_qp_in_set_x0_colmaj(tree_qp_in *qp_in, double *x0)
{
    struct blasfeo_dvec *sx0 = &qp_in->internal_memory.x0;
    blasfeo_pack_dvec(sx0->m, x0, sx0, 0);
    tree_qp_in_set_x0_strvec(qp_in, sx0);
}

/* ------------------------------------------------------------------------------------- */
/* KKT residual (tree_qp_common.c:540-788)                                               */
/* order of entries per node: stationarity x,u | dynamics | bound feas. x,u | compl. x,u |
 * general feas. | general compl.                                                        */
/* ------------------------------------------------------------------------------------- */

static double bound_violation(double v, double lo, double hi) { return v > hi ? v - hi : (v < lo ? lo - v : 0.0); }
static double complementarity(double mu, double v, double lo, double hi) { return mu > 0 ? mu * (v - hi) : mu * (lo - v); }

void tree_qp_out_calculate_KKT_res(const tree_qp_in *const qp_in, const tree_qp_out *const qp_out, double *res)
{
    const int Nn = qp_in->N;
    const int *nx = qp_in->nx, *nu = qp_in->nu, *nc = qp_in->nc;
    const struct node *tree = qp_in->tree;
    const int nKKT = 3 * total_number_of_primal_variables(qp_in) + total_number_of_dynamic_constraints(qp_in)
        + 2 * total_number_of_general_constraints(qp_in);
    for (int i = 0; i < nKKT; i++) res[i] = 1e12;

    struct blasfeo_dvec tx, tu, tg;
    blasfeo_allocate_dvec(max_number_of_states(qp_in), &tx);
    blasfeo_allocate_dvec(max_number_of_controls(qp_in), &tu);
    blasfeo_allocate_dvec(max_number_of_general_constraints(qp_in), &tg);
    struct blasfeo_dvec *x = qp_out->x, *u = qp_out->u, *lam = qp_out->lam;

    int pos = 0;
    for (int k = 0; k < Nn; k++) {
        /* stationarity: Qx + q + S'u + mu_x + C'mu_d - lam_k + sum_kids A'lam_kid  (:589-625) */
        blasfeo_dgemv_n(nx[k], nx[k], 1.0, &qp_in->Q[k], 0, 0, &x[k], 0, 1.0, &qp_in->q[k], 0, &tx, 0);
        blasfeo_dgemv_t(nu[k], nx[k], 1.0, &qp_in->S[k], 0, 0, &u[k], 0, 1.0, &tx, 0, &tx, 0);
        blasfeo_daxpy(nx[k], 1.0, &qp_out->mu_x[k], 0, &tx, 0, &tx, 0);
        blasfeo_dgemv_t(nc[k], nx[k], 1.0, &qp_in->C[k], 0, 0, &qp_out->mu_d[k], 0, 1.0, &tx, 0, &tx, 0);
        if (k > 0) blasfeo_daxpy(nx[k], -1.0, &lam[k - 1], 0, &tx, 0, &tx, 0);
        blasfeo_dgemv_n(nu[k], nu[k], 1.0, &qp_in->R[k], 0, 0, &u[k], 0, 1.0, &qp_in->r[k], 0, &tu, 0);
        blasfeo_dgemv_n(nu[k], nx[k], 1.0, &qp_in->S[k], 0, 0, &x[k], 0, 1.0, &tu, 0, &tu, 0);
        blasfeo_daxpy(nu[k], 1.0, &qp_out->mu_u[k], 0, &tu, 0, &tu, 0);
        blasfeo_dgemv_t(nc[k], nu[k], 1.0, &qp_in->D[k], 0, 0, &qp_out->mu_d[k], 0, 1.0, &tu, 0, &tu, 0);
        for (int c = 0; c < tree[k].nkids; c++) {
            const int kid = tree[k].kids[c];
            blasfeo_dgemv_t(nx[kid], nx[k], 1.0, &qp_in->A[kid - 1], 0, 0, &lam[kid - 1], 0, 1.0, &tx, 0, &tx, 0);
            blasfeo_dgemv_t(nx[kid], nu[k], 1.0, &qp_in->B[kid - 1], 0, 0, &lam[kid - 1], 0, 1.0, &tu, 0, &tu, 0);
        }
        blasfeo_unpack_dvec(nx[k], &tx, 0, &res[pos]); pos += nx[k];
        blasfeo_unpack_dvec(nu[k], &tu, 0, &res[pos]); pos += nu[k];

        /* dynamics (:629-646) */
        if (k > 0) {
            const int p = tree[k].dad;
            blasfeo_dgemv_n(nx[k], nx[p], 1.0, &qp_in->A[k - 1], 0, 0, &x[p], 0, 1.0, &qp_in->b[k - 1], 0, &tx, 0);
            blasfeo_dgemv_n(nx[k], nu[p], 1.0, &qp_in->B[k - 1], 0, 0, &u[p], 0, 1.0, &tx, 0, &tx, 0);
            blasfeo_daxpy(nx[k], -1.0, &x[k], 0, &tx, 0, &tx, 0);
            blasfeo_unpack_dvec(nx[k], &tx, 0, &res[pos]); pos += nx[k];
        }
        /* bounds: feasibility (:651-683) then complementarity (:688-714) */
        for (int j = 0; j < nx[k]; j++)
            res[pos + j] = bound_violation(BLASFEO_DVECEL(&x[k], j), BLASFEO_DVECEL(&qp_in->xmin[k], j), BLASFEO_DVECEL(&qp_in->xmax[k], j));
        pos += nx[k];
        for (int j = 0; j < nu[k]; j++)
            res[pos + j] = bound_violation(BLASFEO_DVECEL(&u[k], j), BLASFEO_DVECEL(&qp_in->umin[k], j), BLASFEO_DVECEL(&qp_in->umax[k], j));
        pos += nu[k];
        for (int j = 0; j < nx[k]; j++)
            res[pos + j] = complementarity(BLASFEO_DVECEL(&qp_out->mu_x[k], j), BLASFEO_DVECEL(&x[k], j), BLASFEO_DVECEL(&qp_in->xmin[k], j), BLASFEO_DVECEL(&qp_in->xmax[k], j));
        pos += nx[k];
        for (int j = 0; j < nu[k]; j++)
            res[pos + j] = complementarity(BLASFEO_DVECEL(&qp_out->mu_u[k], j), BLASFEO_DVECEL(&u[k], j), BLASFEO_DVECEL(&qp_in->umin[k], j), BLASFEO_DVECEL(&qp_in->umax[k], j));
        pos += nu[k];
        /* general constraints (:719-756) */
        blasfeo_dgemv_n(nc[k], nx[k], 1.0, &qp_in->C[k], 0, 0, &x[k], 0, 0.0, &tg, 0, &tg, 0);
        blasfeo_dgemv_n(nc[k], nu[k], 1.0, &qp_in->D[k], 0, 0, &u[k], 0, 1.0, &tg, 0, &tg, 0);
        for (int j = 0; j < nc[k]; j++)
            res[pos + j] = bound_violation(BLASFEO_DVECEL(&tg, j), BLASFEO_DVECEL(&qp_in->dmin[k], j), BLASFEO_DVECEL(&qp_in->dmax[k], j));
        pos += nc[k];
        for (int j = 0; j < nc[k]; j++)
            res[pos + j] = complementarity(BLASFEO_DVECEL(&qp_out->mu_d[k], j), BLASFEO_DVECEL(&tg, j), BLASFEO_DVECEL(&qp_in->dmin[k], j), BLASFEO_DVECEL(&qp_in->dmax[k], j));
        pos += nc[k];
    }
    blasfeo_free_dvec(&tx); blasfeo_free_dvec(&tu); blasfeo_free_dvec(&tg);
    assert(nKKT == pos && "incorrect size of KKT residuals");
}

double tree_qp_out_max_KKT_res(const tree_qp_in *const qp_in, const tree_qp_out *const qp_out)
{
    const int nKKT = 3 * total_number_of_primal_variables(qp_in) + total_number_of_dynamic_constraints(qp_in)
        + 2 * total_number_of_general_constraints(qp_in);
    double *res = malloc(sizeof(double) * (size_t)(nKKT > 0 ? nKKT : 1));
    tree_qp_out_calculate_KKT_res(qp_in, qp_out, res);
    double worst = 0.0;
    for (int i = 0; i < nKKT; i++) { double a = fabs(res[i]); if (a > worst || a != a) worst = a; }
    free(res);
    return worst;
}

/* ------------------------------------------------------------------------------------- */
/* accessors                                                                             */
/* ------------------------------------------------------------------------------------- */

static int tight(int lda, int rows) { return lda <= 0 ? rows : lda; }

/* after a root-coupled datum was (re)written, refresh its pre-elimination copy
 * (tree_qp_common.c:899-911, 1001-1011, 1196-1206, 1274-1283, 1620-1627, 1736-1741) */
static void sync_root_copy_A(tree_qp_in *qp_in, int e)
{
    if (qp_in->tree[e + 1].dad == 0 && qp_in->nx[0] > 0) {
        blasfeo_dgecp(qp_in->A[e].m, qp_in->internal_memory.A0[e].n, &qp_in->A[e], 0, 0, &qp_in->internal_memory.A0[e], 0, 0);
        qp_in->internal_memory.is_A_initialized[e] = 1;
    }
}
static void sync_root_copy_b(tree_qp_in *qp_in, int e)
{
    if (qp_in->tree[e + 1].dad == 0 && qp_in->nx[0] > 0) {
        blasfeo_dveccp(qp_in->b[e].m, &qp_in->b[e], 0, &qp_in->internal_memory.b0[e], 0);
        qp_in->internal_memory.is_b_initialized[e] = 1;
    }
}

#define MAT_SETGET(KIND, NAME, LIMIT, ROWS, COLS, FIELD, AFTER_SET)                                                    \
    void tree_qp_in_set_##KIND##_##NAME##_colmajor(const double *const NAME, const int lda, tree_qp_in *const qp_in, const int indx) \
    {                                                                                                                  \
        assert(indx >= 0 && indx < (LIMIT));                                                                           \
        const int rows = (ROWS), cols = (COLS);                                                                        \
        blasfeo_pack_dmat(rows, cols, (double *)NAME, tight(lda, rows), &qp_in->FIELD[indx], 0, 0);                    \
        AFTER_SET;                                                                                                     \
    }                                                                                                                  \
    void tree_qp_in_get_##KIND##_##NAME##_colmajor(double *const NAME, const int lda, const tree_qp_in *const qp_in, const int indx) \
    {                                                                                                                  \
        assert(indx >= 0 && indx < (LIMIT));                                                                           \
        const int rows = (ROWS), cols = (COLS);                                                                        \
        blasfeo_unpack_dmat(rows, cols, &qp_in->FIELD[indx], 0, 0, NAME, tight(lda, rows));                            \
    }

#define VEC_SETGET(PFX, OBJ, KIND, NAME, LIMIT, LEN, FIELD, AFTER_SET)                                                 \
    void PFX##_set_##KIND##_##NAME(const double *const NAME, OBJ *const qp_in, const int indx)                         \
    {                                                                                                                  \
        assert(indx >= 0 && indx < (LIMIT));                                                                           \
        blasfeo_pack_dvec((LEN), (double *)NAME, &qp_in->FIELD[indx], 0);                                              \
        AFTER_SET;                                                                                                     \
    }                                                                                                                  \
    void PFX##_get_##KIND##_##NAME(double *const NAME, const OBJ *const qp_in, const int indx)                         \
    {                                                                                                                  \
        assert(indx >= 0 && indx < (LIMIT));                                                                           \
        blasfeo_unpack_dvec((LEN), (struct blasfeo_dvec *)&qp_in->FIELD[indx], 0, NAME);                               \
    }

#define DAD(e) (qp_in->tree[(e) + 1].dad)
#define NOTHING ((void)0)

MAT_SETGET(edge, A, qp_in->N - 1, qp_in->nx[indx + 1], qp_in->nx[DAD(indx)], A, sync_root_copy_A(qp_in, indx))
MAT_SETGET(edge, B, qp_in->N - 1, qp_in->nx[indx + 1], qp_in->nu[DAD(indx)], B, NOTHING)
VEC_SETGET(tree_qp_in, tree_qp_in, edge, b, qp_in->N - 1, qp_in->nx[indx + 1], b, sync_root_copy_b(qp_in, indx))
MAT_SETGET(node, Q, qp_in->N, qp_in->nx[indx], qp_in->nx[indx], Q, NOTHING)
MAT_SETGET(node, R, qp_in->N, qp_in->nu[indx], qp_in->nu[indx], R, NOTHING)
MAT_SETGET(node, S, qp_in->N, qp_in->nu[indx], qp_in->nx[indx], S,
    if (indx == 0 && qp_in->nx[0] > 0) {
        blasfeo_dgecp(qp_in->nu[0], qp_in->nx[0], &qp_in->S[0], 0, 0, &qp_in->internal_memory.S0, 0, 0);
        qp_in->internal_memory.is_S_initialized = 1;
    })
VEC_SETGET(tree_qp_in, tree_qp_in, node, q, qp_in->N, qp_in->nx[indx], q, NOTHING)
VEC_SETGET(tree_qp_in, tree_qp_in, node, r, qp_in->N, qp_in->nu[indx], r,
    if (indx == 0 && qp_in->nx[0] > 0) {
        blasfeo_dveccp(qp_in->nu[0], &qp_in->r[0], 0, &qp_in->internal_memory.r0, 0);
        qp_in->internal_memory.is_r_initialized = 1;
    })
VEC_SETGET(tree_qp_in, tree_qp_in, node, xmin, qp_in->N, qp_in->nx[indx], xmin, NOTHING)
VEC_SETGET(tree_qp_in, tree_qp_in, node, xmax, qp_in->N, qp_in->nx[indx], xmax, NOTHING)
VEC_SETGET(tree_qp_in, tree_qp_in, node, umin, qp_in->N, qp_in->nu[indx], umin, NOTHING)
VEC_SETGET(tree_qp_in, tree_qp_in, node, umax, qp_in->N, qp_in->nu[indx], umax, NOTHING)
MAT_SETGET(node, C, qp_in->N, qp_in->nc[indx], qp_in->nx[indx], C,
    if (indx == 0 && qp_in->nx[0] > 0 && qp_in->nc[0] > 0) {
        blasfeo_dgecp(qp_in->nc[0], qp_in->nx[0], &qp_in->C[0], 0, 0, &qp_in->internal_memory.C0, 0, 0);
        qp_in->internal_memory.is_C_initialized = 1;
    })
MAT_SETGET(node, D, qp_in->N, qp_in->nc[indx], qp_in->nu[indx], D, NOTHING)
VEC_SETGET(tree_qp_in, tree_qp_in, node, dmin, qp_in->N, qp_in->nc[indx], dmin,
    if (indx == 0 && qp_in->nx[0] > 0 && qp_in->nc[0] > 0) {
        blasfeo_dveccp(qp_in->nc[0], &qp_in->dmin[0], 0, &qp_in->internal_memory.dmin0, 0);
        qp_in->internal_memory.is_dmin_initialized = 1;
    })
VEC_SETGET(tree_qp_in, tree_qp_in, node, dmax, qp_in->N, qp_in->nc[indx], dmax,
    if (indx == 0 && qp_in->nx[0] > 0 && qp_in->nc[0] > 0) {
        blasfeo_dveccp(qp_in->nc[0], &qp_in->dmax[0], 0, &qp_in->internal_memory.dmax0, 0);
        qp_in->internal_memory.is_dmax_initialized = 1;
    })

/* qp_out vectors carry their own length (they may have been shortened by eliminate_x0) */
VEC_SETGET(tree_qp_out, tree_qp_out, node, x, qp_in->info.Nn, qp_in->x[indx].m, x, NOTHING)
VEC_SETGET(tree_qp_out, tree_qp_out, node, u, qp_in->info.Nn, qp_in->u[indx].m, u, NOTHING)
VEC_SETGET(tree_qp_out, tree_qp_out, edge, lam, qp_in->info.Nn - 1, qp_in->lam[indx].m, lam, NOTHING)
VEC_SETGET(tree_qp_out, tree_qp_out, node, mu_x, qp_in->info.Nn, qp_in->mu_x[indx].m, mu_x, NOTHING)
VEC_SETGET(tree_qp_out, tree_qp_out, node, mu_u, qp_in->info.Nn, qp_in->mu_u[indx].m, mu_u, NOTHING)
VEC_SETGET(tree_qp_out, tree_qp_out, node, mu_d, qp_in->info.Nn, qp_in->mu_d[indx].m, mu_d, NOTHING)

#undef MAT_SETGET
#undef VEC_SETGET
#undef DAD
#undef NOTHING

/* ---- grouped accessors ---- */

void tree_qp_in_set_edge_dynamics_colmajor(const double *const A, const double *const B, const double *const b,
    tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_set_edge_A_colmajor(A, -1, qp_in, indx);
    tree_qp_in_set_edge_B_colmajor(B, -1, qp_in, indx);
    tree_qp_in_set_edge_b(b, qp_in, indx);
}
void tree_qp_in_get_edge_dynamics_colmajor(double *const A, double *const B, double *const b,
    const tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_get_edge_A_colmajor(A, -1, qp_in, indx);
    tree_qp_in_get_edge_B_colmajor(B, -1, qp_in, indx);
    tree_qp_in_get_edge_b(b, qp_in, indx);
}
void tree_qp_in_set_node_objective_colmajor(const double *const Q, const double *const R, const double *const S,
    const double *const q, const double *const r, tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_set_node_Q_colmajor(Q, -1, qp_in, indx);
    tree_qp_in_set_node_R_colmajor(R, -1, qp_in, indx);
    tree_qp_in_set_node_S_colmajor(S, -1, qp_in, indx);
    tree_qp_in_set_node_q(q, qp_in, indx);
    tree_qp_in_set_node_r(r, qp_in, indx);
}
void tree_qp_in_get_node_objective_colmajor(double *const Q, double *const R, double *const S,
    double *const q, double *const r, const tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_get_node_Q_colmajor(Q, -1, qp_in, indx);
    tree_qp_in_get_node_R_colmajor(R, -1, qp_in, indx);
    tree_qp_in_get_node_S_colmajor(S, -1, qp_in, indx);
    tree_qp_in_get_node_q(q, qp_in, indx);
    tree_qp_in_get_node_r(r, qp_in, indx);
}

/* diagonal weights, zero cross term (tree_qp_common.c:1375-1428).  r is routed through the
 * plain setter so that the root's pre-elimination copy stays in sync. */
void tree_qp_in_set_node_objective_diag(const double *const Qd, const double *const Rd,
    const double *const q, const double *const r, tree_qp_in *const qp_in, const int indx)
{
    assert(indx >= 0 && indx < qp_in->N);
    const int nx = qp_in->nx[indx], nu = qp_in->nu[indx];
    if (nx > 0 && nu > 0) blasfeo_dgese(nu, nx, 0.0, &qp_in->S[indx], 0, 0);
    if (nx > 0) {
        blasfeo_dgese(nx, nx, 0.0, &qp_in->Q[indx], 0, 0);
        for (int j = 0; j < nx; j++) BLASFEO_DMATEL(&qp_in->Q[indx], j, j) = Qd[j];
        blasfeo_pack_dvec(nx, (double *)q, &qp_in->q[indx], 0);
    }
    if (nu > 0) {
        blasfeo_dgese(nu, nu, 0.0, &qp_in->R[indx], 0, 0);
        for (int j = 0; j < nu; j++) BLASFEO_DMATEL(&qp_in->R[indx], j, j) = Rd[j];
        blasfeo_pack_dvec(nu, (double *)r, &qp_in->r[indx], 0);
    }
}

void tree_qp_in_set_node_bounds(const double *const xmin, const double *const xmax,
    const double *const umin, const double *const umax, tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_set_node_xmin(xmin, qp_in, indx); tree_qp_in_set_node_xmax(xmax, qp_in, indx);
    tree_qp_in_set_node_umin(umin, qp_in, indx); tree_qp_in_set_node_umax(umax, qp_in, indx);
}
void tree_qp_in_get_node_bounds(double *const xmin, double *const xmax, double *const umin, double *const umax,
    const tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_get_node_xmin(xmin, qp_in, indx); tree_qp_in_get_node_xmax(xmax, qp_in, indx);
    tree_qp_in_get_node_umin(umin, qp_in, indx); tree_qp_in_get_node_umax(umax, qp_in, indx);
}
void tree_qp_in_set_node_general_constraints(const double *const C, const double *const D,
    const double *const dmin, const double *const dmax, tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_set_node_C_colmajor(C, -1, qp_in, indx); tree_qp_in_set_node_D_colmajor(D, -1, qp_in, indx);
    tree_qp_in_set_node_dmin(dmin, qp_in, indx); tree_qp_in_set_node_dmax(dmax, qp_in, indx);
}
void tree_qp_in_get_node_general_constraints(double *const C, double *const D, double *const dmin, double *const dmax,
    const tree_qp_in *const qp_in, const int indx)
{
    tree_qp_in_get_node_C_colmajor(C, -1, qp_in, indx); tree_qp_in_get_node_D_colmajor(D, -1, qp_in, indx);
    tree_qp_in_get_node_dmin(dmin, qp_in, indx); tree_qp_in_get_node_dmax(dmax, qp_in, indx);
}

/* ---- whole-tree setters (tree_qp_common.c:1952-2150) ---- */

void tree_qp_in_set_ltv_dynamics_colmajor(double *A, double *B, double *b, tree_qp_in *qp_in)
{
    for (int e = 0; e < qp_in->N - 1; e++) {
        tree_qp_in_set_edge_dynamics_colmajor(A, B, b, qp_in, e);
        A += qp_in->A[e].m * qp_in->A[e].n; B += qp_in->B[e].m * qp_in->B[e].n; b += qp_in->b[e].m;
    }
}
void tree_qp_in_set_ltv_objective_diag(double *Qd, double *Rd, double *q, double *r, tree_qp_in *qp_in)
{
    for (int k = 0; k < qp_in->N; k++) {
        tree_qp_in_set_node_objective_diag(Qd, Rd, q, r, qp_in, k);
        Qd += qp_in->Q[k].m; q += qp_in->Q[k].m; Rd += qp_in->R[k].m; r += qp_in->R[k].m;
    }
}
void tree_qp_in_set_ltv_objective_colmajor(double *Q, double *R, double *S, double *q, double *r, tree_qp_in *qp_in)
{
    for (int k = 0; k < qp_in->N; k++) {
        tree_qp_in_set_node_objective_colmajor(Q, R, S, q, r, qp_in, k);
        Q += qp_in->Q[k].m * qp_in->Q[k].n; R += qp_in->R[k].m * qp_in->R[k].n; S += qp_in->S[k].m * qp_in->S[k].n;
        q += qp_in->q[k].m; r += qp_in->r[k].m;
    }
}
void tree_qp_in_set_ltv_bounds(double *xmin, double *xmax, double *umin, double *umax, tree_qp_in *qp_in)
{
    int ix = 0, iu = 0;
    for (int k = 0; k < qp_in->N; k++) {
        tree_qp_in_set_node_bounds(xmin + ix, xmax + ix, umin + iu, umax + iu, qp_in, k);
        ix += qp_in->xmin[k].m; iu += qp_in->umin[k].m;
    }
    assert(ix == total_number_of_states(qp_in) && iu == total_number_of_controls(qp_in));
}
void tree_qp_in_set_const_bounds(double *xmin, double *xmax, double *umin, double *umax, tree_qp_in *qp_in)
{
    for (int k = 0; k < qp_in->N; k++) {
        assert(qp_in->nx[k] == qp_in->nx[1] || qp_in->nx[k] == 0);
        assert(qp_in->nu[k] == qp_in->nu[0] || qp_in->nu[k] == 0);
        tree_qp_in_set_node_bounds(xmin, xmax, umin, umax, qp_in, k);
    }
}
void tree_qp_in_set_inf_bounds(tree_qp_in *qp_in)
{
    for (int k = 0; k < qp_in->N; k++) {
        blasfeo_dvecse(qp_in->xmin[k].m, -TREEQP_INF, &qp_in->xmin[k], 0);
        blasfeo_dvecse(qp_in->xmax[k].m, TREEQP_INF, &qp_in->xmax[k], 0);
        blasfeo_dvecse(qp_in->umin[k].m, -TREEQP_INF, &qp_in->umin[k], 0);
        blasfeo_dvecse(qp_in->umax[k].m, TREEQP_INF, &qp_in->umax[k], 0);
    }
}

/* ------------------------------------------------------------------------------------- */
/* LTI filler (tree_qp_common.c:1837-1949)                                               */
/* ------------------------------------------------------------------------------------- */

void tree_qp_in_fill_lti_data_diag_weights(double *A, double *B, double *b,
    double *Q, double *q, double *P, double *p, double *R, double *r,
    double *xmin, double *xmax, double *umin, double *umax, double *x0,
    double *C, double *CN, double *D, double *dmin, double *dmax, tree_qp_in *qp_in)
{
    const int Nn = qp_in->N;
    const struct node *tree = qp_in->tree;
    assert(qp_in->nx[0] > 0 && "Use eliminate_x0 functions instead of passing nx[0] = 0 here!");

    /* leaves = trailing run of nodes sharing the last stage */
    int numberOfLeaves = 1;
    for (int k = Nn - 1; k > 0 && tree[k].stage == tree[k - 1].stage; k--) numberOfLeaves++;

    int stage_open = 0, in_stage = 0;
    for (int k = 0; k < Nn; k++) {
        const int nx = qp_in->nx[k];
        if (k > 0) {
            const int p = tree[k].dad, re = tree[k].real;
            tree_qp_in_set_edge_dynamics_colmajor(A + (size_t)re * nx * qp_in->nx[p], B + (size_t)re * nx * qp_in->nu[p],
                b + (size_t)re * nx, qp_in, k - 1);
        }
        if (tree[k].nkids > 0) tree_qp_in_set_node_objective_diag(Q, R, q, r, qp_in, k);
        else tree_qp_in_set_node_objective_diag(P, NULL, p, NULL, qp_in, k);

        /* when a new stage opens, scale the weights of the stage just closed so that every
         * stage carries the same total weight; the factor is an INTEGER quotient (:1911) */
        if (tree[k].stage > stage_open) {
            const double scale = numberOfLeaves / in_stage;
            for (int j = k - in_stage; j < k; j++) {
                blasfeo_dgesc(qp_in->Q[j].m, qp_in->Q[j].n, scale, &qp_in->Q[j], 0, 0);
                blasfeo_dgesc(qp_in->R[j].m, qp_in->R[j].n, scale, &qp_in->R[j], 0, 0);
                blasfeo_dvecsc(qp_in->q[j].m, scale, &qp_in->q[j], 0);
                blasfeo_dvecsc(qp_in->r[j].m, scale, &qp_in->r[j], 0);
            }
            stage_open = tree[k].stage;
            in_stage = 1;
        } else {
            in_stage++;
        }
        if (k == 0) tree_qp_in_set_node_bounds(x0, x0, umin, umax, qp_in, k);
        else tree_qp_in_set_node_bounds(xmin, xmax, umin, umax, qp_in, k);

        if (tree[k].nkids > 0) tree_qp_in_set_node_general_constraints(C, D, dmin, dmax, qp_in, k);
        else tree_qp_in_set_node_general_constraints(CN, NULL, dmin, dmax, qp_in, k);
    }
    /* keep the root's pre-elimination copy of r consistent with the scaled value */
    if (qp_in->nx[0] > 0) {
        blasfeo_dveccp(qp_in->nu[0], &qp_in->r[0], 0, &qp_in->internal_memory.r0, 0);
        qp_in->internal_memory.is_r_initialized = 1;
    }
}
