/*
 * host_utils.c -- small host-side services of the treeQP C API for the treeqp_amd build:
 * bump allocation (reference API: treeqp/utils/memory.h), text IO + ipow (utils.h),
 * timers (timing.h), min-over-runs profiling record (profiling.h), dmat/dvec inspection
 * (utils/blasfeo.h) and printing (print.h).  Written from the API contracts; behaviour that
 * callers depend on cites the reference line it matches.
 */
#include "treeqp/utils/memory.h"
#include "treeqp/utils/utils.h"
#include "treeqp/utils/timing.h"
#include "treeqp/utils/profiling.h"
#include "treeqp/utils/blasfeo.h"
#include "treeqp/utils/print.h"
#include "treeqp/utils/tree.h"

#include <blasfeo_d_aux.h>
#include <blasfeo_d_aux_ext_dep.h>

#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ bump allocation ---- */

void make_int_multiple_of(int num, int *size) { *size = (*size + num - 1) / num * num; }

/* returns num - (bytes skipped), like the reference (memory.c:58-65) */
int align_char_to(int num, char **c_ptr)
{
    uintptr_t p = (uintptr_t)*c_ptr;
    uintptr_t q = (p + (uintptr_t)num - 1) / (uintptr_t)num * (uintptr_t)num;
    *c_ptr = (char *)q;
    return num - (int)(q - p);
}

void create_int(int m, int **v, char **ptr) { *v = (int *)*ptr; *ptr += sizeof(int) * (size_t)m; }

void create_double(int m, double **v, char **ptr)
{
    assert((uintptr_t)*ptr % 8 == 0 && "double not 8-byte aligned!");
    *v = (double *)*ptr; *ptr += sizeof(double) * (size_t)m;
}

void create_strvec(int m, struct blasfeo_dvec *sv, char **ptr)
{
    assert((uintptr_t)*ptr % 8 == 0 && "strvec not 8-byte aligned!");
    blasfeo_create_dvec(m, sv, *ptr);
    *ptr += sv->memsize;
}

void create_strmat(int m, int n, struct blasfeo_dmat *sM, char **ptr)
{
    assert((uintptr_t)*ptr % 8 == 0 && "strmat not 8-byte aligned!");
    blasfeo_create_dmat(m, n, sM, *ptr);
    *ptr += sM->memsize;
}

#define DEFINE_PTR_TABLE(FN, T)                                              \
    void FN(int m, int n, T ***arr, char **ptr)                              \
    {                                                                        \
        *arr = (T **)*ptr; *ptr += (size_t)m * sizeof(T *);                  \
        for (int i = 0; i < m; i++) { (*arr)[i] = (T *)*ptr; *ptr += (size_t)n * sizeof(T); } \
    }
DEFINE_PTR_TABLE(create_double_ptr_strvec, struct blasfeo_dvec)
DEFINE_PTR_TABLE(create_double_ptr_strmat, struct blasfeo_dmat)
#undef DEFINE_PTR_TABLE

void create_double_ptr_int(int m, int n, int ***arr, char **ptr)
{
    *arr = (int **)*ptr; *ptr += (size_t)m * sizeof(int *);
    for (int i = 0; i < m; i++) {
        (*arr)[i] = (int *)*ptr; *ptr += (size_t)n * sizeof(int);
        memset((*arr)[i], 0, (size_t)n * sizeof(int));      /* zero-initialised (memory.c:136-140) */
    }
}

void wrapper_vec_to_strvec(int m, const double *v, struct blasfeo_dvec *sv, char **ptr)
{
    create_strvec(m, sv, ptr);
    blasfeo_pack_dvec(m, (double *)v, sv, 0);
}
void wrapper_mat_to_strmat(int m, int n, const double *M, struct blasfeo_dmat *sM, char **ptr)
{
    create_strmat(m, n, sM, ptr);
    blasfeo_pack_dmat(m, n, (double *)M, m, sM, 0, 0);
}
void init_strvec(int m, struct blasfeo_dvec *sv, char **ptr)
{
    create_strvec(m, sv, ptr);
    blasfeo_dvecse(m, 0.0, sv, 0);
}
void init_strmat(int m, int n, struct blasfeo_dmat *sM, char **ptr)
{
    create_strmat(m, n, sM, ptr);
    blasfeo_dgese(m, n, 0.0, sM, 0, 0);
}

/* ------------------------------------------------------------------------- utils ------- */

int ipow(int base, int exp)
{
    int out = 1;
    for (; exp > 0; exp >>= 1, base *= base)
        if (exp & 1) out *= base;
    return out;
}

static FILE *open_or_complain(const char *filename, const char *mode, const char *what)
{
    FILE *f = fopen(filename, mode);
    if (!f) printf("Error %s file (%s)\n", what, filename);
    return f;
}

return_t read_int_vector_from_txt(const int *const vec, const int n, const char *filename)
{
    FILE *f = open_or_complain(filename, "r", "reading");
    if (!f) return TREEQP_ERROR_OPENING_FILE;
    int *out = (int *)vec;                       /* the API takes const, the function fills it */
    for (int i = 0; i < n; i++) if (fscanf(f, "%d,", &out[i]) != 1) break;
    fclose(f);
    return TREEQP_OK;
}

return_t read_double_vector_from_txt(const double *const vec, const int n, const char *filename)
{
    FILE *f = open_or_complain(filename, "r", "reading");
    if (!f) return TREEQP_ERROR_OPENING_FILE;
    double *out = (double *)vec;
    for (int i = 0; i < n; i++) if (fscanf(f, "%lf,", &out[i]) != 1) break;
    fclose(f);
    return TREEQP_OK;
}

return_t write_double_vector_to_txt(const double *const vec, const int n, const char *filename)
{
    FILE *f = open_or_complain(filename, "w", "opening");
    if (!f) return TREEQP_ERROR_OPENING_FILE;
    for (int i = 0; i < n; i++) fprintf(f, "%.16e\n", vec[i]);
    fclose(f);
    return TREEQP_OK;
}

return_t write_int_vector_to_txt(const int *const vec, const int n, const char *filename)
{
    FILE *f = open_or_complain(filename, "w", "opening");
    if (!f) return TREEQP_ERROR_OPENING_FILE;
    for (int i = 0; i < n; i++) fprintf(f, "%d\n", vec[i]);
    fclose(f);
    return TREEQP_OK;
}

/* ------------------------------------------------------------------------ timing ------- */

void treeqp_tic(treeqp_timer *t) { clock_gettime(CLOCK_MONOTONIC, &t->tic); }

double treeqp_toc(treeqp_timer *t)
{
    clock_gettime(CLOCK_MONOTONIC, &t->toc);
    return (double)(t->toc.tv_sec - t->tic.tv_sec) + 1e-9 * (double)(t->toc.tv_nsec - t->tic.tv_nsec);
}

/* --------------------------------------------------------------------- profiling ------- */

enum { N_ITER_DOUBLE_ARRAYS = 10 };   /* iter, min_iter, 4 phases, 4 min phases */

int timers_calculate_size(int num_iter)
{
    return num_iter * (N_ITER_DOUBLE_ARRAYS * (int)sizeof(double) + (int)sizeof(int));
}

void timers_create(int num_iter, treeqp_profiling_t *t, void *ptr)
{
    double *d = (double *)ptr;
    t->num_iter = num_iter;
    double **slots[N_ITER_DOUBLE_ARRAYS] = {
        &t->iter_times, &t->min_iter_times,
        &t->stage_qps_times, &t->build_dual_times, &t->newton_direction_times, &t->line_search_times,
        &t->min_stage_qps_times, &t->min_build_dual_times, &t->min_newton_direction_times,
        &t->min_line_search_times};
    for (int s = 0; s < N_ITER_DOUBLE_ARRAYS; s++) { *slots[s] = d; d += num_iter; }
    t->ls_iters = (int *)d;
}

void timers_initialize(treeqp_profiling_t *t)
{
    t->total_time = NAN; t->min_total_time = NAN; t->total_ls_iter = 0; t->run_indx = 0;
    for (int i = 0; i < t->num_iter; i++) {
        t->iter_times[i] = t->min_iter_times[i] = NAN;
        t->stage_qps_times[i] = t->build_dual_times[i] = NAN;
        t->newton_direction_times[i] = t->line_search_times[i] = NAN;
        t->min_stage_qps_times[i] = t->min_build_dual_times[i] = NAN;
        t->min_newton_direction_times[i] = t->min_line_search_times[i] = NAN;
        t->ls_iters[i] = 0;
    }
}

static void fold_min(int first_run, int n, double *mins, const double *cur)
{
    for (int i = 0; i < n; i++)
        if (first_run || cur[i] < mins[i]) mins[i] = cur[i];       /* NaN never replaces */
}

/* min over runs; line-search totals are taken from the first run (profiling.c:164-171) */
void timers_update(treeqp_profiling_t *t)
{
    const int first = (t->run_indx == 0), n = t->num_iter;
    fold_min(first, 1, &t->min_total_time, &t->total_time);
    fold_min(first, n, t->min_iter_times, t->iter_times);
    fold_min(first, n, t->min_stage_qps_times, t->stage_qps_times);
    fold_min(first, n, t->min_build_dual_times, t->build_dual_times);
    fold_min(first, n, t->min_newton_direction_times, t->newton_direction_times);
    fold_min(first, n, t->min_line_search_times, t->line_search_times);
    if (first) {
        t->total_ls_iter = 0;
        for (int i = 0; i < n; i++) t->total_ls_iter += t->ls_iters[i];
    }
    t->run_indx++;
}

void timers_print(treeqp_profiling_t *t)
{
    int iters = 0;
    while (iters < t->num_iter && t->ls_iters[iters] > 0) iters++;
    printf("\nTotal time:\n\n");
    printf("> > > algorithm converged in (%d it):\t %10.4f ms\n\n", iters, t->min_total_time * 1e3);
    if (iters > 0 && !isnan(t->min_iter_times[0])) {
        printf("\nTimings per iteration:\n\n");
        for (int j = 0; j < iters; j++)
            printf("Iteration #%3d - %7.3f ms  (%3d ls iters. )\n", j + 1, t->min_iter_times[j] * 1e3, t->ls_iters[j]);
    }
    if (iters > 0 && !isnan(t->min_stage_qps_times[0])) {
        double s[4] = {0, 0, 0, 0};
        for (int j = 0; j < iters; j++) {
            s[0] += t->min_stage_qps_times[j]; s[1] += t->min_build_dual_times[j];
            s[2] += t->min_newton_direction_times[j]; s[3] += t->min_line_search_times[j];
        }
        const double all = s[0] + s[1] + s[2] + s[3];
        printf("\nTimings per operation:\n\n");
        printf("> > > solved stage QPs in:\t\t %10.4f ms (%5.2f %%)\n", s[0] * 1e3, 100 * s[0] / all);
        printf("> > > built dual problem in:\t\t %10.4f ms (%5.2f %%)\n", s[1] * 1e3, 100 * s[1] / all);
        printf("> > > calculated Newton direction in: \t %10.4f ms (%5.2f %%)\n", s[2] * 1e3, 100 * s[2] / all);
        printf("> > > performed line-search (%d it) in:\t %10.4f ms (%5.2f %%)\n", t->total_ls_iter, s[3] * 1e3, 100 * s[3] / all);
        printf("> > > sum all of the above:\t\t %10.4f ms\n", all * 1e3);
    }
}

void timers_write_to_txt(treeqp_profiling_t *t)
{
    write_double_vector_to_txt(&t->min_total_time, 1, "examples/spring_mass_utils/cputime.txt");
    write_double_vector_to_txt(t->min_iter_times, t->num_iter, "examples/spring_mass_utils/iter_times.txt");
    write_int_vector_to_txt(t->ls_iters, t->num_iter, "examples/spring_mass_utils/ls_iters.txt");
}

/* ------------------------------------------------------------- dmat/dvec inspection ---- */

void convert_strvecs_to_single_vec(int n, const struct blasfeo_dvec *sv, double *v)
{
    for (int k = 0; k < n; k++) { memcpy(v, sv[k].pa, sizeof(double) * (size_t)sv[k].m); v += sv[k].m; }
}
void convert_strmats_to_single_vec(int n, const struct blasfeo_dmat *sM, double *M)
{
    for (int k = 0; k < n; k++) {
        size_t cnt = (size_t)sM[k].m * (size_t)sM[k].n;
        memcpy(M, sM[k].pA, sizeof(double) * cnt); M += cnt;
    }
}
void convert_strmats_tran_to_single_vec(int n, const struct blasfeo_dmat *sM, double *M)
{
    for (int k = 0; k < n; k++) {
        blasfeo_unpack_tran_dmat(sM[k].m, sM[k].n, (struct blasfeo_dmat *)&sM[k], 0, 0, M, sM[k].n);
        M += (size_t)sM[k].m * (size_t)sM[k].n;
    }
}

static double max_abs_diff(size_t n, const double *a, const double *b)
{
    double worst = 0.0;
    for (size_t i = 0; i < n; i++) { double d = fabs(a[i] - b[i]); if (d > worst || d != d) worst = d; }
    return worst;
}
double check_error_strmat(const struct blasfeo_dmat *M1, const struct blasfeo_dmat *M2)
{
    return max_abs_diff((size_t)M1->m * (size_t)M1->n, M1->pA, M2->pA);
}
double check_error_strvec(const struct blasfeo_dvec *v1, const struct blasfeo_dvec *v2)
{
    return max_abs_diff((size_t)v1->m, v1->pa, v2->pa);
}
double check_error_strvec_double(const struct blasfeo_dvec *v1, const double *v2)
{
    return max_abs_diff((size_t)v1->m, v1->pa, v2);
}

answer_t is_strmat_symmetric(const struct blasfeo_dmat *M)
{
    if (M->m != M->n) return NO;
    for (int j = 0; j < M->n; j++) for (int i = j + 1; i < M->m; i++)
        if (BLASFEO_DMATEL(M, i, j) != BLASFEO_DMATEL(M, j, i)) return NO;
    return YES;
}
answer_t is_strmat_diagonal(const struct blasfeo_dmat *M)
{
    for (int j = 0; j < M->n; j++) for (int i = 0; i < M->m; i++)
        if (i != j && BLASFEO_DMATEL(M, i, j) != 0.0) return NO;
    return YES;
}
answer_t is_strmat_zero(const struct blasfeo_dmat *M)
{
    for (int j = 0; j < M->n; j++) for (int i = 0; i < M->m; i++)
        if (BLASFEO_DMATEL(M, i, j) != 0.0) return NO;
    return YES;
}

/* ---------------------------------------------------------------------- printing ------- */

void node_print(const struct node *n)
{
    printf("node %d: dad %d, stage %d, realization %d, kid #%d, %d kids:", n->idx, n->dad, n->stage, n->real, n->idxkid, n->nkids);
    for (int c = 0; c < n->nkids; c++) printf(" %d", n->kids[c]);
    printf("\n");
}

void tree_qp_in_print_dims(const tree_qp_in *qp_in)
{
    printf("tree QP with %d nodes\n", qp_in->N);
    for (int k = 0; k < qp_in->N; k++)
        printf("  node %4d: nx = %d, nu = %d, nc = %d, kids = %d\n", k, qp_in->nx[k], qp_in->nu[k], qp_in->nc[k], qp_in->tree[k].nkids);
}

static void print_vec(const char *name, int k, const struct blasfeo_dvec *v)
{
    printf("%s[%d] = ", name, k);
    blasfeo_print_tran_dvec(v->m, (struct blasfeo_dvec *)v, 0);
}
static void print_mat_named(const char *name, int k, const struct blasfeo_dmat *M)
{
    printf("%s[%d] =\n", name, k);
    blasfeo_print_dmat(M->m, M->n, (struct blasfeo_dmat *)M, 0, 0);
}

void tree_qp_in_print(const tree_qp_in *qp_in)
{
    tree_qp_in_print_dims(qp_in);
    for (int k = 0; k < qp_in->N; k++) {
        printf("---------------- node %d ----------------\n", k);
        if (k > 0) {
            print_mat_named("A", k - 1, &qp_in->A[k - 1]);
            print_mat_named("B", k - 1, &qp_in->B[k - 1]);
            print_vec("b", k - 1, &qp_in->b[k - 1]);
        }
        print_mat_named("Q", k, &qp_in->Q[k]); print_mat_named("R", k, &qp_in->R[k]);
        print_mat_named("S", k, &qp_in->S[k]);
        print_vec("q", k, &qp_in->q[k]); print_vec("r", k, &qp_in->r[k]);
        print_vec("xmin", k, &qp_in->xmin[k]); print_vec("xmax", k, &qp_in->xmax[k]);
        print_vec("umin", k, &qp_in->umin[k]); print_vec("umax", k, &qp_in->umax[k]);
    }
}

void tree_qp_out_print(int Nn, const tree_qp_out *qp_out)
{
    for (int k = 0; k < Nn; k++) {
        printf("---------------- node %d ----------------\n", k);
        print_vec("x", k, &qp_out->x[k]); print_vec("u", k, &qp_out->u[k]);
        if (k > 0) print_vec("lam", k - 1, &qp_out->lam[k - 1]);
        print_vec("mu_x", k, &qp_out->mu_x[k]); print_vec("mu_u", k, &qp_out->mu_u[k]);
    }
}

void tree_qp_out_write_to_txt(const tree_qp_in *qp_in, const tree_qp_out *qp_out, const char *fpath)
{
    char name[512];
    const int Nn = qp_in->N;
    int sx = 0, su = 0;
    for (int k = 0; k < Nn; k++) { sx += qp_out->x[k].m; su += qp_out->u[k].m; }
    double *buf = malloc(sizeof(double) * (size_t)(sx + su + 1));
    convert_strvecs_to_single_vec(Nn, qp_out->x, buf);
    snprintf(name, sizeof name, "%s/x_opt.txt", fpath); write_double_vector_to_txt(buf, sx, name);
    convert_strvecs_to_single_vec(Nn, qp_out->u, buf);
    snprintf(name, sizeof name, "%s/u_opt.txt", fpath); write_double_vector_to_txt(buf, su, name);
    free(buf);
}

void regularization_print_status(regType_t reg_type, reg_result_t reg_res)
{
    printf("regularization type %d: %s\n", (int)reg_type, reg_res == TREEQP_REGULARIZATION_ADDED ? "added" : "not added");
}

void blasfeo_print_target(void) { printf("BLASFEO target: treeqp_amd column-major compat layer (host) + HIP gfx950 (device)\n"); }

/* ------------------------------------------------------------------------------------- */
/* regularised Cholesky utility (dual_Newton_common.c:36-123); one body for both shapes   */
/* ------------------------------------------------------------------------------------- */
#include "treeqp/src/dual_Newton_common.h"
#include <blasfeo_d_aux.h>
#include <blasfeo_d_blas.h>

static reg_result_t potrf_reg(struct blasfeo_dmat *M, struct blasfeo_dmat *CholM, int ncol, regType_t reg_type, double reg_tol, double reg_val)
{
    const int m = M->m;
    reg_result_t res = TREEQP_NO_REGULARIZATION_ADDED;
    if (reg_type == TREEQP_ALWAYS_LEVENBERG_MARQUARDT) { blasfeo_ddiare(m, reg_val, M, 0, 0); res = TREEQP_REGULARIZATION_ADDED; }
    blasfeo_dpotrf_l_mn(m, ncol, M, 0, 0, CholM, 0, 0);
    if (reg_type != TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT) return res;
    int small = 0;
    for (int j = 0; j < m && j < ncol && !small; j++) small = BLASFEO_DMATEL(CholM, j, j) <= reg_tol;
    if (small) {
        blasfeo_ddiare(m, reg_val, M, 0, 0);
        blasfeo_dpotrf_l_mn(m, ncol, M, 0, 0, CholM, 0, 0);
        res = TREEQP_REGULARIZATION_ADDED;
    }
    return res;
}

reg_result_t treeqp_dpotrf_l_with_reg_opts(struct blasfeo_dmat *M, struct blasfeo_dmat *CholM, regType_t reg_type, double reg_tol, double reg_val)
{
    return potrf_reg(M, CholM, M->m, reg_type, reg_tol, reg_val);
}

reg_result_t treeqp_dpotrf_l_mn_with_reg_opts(struct blasfeo_dmat *M, struct blasfeo_dmat *CholM, regType_t reg_type, double reg_tol, double reg_val)
{
    return potrf_reg(M, CholM, M->n, reg_type, reg_tol, reg_val);
}
