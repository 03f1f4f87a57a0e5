/*
 * treeqp_amd BLASFEO-compat: the level-1/2 BLAS subset the *host* layer needs (KKT residual,
 * x0 elimination, data marshalling).  The level-3 / factorization routines of the reference's
 * hot path (dgemm_nd, dgemm_nt, dsyrk_ln, dpotrf_l, dtrsv_*, dtrsm_rltn; SURVEY.md §8 a7')
 * are deliberately NOT provided on the host: in this build they exist only as HIP device code
 * (treeqp_amd/csrc/device), so the product has no CPU solve path to fall back to.
 */
#ifndef TREEQP_AMD_BLASFEO_D_BLAS_H_
#define TREEQP_AMD_BLASFEO_D_BLAS_H_
#include "blasfeo_common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* z = y + alpha*x */
void blasfeo_daxpy(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
double blasfeo_ddot(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi);
/* z = beta*y + alpha*A*x   (A is m x n) */
void blasfeo_dgemv_n(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
/* z = beta*y + alpha*A'*x  (A is m x n, x length m, z length n) */
void blasfeo_dgemv_t(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
/* z = beta*y + alpha*A*x, A symmetric, lower part referenced */
void blasfeo_dsymv_l(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
/* lower Cholesky of the leading m x m part (a non-positive pivot gives a zero column); _mn: m x n, m >= n */
void blasfeo_dpotrf_l(int m, struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dpotrf_l_mn(int m, int n, struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);

#ifdef __cplusplus
}
#endif
#endif
