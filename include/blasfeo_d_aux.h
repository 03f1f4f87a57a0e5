/*
 * treeqp_amd BLASFEO-compat: auxiliary (non-BLAS) double-precision routines.
 * Same names / argument order as the BLASFEO 0.1.x API the reference calls
 * (call sites: tree_qp_common.c, memory.c, dual_Newton_tree.c:1660, blasfeo.c).
 * Note the 4-argument blasfeo_pack_dvec (pre-2020 signature, tree_qp_common.c:1034).
 */
#ifndef TREEQP_AMD_BLASFEO_D_AUX_H_
#define TREEQP_AMD_BLASFEO_D_AUX_H_
#include "blasfeo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

int blasfeo_memsize_dmat(int m, int n);
int blasfeo_memsize_dvec(int m);
void blasfeo_create_dmat(int m, int n, struct blasfeo_dmat *sA, void *memory);
void blasfeo_create_dvec(int m, struct blasfeo_dvec *sa, void *memory);

void blasfeo_pack_dmat(int m, int n, double *A, int lda, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_pack_tran_dmat(int m, int n, double *A, int lda, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_pack_dvec(int m, double *a, struct blasfeo_dvec *sa, int ai);
void blasfeo_unpack_dmat(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, double *A, int lda);
void blasfeo_unpack_tran_dmat(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, double *A, int lda);
void blasfeo_unpack_dvec(int m, struct blasfeo_dvec *sa, int ai, double *a);

void blasfeo_dgese(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_dvecse(int m, double alpha, struct blasfeo_dvec *sx, int xi);
void blasfeo_dgecp(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dgesc(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_dgead(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dgetr(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sC, int ci, int cj);
void blasfeo_dveccp(int m, struct blasfeo_dvec *sa, int ai, struct blasfeo_dvec *sc, int ci);
void blasfeo_dvecsc(int m, double alpha, struct blasfeo_dvec *sa, int ai);
void blasfeo_dveccpsc(int m, double alpha, struct blasfeo_dvec *sa, int ai, struct blasfeo_dvec *sc, int ci);

void blasfeo_ddiaex(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi);
void blasfeo_ddiain(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_ddiaad(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_ddiare(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_drowin(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_drowex(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi);

/* z = x .* y */
void blasfeo_dvecmuldot(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
/* z = clip(x, xm, xp); inclusive comparisons (x >= xp -> xp, x <= xm -> xm) */
void blasfeo_dveccl(int m, struct blasfeo_dvec *sxm, int xim, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sxp, int xip, struct blasfeo_dvec *sz, int zi);
/* same, and mask = +1 / -1 / 0 for upper / lower / inactive */
void blasfeo_dveccl_mask(int m, struct blasfeo_dvec *sxm, int xim, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sxp, int xip, struct blasfeo_dvec *sz, int zi, struct blasfeo_dvec *sm, int mi);
/* e = (mask == 0) ? v : 0 */
void blasfeo_dvecze(int m, struct blasfeo_dvec *sm, int mi, struct blasfeo_dvec *sv, int vi, struct blasfeo_dvec *se, int ei);

#ifdef __cplusplus
}
#endif
#endif
