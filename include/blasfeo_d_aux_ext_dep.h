/* treeqp_amd BLASFEO-compat: routines with external dependencies (malloc / stdio). */
#ifndef TREEQP_AMD_BLASFEO_D_AUX_EXT_DEP_H_
#define TREEQP_AMD_BLASFEO_D_AUX_EXT_DEP_H_
#include "blasfeo_common.h"
#ifdef __cplusplus
extern "C" {
#endif
void blasfeo_allocate_dmat(int m, int n, struct blasfeo_dmat *sA);
void blasfeo_allocate_dvec(int m, struct blasfeo_dvec *sa);
void blasfeo_free_dmat(struct blasfeo_dmat *sA);
void blasfeo_free_dvec(struct blasfeo_dvec *sa);

void blasfeo_print_dmat(int m, int n, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_print_exp_dmat(int m, int n, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_print_dvec(int m, struct blasfeo_dvec *sa, int ai);
void blasfeo_print_exp_dvec(int m, struct blasfeo_dvec *sa, int ai);
void blasfeo_print_tran_dvec(int m, struct blasfeo_dvec *sa, int ai);
void blasfeo_print_exp_tran_dvec(int m, struct blasfeo_dvec *sa, int ai);
void d_print_mat(int m, int n, double *A, int lda);
void d_print_e_mat(int m, int n, double *A, int lda);
void d_print_exp_mat(int m, int n, double *A, int lda);
void d_zeros(double **pA, int row, int col);
void d_free(double *pA);
#ifdef __cplusplus
}
#endif
#endif
