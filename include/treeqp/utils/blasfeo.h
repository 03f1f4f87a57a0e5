/* treeqp_amd: dmat/dvec inspection helpers (reference: treeqp/utils/blasfeo.h:41-61). */
#ifndef TREEQP_UTILS_BLASFEO_H_
#define TREEQP_UTILS_BLASFEO_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/utils/types.h"
#include "treeqp/utils/utils.h"
#include <blasfeo_target.h>
#include <blasfeo_common.h>

void convert_strvecs_to_single_vec(int n, const struct blasfeo_dvec *sv, double *v);
void convert_strmats_to_single_vec(int n, const struct blasfeo_dmat *sM, double *M);
void convert_strmats_tran_to_single_vec(int n, const struct blasfeo_dmat *sM, double *M);

double check_error_strmat(const struct blasfeo_dmat *M1, const struct blasfeo_dmat *M2);
double check_error_strvec(const struct blasfeo_dvec *v1, const struct blasfeo_dvec *v2);
double check_error_strvec_double(const struct blasfeo_dvec *v1, const double *v2);

answer_t is_strmat_symmetric(const struct blasfeo_dmat *M);
answer_t is_strmat_diagonal(const struct blasfeo_dmat *M);
answer_t is_strmat_zero(const struct blasfeo_dmat *M);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_BLASFEO_H_ */
