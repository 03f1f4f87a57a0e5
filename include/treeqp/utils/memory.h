/*
 * treeqp_amd: bump-allocation helpers of the treeQP C API (reference: treeqp/utils/memory.h:43-71).
 * Pattern kept from the reference: X_calculate_size -> caller malloc -> X_create(..., ptr).
 */
#ifndef TREEQP_UTILS_MEMORY_H_
#define TREEQP_UTILS_MEMORY_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/utils/types.h"
#include "treeqp/utils/utils.h"
#include <blasfeo_target.h>
#include <blasfeo_common.h>

void make_int_multiple_of(int num, int *size);
int align_char_to(int num, char **c_ptr);

void create_int(int m, int **v, char **ptr);
void create_double(int m, double **v, char **ptr);
void create_strvec(int m, struct blasfeo_dvec *sv, char **ptr);
void create_strmat(int m, int n, struct blasfeo_dmat *sM, char **ptr);
void create_double_ptr_int(int m, int n, int ***arr, char **ptr);
void create_double_ptr_strvec(int m, int n, struct blasfeo_dvec ***arr, char **ptr);
void create_double_ptr_strmat(int m, int n, struct blasfeo_dmat ***arr, char **ptr);

void wrapper_vec_to_strvec(int m, const double *v, struct blasfeo_dvec *sv, char **ptr);
void wrapper_mat_to_strmat(int m, int n, const double *M, struct blasfeo_dmat *sM, char **ptr);
void init_strvec(int m, struct blasfeo_dvec *sv, char **ptr);
void init_strmat(int m, int n, struct blasfeo_dmat *sM, char **ptr);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_MEMORY_H_ */
