/* treeqp_amd: small helpers of the treeQP C API (reference: treeqp/utils/utils.h:39-53). */
#ifndef TREEQP_UTILS_UTILS_H_
#define TREEQP_UTILS_UTILS_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/utils/types.h"

#define MAX(X, Y) (X > Y ? X:Y)
#define MIN(X, Y) (X < Y ? X:Y)
#define ABS(X) ((X) < 0 ?-(X):X)

int ipow(int base, int exp);

/* one value per line (an optional trailing comma is accepted) */
return_t read_int_vector_from_txt(const int *const vec, const int n, const char *filename);
return_t read_double_vector_from_txt(const double *const vec, const int n, const char *filename);
return_t write_double_vector_to_txt(const double *const vec, const int n, const char *filename);
return_t write_int_vector_to_txt(const int *const vec, const int n, const char *filename);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_UTILS_H_ */
