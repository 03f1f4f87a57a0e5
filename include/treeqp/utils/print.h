/* treeqp_amd: printing helpers of the treeQP C API (reference: treeqp/utils/print.h:41-58). */
#ifndef TREEQP_UTILS_PRINT_H_
#define TREEQP_UTILS_PRINT_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/src/dual_Newton_common.h"
#include "treeqp/src/tree_qp_common.h"
#include "treeqp/utils/tree.h"
#include "treeqp/utils/types.h"
#include "treeqp/utils/profiling.h"

void node_print(const struct node *tree);
void tree_qp_in_print_dims(const tree_qp_in *qp_in);
void tree_qp_in_print(const tree_qp_in *qp_in);
void tree_qp_out_print(int Nn, const tree_qp_out *qp_out);
void tree_qp_out_write_to_txt(const tree_qp_in *qp_in, const tree_qp_out *qp_out, const char *fpath);
void timers_write_to_txt(treeqp_profiling_t *timings);
void regularization_print_status(regType_t reg_type, reg_result_t reg_res);
void blasfeo_print_target(void);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_PRINT_H_ */
