/*
 * treeqp_amd: scenario-tree topology (BFS numbered, children contiguous).
 * API restates the reference's treeqp/utils/tree.h:41-73; semantics follow tree.c:36-280 and
 * are integer-exact (tests/test_tree.py checks every field against the oracle).
 */
#ifndef TREEQP_UTILS_TREE_H_
#define TREEQP_UTILS_TREE_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/utils/types.h"

#ifndef TREE_MPC
#ifndef HPIPM_TREE_H_
struct node {
    int *kids;    /* indices of children (contiguous range) */
    int idx;      /* own index */
    int dad;      /* parent index, -1 at the root */
    int nkids;
    int stage;    /* depth */
    int real;     /* realization id used by the LTI filler */
    int idxkid;   /* ordinal among siblings */
};
#endif
#endif

int calculate_number_of_nodes(int md, int Nr, int Nh);
int get_number_of_parent_nodes(int Nn, const struct node *tree);
int get_robust_horizon(int Nn, const struct node *tree);
int get_prediction_horizon(int Nn, const struct node *tree);
int number_of_nodes_from_nkids(const int *nkids);
int number_of_nodes_from_tree(const struct node *tree);
int tree_calculate_size(const int *nk);
return_t tree_create(const int *nk, struct node *tree, void *ptr);
void setup_multistage_tree(int md, int Nr, int Nh, int *nk);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_TREE_H_ */
