/*
 * treeqp_amd: the HPIPM interior-point backend (reference: treeqp/src/hpipm_tree.h) is OUT OF
 * SCOPE of this build (third-party solver, absent submodule).  This header exists
 * only because some reference drivers include it unconditionally (examples/random_qp.c:32) while
 * using tdunes; it declares nothing.
 */
#ifndef TREEQP_SRC_HPIPM_TREE_H_
#define TREEQP_SRC_HPIPM_TREE_H_
#endif
