/*
 * treeqp_amd: the HPMPC interior-point backend (reference: treeqp/src/hpmpc_tree.h) is OUT OF
 * SCOPE of this build (third-party solver, absent submodule).  This header
 * exists only because some reference drivers include it unconditionally
 * (examples/thesis_example.c:35) while using tdunes; it declares nothing.
 */
#ifndef TREEQP_SRC_HPMPC_TREE_H_
#define TREEQP_SRC_HPMPC_TREE_H_
#endif
