/*
 * tree_topology.c -- scenario-tree numbering for the treeqp_amd host layer.
 *
 * Behaviour (field values for every node) follows the reference's treeqp/utils/tree.c:36-280;
 * tests/test_tree.py compares every integer with the oracle.  Implementation notes:
 *  - children of node i are the next unassigned indices, so one running cursor replaces the
 *    reference's rescan for the first node with stage == -1 (tree.c:205-214);
 *  - tree_create returns TREEQP_OK explicitly (the reference's `return_t TREEQP_OK;` at
 *    tree.c:242 is a declaration, its return value is indeterminate).
 */
#include "treeqp/utils/tree.h"
#include "treeqp/utils/utils.h"

#include <assert.h>
#include <stddef.h>

int calculate_number_of_nodes(int md, int Nr, int Nh)
{
    if (md == 1) return Nh + 1;                       /* chain */
    int scenarios = ipow(md, Nr);
    return (Nh - Nr) * scenarios + (scenarios * md - 1) / (md - 1);
}

int get_number_of_parent_nodes(int Nn, const struct node *tree)
{
    int count = 0;
    for (int k = 0; k < Nn; k++) count += (tree[k].nkids > 0);
    return count;
}

int get_robust_horizon(int Nn, const struct node *tree)
{
    int Nr = 0;
    for (int k = 0; k < Nn && tree[k].nkids > 1; k++) Nr = tree[k].stage + 1;
    return Nr;
}

int get_prediction_horizon(int Nn, const struct node *tree)
{
    int k = Nn - 1, depth = 0;
    while (k != 0) {
        if (depth >= Nn) return -1;
        k = tree[k].dad;
        depth++;
    }
    return depth;
}

/* walk stage by stage until the first leaf; -1 on inconsistent data */
static int count_nodes(const int *nk_plain, const struct node *tree)
{
#define NK(i) (tree ? tree[(i)].nkids : nk_plain[(i)])
    int first = 0, width = 1;
    for (;;) {
        int next_width = 0;
        for (int i = 0; i < width; i++) {
            int c = NK(first + i);
            if (c < 0) return -1;
            if (c == 0) break;                        /* leaves: uniform depth assumed */
            next_width += c;
        }
        first += width;
        if (next_width == 0) return first;
        if (next_width < width) return -1;
        width = next_width;
    }
#undef NK
}

int number_of_nodes_from_nkids(const int *nkids) { return count_nodes(nkids, NULL); }
int number_of_nodes_from_tree(const struct node *tree) { return count_nodes(NULL, tree); }

int tree_calculate_size(const int *nk)
{
    int Nn = number_of_nodes_from_nkids(nk);
    int bytes = 0;
    for (int k = 0; k < Nn; k++) bytes += nk[k] * (int)sizeof(int);
    return bytes;
}

return_t tree_create(const int *nk, struct node *tree, void *ptr)
{
    const int Nn = number_of_nodes_from_nkids(nk);
    if (Nn < 0) return TREEQP_FAILURE;

    int *kid_store = (int *)ptr;
    tree[0].idx = 0; tree[0].dad = -1; tree[0].stage = 0; tree[0].idxkid = 0; tree[0].real = -1;

    int cursor = 1;                                   /* first unassigned node */
    for (int i = 0; i < Nn; i++) {
        tree[i].nkids = nk[i];
        tree[i].kids = nk[i] > 0 ? kid_store : NULL;
        kid_store += nk[i];
        for (int c = 0; c < nk[i]; c++) {
            struct node *kid = &tree[cursor + c];
            tree[i].kids[c] = cursor + c;
            kid->idx = cursor + c;
            kid->dad = i;
            kid->stage = tree[i].stage + 1;
            kid->idxkid = c;
            /* realization: ordinal under a branching parent, inherited otherwise (0 below a
             * non-branching root) */
            kid->real = (nk[i] > 1) ? c : (i > 0 ? tree[i].real : 0);
        }
        cursor += nk[i];
    }
    assert((char *)ptr + tree_calculate_size(nk) == (char *)kid_store);
    return TREEQP_OK;
}

void setup_multistage_tree(int md, int Nr, int Nh, int *nk)
{
    int first = 0, width = 1;
    for (int stage = 0; stage < Nh; stage++) {
        const int fanout = stage < Nr ? md : 1;
        for (int i = 0; i < width; i++) nk[first + i] = fanout;
        first += width;
        width *= fanout;
    }
    for (int i = 0; i < width; i++) nk[first + i] = 0;
}

/* Partition of the persistent launch's workgroups over the ranks of a sharded solve (treeqp_amd.h: tqgpu_pshard_*), host
 * arithmetic only.  Uniform complete md-ary tree with Nh block levels; tiers of the fused path bottom-up (3 levels for md = 2,
 * 2 for md <= 4, else 1; the top tier takes the rest); workgroups are numbered tier by tier from the bottom, one per tier subtree.
 * Tiers whose subtree count is a multiple of nranks go to the ranks by contiguous subtree ranges (the independent subtrees of
 * dual_Newton_tree.c:668-775), the tiers above them to rank 0.  wgs (may be NULL): this rank's workgroups, bottom tier first;
 * *n: how many; *part_top: the highest partitioned tier (-1: one rank); *boundary_level: the tree level of the partitioned
 * subtree roots.  Returns 0, or -1 when the tree is too small for that many ranks. */
int tqgpu_pshard_plan(int md, int Nh, int nranks, int rank, int *wgs, int cap, int *n, int *part_top, int *boundary_level)
{
    if (md < 2 || Nh < 1 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    const int TH = md == 2 ? 3 : (md <= 4 ? 2 : 1);
    const int nt = (Nh + TH - 1) / TH;
    int l0v[64], gridv[64];
    if (nt > 64) return -1;
    for (int i = 0; i < nt; i++) {
        const int l1 = Nh - i * TH, l0 = l1 - TH > 0 ? l1 - TH : 0;
        int grid = 1;
        for (int l = 0; l < l0; l++) grid *= md;
        l0v[i] = l0; gridv[i] = grid;
    }
    int top = nranks == 1 ? nt - 1 : -1;
    if (nranks > 1) for (int i = 0; i < nt - 1; i++) if (gridv[i] % nranks == 0 && gridv[i] >= nranks) top = i;
    if (top < 0) return -1;
    int count = 0, wg0 = 0;
    for (int i = 0; i < nt; i++) {
        for (int q = 0; q < gridv[i]; q++) {
            const int owner = (nranks > 1 && i <= top) ? q / (gridv[i] / nranks) : 0;
            if (owner == rank) { if (wgs && count < cap) wgs[count] = wg0 + q; count++; }
        }
        wg0 += gridv[i];
    }
    if (n) *n = count;
    if (part_top) *part_top = nranks > 1 ? top : -1;
    if (boundary_level) *boundary_level = nranks > 1 ? l0v[top] : 0;
    return 0;
}

