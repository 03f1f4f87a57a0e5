/*
 * tdunes_gpersist.hpp -- the whole solve of a SMALL tree of ANY shape as one launch of ONE workgroup.
 *
 * Included by tdunes_device.hip (uses the bodies of the generic kernels: stage_body, grad_body,
 * hess_body, factor_body, forward_body).
 *
 * The generic path launches one kernel per phase and tree level: ~25 launches per Newton iteration.
 * For the trees real MPC callers bring (the reference's own spring-mass example: 85 nodes; the pruned
 * scenario trees of examples/fault_tolerance.c: a few hundred nodes, <= 40 blocks per level) a launch
 * has a handful of wavefronts of work and costs its ~4.5 us floor: the solve is pure launch overhead.
 * Here one workgroup of GP_WAVES wavefronts walks through all phases, waves taking nodes / blocks
 * round-robin with a workgroup barrier between dependent phases and tree levels.  State stays in global
 * memory (L2 resident; one CU, one L1, so a barrier makes it visible); control flow -- termination,
 * direction test, Armijo backtracking with any number of trials, iteration cap -- runs on the device,
 * and the verdict goes to the pinned host result block like on the uniform-tree persistent path.
 */
#pragma once

#ifndef GP_WAVES
#define GP_WAVES 16
#endif

struct GParams {
    const int *lvl_first;        /* [Nh + 2] first node of every tree level (device copy) */
    const double *lam_init;      /* starting duals of the solve */
    HostRes *hres;               /* pinned host result block */
    unsigned seq;                /* launch number (completion word of the result block) */
    int lds_wave;                /* doubles of LDS window per wave */
    int in_lds;                  /* 1: the mutable state of the solve is mirrored in LDS for the whole launch */
    int small8;                  /* 1: every node has nx <= 8: the gradient sweep takes eight nodes per wave (grad_body8) */
    int small16;                 /* 1: every node has nx + nu <= 16 and is a clipping node: the stage sweep takes four nodes per wave (stage_body16) */
    int tab_in_lds;              /* 1 (in_lds == 0): at least the index tables are -- every node or block step starts with a chain of dependent look-ups */
    int const_in_lds;            /* 1: ... and so are the constants (A, B, b, weights, linear terms, bounds) */
    int sum_nx, sum_nu, sum_W, sum_Ut, sum_A, sum_B;
};

/* The state the phases read and write (block matrices, factors, residuals, steps, duals, node variables) is
 * small for the trees this kernel is for -- it fits the 160 KB of LDS next to the per-wave windows.  The
 * generic bodies address it through ordinary (generic address space) pointers, so pointing those at an LDS
 * mirror turns every dependent global round trip (~1 us each, several per block step) into an LDS access.
 * Constants (A, B, b, weights, bounds, index tables) stay in global memory: read-only, cached. */
/* The stage sweep for SMALL nodes: four nodes per wave, one per row of 16 lanes (lane t of a row: entry t of [x | u] of its
 * node), instead of one node per wave with 64 - (nx + nu) lanes idle.  A sweep over the nodes of a tree is then a quarter of the
 * rounds -- and a line-search trial IS a sweep: the stragglers of a batch of C5-class trees (the launch ends with the slowest
 * tree) are the trees with many trials (one with 95 trials spent 14 of its 18 ms in them).  Same arithmetic as stage_body's
 * clipping branch; the sums of a node are taken over its row (row16_sum) instead of over the wave.  Needs nx + nu <= 16 for
 * every node of the group, clipping nodes only; `win` doubles of the wave's LDS window per row. */
struct SweepCtl { int cur; double step; bool save_s; };      /* what a sweep reads from the control block -- once per wave and sweep, not once per group of nodes */
__device__ __forceinline__ SweepCtl sweep_ctl(const Ctrl *c, int mode) {
    SweepCtl sc;
    sc.cur = c->cur; sc.step = c->tau - c->tauPrev; sc.save_s = mode == 1 && c->ls_iter == 1;
    return sc;
}
__device__ void stage_body16(const Tree &T, const Data &D, int mode, int k0, int count, int lane, double *lds, int win, bool batch, const SweepCtl &sc) {
    const int g = lane >> 4, t = lane & 15;
    const bool act = g < count;
    const int k = act ? k0 + g : k0;
    const int nxk = T.nx[k], nuk = T.nu[k], xo = T.xoff[k], uo = T.uoff[k];
    const int nkid = T.nk[k], d = T.bdim[k];
    const double *lamc = sc.cur ? D.lam1 : D.lam0;
    double *lamn = sc.cur ? D.lam0 : D.lam1;
    const double step = sc.step;
    const bool save_s = sc.save_s;
    double *lk = lds + (size_t)g * win;         /* d doubles */
    double *lown = lk + d;                      /* nxk doubles */
    const int kid0k = nkid > 0 ? T.kid0[k] : 0;
    const int ko = nkid > 0 ? T.xoff[kid0k] : 0;
    /* everything of the lane's entry that does not depend on the duals is requested NOW, in one round of loads with the duals below: taken where
     * it is used -- behind stores the compiler must assume to alias -- each of them was a round trip of its own to the L2 (six to eight per
     * group of four nodes; the trees that do not fit the LDS mirror are the stragglers of a batch) */
    const bool ent = act && t < nxk + nuk, isx0 = t < nxk;
    const int je = ent ? (isx0 ? xo + t : uo + t - nxk) : (isx0 ? xo : uo);
    const double c_lin = ent ? (isx0 ? D.q : D.r)[je] : 0.0, c_inv = ent ? (isx0 ? D.Qinv : D.Rinv)[je] : 0.0;
    const double c_lo = ent ? (isx0 ? D.xmin : D.umin)[je] : 0.0, c_hi = ent ? (isx0 ? D.xmax : D.umax)[je] : 0.0;
    const double c_w = ent ? (isx0 ? D.Qd : D.Rd)[je] : 0.0;
    const double c_unc = (ent && save_s) ? (isx0 ? D.xUnc : D.uUnc)[je] : 0.0;
    const double c_b0 = (act && t < d) ? D.b[ko + t] : 0.0, c_b1 = (act && t + 16 < d) ? D.b[ko + t + 16] : 0.0;      /* b of the children (the node's constant term), first two rounds of the loop at the end */
    if (act) {
        for (int tt = t; tt < d; tt += 16) {
            double v = lamc[ko + tt];
            if (mode == 1) v = fma(step, D.dlam[ko + tt], v);
            lk[tt] = v;
        }
        for (int tt = t; tt < nxk; tt += 16) {
            double v = 0.0;
            if (k > 0) {
                v = lamc[xo + tt];
                if (mode == 1) { v = fma(step, D.dlam[xo + tt], v); lamn[xo + tt] = v; }
            }
            lown[tt] = v;
        }
    }
    WSYNC();
    double p_qx = 0.0, p_hx = 0.0, p_ru = 0.0, p_hu = 0.0, p_c = 0.0;
    if (act && t < nxk + nuk) {
        const bool isx = t < nxk;
        const int j = isx ? t : t - nxk;
        double v = isx ? fma(-1.0, c_lin, lown[j]) : -1.0 * c_lin;
        int rowoff = 0;
        for (int cc = 0; cc < nkid; cc++) {
            const int kid = kid0k + cc, nxc = T.nx[kid];
            const double *col = isx ? D.A + T.aoff[kid] + (size_t)j * nxc : D.B + T.boff[kid] + (size_t)j * nxc;
            double acc = 0.0;
            acc = dot_batched(col, 1, lk + rowoff, 1, nxc, acc, batch);
            v = fma(-1.0, acc, v);
            rowoff += nxc;
        }
        if (isx) {
            D.qmod[xo + j] = v;
            const double qi = c_inv;
            const double unc = qi * v, lo = c_lo, hi = c_hi;
            double xv, cal;
            if (unc >= hi) { xv = hi; cal = 0.0; } else if (unc <= lo) { xv = lo; cal = 0.0; } else { xv = unc; cal = qi; }
            if (save_s) D.xUncS[xo + j] = c_unc;
            D.xUnc[xo + j] = unc; D.x[xo + j] = xv; D.QinvCal[xo + j] = cal;
            p_qx = (c_w * xv) * xv;
            p_hx = v * xv;
        } else {
            D.rmod[uo + j] = v;
            const double ri = c_inv;
            const double unc = ri * v, lo = c_lo, hi = c_hi;
            double uv, cal;
            if (unc >= hi) { uv = hi; cal = 0.0; } else if (unc <= lo) { uv = lo; cal = 0.0; } else { uv = unc; cal = ri; }
            if (save_s) D.uUncS[uo + j] = c_unc;
            D.uUnc[uo + j] = unc; D.u[uo + j] = uv; D.RinvCal[uo + j] = cal;
            p_ru = (c_w * uv) * uv;
            p_hu = v * uv;
        }
    }
    if (act) {
        if (t < d) p_c = fma(c_b0, lk[t], p_c);
        if (t + 16 < d) p_c = fma(c_b1, lk[t + 16], p_c);
        for (int tt = t + 32; tt < d; tt += 16) p_c = fma(D.b[ko + tt], lk[tt], p_c);
    }
    p_qx = row16_sum(p_qx); p_hx = row16_sum(p_hx); p_ru = row16_sum(p_ru); p_hu = row16_sum(p_hu); p_c = row16_sum(p_c);
    if (act && t == 0) {
        double f = -0.5 * p_qx - p_c;       /* clipping.c:375 */
        f += p_hx;                          /* :376 */
        f -= 0.5 * p_ru;                    /* :380 */
        f += p_hu;                          /* :381 */
        D.fval[k] = f;
    }
    WSYNC();
}

/* The gradient sweep for nodes with nx <= 8: eight nodes per wave, one per group of 8 lanes (grad_body keeps nx lanes of 64 busy).
 * The node's termination partial is taken over its group: after row_ror 4, 2, 1 the LAST lane of a group holds the group's
 * sum / maximum.  Same arithmetic and order of the sums per entry as grad_body. */
__device__ void grad_body8(const Tree &T, const Data &D, int termCondition, int k0, int count, int lane, bool batch) {
    const int g = lane >> 3, i = lane & 7;
    const bool act = g < count;
    const int k = act ? k0 + g : k0;
    const int p = T.dad[k], nxk = T.nx[k], nxp = T.nx[p], nup = T.nu[p];
    const int xo = T.xoff[k], xp = T.xoff[p], up = T.uoff[p];
    const double *A = D.A + T.aoff[k], *B = D.B + T.boff[k];
    double part = 0.0;
    if (act && i < nxk) {
        double rv = fma(-1.0, D.x[xo + i], D.b[xo + i]);
        double acc = 0.0;
        acc = dot_batched(A + i, nxk, D.x + xp, 1, nxp, acc, batch);
        rv += acc;
        acc = 0.0;
        acc = dot_batched(B + i, nxk, D.u + up, 1, nup, acc, batch);
        rv += acc;
        D.res[xo + i] = rv;
        D.resMod[xo + i] = rv;
        part = (termCondition == 2) ? fabs(rv) : rv * rv;
    }
    if (termCondition == 2) { part = nanmax(part, dpp_mov<0x124>(part)); part = nanmax(part, dpp_mov<0x122>(part)); part = nanmax(part, dpp_mov<0x121>(part)); }
    else { part += dpp_mov<0x124>(part); part += dpp_mov<0x122>(part); part += dpp_mov<0x121>(part); }
    if (act && i == 7) D.part_err[k] = part;
}

__device__ __forceinline__ double *gp_take(double *&cursor, int n) { double *p = cursor; cursor += (n + 1) & ~1; return p; }

/* The two descriptor structs hold ~60 pointers.  As kernel arguments (or locals) they would have to live in
 * scalar registers for the whole kernel -- 120 SGPRs, more than a wave has -- and every use would be a spill
 * reload.  They live in LDS instead: the bodies fetch the pointer they need when they need it. */
__device__ __forceinline__ void g_persist_body(Tree T_in, Data D_in, Opts O, GParams G) {
    __shared__ Data sD;
    __shared__ Tree sT;
    Data D = D_in;
    Tree T = T_in;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    __shared__ double sh[GP_WAVES];
    __shared__ int flag;
    Ctrl *c = D.ctrl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *lds = lds_all + (size_t)wave * G.lds_wave;
    double *cur_tab = lds_all + (size_t)GP_WAVES * G.lds_wave;      /* where the LDS copy of the index tables goes (behind the state mirror, if there is one) */
    if (G.in_lds) {
        double *cur_p = lds_all + (size_t)GP_WAVES * G.lds_wave;
        const int sx = G.sum_nx, su = G.sum_nu, Nn_ = T.Nn;
        D.Qinv = gp_take(cur_p, sx); D.Rinv = gp_take(cur_p, su);
        D.qmod = gp_take(cur_p, sx); D.rmod = gp_take(cur_p, su); D.x = gp_take(cur_p, sx); D.u = gp_take(cur_p, su);
        D.xUnc = gp_take(cur_p, sx); D.uUnc = gp_take(cur_p, su); D.QinvCal = gp_take(cur_p, sx); D.RinvCal = gp_take(cur_p, su);
        D.lam0 = gp_take(cur_p, sx); D.lam1 = gp_take(cur_p, sx); D.dlam = gp_take(cur_p, sx); D.res = gp_take(cur_p, sx); D.resMod = gp_take(cur_p, sx);
        D.invd = gp_take(cur_p, sx);
        D.W = gp_take(cur_p, G.sum_W); D.CholW = gp_take(cur_p, G.sum_W); D.Ut = gp_take(cur_p, G.sum_Ut); D.CholUt = gp_take(cur_p, G.sum_Ut);
        D.fval = gp_take(cur_p, Nn_); D.part_err = gp_take(cur_p, sx + Nn_ + 1); D.part_dot = gp_take(cur_p, Nn_);
        if (G.const_in_lds) {
            /* the bodies walk short runtime-bounded loops with one load per trip: from global memory that is one
             * cache latency per trip, from LDS a tenth of it */
            double *cA = gp_take(cur_p, G.sum_A), *cB = gp_take(cur_p, G.sum_B), *cb = gp_take(cur_p, sx);
            double *cQd = gp_take(cur_p, sx), *cq = gp_take(cur_p, sx), *cxl = gp_take(cur_p, sx), *cxu = gp_take(cur_p, sx);
            double *cRd = gp_take(cur_p, su), *cr = gp_take(cur_p, su), *cul = gp_take(cur_p, su), *cuu = gp_take(cur_p, su);
            for (int i = threadIdx.x; i < G.sum_A; i += GP_WAVES * WAVE) cA[i] = D_in.A[i];
            for (int i = threadIdx.x; i < G.sum_B; i += GP_WAVES * WAVE) cB[i] = D_in.B[i];
            for (int i = threadIdx.x; i < sx; i += GP_WAVES * WAVE) { cb[i] = D_in.b[i]; cQd[i] = D_in.Qd[i]; cq[i] = D_in.q[i]; cxl[i] = D_in.xmin[i]; cxu[i] = D_in.xmax[i]; }
            for (int i = threadIdx.x; i < su; i += GP_WAVES * WAVE) { cRd[i] = D_in.Rd[i]; cr[i] = D_in.r[i]; cul[i] = D_in.umin[i]; cuu[i] = D_in.umax[i]; }
            D.A = cA; D.B = cB; D.b = cb; D.Qd = cQd; D.q = cq; D.xmin = cxl; D.xmax = cxu; D.Rd = cRd; D.r = cr; D.umin = cul; D.umax = cuu;
        }
        cur_tab = cur_p;
    }
    if (G.in_lds || G.tab_in_lds) {
        /* the index tables: every node or block step starts with a chain of dependent table look-ups -- from global memory that is
         * a round trip per hop, whatever else the step reads.  (They fit next to the per-wave windows even when the state does not:
         * the members of a batch of C5-class trees.) */
        int *ip = reinterpret_cast<int *>(cur_tab);
        const int Nn_ = T.Nn, n1 = Nn_ + 1;
        const int *src[13] = {T_in.dad, T_in.nk, T_in.kid0, T_in.nx, T_in.nu, T_in.xoff, T_in.uoff, T_in.aoff, T_in.boff, T_in.pos, T_in.bdim, T_in.woff, T_in.utoff};
        const int len[13] = {Nn_, Nn_, Nn_, Nn_, Nn_, n1, n1, n1, n1, Nn_, Nn_, n1, n1};
        int *dst[13];
        for (int a = 0; a < 13; a++) { dst[a] = ip; ip += (len[a] + 1) & ~1; }
        for (int a = 0; a < 13; a++) for (int i = threadIdx.x; i < len[a]; i += GP_WAVES * WAVE) dst[a][i] = src[a][i];
        T.dad = dst[0]; T.nk = dst[1]; T.kid0 = dst[2]; T.nx = dst[3]; T.nu = dst[4]; T.xoff = dst[5]; T.uoff = dst[6];
        T.aoff = dst[7]; T.boff = dst[8]; T.pos = dst[9]; T.bdim = dst[10]; T.woff = dst[11]; T.utoff = dst[12];
        __syncthreads();
    }
    if (threadIdx.x == 0) { sD = D; sT = T; }
    __syncthreads();
    const unsigned long long t_start = wall_clock64();
    const int Nn = T_in.Nn, Np = T_in.Np, Nh = T_in.Nh;
    /* diagnostic (TREEQP_AMD_STAMPS): wall clock per phase, summed over the solve: 0 init+first sweep, 1 G, 2 H,
     * 3 F backward, 4 F forward, 5 L */
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, t_prev = t_start;
#define GP_MARK(i) do { if (O.stamps && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); acc_t[i] += now_ - t_prev; t_prev = now_; } } while (0)

    /* fresh control block, current duals = lambda0 */
    if (threadIdx.x == 0) {
        c->done = 0; c->status = 0; c->iter = 0; c->cur = 0; c->ls_pending = 0; c->ls_iter = 0; c->ls_total = 0; c->ls_last = 0;
        c->restart_counter = 0; c->n_reg = 0; c->tau = 0.0; c->tauPrev = 0.0; c->fval0 = 0.0; c->fval = 0.0; c->dot = 0.0; c->err = 0.0;
    }
    for (int i = threadIdx.x; i < sT.xoff[Nn]; i += GP_WAVES * WAVE) sD.lam0[i] = G.lam_init[i];
    if (sD.Qinv) {
        for (int i = threadIdx.x; i < sT.xoff[Nn]; i += GP_WAVES * WAVE) sD.Qinv[i] = 1.0 / sD.Qd[i];     /* k_init */
        for (int i = threadIdx.x; i < sT.uoff[Nn]; i += GP_WAVES * WAVE) sD.Rinv[i] = 1.0 / sD.Rd[i];
    }
    __syncthreads();

    /* first sweep at lambda0 (phase S of iteration 0) and fval0 */
    const int win16 = G.lds_wave / 4;           /* LDS per row of 16 lanes in stage_body16 */
    auto stage_sweep = [&](int mode) {
        if (G.small16 && !sD.strict) { const SweepCtl sc = sweep_ctl(c, mode); for (int k0 = 4 * wave; k0 < Nn; k0 += 4 * GP_WAVES) stage_body16(sT, sD, mode, k0, min(4, Nn - k0), lane, lds, win16, !G.const_in_lds, sc); }
        else for (int k = wave; k < Nn; k += GP_WAVES) stage_body(sT, sD, mode, k, lane, lds, !G.const_in_lds);
    };
    stage_sweep(0);
    __syncthreads();
    {
        const double f = sD.strict ? bcast0(threadIdx.x == 0 ? strict_sum(sD.fval, Nn) : 0.0, sh) : block_reduce<false>(sD.fval, Nn, sh);
        if (threadIdx.x == 0) { c->fval0 = f; c->fval = f; }
    }
    __syncthreads();
    GP_MARK(0);

    for (;;) {
        /* ---- G: dual gradient + termination test (dual_Newton_tree.c:519-543) ---- */
        if (G.small8) { for (int k0 = 1 + 8 * wave; k0 < Nn; k0 += 8 * GP_WAVES) grad_body8(sT, sD, O.termCondition, k0, min(8, Nn - k0), lane, !G.const_in_lds); }
        else for (int k = 1 + wave; k < Nn; k += GP_WAVES) grad_body(sT, sD, O.termCondition, k, lane, !G.const_in_lds);
        __syncthreads();
        {
            double err = (O.termCondition == 2) ? block_reduce<true>(sD.part_err + 1, Nn - 1, sh)
                                                : (sD.strict ? bcast0(threadIdx.x == 0 ? strict_block_dots(sT, sD.res, sD.res) : 0.0, sh) : block_reduce<false>(sD.part_err + 1, Nn - 1, sh));
            if (threadIdx.x == 0) {
                if (O.termCondition == 1) err = sqrt(err);
                c->err = err;
                if (err < O.tol) { c->done = 1; c->status = 0; }
                flag = c->done;
            }
        }
        __syncthreads();
        GP_MARK(1);
        if (flag) break;

        /* ---- H: block dual Hessian (:551-615) ---- */
        for (int p = wave; p < Np; p += GP_WAVES) hess_body(sT, sD, p, lane, lds);      /* (side by side in lane groups, as the sweeps below: measured slower, 67 against 56 us per iteration of a 308-node tree) */
        __syncthreads();
        GP_MARK(2);

        /* ---- F: backward factorisation level by level (children push their Schur complements into the
         * parent's block), then forward substitution (:641-805) ---- */
        for (int lvl = Nh - 1; lvl >= 0; lvl--) {
            const int first = G.lvl_first[lvl], count = G.lvl_first[lvl + 1] - first;
            const int grp = G.lvl_first[Nh + 2 + lvl];          /* blocks of this level one wave takes side by side (1: one block per wave) */
            if (grp == 3 && sT.bdim[first] == 8 && sT.nx[first] == 8) { for (int b = 3 * wave; b < count; b += 3 * GP_WAVES) factor_body_g<3, 8, 8>(sT, sD, O, first + b, min(3, count - b), lane, lds, G.lds_wave / 3); }
            else if (grp == 3) { for (int b = 3 * wave; b < count; b += 3 * GP_WAVES) factor_body_g<3>(sT, sD, O, first + b, min(3, count - b), lane, lds, G.lds_wave / 3); }
            else if (grp == 2) { for (int b = 2 * wave; b < count; b += 2 * GP_WAVES) factor_body_g<2>(sT, sD, O, first + b, min(2, count - b), lane, lds, G.lds_wave / 2); }
            else for (int b = wave; b < count; b += GP_WAVES) factor_body(sT, sD, O, first + b, lane, lds);
            __syncthreads();
        }
        GP_MARK(3);
        for (int lvl = 1; lvl < Nh; lvl++) {
            const int first = G.lvl_first[lvl], count = G.lvl_first[lvl + 1] - first;
            const int grp = G.lvl_first[Nh + 2 + lvl];
            if (grp == 3) { for (int b = 3 * wave; b < count; b += 3 * GP_WAVES) forward_body_g<3>(sT, sD, first + b, min(3, count - b), lane, lds, G.lds_wave / 3); }
            else if (grp == 2) { for (int b = 2 * wave; b < count; b += 2 * GP_WAVES) forward_body_g<2>(sT, sD, first + b, min(2, count - b), lane, lds, G.lds_wave / 2); }
            else for (int b = wave; b < count; b += GP_WAVES) forward_body(sT, sD, first + b, lane, lds);
            __syncthreads();
        }
        GP_MARK(4);

        /* ---- L: direction test, then Armijo backtracking; every trial is a full stage sweep (:922-1019) ---- */
        {
            const double s = sD.strict ? bcast0(threadIdx.x == 0 ? strict_block_dots(sT, sD.res, sD.dlam) : 0.0, sh) : block_reduce<false>(sD.part_dot, Np, sh);
            if (threadIdx.x == 0) {
                const double dotp = -s;
                c->dot = dotp;
                if (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10))) { c->done = 1; c->status = 2; }
                else { c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1; }
                flag = c->done;
            }
        }
        __syncthreads();
        if (flag) break;
        for (;;) {
            stage_sweep(1);
            __syncthreads();
            const double f = sD.strict ? bcast0(threadIdx.x == 0 ? strict_sum(sD.fval, Nn) : 0.0, sh) : block_reduce<false>(sD.fval, Nn, sh);
            if (threadIdx.x == 0) { ls_decide_tail(c, sD, O, f); flag = c->ls_pending; }
            __syncthreads();
            if (!flag) break;
        }
        if (threadIdx.x == 0) flag = c->done;
        __syncthreads();
        GP_MARK(5);
        if (flag) break;
        __syncthreads();
    }
    if (O.stamps && threadIdx.x == 0) for (int i = 0; i < 6; i++) { sD.stamps[2 * i] = acc_t[i]; sD.stamps[2 * i + 1] = 1ull; }
#undef GP_MARK

    /* the solution goes back to global memory (what tqgpu_get_solution and a warm-started next solve read) */
    __syncthreads();
    if (G.in_lds) {
        const int sx = G.sum_nx, su = G.sum_nu;
        for (int i = threadIdx.x; i < sx; i += GP_WAVES * WAVE) {
            D_in.x[i] = sD.x[i]; D_in.xUnc[i] = sD.xUnc[i]; D_in.qmod[i] = sD.qmod[i]; D_in.QinvCal[i] = sD.QinvCal[i];
            D_in.lam0[i] = sD.lam0[i]; D_in.lam1[i] = sD.lam1[i]; D_in.dlam[i] = sD.dlam[i]; D_in.res[i] = sD.res[i];
        }
        for (int i = threadIdx.x; i < su; i += GP_WAVES * WAVE) { D_in.u[i] = sD.u[i]; D_in.uUnc[i] = sD.uUnc[i]; D_in.rmod[i] = sD.rmod[i]; D_in.RinvCal[i] = sD.RinvCal[i]; }
    }
    /* verdict to the host */
    __syncthreads();
    if (threadIdx.x == 0) {
        HostRes *hr = G.hres;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(c);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&hr->c);
        for (int i = 0; i < (int)(sizeof(Ctrl) / 8); i++) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hr->t_start, t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hr->t_end, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* every store of the block acknowledged, then the sequence word (no L2 write-back: see f_persist) */
        __hip_atomic_store(&hr->seq, G.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(GP_WAVES * WAVE) g_persist(Tree T_in, Data D_in, Opts O, GParams G)
#if !TQ_HAS(TQP_GP)
;
#else
{ g_persist_body(T_in, D_in, O, G); }
#endif

/* a batch of independent small trees (fault_tolerance.c keeps one QP per spring configuration, :486-530): ONE launch, one
 * workgroup per tree, each working from its own descriptors.  (One launch per tree on its own stream only overlaps as
 * many trees as the runtime has hardware queues -- four.) */
struct GItem { Tree T; Data D; GParams G; };
__global__ void __launch_bounds__(GP_WAVES * WAVE) g_persist_batch(const GItem *items, Opts O)
#if !TQ_HAS(TQP_GP)
;
#else
{
    const GItem *it = items + blockIdx.x;
    g_persist_body(it->T, it->D, O, it->G);
}
#endif
