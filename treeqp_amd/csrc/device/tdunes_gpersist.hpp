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

#define GP_WAVES 16

struct GParams {
    const int *lvl_first;        /* [Nh + 2] first node of every tree level (device copy) */
    const double *lam_init;      /* starting duals of the solve */
    HostRes *hres;               /* pinned host result block */
    unsigned seq;                /* launch number (completion word of the result block) */
    int lds_wave;                /* doubles of LDS window per wave */
};

__global__ void __launch_bounds__(GP_WAVES * WAVE) g_persist(Tree T, Data D, Opts O, GParams G) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    __shared__ double sh[GP_WAVES];
    __shared__ int flag;
    Ctrl *c = D.ctrl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *lds = lds_all + (size_t)wave * G.lds_wave;
    const unsigned long long t_start = wall_clock64();
    const int Nn = T.Nn, Np = T.Np, Nh = T.Nh;
    /* diagnostic (TREEQP_AMD_STAMPS): wall clock per phase, summed over the solve: 0 init+first sweep, 1 G, 2 H,
     * 3 F backward, 4 F forward, 5 L */
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, t_prev = t_start;
#define GP_MARK(i) do { if (O.stamps && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); acc_t[i] += now_ - t_prev; t_prev = now_; } } while (0)

    /* fresh control block, current duals = lambda0 */
    if (threadIdx.x == 0) {
        c->done = 0; c->status = 0; c->iter = 0; c->cur = 0; c->ls_pending = 0; c->ls_iter = 0; c->ls_total = 0; c->ls_last = 0;
        c->restart_counter = 0; c->n_reg = 0; c->tau = 0.0; c->tauPrev = 0.0; c->fval0 = 0.0; c->fval = 0.0; c->dot = 0.0; c->err = 0.0;
    }
    for (int i = threadIdx.x; i < T.xoff[Nn]; i += GP_WAVES * WAVE) D.lam0[i] = G.lam_init[i];
    if (D.Qinv) {
        for (int i = threadIdx.x; i < T.xoff[Nn]; i += GP_WAVES * WAVE) D.Qinv[i] = 1.0 / D.Qd[i];     /* k_init */
        for (int i = threadIdx.x; i < T.uoff[Nn]; i += GP_WAVES * WAVE) D.Rinv[i] = 1.0 / D.Rd[i];
    }
    __syncthreads();

    /* first sweep at lambda0 (phase S of iteration 0) and fval0 */
    for (int k = wave; k < Nn; k += GP_WAVES) stage_body(T, D, 0, k, lane, lds);
    __syncthreads();
    {
        const double f = block_reduce<false>(D.fval, Nn, sh);
        if (threadIdx.x == 0) { c->fval0 = f; c->fval = f; }
    }
    __syncthreads();
    GP_MARK(0);

    for (;;) {
        /* ---- G: dual gradient + termination test (dual_Newton_tree.c:519-543) ---- */
        for (int k = 1 + wave; k < Nn; k += GP_WAVES) grad_body(T, D, O.termCondition, k, lane);
        __syncthreads();
        {
            double err = (O.termCondition == 2) ? block_reduce<true>(D.part_err + 1, Nn - 1, sh) : block_reduce<false>(D.part_err + 1, Nn - 1, sh);
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
        for (int p = wave; p < Np; p += GP_WAVES) hess_body(T, D, p, lane, lds);
        __syncthreads();
        GP_MARK(2);

        /* ---- F: backward factorisation level by level (children push their Schur complements into the
         * parent's block), then forward substitution (:641-805) ---- */
        for (int lvl = Nh - 1; lvl >= 0; lvl--) {
            const int first = G.lvl_first[lvl], count = G.lvl_first[lvl + 1] - first;
            for (int b = wave; b < count; b += GP_WAVES) factor_body(T, D, O, first + b, lane, lds);
            __syncthreads();
        }
        GP_MARK(3);
        for (int lvl = 1; lvl < Nh; lvl++) {
            const int first = G.lvl_first[lvl], count = G.lvl_first[lvl + 1] - first;
            for (int b = wave; b < count; b += GP_WAVES) forward_body(T, D, first + b, lane, lds);
            __syncthreads();
        }
        GP_MARK(4);

        /* ---- L: direction test, then Armijo backtracking; every trial is a full stage sweep (:922-1019) ---- */
        {
            const double s = block_reduce<false>(D.part_dot, Np, sh);
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
            for (int k = wave; k < Nn; k += GP_WAVES) stage_body(T, D, 1, k, lane, lds);
            __syncthreads();
            const double f = block_reduce<false>(D.fval, Nn, sh);
            if (threadIdx.x == 0) { ls_decide_tail(c, D, O, f); flag = c->ls_pending; }
            __syncthreads();
            if (!flag) break;
        }
        if (threadIdx.x == 0) flag = c->done;
        __syncthreads();
        GP_MARK(5);
        if (flag) break;
        __syncthreads();
    }
    if (O.stamps && threadIdx.x == 0) for (int i = 0; i < 6; i++) { D.stamps[2 * i] = acc_t[i]; D.stamps[2 * i + 1] = 1ull; }
#undef GP_MARK

    /* verdict to the host */
    __syncthreads();
    if (threadIdx.x == 0) {
        HostRes *hr = G.hres;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(c);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&hr->c);
        for (int i = 0; i < (int)(sizeof(Ctrl) / 8); i++) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hr->t_start, t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hr->t_end, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hr->seq, G.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
