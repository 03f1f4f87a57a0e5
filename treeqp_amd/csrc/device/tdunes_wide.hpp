/*
 * tdunes_wide.hpp -- workgroup-per-block kernels for LARGER dual Hessian blocks (16 < d <= 64), e.g.
 * BASELINE config C4: nx = 20, nu = 10, three children per node -> d = 60, tall matrix 81 x 60.
 *
 * Included by tdunes_device.hip after tdunes_persist.hpp.  Same phases, same global layout and same
 * launch-per-level protocol as the wave-per-block kernels (k_hess / k_factor / k_forward) they stand in
 * for; what changes is how a block is worked on:
 *   - one 4-wave workgroup per block, the block padded to 16 x 16 tiles in LDS (leading dimensions
 *     == 16 mod 32 doubles, so the four k-groups of an MFMA operand fetch hit disjoint banks);
 *   - H:  W = C P C' by v_mfma_f64_16x16x4_f64, lower tiles dealt over the waves;
 *   - F:  blocked right-looking tall Cholesky, panel width 16.  A panel is factorised in REGISTERS, one row
 *     per lane, pivot rows broadcast by v_readlane (p_potrf_rows<16>: lanes 0..15 hold the diagonal tile,
 *     lanes 16..63 forty-eight rows below it; further rows go to further waves, which repeat the diagonal
 *     tile); the trailing update T22 -= L21 L21' is four MFMAs per 16 x 16 tile, the Schur complement into the
 *     parent X X' likewise.  d = 60: ~15 us of factorisation per block instead of ~72 us for the
 *     column-by-column LDS version;
 *   - substitutions (root solve, forward sweep) keep the vector in registers, one entry per lane, and
 *     broadcast z_k by v_readlane: one LDS read and one FMA per step, no barrier;
 *   - a dependent global round trip costs ~2.4 us here (data written by other XCDs), so every kernel reads ONE
 *     128-byte node record (Tree::desc) instead of walking the index tables, and issues all its data loads as one
 *     batch of clamped, branch-free loads (LOADS_DONE keeps the optimiser from re-serialising them).
 * Reference: calculate_hessian_blocks / build W (dual_Newton_tree.c:531-760), factorise + substitute
 * (:763-913).  Results differ from the wave-per-block kernels by summation order only.
 */
#pragma once

#define WW 4
#define WT (WW * WAVE)

#ifdef TQ_WIDE_STAMPS
#define WSTAMP(slot) do { if (first == 0 && threadIdx.x == 0) { D.stamps[2 * (slot)] = clock64(); D.stamps[2 * (slot) + 1] = wall_clock64(); } } while (0)
#else
#define WSTAMP(slot) do { } while (0)
#endif

/* keeps a batch of independent global loads a batch: the optimiser otherwise sinks each (clamped, unconditional) load
 * under the predicate that masks its result, one branch and one full wait per load */
#define LOADS_DONE() asm volatile("" ::: "memory")

__device__ __forceinline__ int up16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ int wide_ld(int rows_padded) { return rows_padded | 16; }
/* leading dimension of the tall matrix in k_factor_w: == 16 (mod 32) where that costs nothing, == 8 (mod 32) otherwise (the four
 * k-groups of an operand fetch then still spread over all banks, two groups per bank like with 16; and 81 x 60 -- C4 -- takes
 * 52 KB instead of 56: three workgroups per CU, i.e. the 729 blocks of C4's last level resident at once) */
#ifdef TQ_WIDE_LD16
__host__ __device__ __forceinline__ int wide_ldf(int rows_padded) { return rows_padded | 16; }
#else
#ifndef TQ_WIDE_PAD
#define TQ_WIDE_PAD 8
#endif
__host__ __device__ __forceinline__ int wide_ldf(int rows_padded) { return (rows_padded & 16) ? rows_padded : rows_padded + TQ_WIDE_PAD; }
#endif

/* z <- L^-T z for one wave, entry j of z on lane j (d <= 64): the strictly-lower part of column `lane` of L is
 * fetched into registers in one go, then every step is two readlanes and one FMA -- no memory in the chain.
 * L column major in LDS with leading dimension ld; myinv = 1 / L[lane][lane].  Returns the solution entry. */
__device__ __forceinline__ void wide_backsolve_load(lds_cptr L, int ld, int d, int lane, double (&Lc)[64]) {
    const int lc = lane < d ? lane : 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const bool use = k < d && k > lane;
        const double v = L[(use ? k : 0) + lc * ld];
        Lc[k] = use ? -v : 0.0;
    }
}
__device__ __forceinline__ double wide_backsolve_chain(const double (&Lc)[64], int d, double z, double myinv) {
#pragma unroll
    for (int k = 63; k >= 1; k--) {
        if (k < d) {
            const double zk = rdlane(z * myinv, k);
            z = fma(Lc[k], zk, z);
        }
    }
    return z * myinv;
}
__device__ __forceinline__ double wide_backsolve(lds_cptr L, int ld, int d, int lane, double z, double myinv) {
    double Lc[64];
    wide_backsolve_load(L, ld, d, lane, Lc);
    return wide_backsolve_chain(Lc, d, z, myinv);
}

/* ------------------------------------------------------------------------------------------ */
/* H                                                                                          */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(WT) k_hess_w(Tree T, Data D, int h)
#if !TQ_HAS(TQP_WIDE)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int e[28];
#pragma unroll
    for (int i = 0; i < 28; i++) e[i] = T.desc[DESC_INTS * p + i];       /* node record: requested together with the control block */
    if (!phase_main(D.ctrl, h)) return;
    const int d = e[0], nxp = e[1], nup = e[2], nz = nxp + nup;
    const int dp = up16(d), kz = (nz + 3) & ~3, ldc = wide_ld(dp);
    lds_ptr Cs = to_lds(lds);     /* C only: C P is formed when an operand is fetched (one multiply per MFMA) -- half the LDS, so that all
                                     parents of C4 (1093) are resident at once instead of in two rounds */
    const int k0 = e[4], ko = e[7];
    const double *Qc = D.QinvCal + e[5], *Rc = D.RinvCal + e[6];
    for (int e = tid; e < ldc * kz; e += WT) Cs[e] = 0.0;
    /* this lane's entries of P for the k-steps of the product below: column s + g, s = 0, 4, .. (requested before the staging) */
    const int g_ = lane >> 4;
    double pcs[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { const int col = 4 * m + g_; const int cs = col < nz ? col : 0; pcs[m] = cs < nxp ? Qc[cs] : Rc[cs - nxp]; }
    __syncthreads();
    /* children's [A B] rows: rows on the lanes, columns dealt over the waves; branch-free (clamped) loads with 32-bit
     * offsets, eight in flight per thread and child (more in flight costs registers, i.e. workgroups per CU, and this
     * launch is one block per parent: throughput matters more than one block's latency) */
    const int nkp = e[3];
    for (int cc = 0, rowoff = 0; cc < nkp; cc++) {
        const int kid = k0 + cc;
        const bool rec = cc < 4;                                                  /* the first four children are in the record */
        /* (static indices: a dynamically indexed register array would live in scratch memory) */
        const int rnx = cc == 0 ? e[16] : cc == 1 ? e[19] : cc == 2 ? e[22] : e[25];
        const int rao = cc == 0 ? e[17] : cc == 1 ? e[20] : cc == 2 ? e[23] : e[26];
        const int rbo = cc == 0 ? e[18] : cc == 1 ? e[21] : cc == 2 ? e[24] : e[27];
        const int nxc = rec ? rnx : T.nx[kid];
        const double *A = D.A + (rec ? rao : T.aoff[kid]), *B = D.B + (rec ? rbo : T.boff[kid]);
        const bool rowok = lane < nxc;
        const int i = rowok ? lane : 0;
        for (int c0 = 0; c0 < nz; c0 += 8 * WW) {
            double a[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int col = c0 + wave + WW * m;
                const int cs = (rowok && col < nz) ? col : 0;
                const bool st = cs < nxp;
                a[m] = (st ? A : B)[i + (st ? cs : cs - nxp) * nxc];
            }
            LOADS_DONE();
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int col = c0 + wave + WW * m;
                if (rowok && col < nz) Cs[rowoff + i + col * ldc] = a[m];
            }
        }
        rowoff += nxc;
    }
    __syncthreads();
    /* lower tiles (I >= J); computed transposed (A = tile J of CP, B = tile I of C) so that a lane's results lie in
     * one row of W and lanes run down a column: coalesced stores */
    const int nt = dp >> 4, r = lane & 15, g = lane >> 4;
    double *W = D.W + e[8];
    int t = 0;
    for (int J = 0; J < nt; J++) {
        for (int I = J; I < nt; I++, t++) {
            if ((t & (WW - 1)) != wave) continue;
            const int i = 16 * I + r;
            const double qd = D.QinvCal[ko + (i < d ? i : 0)];          /* diagonal term, in flight during the MFMAs */
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < 8; m++) {                            /* kz <= 32 (checked at create time) */
                const int s = 4 * m;
                if (s < kz) {
                    const double a = Cs[16 * J + r + (s + g) * ldc] * pcs[m];
                    const double b = Cs[16 * I + r + (s + g) * ldc];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = 16 * J + g + 4 * q;
                if (i < d && j <= i) W[i + (size_t)j * d] = acc[q] + (i == j ? qd : 0.0);
            }
        }
    }
    if (p > 0) {
        double *Ut = D.Ut + e[9];
        for (int f = tid; f < nxp * d; f += WT) {
            const int i = f % nxp, rr = f / nxp;
            Ut[i + (size_t)rr * nxp] = -1.0 * (Cs[rr + i * ldc] * Qc[i]);
        }
    }
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* F: tall Cholesky of [W ; resMod' ; Ut], Schur complement into the parent, root solve        */
/* ------------------------------------------------------------------------------------------ */
/* FUSED: all levels of the backward sweep in ONE launch (k_factor_all_w).  A block's Schur complement then does not go into
 * the parent's W / resMod in global memory (read-modify-write, visible to the parent only across a kernel boundary) but
 * travels as a record of tagged words (st_tag / ld_tag of the persistent path): entry (gi, gj), 1 <= gi <= nx, 0 <= gj <= gi, of
 * G = Xt Xt' at gi * (nx + 1) + gj of the block's record (rs doubles per node).  The parent loads its block -- which does not
 * depend on its children -- and subtracts the records in LDS as they arrive.  Workgroups are numbered from the LAST block to
 * the root and the hardware starts them in that order, so the children a workgroup waits for are running or done whatever
 * part of the grid is resident. */
template <bool FUSED>
__device__ __forceinline__ void factor_w_body(const Tree &T, const Data &D, const Opts &O, int ii, int first, int h, u64 *sch, int rs, unsigned tag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int small_flag;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NE = FUSED ? 28 : 14;
    int e[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) e[i] = T.desc[(size_t)DESC_INTS * ii + i];      /* node record: requested together with the control block */
    if (!phase_main(D.ctrl, h)) return;
    Ctrl *c = D.ctrl;
    const int d = e[0], nxi = ii > 0 ? e[1] : 0;
    /* rows in LDS: 0..d-1 the block, d..dp-1 identity padding (so that the padding columns stay inert), dp the right-hand
     * side, dp+1.. the Ut rows */
    const int dp = up16(d), R = dp + 1 + nxi, Rp = up16(R), ld = wide_ldf(Rp);
    lds_ptr Tm = to_lds(lds);                   /* ld x dp, column major; typed LDS pointer and 32-bit index arithmetic: ds_read / ds_write with
                                                   immediate offsets instead of flat_* with 64-bit address arithmetic per access */
    const double *W = D.W + e[8];
    const int bo = e[7];
    const double *Ut = D.Ut + e[9];
    const int r16 = lane & 15, g = lane >> 4;
    double *Lout = D.CholW + e[8], *CUt = D.CholUt + e[9];
    const int pos = e[11], ddim = e[12], xo = e[5];
    double *Wd = D.W + e[13];
    const int nt2 = ii > 0 ? (nxi + 1 + 15) >> 4 : 0;
    WSTAMP(0);
    for (int pass = 0; pass < 2; pass++) {
        const double shift = (O.regType == 1 || pass == 1) ? O.regValue : 0.0;         /* ddiare */
        if (tid == 0) small_flag = 0;
        /* rows on the lanes, columns dealt over the waves (j = wave + 4 m); 32-bit offsets from uniform bases, addresses
         * clamped instead of branched, so that all 32 loads of a thread are in flight together (the block was written
         * by other XCDs: ~2.4 us per dependent round trip).  Part A: the block itself (lower triangle), part B: the
         * right-hand side row and the Ut rows (LDS rows dp + rr, rr = lane). */
        {
            double va[16], vb[16];
            const int rr = lane;
            const double *bbase = rr == 0 ? D.resMod + bo : Ut + (rr - 1);
            const int bstride = rr == 0 ? 1 : nxi;
            const bool brow = rr <= nxi;
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const int j = wave + WW * m;
                const bool oka = lane < d && j < d && lane >= j, okb = brow && j < d;
                va[m] = W[oka ? lane + j * d : 0];
                vb[m] = bbase[okb ? j * bstride : 0];
            }
            LOADS_DONE();
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const int j = wave + WW * m;
                const bool oka = lane < d && j < d && lane >= j, okb = brow && j < d;
                const double xa = oka ? va[m] + (lane == j ? shift : 0.0) : ((lane == j && lane >= d) ? 1.0 : 0.0);   /* padding: identity */
                if (lane < dp && j < dp) Tm[lane + j * ld] = xa;
                if (dp + rr < Rp && j < dp) Tm[dp + rr + j * ld] = okb ? vb[m] : 0.0;
            }
        }
        __syncthreads();
        if (FUSED) {
            /* children that own a block (kid < Np) have posted G = Xt Xt' of their tall factor: subtract G[gi][gj], gj >= 1, from the
             * diagonal sub-block of W at the child's position, G[gi][0] from the right-hand side row.  Every thread polls its own
             * entries (at most ceil((nx + 1)^2 / 256) per child); distinct entries, distinct LDS words. */
            const int nkp = e[3], k0 = e[4];
            int posc = 0;
            bool dead = false;
            for (int cc = 0; cc < nkp; cc++) {
                const int kid = k0 + cc;
                const int nxc = cc == 0 ? e[16] : cc == 1 ? e[19] : cc == 2 ? e[22] : cc == 3 ? e[25] : T.nx[kid];
                if (kid < T.Np) {
                    const int w = nxc + 1;
                    const u64 *rec = sch + (size_t)kid * rs * 2;
                    for (int f = tid; f < w * w; f += WT) {
                        const int gi = f / w, gj = f - gi * w;
                        if (gi < 1 || gj > gi) continue;
                        double val = 0.0;
                        const u64 t0 = wall_clock64();
                        for (;;) {
                            bool ok = true;
                            val = ld_tag(rec + (size_t)f * 2, tag, ok);
                            if (ok || dead) break;
                            if (wall_clock64() - t0 > 50000000ull) { dead = true; val = 0.0; break; }      /* 0.5 s at 100 MHz: cannot happen (see above) */
                            __builtin_amdgcn_s_sleep(TQ_WIDE_NAP);
                        }
                        lds_ptr dst = gj == 0 ? Tm + dp + (posc + gi - 1) * ld : Tm + (posc + gi - 1) + (posc + gj - 1) * ld;
                        *dst -= val;
                    }
                }
                posc += nxc;
            }
            if (dead) { D.ctrl->status = 3; __hip_atomic_store(&D.ctrl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            __syncthreads();
        }
        WSTAMP(1);
        for (int kb = 0; kb < dp; kb += 16) {
            /* ---- panel kb: rows kb.. , columns kb..kb+15, one row per lane, in registers ---- */
            const int nbelow = Rp - kb - 16;
            const bool mine = wave == 0 || wave * 48 < nbelow;
            const int bi = wave * 48 + lane - 16;
            const bool diag = lane < 16, valid = diag || bi < nbelow;
            const int row = diag ? kb + lane : kb + 16 + (valid ? bi : 0);
            double Tr[16];
            if (mine) {
#pragma unroll
                for (int j = 0; j < 16; j++) { const double v = Tm[row + (kb + j) * ld]; Tr[j] = valid ? v : 0.0; }
            }
            __syncthreads();                       /* every wave holds its copy of the diagonal tile before wave 0 overwrites it */
            WSTAMP(2 + 3 * (kb >> 4));
            if (mine) {
                const double pmin = p_potrf_rows<16>(Tr, lane);
                if (valid && (wave == 0 || !diag)) {
#pragma unroll
                    for (int j = 0; j < 16; j++) Tm[row + (kb + j) * ld] = Tr[j];
                }
                if (wave == 0 && lane == 0 && pmin <= O.regTol * O.regTol) small_flag = 1;   /* sqrt(pivot) <= regTol, incl. non-positive pivots */
            }
            __syncthreads();
            WSTAMP(3 + 3 * (kb >> 4));
            /* ---- trailing update: tile (I, J) -= P_I P_J', J > kb/16, I >= J; transposed product so that the lanes
             * of a C access run down a column of the tile ---- */
            const int ct = dp >> 4, rt = Rp >> 4, kt = kb >> 4;
            int t = 0;
            for (int J = kt + 1; J < ct; J++) {
                for (int I = J; I < rt; I++, t++) {
                    if ((t & (WW - 1)) != wave) continue;
                    f64x4 acc;
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[q] = Tm[16 * I + r16 + (16 * J + g + 4 * q) * ld];
#pragma unroll
                    for (int s = 0; s < 16; s += 4) {
                        const double a = -1.0 * Tm[16 * J + r16 + (kb + s + g) * ld];
                        const double b = Tm[16 * I + r16 + (kb + s + g) * ld];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) Tm[16 * I + r16 + (16 * J + g + 4 * q) * ld] = acc[q];
                }
            }
            __syncthreads();
            WSTAMP(4 + 3 * (kb >> 4));
        }
        WSTAMP(14);
        /* on-the-fly Levenberg-Marquardt: any diagonal entry <= regTol -> shift and refactorise */
        if (O.regType != 2 || pass == 1 || !small_flag) break;
        __syncthreads();
        if (tid == 0) atomicAdd(&c->n_reg, 1);
    }

    /* destination values of the Schur update (nobody else touches them during this launch), requested before the
     * outputs are written so that part of their latency is hidden; same tile walk as the update below (at most 15
     * lower tiles: 4 per wave).  (Requested at kernel start they cost 32 registers across the factorisation and
     * one workgroup per CU less.) */
    double pre[4][4];
    if (!FUSED) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            int J = 0, rem = wave + WW * mt;
            while (J < nt2 && rem >= nt2 - J) { rem -= nt2 - J; J++; }      /* tile number -> (I, J), lower tiles column by column */
            const int gi = 16 * (J + rem) + r16;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int gj = 16 * J + g + 4 * q;
                const bool ok = J < nt2 && gi >= 1 && gi <= nxi && gj <= gi;
                const double *src = !ok ? D.resMod + xo : (gj == 0 ? D.resMod + xo + gi - 1 : Wd + (pos + gi - 1) + (size_t)(pos + gj - 1) * ddim);
                pre[mt][q] = *src;
            }
        }
    }
    const double rv0 = D.res[bo + (lane < d ? lane : 0)];        /* root: residual for res' * dlam, requested early */
    LOADS_DONE();
    /* outputs: factor (rows on the lanes, columns dealt over the waves), reciprocal diagonal, backward solution, CholUt */
    auto write_outputs = [&]() {
        for (int j = wave; j < d; j += WW) {
            if (lane < d && lane >= j) Lout[lane + j * d] = Tm[lane + j * ld];
            if (ii > 0 && lane >= 1 && lane <= nxi) CUt[(lane - 1) + j * nxi] = Tm[dp + lane + j * ld];
        }
        for (int j = tid; j < d; j += WT) {
            const double l = Tm[j + j * ld];
            D.invd[bo + j] = l > 0.0 ? 1.0 / l : 0.0;
            if (ii > 0) D.dlam[bo + j] = Tm[dp + j * ld];
        }
    };
    if (!FUSED || ii == 0) write_outputs();      /* fused sweep: the Schur record first -- the parent is waiting for it, nobody in this launch for the rest */

    if (ii > 0) {
        /* Schur complement into the parent's diagonal sub-block and right-hand side: G = Xt Xt', Xt = rows dp .. R-1
         * (the backward solution y, then Ut L^-T); G[i][0] goes to resMod, G[i][j] (i >= j >= 1) to the parent's W.
         * MFMA tiles over the padded columns (zeros beyond d); the destination values were fetched at the start. */
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            int J = 0, rem = wave + WW * mt;
            while (J < nt2 && rem >= nt2 - J) { rem -= nt2 - J; J++; }
            if (J >= nt2) continue;
            const int I = J + rem;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            const int ra = dp + 16 * J + r16, rb = dp + 16 * I + r16;
            const bool oka = ra < R, okb = rb < R;
            for (int k = 0; k < dp; k += 4) {
                const double a = Tm[(oka ? ra : 0) + (k + g) * ld], b = Tm[(okb ? rb : 0) + (k + g) * ld];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(oka ? a : 0.0, okb ? b : 0.0, acc, 0, 0, 0);
            }
            const int gi = 16 * I + r16;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int gj = 16 * J + g + 4 * q;
                if (gi >= 1 && gi <= nxi && gj <= gi) {
                    if (FUSED) st_tag(sch + ((size_t)ii * rs + (size_t)gi * (nxi + 1) + gj) * 2, acc[q], tag);
                    else if (gj == 0) D.resMod[xo + gi - 1] = pre[mt][q] - acc[q];
                    else Wd[(pos + gi - 1) + (size_t)(pos + gj - 1) * ddim] = pre[mt][q] - acc[q];
                }
            }
        }
        if (FUSED) write_outputs();
    } else if (wave == 0) {
        /* root: dlam_0 = L^-T (L^-1 resMod_0), the vector in registers (entry j on lane j) */
        const int lc = lane < d ? lane : 0;
        const double l = Tm[lc + lc * ld];
        const double myinv = l > 0.0 ? 1.0 / l : 0.0;
        /* (one block per solve: the plain loop, not wide_backsolve's 64-register column) */
        double z = lane < d ? Tm[dp + lc * ld] : 0.0;
        for (int k = d - 1; k >= 1; k--) {
            const double lk = Tm[k + lc * ld];
            const double zk = rdlane(z * myinv, k);
            z = fma(lane < k ? -lk : 0.0, zk, z);
        }
        const double mine = z * myinv;
        double pd = 0.0;
        if (lane < d) { D.dlam[bo + lane] = mine; pd = rv0 * mine; }
        pd = wave_sum(pd);
        if (lane == 0) D.part_dot[0] = pd;
    }
    WSTAMP(15);
}

__global__ void __launch_bounds__(WT) k_factor_w(Tree T, Data D, Opts O, int first, int h)
#if !TQ_HAS(TQP_WIDE)
;
#else
{
    factor_w_body<false>(T, D, O, first + blockIdx.x, first, h, nullptr, 0, 0u);
}
#endif
/* the backward sweep of ALL levels as one launch: workgroup b takes block Np - 1 - b */
#ifndef TQ_WIDE_WPS
#define TQ_WIDE_WPS 2
#endif
__global__ void __launch_bounds__(WT, TQ_WIDE_WPS) k_factor_all_w(Tree T, Data D, Opts O, u64 *sch, int rs, unsigned tag, int h)
#if !TQ_HAS(TQP_WIDE)
;
#else
{
    factor_w_body<true>(T, D, O, T.Np - 1 - (int)blockIdx.x, T.Np - 1 - (int)blockIdx.x, h, sch, rs, tag);
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* forward substitution of one level:  dlam_ii = L^-T ( y_ii - CholUt_ii' * dlam_dad[pos..] )  */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(WT) k_forward_w(Tree T, Data D, int first, int h)
#if !TQ_HAS(TQP_WIDE)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ii = first + blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int e[10];
#pragma unroll
    for (int i = 0; i < 10; i++) e[i] = T.desc[(size_t)DESC_INTS * ii + i];      /* node record: requested together with the control block */
    if (!phase_main(D.ctrl, h)) return;
    const int d = e[0], nxi = e[1], ld = d | 1;
    lds_ptr L = to_lds(lds);              /* ld x d, lower part */
    lds_ptr zz = L + ld * d;              /* WW x 64 : per-wave partials of CholUt' * dlam_dad */
    const int bo = e[7], xo = e[5];
    const double *Lg = D.CholW + e[8];
    const int lc = lane < d ? lane : 0;
    /* everything this workgroup reads from global memory is requested up front (the data was written by other XCDs:
     * ~2.4 us per dependent round trip): the factor (rows on the lanes, columns dealt over the waves), this wave's
     * share of CholUt and of the parent's step, the reciprocal diagonal, the backward solution, the residual */
    double v[16], cu[16], dl[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { const int j = wave + WW * m; const bool ok = lane < d && j < d && lane >= j; v[m] = Lg[ok ? lane + (size_t)j * d : 0]; }
    const double *CUt = D.CholUt + e[9] + (size_t)lc * nxi;
#pragma unroll
    for (int m = 0; m < 16; m++) { const int i = wave + WW * m; const bool ok = i < nxi; cu[m] = CUt[ok ? i : 0]; dl[m] = D.dlam[xo + (ok ? i : 0)]; }
    const double myinv = D.invd[bo + lc], yv = D.dlam[bo + lc], rv = D.res[bo + lc];
    LOADS_DONE();
#pragma unroll
    for (int m = 0; m < 16; m++) { const int j = wave + WW * m; if (lane < d && j < d && lane >= j) L[lane + j * ld] = v[m]; }
    {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int m = 0; m < 16; m += 2) {
            a0 = fma(wave + WW * m < nxi ? cu[m] : 0.0, dl[m], a0);
            a1 = fma(wave + WW * (m + 1) < nxi ? cu[m + 1] : 0.0, dl[m + 1], a1);
        }
        zz[wave * 64 + lane] = a0 + a1;
    }
    __syncthreads();
    if (wave != 0) return;
    const double rhs = lane < d ? fma(-1.0, (zz[lane] + zz[64 + lane]) + (zz[128 + lane] + zz[192 + lane]), yv) : 0.0;
    const double mine = wide_backsolve(L, ld, d, lane, rhs, myinv);
    double pd = 0.0;
    if (lane < d) { D.dlam[bo + lane] = mine; pd = rv * mine; }
    pd = wave_sum(pd);
    if (lane == 0) D.part_dot[ii] = pd;
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* the forward sweep of ALL levels as one launch                                               */
/* ------------------------------------------------------------------------------------------ */
/* One launch per level is a kernel boundary (~3 us) plus a batch of cold loads (~2.4 us) per level for ~1 us of dependent
 * work.  Here every block below the root has its workgroup in ONE launch: it requests everything that does not depend on
 * its parent (factor, CholUt, backward solution, residual), then waits for the parent's step, which travels as tagged words
 * (st_tag / ld_tag of the persistent path: the consumer polls the payload itself until every word carries this launch's tag;
 * relaxed agent-scope accesses, so nothing depends on one XCD's L2 seeing another's).  Workgroups are numbered in BFS order
 * and the hardware starts them in order, so a waiting workgroup's parent is always running or done: no deadlock whatever
 * part of the grid is resident.  Children of the root read the root's step from D.dlam (k_factor_w wrote it in an earlier
 * launch).  A wait that never ends (it cannot) gives up after 0.5 s and ends the solve with UNKNOWN_ERROR. */
__global__ void __launch_bounds__(WT) k_forward_all_w(Tree T, Data D, u64 *fw, unsigned tag, int h, Fuse F)
#if !TQ_HAS(TQP_WIDE)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ii = 1 + blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = T.desc[(size_t)DESC_INTS * ii + i];
    if (!phase_main(D.ctrl, h)) return;
    const int d = e[0], nxi = e[1], ld = d | 1, dad = e[10];
    lds_ptr L = to_lds(lds);
    lds_ptr zz = L + ld * d;
    const int bo = e[7], xo = e[5];
    const double *Lg = D.CholW + e[8];
    const int lc = lane < d ? lane : 0;
    double v[16], cu[16], dl[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { const int j = wave + WW * m; const bool ok = lane < d && j < d && lane >= j; v[m] = Lg[ok ? lane + (size_t)j * d : 0]; }
    const double *CUt = D.CholUt + e[9] + (size_t)lc * nxi;
#pragma unroll
    for (int m = 0; m < 16; m++) { const int i = wave + WW * m; cu[m] = CUt[i < nxi ? i : 0]; }
    const double myinv = D.invd[bo + lc], yv = D.dlam[bo + lc], rv = D.res[bo + lc];
    LOADS_DONE();
#pragma unroll
    for (int m = 0; m < 16; m++) { const int j = wave + WW * m; if (lane < d && j < d && lane >= j) L[lane + j * ld] = v[m]; }
    __syncthreads();
    double Lc[64];                                        /* wave 0: its column of the factor, in registers before the wait */
    if (wave == 0) wide_backsolve_load(L, ld, d, lane, Lc);
    /* the parent's step: entries xo .. xo + nxi - 1 of the step vector (this wave's share: i = wave + 4 m) */
    if (dad == 0) {
#pragma unroll
        for (int m = 0; m < 16; m++) { const int i = wave + WW * m; dl[m] = D.dlam[xo + (i < nxi ? i : 0)]; }
    } else {
        /* lane m (mod 16) fetches entry i = wave + 4 m: ONE tagged double per lane and round (a workgroup's poll is 2 x nxi words,
         * not 32 per thread: hundreds of workgroups poll at once while the levels above them are still at work) */
        const int i = wave + WW * (lane & 15);
        const bool need = i < nxi;
        const u64 *src = fw + (size_t)(xo + (need ? i : wave)) * 2;
        double val = 0.0;
        const u64 t0 = wall_clock64();
        for (;;) {
            bool ok = true;
            val = ld_tag(src, tag, ok);
            if (__all(ok || !need)) break;
            if (wall_clock64() - t0 > 50000000ull) {                      /* 0.5 s at 100 MHz */
                if (tid == 0) { D.ctrl->status = 3; __hip_atomic_store(&D.ctrl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                break;
            }
            __builtin_amdgcn_s_sleep(TQ_WIDE_NAP);
        }
#pragma unroll
        for (int m = 0; m < 16; m++) dl[m] = rdlane(val, m);
    }
    {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int m = 0; m < 16; m += 2) {
            a0 = fma(wave + WW * m < nxi ? cu[m] : 0.0, dl[m], a0);
            a1 = fma(wave + WW * (m + 1) < nxi ? cu[m + 1] : 0.0, dl[m + 1], a1);
        }
        zz[wave * 64 + lane] = a0 + a1;
    }
    __syncthreads();
    if (wave != 0) return;
    const double rhs = lane < d ? fma(-1.0, (zz[lane] + zz[64 + lane]) + (zz[128 + lane] + zz[192 + lane]), yv) : 0.0;
    const double mine = wide_backsolve_chain(Lc, d, rhs, myinv);
    double pd = 0.0;
    if (lane < d) {
        st_tag(fw + (size_t)(bo + lane) * 2, mine, tag);                  /* first: my children wait for it */
        D.dlam[bo + lane] = mine; pd = rv * mine;
    }
    pd = wave_sum(pd);
    if (lane == 0) D.part_dot[ii] = pd;
    if (F.on) fuse_ls_begin(T, D, F, ii, lane);          /* small trees: k_ls_begin as the tail of the sweep (wave 0 is the one left here) */
}
#endif

/* LDS a block of dimension d (tall matrix of R rows, nz parent columns) needs in the wide kernels */
static inline size_t wide_lds_hess(int d, int nz) { const int dp = (d + 15) & ~15, kz = (nz + 3) & ~3; return (size_t)(dp | 16) * kz * sizeof(double); }
static inline int wide_rows(int d, int nxi) { const int dp = (d + 15) & ~15; return (dp + 1 + nxi + 15) & ~15; }      /* padded rows of the tall matrix */
static inline size_t wide_lds_factor(int d, int nxi) { const int dp = (d + 15) & ~15; return (size_t)wide_ldf(wide_rows(d, nxi)) * dp * sizeof(double); }
static inline size_t wide_lds_forward(int d) { return ((size_t)(d | 1) * d + WW * 64 + 2) * sizeof(double); }
