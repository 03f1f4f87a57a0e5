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
 *     tile); the trailing update T22 -= L21 L21' is four MFMAs per 16 x 16 tile.  d = 60: ~9 us per level
 *     instead of ~72 us for the column-by-column LDS version;
 *   - substitutions (root solve, forward sweep) keep the vector in registers, one entry per lane, and
 *     broadcast z_k by v_readlane: one LDS read and one FMA per step, no barrier.
 * Reference: calculate_hessian_blocks / build W (dual_Newton_tree.c:531-760), factorise + substitute
 * (:763-913).  Results differ from the wave-per-block kernels by summation order only.
 */
#pragma once

#define WW 4
#define WT (WW * WAVE)

__device__ __forceinline__ int up16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ int wide_ld(int rows_padded) { return rows_padded | 16; }

/* ------------------------------------------------------------------------------------------ */
/* H                                                                                          */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(WT) k_hess_w(Tree T, Data D, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int d = T.bdim[p], nxp = T.nx[p], nup = T.nu[p], nz = nxp + nup;
    const int dp = up16(d), kz = (nz + 3) & ~3, ldc = wide_ld(dp);
    double *Cs = lds, *CP = lds + (size_t)ldc * kz;
    const int k0 = T.kid0[p], ko = T.xoff[k0];
    const double *Qc = D.QinvCal + T.xoff[p], *Rc = D.RinvCal + T.uoff[p];
    for (int e = tid; e < 2 * ldc * kz; e += WT) lds[e] = 0.0;
    __syncthreads();
    int rowoff = 0;
    for (int cc = 0; cc < T.nk[p]; cc++) {
        const int kid = k0 + cc, nxc = T.nx[kid];
        const double *A = D.A + T.aoff[kid], *B = D.B + T.boff[kid];
        for (int e = tid; e < nxc * nz; e += WT) {
            const int i = e % nxc, col = e / nxc;
            const double a = col < nxp ? A[i + (size_t)col * nxc] : B[i + (size_t)(col - nxp) * nxc];
            const double pc = col < nxp ? Qc[col] : Rc[col - nxp];
            Cs[rowoff + i + (size_t)col * ldc] = a;
            CP[rowoff + i + (size_t)col * ldc] = a * pc;
        }
        rowoff += nxc;
    }
    __syncthreads();
    /* lower tiles (I >= J); computed transposed (A = tile J of CP, B = tile I of C) so that a lane's results lie in
     * one row of W and lanes run down a column: coalesced stores */
    const int nt = dp >> 4, r = lane & 15, g = lane >> 4;
    double *W = D.W + T.woff[p];
    int t = 0;
    for (int J = 0; J < nt; J++) {
        for (int I = J; I < nt; I++, t++) {
            if ((t & (WW - 1)) != wave) continue;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int s = 0; s < kz; s += 4) {
                const double a = CP[16 * J + r + (size_t)(s + g) * ldc];
                const double b = Cs[16 * I + r + (size_t)(s + g) * ldc];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            const int i = 16 * I + r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = 16 * J + g + 4 * q;
                if (i < d && j <= i) W[i + (size_t)j * d] = acc[q] + (i == j ? D.QinvCal[ko + i] : 0.0);
            }
        }
    }
    if (p > 0) {
        double *Ut = D.Ut + T.utoff[p];
        for (int e = tid; e < nxp * d; e += WT) {
            const int i = e % nxp, rr = e / nxp;
            Ut[i + (size_t)rr * nxp] = -1.0 * CP[rr + (size_t)i * ldc];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* F: tall Cholesky of [W ; resMod' ; Ut], Schur complement into the parent, root solve        */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(WT) k_factor_w(Tree T, Data D, Opts O, int first, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int small_flag;
    if (!phase_main(D.ctrl, h)) return;
    Ctrl *c = D.ctrl;
    const int ii = first + blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int d = T.bdim[ii], nxi = ii > 0 ? T.nx[ii] : 0;
    const int R = d + 1 + nxi, dp = up16(d), Rp = up16(R), ld = wide_ld(Rp);
    double *Tm = lds;                           /* ld x dp, column major */
    const double *W = D.W + T.woff[ii];
    const int bo = T.xoff[T.kid0[ii]];
    const double *Ut = D.Ut + T.utoff[ii];
    const int r16 = lane & 15, g = lane >> 4;

    for (int pass = 0; pass < 2; pass++) {
        const double shift = (O.regType == 1 || pass == 1) ? O.regValue : 0.0;         /* ddiare */
        if (tid == 0) small_flag = 0;
        for (int e = tid; e < ld * dp; e += WT) {
            const int i = e % ld, j = e / ld;
            double v = 0.0;
            if (j < d) {
                if (i < d) { if (i >= j) v = W[i + (size_t)j * d] + (i == j ? shift : 0.0); }
                else if (i == d) v = D.resMod[bo + j];
                else if (i < R) v = Ut[(i - d - 1) + (size_t)j * nxi];
            } else if (i == j) v = 1.0;                                                /* padding columns: unit pivots */
            Tm[i + (size_t)j * ld] = v;
        }
        __syncthreads();
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
                for (int j = 0; j < 16; j++) { const double v = Tm[row + (size_t)(kb + j) * ld]; Tr[j] = valid ? v : 0.0; }
            }
            __syncthreads();                       /* every wave holds its copy of the diagonal tile before wave 0 overwrites it */
            if (mine) {
                const double pmin = p_potrf_rows<16>(Tr, lane);
                if (valid && (wave == 0 || !diag)) {
#pragma unroll
                    for (int j = 0; j < 16; j++) Tm[row + (size_t)(kb + j) * ld] = Tr[j];
                }
                if (wave == 0 && lane == 0 && pmin <= O.regTol * O.regTol) small_flag = 1;   /* sqrt(pivot) <= regTol, incl. non-positive pivots */
            }
            __syncthreads();
            /* ---- trailing update: tile (I, J) -= P_I P_J', J > kb/16, I >= J; transposed product so that the lanes
             * of a C access run down a column of the tile ---- */
            const int ct = dp >> 4, rt = Rp >> 4, kt = kb >> 4;
            int t = 0;
            for (int J = kt + 1; J < ct; J++) {
                for (int I = J; I < rt; I++, t++) {
                    if ((t & (WW - 1)) != wave) continue;
                    f64x4 acc;
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[q] = Tm[16 * I + r16 + (size_t)(16 * J + g + 4 * q) * ld];
#pragma unroll
                    for (int s = 0; s < 16; s += 4) {
                        const double a = -1.0 * Tm[16 * J + r16 + (size_t)(kb + s + g) * ld];
                        const double b = Tm[16 * I + r16 + (size_t)(kb + s + g) * ld];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) Tm[16 * I + r16 + (size_t)(16 * J + g + 4 * q) * ld] = acc[q];
                }
            }
            __syncthreads();
        }
        /* on-the-fly Levenberg-Marquardt: any diagonal entry <= regTol -> shift and refactorise */
        if (O.regType != 2 || pass == 1 || !small_flag) break;
        __syncthreads();
        if (tid == 0) atomicAdd(&c->n_reg, 1);
    }

    /* outputs: factor, reciprocal diagonal */
    double *L = D.CholW + T.woff[ii];
    for (int e = tid; e < d * d; e += WT) {
        const int i = e % d, j = e / d;
        if (i >= j) L[i + (size_t)j * d] = Tm[i + (size_t)j * ld];
    }
    for (int j = tid; j < d; j += WT) { const double l = Tm[j + (size_t)j * ld]; D.invd[bo + j] = l > 0.0 ? 1.0 / l : 0.0; }

    if (ii > 0) {
        for (int j = tid; j < d; j += WT) D.dlam[bo + j] = Tm[d + (size_t)j * ld];
        double *CUt = D.CholUt + T.utoff[ii];
        for (int e = tid; e < nxi * d; e += WT) {
            const int i = e % nxi, j = e / nxi;
            CUt[i + (size_t)j * nxi] = Tm[d + 1 + i + (size_t)j * ld];
        }
        /* Schur complement into the parent's diagonal sub-block and right-hand side */
        const int dd = T.dad[ii], pos = T.pos[ii], ddim = T.bdim[dd];
        double *Wd = D.W + T.woff[dd];
        for (int e = tid; e < nxi * nxi; e += WT) {
            const int i = e % nxi, j = e / nxi;
            if (i < j) continue;
            double a0 = 0.0, a1 = 0.0;
            int cidx = 0;
            for (; cidx + 1 < d; cidx += 2) {
                a0 = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + 1 + j + (size_t)cidx * ld], a0);
                a1 = fma(Tm[d + 1 + i + (size_t)(cidx + 1) * ld], Tm[d + 1 + j + (size_t)(cidx + 1) * ld], a1);
            }
            if (cidx < d) a0 = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + 1 + j + (size_t)cidx * ld], a0);
            Wd[(pos + i) + (size_t)(pos + j) * ddim] -= a0 + a1;
        }
        const int xo = T.xoff[ii];
        for (int i = tid; i < nxi; i += WT) {
            double acc = 0.0;
            for (int cidx = 0; cidx < d; cidx++) acc = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + (size_t)cidx * ld], acc);
            D.resMod[xo + i] -= acc;
        }
    } else if (wave == 0) {
        /* root: dlam_0 = L^-T (L^-1 resMod_0), the vector in registers (entry j on lane j) */
        const int lc = lane < d ? lane : 0;
        const double l = Tm[lc + (size_t)lc * ld];
        const double myinv = l > 0.0 ? 1.0 / l : 0.0;
        double z = lane < d ? Tm[d + (size_t)lc * ld] : 0.0;
        for (int k = d - 1; k >= 1; k--) {
            const double lk = Tm[k + (size_t)lc * ld];
            const double zk = rdlane(z * myinv, k);
            z = fma(lane < k ? -lk : 0.0, zk, z);
        }
        const double mine = z * myinv;
        double pd = 0.0;
        if (lane < d) { D.dlam[bo + lane] = mine; pd = D.res[bo + lane] * mine; }
        pd = wave_sum(pd);
        if (lane == 0) D.part_dot[0] = pd;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* forward substitution of one level:  dlam_ii = L^-T ( y_ii - CholUt_ii' * dlam_dad[pos..] )  */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(WT) k_forward_w(Tree T, Data D, int first, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    const int ii = first + blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int d = T.bdim[ii], nxi = T.nx[ii], ld = d | 1;
    double *L = lds;                      /* ld x d, lower part */
    double *zz = lds + (size_t)ld * d;    /* d : right-hand side */
    const int bo = T.xoff[T.kid0[ii]], xo = T.xoff[ii];
    const double *Lg = D.CholW + T.woff[ii];
    for (int e = tid; e < d * d; e += WT) {
        const int i = e % d, j = e / d;
        if (i >= j) L[i + (size_t)j * ld] = Lg[i + (size_t)j * d];
    }
    const double *CUt = D.CholUt + T.utoff[ii];
    for (int j = tid; j < d; j += WT) {
        double a0 = 0.0, a1 = 0.0;
        int i = 0;
        for (; i + 1 < nxi; i += 2) { a0 = fma(CUt[i + (size_t)j * nxi], D.dlam[xo + i], a0); a1 = fma(CUt[i + 1 + (size_t)j * nxi], D.dlam[xo + i + 1], a1); }
        if (i < nxi) a0 = fma(CUt[i + (size_t)j * nxi], D.dlam[xo + i], a0);
        zz[j] = fma(-1.0, a0 + a1, D.dlam[bo + j]);
    }
    __syncthreads();
    if (wave != 0) return;
    const int lc = lane < d ? lane : 0;
    const double myinv = D.invd[bo + lc];
    double z = lane < d ? zz[lc] : 0.0;
    for (int k = d - 1; k >= 1; k--) {
        const double lk = L[k + (size_t)lc * ld];
        const double zk = rdlane(z * myinv, k);
        z = fma(lane < k ? -lk : 0.0, zk, z);
    }
    const double mine = z * myinv;
    double pd = 0.0;
    if (lane < d) { D.dlam[bo + lane] = mine; pd = D.res[bo + lane] * mine; }
    pd = wave_sum(pd);
    if (lane == 0) D.part_dot[ii] = pd;
}

/* LDS a block of dimension d (tall matrix of R rows, nz parent columns) needs in the wide kernels */
static inline size_t wide_lds_hess(int d, int nz) { const int dp = (d + 15) & ~15, kz = (nz + 3) & ~3; return (size_t)2 * (dp | 16) * kz * sizeof(double); }
static inline size_t wide_lds_factor(int d, int R) { const int dp = (d + 15) & ~15, Rp = (R + 15) & ~15; return (size_t)(Rp | 16) * dp * sizeof(double); }
static inline size_t wide_lds_forward(int d) { return ((size_t)(d | 1) * d + d + 2) * sizeof(double); }
