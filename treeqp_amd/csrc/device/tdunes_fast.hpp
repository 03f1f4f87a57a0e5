/*
 * tdunes_fast.hpp -- fused gfx950 kernels for UNIFORM complete trees (every node nx = NX, every
 * parent nu = NU and MD children, leaves at one depth): BASELINE configs C2 / C3.
 *
 * Included by tdunes_device.hip (shares Tree/Data/Ctrl/Opts and the phase guards).
 *
 * Why a second path: on this workload a Newton iteration is a chain of 2*levels dependent
 * block steps (SURVEY.md §7 "Latency, not bandwidth").  The generic path pays one kernel boundary
 * plus an LDS round-trip heavy block step per level (13 us per level measured, profiles/r01_v1_*).
 * Here
 *   - a dual-Hessian block lives in the REGISTERS of one wavefront while it is factorised: lane i
 *     owns row i of the tall matrix T = [W ; resMod' ; Ut] (D + 1 + NX <= 64 rows, D doubles per
 *     lane); the rank-1 updates read the pivot column of other rows with v_readlane_b32 (SGPR
 *     broadcast, no LDS in the dependency chain); the reciprocal pivot is v_rsq_f64 + two Newton
 *     steps instead of sqrt + divide (the latency-critical part of every pivot);
 *   - W = C P C' for a block is ONE 16x16 f64 MFMA tile (v_mfma_f64_16x16x4_f64, K = NX+NU padded
 *     to a multiple of 4), operands loaded straight from the packed [A B] edge data;
 *   - the tree is cut at level `lcut`: every subtree below the cut is owned by one workgroup that
 *     walks its levels bottom-up with workgroup barriers only (f_up), the levels above the cut are
 *     owned by a single workgroup (f_top), the forward sweep mirrors that (f_top, f_down).  One
 *     Newton iteration = 4 launches (f_up, f_top, f_down, k_ls_decide) instead of ~31;
 *   - children hand their Schur complement to the parent through small per-block buffers
 *     (Sbuf/vbuf) instead of read-modify-write on the parent's block, so no two workgroups ever
 *     write the same words;
 *   - the first line-search trial (tau = 1) is evaluated speculatively inside f_top/f_down; the
 *     accepted trial sweep doubles as phase S of the next iteration.
 *
 * Arithmetic follows the same operation order as the generic path except (i) the reciprocal
 * pivot (<= 1 ulp from 1/sqrt) and (ii) MFMA / cross-lane summation order in W and res; the parity
 * tests hold it to the same 1e-10 tolerance against the oracle.
 */
#pragma once

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define FAST_WAVES 16      /* waves per workgroup of the fused kernels (1024 threads) */

template <int NX, int NU, int MD>
struct Uni {
    static constexpr int D = NX * MD;        /* dual block dimension            */
    static constexpr int NZ = NX + NU;
    static constexpr int R = D + 1 + NX;     /* rows of the tall matrix         */
    static constexpr int KS = (NZ + 3) / 4;  /* MFMA k-steps                    */
    static constexpr int LDW = D + 1;        /* LDS row stride (bank spread)    */
    static constexpr int WAVE_LDS = (NX + 1) * (D + 1) + D + NX + 8;   /* doubles of LDS per wave */
    static_assert(D <= 16, "the MFMA tile path needs a dual block of at most 16 rows");
    static_assert(R <= 64, "tall matrix must fit one wavefront");
    __device__ static __forceinline__ int kid0(int k) { return MD * k + 1; }
    __device__ static __forceinline__ int dad(int k) { return (k - 1) / MD; }
    __device__ static __forceinline__ int bo(int p) { return NX * (MD * p + 1); }      /* block vector offset */
    __device__ static __forceinline__ int first(int level) {                            /* first node of a level */
        int n = 0, w = 1;
        for (int l = 0; l < level; l++) { n += w; w *= MD; }
        return n;
    }
    __device__ static __forceinline__ int width(int level) { int w = 1; for (int l = 0; l < level; l++) w *= MD; return w; }
};

__device__ __forceinline__ double rdlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

/* 1/sqrt(p) for p > 0, 0 otherwise (non-positive pivot -> zero column, as dpotrf_l) */
__device__ __forceinline__ double pivot_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    double e = fma(-(p * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(p * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return p > 0.0 ? y : 0.0;
}

/* diagnostic stamp: slot of kernel `kern` (0 f_up, 1 f_top, 2 f_down), workgroup 0 / thread 0 only */
__device__ __forceinline__ void stamp(const Data &Dt, const Opts &O, int kern, int slot) {
    if (O.stamps && threadIdx.x == 0 && blockIdx.x == 0 && slot < 64) {
        Dt.stamps[(kern * 64 + slot) * 2 + 0] = clock64();
        Dt.stamps[(kern * 64 + slot) * 2 + 1] = wall_clock64();
    }
}

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

/* ------------------------------------------------------------------------------------------ */
/* G + H for one parent block p (one wave): res/resMod of its children, W_p, Ut_p             */
/* ------------------------------------------------------------------------------------------ */
template <int NX, int NU, int MD>
__device__ __forceinline__ void fast_gh(const Data &Dt, int p, int lane, int termCondition) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    const int row = lane & 15, g = lane >> 4;
    const int cidx = row / NX, r = row - cidx * NX;
    const int k = U::kid0(p) + cidx;                     /* child owning this row */
    const bool live = row < D;
    const double *A = Dt.A + (size_t)(k - 1) * NX * NX + r;
    const double *B = Dt.B + (size_t)(k - 1) * NX * NU + r;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;                                   /* this lane's share of (A x_p + B u_p)[row] */
    double *Ut = Dt.Ut + (size_t)(p > 0 ? p - 1 : 0) * NX * D;
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        double a = 0.0, pc = 0.0, z = 0.0;
        if (live && cc < NZ) {
            if (cc < NX) { a = A[(size_t)cc * NX]; pc = Dt.QinvCal[NX * p + cc]; z = Dt.x[NX * p + cc]; }
            else { a = B[(size_t)(cc - NX) * NX]; pc = Dt.RinvCal[NU * p + cc - NX]; z = Dt.u[NU * p + cc - NX]; }
        }
        const double ap = a * pc;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ap, acc, 0, 0, 0);
        part = fma(a, z, part);
        if (p > 0 && live && cc < NX) Ut[cc + (size_t)row * NX] = -1.0 * ap;      /* Ut = -(A Qcal)' */
    }
    /* residual: reduce the 4 k-groups of a row (lanes row, row+16, row+32, row+48) */
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    const int bo = U::bo(p);
    double e = 0.0;
    if (live && g == 0) {
        const double rv = fma(-1.0, Dt.x[bo + row], Dt.b[bo + row]) + part;
        Dt.res[bo + row] = rv;
        Dt.resMod[bo + row] = rv;
        e = (termCondition == 2) ? fabs(rv) : rv * rv;
        Dt.part_err[bo + row] = e;
    }
    /* W tile: lane holds W[i = g + 4 rr][j = row] */
    double *W = Dt.W + (size_t)p * D * D;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = g + 4 * rr;
        if (live && i < D) {
            double w = acc[rr];
            if (i == row) w += Dt.QinvCal[bo + i];
            W[i + (size_t)row * D] = w;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* factor one block (one wave, block in registers)                                            */
/* ------------------------------------------------------------------------------------------ */
template <int NX, int NU, int MD>
__device__ __forceinline__ void fast_factor(const Data &Dt, const Opts &O, int ii, int Np, int lane, double *lds, bool is_root) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R, LDW = U::LDW;
    const int bo = U::bo(ii);
    const bool has_parent_kids = U::kid0(ii) < Np;        /* children are parents themselves -> Schur inputs */
    /* per-lane row source */
    const double *src; int stride;
    if (lane < D) { src = Dt.W + (size_t)ii * D * D + lane; stride = D; }
    else if (lane == D) { src = Dt.resMod + bo; stride = 1; }
    else if (lane < R && !is_root) { src = Dt.Ut + (size_t)(ii - 1) * NX * D + (lane - D - 1); stride = NX; }
    else { src = Dt.W + (size_t)ii * D * D; stride = D; }
    double T[D];
    double myinv = 0.0;
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int j = 0; j < D; j++) T[j] = src[(size_t)j * stride];
        if (has_parent_kids) {
            /* subtract the children's Schur complements: rows of child c get -S_c in its column
             * range, the right-hand side row gets -v_c */
#pragma unroll
            for (int c = 0; c < MD; c++) {
                const int kid = U::kid0(ii) + c;
                const double *S = Dt.Sbuf + (size_t)kid * NX * NX;
                const double *v = Dt.vbuf + (size_t)kid * NX;
                const int r = lane - c * NX;
                if (lane < D && r >= 0 && r < NX) {
#pragma unroll
                    for (int j = 0; j < NX; j++) T[c * NX + j] -= S[r + (size_t)j * NX];
                }
                if (lane == D) {
#pragma unroll
                    for (int j = 0; j < NX; j++) T[c * NX + j] -= v[j];
                }
            }
        }
        if (O.regType == 1 || pass == 1) {
#pragma unroll
            for (int j = 0; j < D; j++) if (lane == j) T[j] += O.regValue;           /* ddiare */
        }
        int small = 0;
#pragma unroll
        for (int j = 0; j < D; j++) {
            const double pj = rdlane(T[j], j);
            const double finv = pivot_rsqrt(pj);
            small |= (pj * finv <= O.regTol);
            T[j] *= finv;
            if (lane == j) myinv = finv;
#pragma unroll
            for (int c = j + 1; c < D; c++) {
                const double lc = rdlane(T[j], c);
                T[c] = fma(-T[j], lc, T[c]);
            }
        }
        if (O.regType != 2 || pass == 1 || !small) break;
        if (lane == 0) atomicAdd(&Dt.ctrl->n_reg, 1);
    }
    /* stores: factor rows, reciprocal diagonal */
    if (lane < D) {
        double *L = Dt.CholW + (size_t)ii * D * D + lane;
#pragma unroll
        for (int j = 0; j < D; j++) L[(size_t)j * D] = T[j];
        Dt.invd[bo + lane] = myinv;
    }
    if (!is_root) {
        if (lane == D) {
#pragma unroll
            for (int j = 0; j < D; j++) Dt.dlam[bo + j] = T[j];
        }
        if (lane > D && lane < R) {
            double *CUt = Dt.CholUt + (size_t)(ii - 1) * NX * D + (lane - D - 1);
#pragma unroll
            for (int j = 0; j < D; j++) CUt[(size_t)j * NX] = T[j];
        }
        /* Schur complement for the parent: S = CUt CUt', v = CUt y, through LDS (rows D..R-1) */
        if (lane >= D && lane < R) {
#pragma unroll
            for (int j = 0; j < D; j++) lds[(lane - D) * LDW + j] = T[j];
        }
        wave_lds_fence();
        if (lane < NX * NX) {
            const int i = lane % NX, j = lane / NX;
            const double *ri = lds + (1 + i) * LDW, *rj = lds + (1 + j) * LDW;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < D; c++) acc = fma(ri[c], rj[c], acc);
            Dt.Sbuf[(size_t)ii * NX * NX + i + (size_t)j * NX] = acc;
        }
        if (lane < NX) {
            const double *ri = lds + (1 + lane) * LDW;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < D; c++) acc = fma(ri[c], lds[c], acc);
            Dt.vbuf[(size_t)ii * NX + lane] = acc;
        }
        wave_lds_fence();
    } else {
        /* root: dlam_0 = L^-T y, k descending; y and L rows are broadcast with readlane */
        double y[D];
#pragma unroll
        for (int j = 0; j < D; j++) y[j] = rdlane(T[j], D);
        double mine = 0.0;
#pragma unroll
        for (int k = D - 1; k >= 0; k--) {
            const double zk = y[k] * rdlane(myinv, k);
            if (lane == k) mine = zk;
#pragma unroll
            for (int i = 0; i < k; i++) y[i] = fma(-rdlane(T[i], k), zk, y[i]);
        }
        double pd = 0.0;
        if (lane < D) { Dt.dlam[bo + lane] = mine; pd = Dt.res[bo + lane] * mine; }
        pd = wave_sum(pd);
        if (lane == 0) Dt.part_dot[0] = pd;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* forward substitution of one block (one wave): lane i owns column i of L                     */
/* ------------------------------------------------------------------------------------------ */
template <int NX, int NU, int MD>
__device__ __forceinline__ void fast_forward(const Data &Dt, int ii, int lane) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int bo = U::bo(ii), xo = NX * ii;
    const int li = lane < D ? lane : 0;
    const double *Lc = Dt.CholW + (size_t)ii * D * D + (size_t)li * D;
    const double *Cc = Dt.CholUt + (size_t)(ii - 1) * NX * D + (size_t)li * NX;
    double Lcol[D];
#pragma unroll
    for (int k = 0; k < D; k++) Lcol[k] = Lc[k];
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < NX; r++) acc = fma(Cc[r], Dt.dlam[xo + r], acc);
    double s = fma(-1.0, acc, Dt.dlam[bo + li]);
    const double inv = Dt.invd[bo + li];
    double mine = 0.0;
#pragma unroll
    for (int k = D - 1; k >= 0; k--) {
        const double zk = rdlane(s * inv, k);
        if (lane == k) mine = zk;
        if (lane < k) s = fma(-Lcol[k], zk, s);
    }
    double pd = 0.0;
    if (lane < D) { Dt.dlam[bo + lane] = mine; pd = Dt.res[bo + lane] * mine; }
    pd = wave_sum(pd);
    if (lane == 0) Dt.part_dot[ii] = pd;
}

/* ------------------------------------------------------------------------------------------ */
/* stage QP of one node at the trial point lam_cur + step * dlam (one wave); writes lam_next  */
/* ------------------------------------------------------------------------------------------ */
template <int NX, int NU, int MD>
__device__ __forceinline__ void fast_stage(const Data &Dt, int k, int Np, int lane, double *lds, double step,
                                           const double *lamc, double *lamn) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    const bool parent = k < Np;
    const int nuk = parent ? NU : 0;
    const int xo = NX * k, uo = NU * k, ko = U::bo(k);
    double *lk = lds, *lown = lds + D;
    if (parent && lane < D) lk[lane] = fma(step, Dt.dlam[ko + lane], lamc[ko + lane]);
    if (lane < NX) {
        double v = 0.0;
        if (k > 0) { v = fma(step, Dt.dlam[xo + lane], lamc[xo + lane]); lamn[xo + lane] = v; }
        lown[lane] = v;
    }
    wave_lds_fence();
    double p_qx = 0.0, p_hx = 0.0, p_ru = 0.0, p_hu = 0.0, p_c = 0.0;
    if (lane < NX + nuk) {
        const bool isx = lane < NX;
        const int j = isx ? lane : lane - NX;
        double v = isx ? fma(-1.0, Dt.q[xo + j], lown[j]) : -1.0 * Dt.r[uo + j];
        if (parent) {
#pragma unroll
            for (int cc = 0; cc < MD; cc++) {
                const int kid = U::kid0(k) + cc;
                const double *col = isx ? Dt.A + (size_t)(kid - 1) * NX * NX + (size_t)j * NX
                                        : Dt.B + (size_t)(kid - 1) * NX * NU + (size_t)j * NX;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NX; i++) acc = fma(col[i], lk[cc * NX + i], acc);
                v = fma(-1.0, acc, v);
            }
        }
        if (isx) {
            Dt.qmod[xo + j] = v;
            const double qi = Dt.Qinv[xo + j];
            const double unc = qi * v, lo = Dt.xmin[xo + j], hi = Dt.xmax[xo + j];
            double xv, cal;
            if (unc >= hi) { xv = hi; cal = 0.0; } else if (unc <= lo) { xv = lo; cal = 0.0; } else { xv = unc; cal = qi; }
            Dt.xUnc[xo + j] = unc; Dt.x[xo + j] = xv; Dt.QinvCal[xo + j] = cal;
            p_qx = (Dt.Qd[xo + j] * xv) * xv;
            p_hx = v * xv;
        } else {
            Dt.rmod[uo + j] = v;
            const double ri = Dt.Rinv[uo + j];
            const double unc = ri * v, lo = Dt.umin[uo + j], hi = Dt.umax[uo + j];
            double uv, cal;
            if (unc >= hi) { uv = hi; cal = 0.0; } else if (unc <= lo) { uv = lo; cal = 0.0; } else { uv = unc; cal = ri; }
            Dt.uUnc[uo + j] = unc; Dt.u[uo + j] = uv; Dt.RinvCal[uo + j] = cal;
            p_ru = (Dt.Rd[uo + j] * uv) * uv;
            p_hu = v * uv;
        }
    }
    if (parent && lane < D) p_c = Dt.b[ko + lane] * lk[lane];
    p_qx = wave_sum(p_qx); p_hx = wave_sum(p_hx); p_ru = wave_sum(p_ru); p_hu = wave_sum(p_hu); p_c = wave_sum(p_c);
    if (lane == 0) {
        double f = -0.5 * p_qx - p_c;
        f += p_hx;
        f -= 0.5 * p_ru;
        f += p_hu;
        Dt.fval[k] = f;
    }
    wave_lds_fence();
}

/* ------------------------------------------------------------------------------------------ */
/* fused kernels                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* f_up: one workgroup per subtree rooted at level lcut: G+H for its blocks (and a share of the
 * top blocks), then the backward sweep over its levels, bottom-up. */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FAST_WAVES * WAVE) f_up(Tree T, Data Dt, Opts O, int lcut, int h) {
    using U = Uni<NX, NU, MD>;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    if (!phase_main(Dt.ctrl, h)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *lds = lds_all + wave * U::WAVE_LDS;
    const int s = blockIdx.x, Nh = T.Nh, depth = Nh - lcut;       /* block levels lcut .. Nh-1 */
    int sl = 0;
    stamp(Dt, O, 0, sl++);
    for (int t = 0; t < depth; t++) {
        const int nb = U::width(t), f0 = U::first(lcut + t) + s * nb;
        for (int b = wave; b < nb; b += FAST_WAVES) fast_gh<NX, NU, MD>(Dt, f0 + b, lane, O.termCondition);
    }
    const int ntop = U::first(lcut);
    for (int p = s * FAST_WAVES + wave; p < ntop; p += gridDim.x * FAST_WAVES) fast_gh<NX, NU, MD>(Dt, p, lane, O.termCondition);
    __syncthreads();
    stamp(Dt, O, 0, sl++);
    for (int t = depth - 1; t >= 0; t--) {
        const int nb = U::width(t), f0 = U::first(lcut + t) + s * nb;
        for (int b = wave; b < nb; b += FAST_WAVES) fast_factor<NX, NU, MD>(Dt, O, f0 + b, T.Np, lane, lds, false);
        __syncthreads();
        stamp(Dt, O, 0, sl++);
    }
}

/* f_top: single workgroup: termination test, levels above the cut backward + root + forward,
 * then the speculative first line-search trial (tau = 1) for the nodes above the cut. */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FAST_WAVES * WAVE) f_top(Tree T, Data Dt, Opts O, int lcut, int h) {
    using U = Uni<NX, NU, MD>;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    __shared__ double sh[FAST_WAVES * WAVE];
    __shared__ int stop;
    Ctrl *c = Dt.ctrl;
    if (!phase_main(c, h)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *lds = lds_all + wave * U::WAVE_LDS;
    int sl = 0;
    stamp(Dt, O, 1, sl++);
    /* termination test (calculate_error_in_residuals + :542-546) */
    {
        const int n0 = NX, n1 = NX * T.Nn;
        double err = (O.termCondition == 2) ? block_reduce<true>(Dt.part_err + n0, n1 - n0, sh)
                                            : block_reduce<false>(Dt.part_err + n0, n1 - n0, sh);
        if (threadIdx.x == 0) {
            if (O.termCondition == 1) err = sqrt(err);
            c->err = err;
            stop = err < O.tol;
            if (stop) { c->done = 1; c->status = 0; }
        }
        __syncthreads();
        if (stop) return;
    }
    stamp(Dt, O, 1, sl++);
    for (int l = lcut - 1; l >= 1; l--) {
        const int nb = U::width(l), f0 = U::first(l);
        for (int b = wave; b < nb; b += FAST_WAVES) fast_factor<NX, NU, MD>(Dt, O, f0 + b, T.Np, lane, lds, false);
        __syncthreads();
        stamp(Dt, O, 1, sl++);
    }
    if (wave == 0) fast_factor<NX, NU, MD>(Dt, O, 0, T.Np, lane, lds, true);
    __syncthreads();
    stamp(Dt, O, 1, sl++);
    for (int l = 1; l < lcut; l++) {
        const int nb = U::width(l), f0 = U::first(l);
        for (int b = wave; b < nb; b += FAST_WAVES) fast_forward<NX, NU, MD>(Dt, f0 + b, lane);
        __syncthreads();
        stamp(Dt, O, 1, sl++);
    }
    /* first trial: nodes of levels 0 .. lcut-1 (their own and their children's duals are final) */
    const double *lamc = c->cur ? Dt.lam1 : Dt.lam0;
    double *lamn = c->cur ? Dt.lam0 : Dt.lam1;
    const int ntop = U::first(lcut);
    for (int k = wave; k < ntop; k += FAST_WAVES) fast_stage<NX, NU, MD>(Dt, k, T.Np, lane, lds, 1.0, lamc, lamn);
    __syncthreads();
    stamp(Dt, O, 1, sl++);
    if (threadIdx.x == 0) { c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1; }
}

/* f_down: one workgroup per subtree: forward sweep top-down, then the first trial for its nodes */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FAST_WAVES * WAVE) f_down(Tree T, Data Dt, Opts O, int lcut, int h) {
    using U = Uni<NX, NU, MD>;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const Ctrl *c = Dt.ctrl;
    if (!phase_trial(c, h, 1)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *lds = lds_all + wave * U::WAVE_LDS;
    const int s = blockIdx.x, Nh = T.Nh, depth = Nh - lcut;
    int sl = 0;
    stamp(Dt, O, 2, sl++);
    for (int t = 0; t < depth; t++) {
        const int nb = U::width(t), f0 = U::first(lcut + t) + s * nb;
        for (int b = wave; b < nb; b += FAST_WAVES) fast_forward<NX, NU, MD>(Dt, f0 + b, lane);
        __syncthreads();
        stamp(Dt, O, 2, sl++);
    }
    const double *lamc = c->cur ? Dt.lam1 : Dt.lam0;
    double *lamn = c->cur ? Dt.lam0 : Dt.lam1;
    for (int t = 0; t <= depth; t++) {                 /* node levels lcut .. Nh (leaves included) */
        const int nb = U::width(t), f0 = U::first(lcut + t) + s * nb;
        for (int b = wave; b < nb; b += FAST_WAVES) fast_stage<NX, NU, MD>(Dt, f0 + b, T.Np, lane, lds, 1.0, lamc, lamn);
    }
    __syncthreads();
    stamp(Dt, O, 2, sl++);
}
