/*
 * tdunes_fast.hpp -- fused gfx950 kernels for UNIFORM complete trees (every node nx = NX, every
 * parent nu = NU and MD children, leaves at one depth): BASELINE configs C2 / C3.
 *
 * Included by tdunes_device.hip (shares Tree/Data/Ctrl/Opts and the phase guards).
 *
 * Why a second path: on this workload a Newton iteration is a chain of 2*levels dependent block
 * steps (SURVEY.md §7 "Latency, not bandwidth").  The generic path pays one kernel boundary plus
 * an LDS round-trip heavy block step per level (13 us per level, profiles/r01_v1_*).  Here
 *   - a dual-Hessian block lives in the REGISTERS of one wavefront while it is factorised: lane i
 *     owns row i of the tall matrix T = [W ; resMod' ; Ut] (D + 1 + NX <= 64 rows, D doubles per
 *     lane); row j of the factor is broadcast with v_readlane_b32 (SGPR operand of the FMAs, no
 *     LDS in the dependency chain; measured 3.3k cycles per 25x16 block against 6.3k-12.9k for
 *     LDS / ds_bpermute broadcasts, tools/microbench/potrf_bench.hip); the reciprocal pivot is
 *     v_rsq_f64 + two Newton steps instead of sqrt + divide;
 *   - W = C P C' of a block is ONE 16x16 f64 MFMA tile (v_mfma_f64_16x16x4_f64, K = NX+NU padded
 *     to a multiple of 4), operands loaded straight from the packed [A B] edge data;
 *   - the block levels are grouped into TIERS of TH levels (MD^(TH-1) <= 4 blocks at the widest
 *     level of a tier subtree): one 4-wave workgroup owns one tier subtree, one wave per block, one
 *     wave per SIMD; inside a tier children hand their Schur complement to the parent through LDS
 *     and the next level's rows are prefetched before the workgroup barrier; tiers are separate
 *     launches (f_back ... f_top ... f_fwd), a kernel boundary costs about one block step;
 *   - the first line-search trial (tau = 1) is evaluated speculatively by f_stage right after the
 *     forward sweep; the accepted trial sweep doubles as phase S of the next iteration.
 *
 * Arithmetic follows the generic path's operation order except (i) the reciprocal pivot (<= 1 ulp
 * from 1/sqrt) and (ii) MFMA / cross-lane summation order in W and res; the parity tests hold it
 * to the same 1e-10 tolerance against the oracle.
 */
#pragma once

typedef double f64x4 __attribute__((ext_vector_type(4)));
/* LDS pointers carry their address space so that hipcc emits ds_read/ds_write, never flat_* */
typedef __attribute__((address_space(3))) double lds_f64;
typedef lds_f64 *lds_ptr;
typedef const lds_f64 *lds_cptr;
typedef __attribute__((address_space(3))) int *lds_iptr;
__device__ __forceinline__ lds_ptr to_lds(double *p) { return (lds_ptr)p; }

#define FW 4            /* waves per workgroup of the tier kernels: one per SIMD */

template <int NX, int NU, int MD>
struct Uni {
    static constexpr int D = NX * MD;        /* dual block dimension            */
    static constexpr int NZ = NX + NU;
    static constexpr int R = D + 1 + NX;     /* rows of the tall matrix         */
    static constexpr int KS = (NZ + 3) / 4;  /* MFMA k-steps                    */
    static constexpr int LDW = D + 1;        /* LDS row stride (bank spread)    */
    static constexpr int SCH = NX * NX + NX; /* Schur hand-off record: S (NX x NX) then v (NX) */
    static constexpr int TH = (MD == 1) ? 8 : ((MD == 2) ? 3 : ((MD <= 4) ? 2 : 1));     /* tier height: MD^(TH-1) <= FW; chains (MD == 1): 8 levels per tier */
    static constexpr int NBT = (MD == 1) ? 8 : ((MD == 2) ? 7 : (1 + MD));               /* blocks of a full tier subtree */
    static constexpr int WAVE_LDS = (D + 1) * LDW + D + NX + 8;        /* per-wave scratch (doubles) */
    static_assert(D % 4 == 0 && NX < 16, "Schur MFMA tile needs D % 4 == 0 and NX < 16");
    static constexpr int TIER_LDS = NBT * SCH + NBT * D + FW * WAVE_LDS + 16;   /* doubles per workgroup */
    static_assert(D <= 16, "the MFMA tile path needs a dual block of at most 16 rows");
    static_assert(R <= 64, "tall matrix must fit one wavefront");
    __device__ static __forceinline__ int kid0(int k) { return MD * k + 1; }
    __device__ static __forceinline__ int bo(int p) { return NX * (MD * p + 1); }      /* block vector offset */
    /* first node / number of nodes of a level; closed forms where MD is a power of two (these sit in every level
     * step of the persistent kernel: as loops they cost hundreds of scalar cycles per call) */
    __device__ static __forceinline__ int width(int level) {
        if (MD == 1) return 1;
        if (MD == 2) return 1 << level;
        if (MD == 4) return 1 << (2 * level);
        int w = 1; for (int l = 0; l < level; l++) w *= MD; return w;
    }
    __device__ static __forceinline__ int first(int level) {
        if (MD == 1) return level;
        if (MD == 2) return (1 << level) - 1;
        if (MD == 4) return ((1 << (2 * level)) - 1) / 3;
        int n = 0, w = 1;
        for (int l = 0; l < level; l++) { n += w; w *= MD; }
        return n;
    }
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

/* diagnostic stamp: slot of kernel `kern`, workgroup 0 / thread 0 only; the buffer is never read
 * by any kernel */
__device__ __forceinline__ void stamp(const Data &Dt, const Opts &O, int kern, int slot) {
#ifndef TQ_STAMPS      /* diagnostic builds only, as in the persistent kernel: the never-taken branches are not free */
    return;
#endif
    if (O.stamps && threadIdx.x == 0 && blockIdx.x == 0 && slot < 32 && kern < 8) {
        Dt.stamps[(kern * 32 + slot) * 2 + 0] = clock64();
        Dt.stamps[(kern * 32 + slot) * 2 + 1] = wall_clock64();
    }
}

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
/* workgroup barrier that orders LDS traffic only: global stores stay in flight (they are consumed
 * by a later kernel) */
__device__ __forceinline__ void lds_barrier() { lds_fence(); __builtin_amdgcn_s_barrier(); lds_fence(); }

/* ------------------------------------------------------------------------------------------ */
/* G + H for one parent block p (one wave): res/resMod of its children, W_p, Ut_p             */
/* returns this wave's contribution to the termination norm (valid in every lane)             */
/* ------------------------------------------------------------------------------------------ */
template <int NX, int NU, int MD>
__device__ __forceinline__ double fast_gh(const Data &Dt, int p, int lane, int termCondition) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    const int row = lane & 15, g = lane >> 4;
    const int cidx = row / NX, r = row - cidx * NX;
    const int k = U::kid0(p) + cidx;                     /* child owning this row */
    const bool live = row < D;
    const double *A = Dt.A + (size_t)(k - 1) * NX * NX + r;
    const double *B = Dt.B + (size_t)(k - 1) * NX * NU + r;
    const int bo = U::bo(p);
    /* issue every load before the first use */
    double a[U::KS], pc[U::KS], z[U::KS];
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        a[s] = 0.0; pc[s] = 0.0; z[s] = 0.0;
        if (live && cc < NZ) {
            if (cc < NX) { a[s] = A[(size_t)cc * NX]; pc[s] = Dt.QinvCal[NX * p + cc]; z[s] = Dt.x[NX * p + cc]; }
            else { a[s] = B[(size_t)(cc - NX) * NX]; pc[s] = Dt.RinvCal[NU * p + cc - NX]; z[s] = Dt.u[NU * p + cc - NX]; }
        }
    }
    double xk = 0.0, bk = 0.0, qk = 0.0;
    if (live && g == 0) { xk = Dt.x[bo + row]; bk = Dt.b[bo + row]; }
    if (live) qk = Dt.QinvCal[bo + row];
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;                                   /* this lane's share of (A x_p + B u_p)[row] */
    double *Ut = Dt.Ut + (size_t)(p > 0 ? p - 1 : 0) * NX * D;
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const double ap = a[s] * pc[s];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], ap, acc, 0, 0, 0);
        part = fma(a[s], z[s], part);
        if (p > 0 && live && cc < NX) Ut[cc + (size_t)row * NX] = -1.0 * ap;      /* Ut = -(A Qcal)' */
    }
    /* residual: reduce the 4 k-groups of a row (lanes row, row+16, row+32, row+48) */
    part = rows_fold<false>(part);
    double e = 0.0;
    if (live && g == 0) {
        const double rv = fma(-1.0, xk, bk) + part;
        Dt.res[bo + row] = rv;
        Dt.resMod[bo + row] = rv;
        e = (termCondition == 2) ? fabs(rv) : rv * rv;
    }
    /* W tile: lane holds W[i = g + 4 rr][j = row] */
    double *W = Dt.W + (size_t)p * D * D;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = g + 4 * rr;
        if (live && i < D) {
            double w = acc[rr];
            if (i == row) w += qk;
            W[i + (size_t)row * D] = w;
        }
    }
    return (termCondition == 2) ? wmax(e) : wsum(e);
}

/* ------------------------------------------------------------------------------------------ */
/* backward step of one block, split into pieces so that loads can be issued early            */
/* ------------------------------------------------------------------------------------------ */

/* rows of T = [W ; resMod' ; Ut] of block ii: lane i < D row i of W, lane D the right-hand side,
 * lanes D+1 .. R-1 the rows of Ut (none for the root) */
template <int NX, int NU, int MD>
__device__ __forceinline__ void load_rows(const Data &Dt, int ii, int lane, bool is_root, double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R;
    const double *src; int stride;
    if (lane < D) { src = Dt.W + (size_t)ii * D * D + lane; stride = D; }
    else if (lane == D) { src = Dt.resMod + U::bo(ii); stride = 1; }
    else if (lane < R && !is_root) { src = Dt.Ut + (size_t)(ii - 1) * NX * D + (lane - D - 1); stride = NX; }
    else { src = Dt.W + (size_t)ii * D * D; stride = D; }
#pragma unroll
    for (int j = 0; j < D; j++) T[j] = src[(size_t)j * stride];
}

/* subtract the children's Schur complements; `sch` points at MD consecutive records (S then v).
 * Branch-free: every lane issues all MD*NX loads from a valid address (per-lane base + stride) and
 * masks afterwards, so the loads overlap instead of serialising behind divergent branches.
 * SC1: records written by another workgroup of the same launch (agent-scope loads). */
template <int NX, int NU, int MD, bool SC1 = false, typename P>
__device__ __forceinline__ void sub_children(P sch, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const bool vrow = lane == D;
    const int lc = lane < D ? lane / NX : 0;
    const int r = lane < D ? lane - lc * NX : 0;
    const int off = vrow ? NX * NX : r, stride = vrow ? 1 : NX;
    double v[MD][NX];
#pragma unroll
    for (int c = 0; c < MD; c++) {
        const P src = sch + c * U::SCH + off;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            if constexpr (SC1) v[c][j] = __hip_atomic_load(src + j * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v[c][j] = src[j * stride];
        }
    }
    /* a 0/1 factor instead of a select: one fma per entry (the records hold finite numbers) */
#pragma unroll
    for (int c = 0; c < MD; c++) {
        const double m = (vrow || (lane < D && lc == c)) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < NX; j++) T[c * NX + j] = fma(-m, v[c][j], T[c * NX + j]);
    }
}

/* in-register tall Cholesky, left-looking, row broadcasts by readlane; returns true when a diagonal
 * entry of the factor is <= regTol (on-the-fly regularisation trigger) */
template <int D>
__device__ __forceinline__ bool potrf_rows(double (&T)[D], int lane, double regTol, double &myinv) {
    int small = 0;
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        const double finv = pivot_rsqrt(pj);
        small |= (pj * finv <= regTol);
        T[j] = s * finv;
        if (lane == j) myinv = finv;
    }
    return small != 0;
}

/* factor rows, reciprocal diagonal, backward solution and CholUt of block ii to global memory:
 * ONE store per column with a per-lane base + stride */
template <int NX, int NU, int MD>
__device__ __forceinline__ void store_factor(const Data &Dt, int ii, int lane, const double (&T)[Uni<NX, NU, MD>::D], double myinv) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R;
    const int bo = U::bo(ii);
    double *dst; int stride;
    if (lane < D) { dst = Dt.CholW + (size_t)ii * D * D + lane; stride = D; }
    else if (lane == D) { dst = Dt.ybuf + bo; stride = 1; }
    else { dst = Dt.CholUt + (size_t)(ii - 1) * NX * D + (lane - D - 1); stride = NX; }
    if (lane < R) {
#pragma unroll
        for (int j = 0; j < D; j++) dst[(size_t)j * stride] = T[j];
    }
    if (lane < D) Dt.invd[bo + lane] = myinv;
}

/* Schur record for the parent: [S | v] = CUt * [CUt' | y] as one f64 MFMA tile (K = D).  Rows
 * D .. R-1 of T go through the wave's LDS scratch; lane (i, g) feeds row 1+i as A and row 1+i
 * (i < NX) or row 0 (= y, i == NX) as B.  sdst: LDS record inside a tier or global Sbuf. */
template <int NX, int NU, int MD, typename PD>
__device__ __forceinline__ void schur_record(int lane, const double (&T)[Uni<NX, NU, MD>::D], lds_ptr lds, PD sdst) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R, LDW = U::LDW;
    if (lane >= D && lane < R) {
#pragma unroll
        for (int j = 0; j < D; j++) lds[(lane - D) * LDW + j] = T[j];
    }
    lds_fence();
    const int i = lane & 15, g = lane >> 4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < D / 4; st++) {
        const int kk = g + 4 * st;
        const double m = (i <= NX) ? lds[(i < NX ? 1 + i : 0) * LDW + kk] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(i < NX ? m : 0.0, m, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int ip = g + 4 * rr;                 /* acc[rr] = [S | v][ip][i] */
        if (ip < NX) {
            if (i < NX) sdst[ip + i * NX] = acc[rr];
            else if (i == NX) sdst[NX * NX + ip] = acc[rr];
        }
    }
    lds_fence();
}

/* regularised factorisation of the rows in T (treeqp_dpotrf_l_with_reg_opts, dual_Newton_common.c:36-78);
 * ONE copy of the unrolled factorisation in the instruction stream: the on-the-fly retry loops back */
template <int NX, int NU, int MD>
__device__ __forceinline__ void factor_rows(Ctrl *ctrl, const Opts &O, int lane, double (&T)[Uni<NX, NU, MD>::D], double &myinv) {
    constexpr int D = Uni<NX, NU, MD>::D;
    if (O.regType == 1) {
#pragma unroll
        for (int j = 0; j < D; j++) if (lane == j) T[j] += O.regValue;            /* ddiare (ALWAYS) */
    }
    double K[D];
#pragma unroll
    for (int j = 0; j < D; j++) K[j] = T[j];
    for (int pass = 0; pass < 2; pass++) {
        const bool small = potrf_rows<D>(T, lane, O.regTol, myinv);
        if (O.regType != 2 || !small || pass == 1) break;
#pragma unroll
        for (int j = 0; j < D; j++) T[j] = (lane == j) ? K[j] + O.regValue : K[j];   /* rare: shift and refactorise */
        if (lane == 0) atomicAdd(&ctrl->n_reg, 1);
    }
}

/* root: T holds [W_0 ; resMod_0'] minus the children's records; factor, then dlam_0 = L^-T y (k
 * descending) with lane i owning column i of L after a transpose through LDS; dot-product partial */
template <int NX, int NU, int MD>
__device__ __forceinline__ void root_block(const Data &Dt, const Opts &O, int lane, double (&T)[Uni<NX, NU, MD>::D],
                                           lds_ptr lds, lds_ptr dl_out) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int bo = U::bo(0);
    double myinv = 0.0;
    factor_rows<NX, NU, MD>(Dt.ctrl, O, lane, T, myinv);
    if (lane < D) {
        double *L = Dt.CholW + lane;
#pragma unroll
        for (int j = 0; j < D; j++) L[(size_t)j * D] = T[j];
        Dt.invd[bo + lane] = myinv;
    }
    if (lane <= D) {
#pragma unroll
        for (int j = 0; j < D; j++) lds[lane * U::LDW + j] = T[j];
    }
    lds_fence();
    double s = 0.0, Lcol[D];
    if (lane < D) {
        s = lds[D * U::LDW + lane];
#pragma unroll
        for (int k = 0; k < D; k++) Lcol[k] = lds[k * U::LDW + lane];
    } else {
#pragma unroll
        for (int k = 0; k < D; k++) Lcol[k] = 0.0;
    }
    double mine = 0.0;
#pragma unroll
    for (int k = D - 1; k >= 0; k--) {
        const double zk = rdlane(s * myinv, k);
        if (lane == k) mine = zk;
        if (lane < k) s = fma(-Lcol[k], zk, s);
    }
    lds_fence();
    double pd = 0.0;
    if (lane < D) { Dt.dlam[bo + lane] = mine; dl_out[lane] = mine; pd = Dt.res[bo + lane] * mine; }
    pd = wsum(pd);
    if (lane == 0) Dt.part_dot[0] = pd;
}

/* ------------------------------------------------------------------------------------------ */
/* forward step of one block: lane i owns column i of L                                       */
/* ------------------------------------------------------------------------------------------ */
template <int NX, int NU, int MD>
struct FwdRegs {
    double Lcol[Uni<NX, NU, MD>::D];
    double Ccol[NX];
    double y, inv, res;
};

template <int NX, int NU, int MD>
__device__ __forceinline__ void load_fwd(const Data &Dt, int ii, int lane, FwdRegs<NX, NU, MD> &F) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int bo = U::bo(ii), li = lane < D ? lane : 0;
    const double *Lc = Dt.CholW + (size_t)ii * D * D + (size_t)li * D;
    const double *Cc = Dt.CholUt + (size_t)(ii - 1) * NX * D + (size_t)li * NX;
#pragma unroll
    for (int k = 0; k < D; k++) F.Lcol[k] = Lc[k];
#pragma unroll
    for (int r = 0; r < NX; r++) F.Ccol[r] = Cc[r];
    F.y = Dt.ybuf[bo + li]; F.inv = Dt.invd[bo + li]; F.res = Dt.res[bo + li];
}

/* delta: the NX entries of the parent's solution that belong to node ii (LDS or global) */
template <int NX, int NU, int MD, typename PDelta>
__device__ __forceinline__ void forward_block(const Data &Dt, int ii, int lane, const FwdRegs<NX, NU, MD> &F,
                                              PDelta delta, lds_ptr dl_out) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < NX; r++) acc = fma(F.Ccol[r], delta[r], acc);
    double s = fma(-1.0, acc, F.y);
    double mine = 0.0;
#pragma unroll
    for (int k = D - 1; k >= 0; k--) {
        const double zk = rdlane(s * F.inv, k);
        if (lane == k) mine = zk;
        if (lane < k) s = fma(-F.Lcol[k], zk, s);
    }
    double pd = 0.0;
    if (lane < D) { Dt.dlam[U::bo(ii) + lane] = mine; dl_out[lane] = mine; pd = F.res * mine; }
    pd = wsum(pd);
    if (lane == 0) Dt.part_dot[ii] = pd;
}

/* ------------------------------------------------------------------------------------------ */
/* termination test run by the kernel that comes second in an iteration: every workgroup        */
/* reduces the per-workgroup partials of the first kernel; workgroup 0 records the decision     */
/* ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ bool converged_now(const Data &Dt, const Opts &O, const double *parts, int nparts, lds_iptr flag_lds) {
    if (threadIdx.x < WAVE) {
        double e = 0.0;
        for (int i = threadIdx.x; i < nparts; i += WAVE) e = (O.termCondition == 2) ? nanmax(e, parts[i]) : e + parts[i];
        e = (O.termCondition == 2) ? wmax(e) : wsum(e);
        if (O.termCondition == 1) e = sqrt(e);
        if (threadIdx.x == 0) {
            *flag_lds = e < O.tol;
            if (blockIdx.x == 0) {
                Dt.ctrl->err = e;
                if (e < O.tol) { Dt.ctrl->status = 0; Dt.ctrl->done = 1; }
            }
        }
    }
    lds_barrier();
    return *flag_lds != 0;
}

/* ------------------------------------------------------------------------------------------ */
/* tier kernels                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* LDS carve of a tier workgroup */
template <int NX, int NU, int MD>
struct TierLds {
    using U = Uni<NX, NU, MD>;
    lds_ptr sch;       /* NBT Schur records, heap order inside the tier subtree */
    lds_ptr dl;        /* NBT block solutions (forward sweep)                   */
    lds_ptr wave0;     /* scratch of wave 0                                     */
    lds_ptr wave;      /* this wave's scratch                                   */
    lds_iptr flag;
    __device__ TierLds(double *base, int wave_id) {
        sch = to_lds(base); dl = sch + U::NBT * U::SCH; wave0 = dl + U::NBT * U::D; wave = wave0 + wave_id * U::WAVE_LDS;
        flag = (lds_iptr)(wave0 + FW * U::WAVE_LDS);
    }
};

/* Sharding of one tree over several devices: a rank owns a contiguous range of
 * the subtrees of every partitioned tier and computes the replicated tiers in full.  The same
 * kernels serve the single-device case with the neutral descriptor {0, nullptr, 0, 0, nullptr, 0}. */
struct Shard {
    int wg_off;             /* first subtree of this rank in a partitioned tier (added to blockIdx.x) */
    const int *gh_list;     /* blocks above tier 0 whose G+H this rank computes (nullptr: all of them) */
    int gh_n, gh_counted;   /* list length; entries >= gh_counted do not enter the termination norm here */
    const double *err_src;  /* termination partials for the check (nullptr: Dt.part_err)               */
    int pad;
};

/* f_back: one workgroup per subtree of block levels [l0, l1): backward sweep bottom-up.
 * first != 0: this is the first kernel of the iteration: G+H (and norm partials) before the sweep.
 * check != 0: this is the second kernel of the iteration: termination test first. */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_back(Tree T, Data Dt, Opts O, Shard Sh, int l0, int l1, int first, int check, int nparts, int kern, int h)
#if !TQ_HAS(TQP_TIER)
;
#else
{
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    if (!phase_main(Dt.ctrl, h)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    TierLds<NX, NU, MD> L(lds_all, wave);
    const int s = blockIdx.x + Sh.wg_off, th = l1 - l0;
    int sl = 0;
    stamp(Dt, O, kern, sl++);
    if (check && converged_now(Dt, O, Sh.err_src ? Sh.err_src : Dt.part_err, nparts, L.flag)) return;
    if (first) {
        /* G + H: the blocks of this subtree, plus a share of the blocks above this tier */
        double e = 0.0;
        int cnt = 0;
        for (int t = 0; t < th; t++) {
            const int nb = U::width(t), f0 = U::first(l0 + t) + s * nb;
            for (int b = 0; b < nb; b++, cnt++)
                if ((cnt & (FW - 1)) == wave) { const double v = fast_gh<NX, NU, MD>(Dt, f0 + b, lane, O.termCondition); e = (O.termCondition == 2) ? nanmax(e, v) : e + v; }
        }
        if (Sh.gh_list) {
            for (int q = blockIdx.x * FW + wave; q < Sh.gh_n; q += gridDim.x * FW) {
                const double v = fast_gh<NX, NU, MD>(Dt, Sh.gh_list[q], lane, O.termCondition);
                if (q < Sh.gh_counted) e = (O.termCondition == 2) ? nanmax(e, v) : e + v;
            }
        } else {
            const int nup = U::first(l0);
            for (int p = s * FW + wave; p < nup; p += gridDim.x * FW) { const double v = fast_gh<NX, NU, MD>(Dt, p, lane, O.termCondition); e = (O.termCondition == 2) ? nanmax(e, v) : e + v; }
        }
        if (lane == 0) L.dl[wave] = e;       /* L.dl is not used by the backward sweep */
        __syncthreads();                     /* W / Ut / resMod of this subtree are read back below */
        if (threadIdx.x == 0) {
            double acc = 0.0;
            for (int w = 0; w < FW; w++) { const double v = L.dl[w]; acc = (O.termCondition == 2) ? nanmax(acc, v) : acc + v; }
            Dt.part_err[blockIdx.x] = acc;
        }
        stamp(Dt, O, kern, sl++);
    }
    double Tc[D], Tn[D];
    {
        const int t = th - 1, nb = U::width(t);
        if (wave < nb) load_rows<NX, NU, MD>(Dt, U::first(l0 + t) + s * nb + wave, lane, false, Tc);
    }
    for (int t = th - 1; t >= 0; t--) {
        const int nb = U::width(t), nbn = nb / MD;
        const bool act = wave < nb, act_next = t > 0 && wave < nbn;
        if (act_next) load_rows<NX, NU, MD>(Dt, U::first(l0 + t - 1) + s * nbn + wave, lane, false, Tn);     /* prefetch */
        if (act) {
            const int ii = U::first(l0 + t) + s * nb + wave;
            const int loc = U::first(t) + wave;                       /* heap index inside the tier subtree */
            if (t < th - 1) sub_children<NX, NU, MD>(L.sch + (U::first(t + 1) + MD * wave) * U::SCH, lane, Tc);     /* children in this tier: LDS */
            else if (l1 < T.Nh) sub_children<NX, NU, MD>((const double *)(Dt.Sbuf + (size_t)U::kid0(ii) * U::SCH), lane, Tc);   /* tier below */
            double myinv = 0.0;
            factor_rows<NX, NU, MD>(Dt.ctrl, O, lane, Tc, myinv);
            store_factor<NX, NU, MD>(Dt, ii, lane, Tc, myinv);
            if (t > 0) schur_record<NX, NU, MD>(lane, Tc, L.wave, L.sch + loc * U::SCH);
            else schur_record<NX, NU, MD>(lane, Tc, L.wave, Dt.Sbuf + (size_t)ii * U::SCH);
        }
        lds_barrier();
        if (act_next) {
#pragma unroll
            for (int j = 0; j < D; j++) Tc[j] = Tn[j];
        }
        stamp(Dt, O, kern, sl++);
    }
}
#endif

/* f_top: single workgroup, block levels [0, l1): (G+H if it is the only tier) termination test,
 * backward sweep, root, forward sweep of its levels; arms the line search. */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_top(Tree T, Data Dt, Opts O, Shard Sh, int l1, int first, int check, int nparts, int kern, int h)
#if !TQ_HAS(TQP_TIER)
;
#else
{
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Ctrl *c = Dt.ctrl;
    if (!phase_main(c, h)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    TierLds<NX, NU, MD> L(lds_all, wave);
    int sl = 0;
    stamp(Dt, O, kern, sl++);
    if (first) {
        double e = 0.0;
        const int nblk = U::first(l1);
        for (int p = wave; p < nblk; p += FW) { const double v = fast_gh<NX, NU, MD>(Dt, p, lane, O.termCondition); e = (O.termCondition == 2) ? nanmax(e, v) : e + v; }
        if (lane == 0) Dt.part_err[wave] = e;
        __syncthreads();
        nparts = FW;
        check = 1;
    }
    if (check && converged_now(Dt, O, (Sh.err_src && !first) ? Sh.err_src : Dt.part_err, nparts, L.flag)) return;
    stamp(Dt, O, kern, sl++);
    double Tc[D], Tn[D];
    {
        const int t = l1 - 1, nb = U::width(t);
        if (wave < nb) load_rows<NX, NU, MD>(Dt, U::first(t) + wave, lane, t == 0, Tc);
    }
    for (int t = l1 - 1; t >= 1; t--) {
        const int nb = U::width(t), nbn = nb / MD;
        const bool act = wave < nb, act_next = wave < nbn;
        if (act_next) load_rows<NX, NU, MD>(Dt, U::first(t - 1) + wave, lane, t - 1 == 0, Tn);
        if (act) {
            const int ii = U::first(t) + wave;
            if (t < l1 - 1) sub_children<NX, NU, MD>(L.sch + (U::first(t + 1) + MD * wave) * U::SCH, lane, Tc);
            else if (l1 < T.Nh) sub_children<NX, NU, MD>((const double *)(Dt.Sbuf + (size_t)U::kid0(ii) * U::SCH), lane, Tc);
            double myinv = 0.0;
            factor_rows<NX, NU, MD>(Dt.ctrl, O, lane, Tc, myinv);
            store_factor<NX, NU, MD>(Dt, ii, lane, Tc, myinv);
            schur_record<NX, NU, MD>(lane, Tc, L.wave, L.sch + ii * U::SCH);
        }
        lds_barrier();
        if (act_next) {
#pragma unroll
            for (int j = 0; j < D; j++) Tc[j] = Tn[j];
        }
        stamp(Dt, O, kern, sl++);
    }
    if (wave == 0) {
        if (l1 > 1) sub_children<NX, NU, MD>(L.sch + 1 * U::SCH, lane, Tc);
        else if (l1 < T.Nh) sub_children<NX, NU, MD>((const double *)(Dt.Sbuf + (size_t)1 * U::SCH), lane, Tc);
        root_block<NX, NU, MD>(Dt, O, lane, Tc, L.wave, L.dl);
    }
    __syncthreads();                             /* factors written above are re-read below */
    stamp(Dt, O, kern, sl++);
    for (int t = 1; t < l1; t++) {
        const int nb = U::width(t);
        if (wave < nb) {
            const int ii = U::first(t) + wave;
            FwdRegs<NX, NU, MD> F;
            load_fwd<NX, NU, MD>(Dt, ii, lane, F);
            const int par = (ii - 1) / MD, pos = ((ii - 1) % MD) * NX;
            forward_block<NX, NU, MD>(Dt, ii, lane, F, (lds_cptr)(L.dl + par * D + pos), L.dl + ii * D);
        }
        lds_barrier();
        stamp(Dt, O, kern, sl++);
    }
    if (threadIdx.x == 0) { c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1; }
}
#endif

/* f_fwd: one workgroup per subtree of block levels [l0, l1): forward sweep top-down */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_fwd(Tree T, Data Dt, Opts O, Shard Sh, int l0, int l1, int kern, int h)
#if !TQ_HAS(TQP_TIER)
;
#else
{
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    if (!phase_trial(Dt.ctrl, h, 1)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    TierLds<NX, NU, MD> L(lds_all, wave);
    const int s = blockIdx.x + Sh.wg_off, th = l1 - l0;
    int sl = 0;
    stamp(Dt, O, kern, sl++);
    FwdRegs<NX, NU, MD> Fc, Fn;
    if (wave == 0) load_fwd<NX, NU, MD>(Dt, U::first(l0) + s, lane, Fc);
    for (int t = 0; t < th; t++) {
        const int nb = U::width(t), nbn = nb * MD;
        const bool act = wave < nb, act_next = t + 1 < th && wave < nbn;
        if (act_next) load_fwd<NX, NU, MD>(Dt, U::first(l0 + t + 1) + s * nbn + wave, lane, Fn);
        if (act) {
            const int ii = U::first(l0 + t) + s * nb + wave;
            const int loc = U::first(t) + wave;
            if (t == 0) forward_block<NX, NU, MD>(Dt, ii, lane, Fc, (const double *)(Dt.dlam + NX * ii), L.dl + loc * D);     /* parent in the tier above: memory */
            else forward_block<NX, NU, MD>(Dt, ii, lane, Fc, (lds_cptr)(L.dl + (U::first(t - 1) + wave / MD) * D + (wave % MD) * NX), L.dl + loc * D);   /* this tier: LDS */
        }
        lds_barrier();
        if (act_next) Fc = Fn;
        stamp(Dt, O, kern, sl++);
    }
}
#endif

/* f_stage: trial point lam_cur + (tau - tauPrev) dlam for every node, one wave per node: stage QP by
 * clipping, elimination vectors, dual-function term; writes lam_next.  All loads are issued before
 * the first dependent use. */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_stage(Tree T, Data Dt, Opts O, const int *node_list, int n_nodes, int kern, int h, int trial)
#if !TQ_HAS(TQP_TIER)
;
#else
{
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const Ctrl *c = Dt.ctrl;
    if (!phase_trial(c, h, trial)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = blockIdx.x * FW + wave;
    stamp(Dt, O, kern, 0);
    if (q >= n_nodes) return;
    const int k = node_list ? node_list[q] : q;
    double *lds = lds_all + wave * (D + NX + 8);
    const double *lamc = c->cur ? Dt.lam1 : Dt.lam0;
    double *lamn = c->cur ? Dt.lam0 : Dt.lam1;
    const double step = c->tau - c->tauPrev;
    const bool parent = k < T.Np;
    const int nuk = parent ? NU : 0;
    const int xo = NX * k, uo = NU * k, ko = U::bo(k);
    const bool isx = lane < NX, live = lane < NX + nuk;
    const int j = isx ? lane : lane - NX;
    /* loads */
    double lkv = 0.0, dkv = 0.0, bkv = 0.0, lo_ = 0.0, do_ = 0.0;
    if (parent && lane < D) { lkv = lamc[ko + lane]; dkv = Dt.dlam[ko + lane]; bkv = Dt.b[ko + lane]; }
    if (isx && k > 0) { lo_ = lamc[xo + lane]; do_ = Dt.dlam[xo + lane]; }
    double col[MD][NX];
    if (parent && live) {
#pragma unroll
        for (int cc = 0; cc < MD; cc++) {
            const int kid = U::kid0(k) + cc;
            const double *cp = isx ? Dt.A + (size_t)(kid - 1) * NX * NX + (size_t)j * NX
                                   : Dt.B + (size_t)(kid - 1) * NX * NU + (size_t)j * NX;
#pragma unroll
            for (int i = 0; i < NX; i++) col[cc][i] = cp[i];
        }
    }
    double lin = 0.0, winv = 0.0, wd = 0.0, lob = 0.0, hib = 0.0;
    if (live) {
        if (isx) { lin = Dt.q[xo + j]; winv = Dt.Qinv[xo + j]; wd = Dt.Qd[xo + j]; lob = Dt.xmin[xo + j]; hib = Dt.xmax[xo + j]; }
        else { lin = Dt.r[uo + j]; winv = Dt.Rinv[uo + j]; wd = Dt.Rd[uo + j]; lob = Dt.umin[uo + j]; hib = Dt.umax[uo + j]; }
    }
    /* trial multipliers */
    double *lk = lds, *lown = lds + D;
    double p_c = 0.0;
    if (parent && lane < D) { const double v = fma(step, dkv, lkv); lk[lane] = v; p_c = bkv * v; }
    if (isx) {
        double v = 0.0;
        if (k > 0) { v = fma(step, do_, lo_); lamn[xo + lane] = v; }
        lown[lane] = v;
    }
    lds_fence();
    double p_q = 0.0, p_h = 0.0;
    if (live) {
        double v = isx ? fma(-1.0, lin, lown[j]) : -1.0 * lin;
        if (parent) {
#pragma unroll
            for (int cc = 0; cc < MD; cc++) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NX; i++) acc = fma(col[cc][i], lk[cc * NX + i], acc);
                v = fma(-1.0, acc, v);
            }
        }
        const double unc = winv * v;
        double val, cal;
        if (unc >= hib) { val = hib; cal = 0.0; } else if (unc <= lob) { val = lob; cal = 0.0; } else { val = unc; cal = winv; }
        if (trial == 1) { if (isx) Dt.xUncS[xo + j] = Dt.xUnc[xo + j]; else Dt.uUncS[uo + j] = Dt.uUnc[uo + j]; }      /* phase S of this iteration (see Data) */
        if (isx) { Dt.qmod[xo + j] = v; Dt.xUnc[xo + j] = unc; Dt.x[xo + j] = val; Dt.QinvCal[xo + j] = cal; }
        else { Dt.rmod[uo + j] = v; Dt.uUnc[uo + j] = unc; Dt.u[uo + j] = val; Dt.RinvCal[uo + j] = cal; }
        p_q = (wd * val) * val;
        p_h = v * val;
    }
    /* dual-function term (clipping.c:371-381): x part then u part, summed per part */
    const double qx = wsum(isx ? p_q : 0.0), hx = wsum(isx ? p_h : 0.0);
    const double ru = wsum(isx ? 0.0 : p_q), hu = wsum(isx ? 0.0 : p_h);
    p_c = wsum(p_c);
    if (lane == 0) {
        double f = -0.5 * qx - p_c;
        f += hx;
        f -= 0.5 * ru;
        f += hu;
        Dt.fval[k] = f;
    }
    stamp(Dt, O, kern, 1);
}
#endif

/* the part that owns the family instantiates it (tdunes_parts.hpp) */
#if TQ_HAS(TQP_TIER)
#define X(idx, nx, nu, md) \
    template __global__ void f_back<nx, nu, md>(Tree, Data, Opts, Shard, int, int, int, int, int, int, int); \
    template __global__ void f_top<nx, nu, md>(Tree, Data, Opts, Shard, int, int, int, int, int, int); \
    template __global__ void f_fwd<nx, nu, md>(Tree, Data, Opts, Shard, int, int, int, int); \
    template __global__ void f_stage<nx, nu, md>(Tree, Data, Opts, const int *, int, int, int, int);
FAST_TABLE(X)
#undef X
#endif
