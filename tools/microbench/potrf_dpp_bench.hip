// Microbenchmark: the tall factorisation of the persistent kernel (W 16x16, 9 rows rhs/Ut, 16 identity rows; f64) with
// row broadcasts by v_readlane (production, variant 0) against DPP row_newbcast fused into v_fmac_f64 (variants 1..).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-sched-strategy=max-ilp -o potrf_dpp_bench potrf_dpp_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define D 16
#define NXR 9          /* rhs + Ut rows */

__device__ __forceinline__ double rdlane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pivot_rsqrt3(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y0), y0, 1.0);
    const double h = fma(0.375, e, 0.5);
    const double y = fma(y0 * e, h, y0);
    return p > 0.0 ? y : 0.0;
}

/* production: row q of the tall matrix in lane q (41 lanes), left-looking, readlane broadcasts */
__device__ __forceinline__ void potrf_v0(double (&T)[D]) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        T[j] = s * pivot_rsqrt3(pj);
    }
}

/* acc -= (lane J of my 16-lane row of b) * t, one instruction */
template <int J, bool NOP>
__device__ __forceinline__ void fmac_bc(double &acc, double b, double t) {
    if (NOP) asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(t), "n"(J));
    else asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(t), "n"(J));
}
template <int J, bool NOP>
__device__ __forceinline__ double mov_bc(double b) {
    double r;
    if (NOP) asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "n"(J));
    else asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "n"(J));
    return r;
}

/* DPP left-looking: B = rows of W replicated in every 16-lane row (lane 16 m + c holds row c), T = the other rows */
template <int J, int K, bool NOP>
struct ColUpd {
    static __device__ __forceinline__ void run(double &sT, double &sB, const double (&T)[D], const double (&B)[D]) {
        fmac_bc<J, NOP>(sT, B[K], T[K]);
        fmac_bc<J, NOP>(sB, B[K], B[K]);
        ColUpd<J, K + 1, NOP>::run(sT, sB, T, B);
    }
};
template <int J, bool NOP>
struct ColUpd<J, J, NOP> { static __device__ __forceinline__ void run(double &, double &, const double (&)[D], const double (&)[D]) {} };

template <int J, bool NOP>
struct DppLeft {
    static __device__ __forceinline__ void run(double (&T)[D], double (&B)[D]) {
        double sT = T[J], sB = B[J];
        ColUpd<J, 0, NOP>::run(sT, sB, T, B);
        const double pj = mov_bc<J, NOP>(sB);
        const double y = pivot_rsqrt3(pj);
        T[J] = sT * y; B[J] = sB * y;
        DppLeft<J + 1, NOP>::run(T, B);
    }
};
template <bool NOP>
struct DppLeft<D, NOP> { static __device__ __forceinline__ void run(double (&)[D], double (&)[D]) {} };

/* DPP right-looking */
template <int J, int C, bool NOP>
struct TrailUpd {
    static __device__ __forceinline__ void run(double (&T)[D], double (&B)[D]) {
        fmac_bc<C, NOP>(T[C], B[J], T[J]);
        fmac_bc<C, NOP>(B[C], B[J], B[J]);
        TrailUpd<J, C + 1, NOP>::run(T, B);
    }
};
template <int J, bool NOP>
struct TrailUpd<J, D, NOP> { static __device__ __forceinline__ void run(double (&)[D], double (&)[D]) {} };
template <int J, bool NOP>
struct DppRight {
    static __device__ __forceinline__ void run(double (&T)[D], double (&B)[D]) {
        const double pj = mov_bc<J, NOP>(B[J]);
        const double y = pivot_rsqrt3(pj);
        T[J] *= y; B[J] *= y;
        TrailUpd<J, J + 1, NOP>::run(T, B);
        DppRight<J + 1, NOP>::run(T, B);
    }
};
template <bool NOP>
struct DppRight<D, NOP> { static __device__ __forceinline__ void run(double (&)[D], double (&)[D]) {} };

/* DPP left-looking, two columns per step: both pivots' reciprocal square roots start together.
 * p11, p21, p22 of the 2 x 2 pivot block; l11 = sqrt(p11), l21 = p21 / l11, l22 = sqrt(p22 - l21^2) = sqrt(det / p11):
 * 1 / l22 = l11 * rsqrt(det) = p11 * r1 * rsqrt(det), det = p11 p22 - p21^2 */
template <int J, int K, bool NOP>
struct ColUpd2 {
    static __device__ __forceinline__ void run(double &sT0, double &sB0, double &sT1, double &sB1, const double (&T)[D], const double (&B)[D]) {
        fmac_bc<J, NOP>(sT0, B[K], T[K]);
        fmac_bc<J, NOP>(sB0, B[K], B[K]);
        fmac_bc<J + 1, NOP>(sT1, B[K], T[K]);
        fmac_bc<J + 1, NOP>(sB1, B[K], B[K]);
        ColUpd2<J, K + 1, NOP>::run(sT0, sB0, sT1, sB1, T, B);
    }
};
template <int J, bool NOP>
struct ColUpd2<J, J, NOP> { static __device__ __forceinline__ void run(double &, double &, double &, double &, const double (&)[D], const double (&)[D]) {} };
template <int J, bool NOP>
struct DppLeft2 {
    static __device__ __forceinline__ void run(double (&T)[D], double (&B)[D]) {
        double sT0 = T[J], sB0 = B[J], sT1 = T[J + 1], sB1 = B[J + 1];
        ColUpd2<J, 0, NOP>::run(sT0, sB0, sT1, sB1, T, B);
        const double p11 = mov_bc<J, NOP>(sB0), p21 = mov_bc<J + 1, NOP>(sB0), p22 = mov_bc<J + 1, NOP>(sB1);
        const double det = fma(p11, p22, -(p21 * p21));
        const double r1 = pivot_rsqrt3(p11), rd = pivot_rsqrt3(det);
        const double l21 = p21 * r1;                       /* entry (J+1, J) of the factor */
        const double r2 = (p11 * r1) * rd;
        T[J] = sT0 * r1; B[J] = sB0 * r1;
        T[J + 1] = fma(-T[J], l21, sT1) * r2; B[J + 1] = fma(-B[J], l21, sB1) * r2;
        DppLeft2<J + 2, NOP>::run(T, B);
    }
};
template <bool NOP>
struct DppLeft2<D, NOP> { static __device__ __forceinline__ void run(double (&)[D], double (&)[D]) {} };

/* readlane, two columns per step (same algebra) */
__device__ __forceinline__ void potrf_v4(double (&T)[D]) {
#pragma unroll
    for (int j = 0; j < D; j += 2) {
        double s0 = T[j], s1 = T[j + 1];
#pragma unroll
        for (int k = 0; k < j; k++) { s0 = fma(-T[k], rdlane(T[k], j), s0); s1 = fma(-T[k], rdlane(T[k], j + 1), s1); }
        const double p11 = rdlane(s0, j), p21 = rdlane(s0, j + 1), p22 = rdlane(s1, j + 1);
        const double det = fma(p11, p22, -(p21 * p21));
        const double r1 = pivot_rsqrt3(p11), rd = pivot_rsqrt3(det);
        const double l21 = p21 * r1, r2 = (p11 * r1) * rd;
        T[j] = s0 * r1;
        T[j + 1] = fma(-T[j], l21, s1) * r2;
    }
}


/* LDS row broadcast, left-looking: every finished column of the factor (its W rows, lanes 0..15) is stored once, row-major with
 * stride LS; row j is read back as a broadcast (every lane the same address), two entries per ds_read_b128, one step ahead of
 * its use; only the newest column (k = j - 1) comes by readlane */
typedef __attribute__((address_space(3))) double lds_f64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2 lds_d2;
constexpr int LS = 18;
__device__ __forceinline__ double pivot_rsqrt3m(double p) {        /* NaN (p < 0) / inf (p == 0) -> 0 through v_max_f64 */
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y0), y0, 1.0);
    const double h = fma(0.375, e, 0.5);
    const double y = fma(y0 * e, h, y0);
    return __builtin_fmax(y, 0.0);
}
template <int J, bool MX>
struct LdsLeft {
    static __device__ __forceinline__ void run(double (&T)[D], lds_f64 *L, int lane, const double (&bc)[D]) {
        double nx[D];
#pragma unroll
        for (int k = 0; k < D; k++) nx[k] = 0.0;
        if (J + 1 < D) {
#pragma unroll
            for (int k = 0; k + 1 <= J; k += 2) {                  /* row J + 1, entries 0 .. J - 1 (pairs; an odd last one alone) */
                const d2 v = *(const lds_d2 *)(L + (J + 1) * LS + k);
                nx[k] = v.x; nx[k + 1] = v.y;
            }
            if (J & 1) nx[J - 1] = L[(J + 1) * LS + J - 1];
        }
        double s = T[J];
#pragma unroll
        for (int k = 0; k + 1 < J; k++) s = fma(-T[k], bc[k], s);
        if (J >= 1) s = fma(-T[J - 1], rdlane(T[J - 1], J), s);
        const double pj = rdlane(s, J);
        T[J] = s * (MX ? pivot_rsqrt3m(pj) : pivot_rsqrt3(pj));
        L[(lane < D ? lane : D) * LS + J] = T[J];              /* lanes beyond the W rows write a dummy row: no exec juggling */
        LdsLeft<J + 1, MX>::run(T, L, lane, nx);
    }
};
template <bool MX>
struct LdsLeft<D, MX> { static __device__ __forceinline__ void run(double (&)[D], lds_f64 *, int, const double (&)[D]) {} };

/* readlane production with the max-based pivot select */
__device__ __forceinline__ void potrf_v0m(double (&T)[D]) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        T[j] = s * pivot_rsqrt3m(pj);
    }
}

/* variant 11 (round 3; asked for in round 2's review): 4-column panels factorised on VALU (row per lane, readlane broadcasts INSIDE the
 * panel only: 6 per panel instead of 120 for the block), the rank-4 trailing update of the later columns as v_mfma_f64_16x16x4 tiles on
 * the matrix pipe (three row tiles of 16 for the 41 rows).  The wave holds the block one row per lane; an MFMA wants lane (r, g) = (row or
 * column r of the tile, k-group g): the panel goes to LDS row by row (two ds_write_b128 per lane), comes back as A / B operands, the
 * product goes to LDS in the accumulator layout and comes back as rows. */
typedef double f64x4_b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void potrf_v11(double (&T)[D], lds_f64 *P /* 48 x 4 */, lds_f64 *U /* 48 x 17 */, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        /* the panel: columns 4p .. 4p + 3, left-looking inside the panel */
#pragma unroll
        for (int j = 4 * p; j < 4 * p + 4; j++) {
            double s = T[j];
#pragma unroll
            for (int k = 4 * p; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
            const double pj = rdlane(s, j);
            T[j] = s * pivot_rsqrt3m(pj);
        }
        if (p == 3) break;
        /* rows of the panel -> LDS (row i: P[4 i .. 4 i + 3]) */
        if (lane < 48) {
#pragma unroll
            for (int k = 0; k < 4; k++) P[4 * lane + k] = T[4 * p + k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        /* update = L[rows][panel] * L[cols][panel]' for the columns behind the panel; A: lane (j, k) = L[j][4p + k] for columns j >= 4 (p + 1), else 0 */
        const double a = (r16 >= 4 * (p + 1)) ? P[4 * r16 + g] : 0.0;
        f64x4_b acc[3];
#pragma unroll
        for (int I = 0; I < 3; I++) {
            const double b = P[4 * (16 * I + r16) + g];
            acc[I] = f64x4_b{0.0, 0.0, 0.0, 0.0};
            acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[I], 0, 0, 0);
        }
        /* accumulator layout: lane (r, g), register q = element (row 16 I + r, column g + 4 q) -> LDS U[row][column] (ld 17) */
#pragma unroll
        for (int I = 0; I < 3; I++)
#pragma unroll
            for (int q = 0; q < 4; q++) U[(16 * I + r16) * 17 + g + 4 * q] = acc[I][q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        /* back as rows: lane i takes its row of the update for the columns behind the panel */
        if (lane < 48) {
#pragma unroll
            for (int c = 4 * (p + 1); c < D; c++) T[c] -= U[lane * 17 + c];
        }
    }
}

template <int V>
__global__ void bench(const double *in, double *out, long long *cycles, int reps) {
    __shared__ __attribute__((aligned(16))) double lds_raw[17 * 18 + 8];
    __shared__ __attribute__((aligned(16))) double lds_p[48 * 4], lds_u[48 * 17 + 8];
    lds_f64 *Lw = (lds_f64 *)lds_raw;
    const int lane = threadIdx.x, c = lane & 15;
    /* `in`: 41 rows x 16: rows 0..15 W, 16..24 rhs / Ut, 25..40 identity */
    double T0[D], B0[D], T[D], B[D];
    int rowT, rowB = c;
    if (V == 0 || V == 4 || V >= 8) rowT = lane < 41 ? lane : 41;                       /* row 41: zeros */
    else rowT = lane < 16 ? 25 + lane : (lane < 16 + NXR ? lane : 41);       /* lanes 0..15 identity rows, 16..24 rhs / Ut */
#pragma unroll
    for (int j = 0; j < D; j++) { T0[j] = in[rowT * D + j]; B0[j] = in[rowB * D + j]; }
    double acc = 0.0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < D; j++) { T[j] = T0[j] + acc * 1e-300; B[j] = B0[j] + acc * 1e-300; }
        if (V == 0) potrf_v0(T);
        if (V == 1) DppLeft<0, true>::run(T, B);
        if (V == 2) DppLeft<0, false>::run(T, B);
        if (V == 3) DppRight<0, true>::run(T, B);
        if (V == 4) potrf_v4(T);
        if (V == 5) DppLeft2<0, true>::run(T, B);
        if (V == 6) DppLeft2<0, false>::run(T, B);
        if (V == 7) DppRight<0, false>::run(T, B);
        if (V == 8) { double z[D]; for (int k = 0; k < D; k++) z[k] = 0.0; LdsLeft<0, false>::run(T, Lw, lane, z); }
        if (V == 9) { double z[D]; for (int k = 0; k < D; k++) z[k] = 0.0; LdsLeft<0, true>::run(T, Lw, lane, z); }
        if (V == 10) potrf_v0m(T);
        if (V == 11) potrf_v11(T, (lds_f64 *)lds_p, (lds_f64 *)lds_u, lane);
#pragma unroll
        for (int j = 0; j < D; j++) acc += T[j];
    }
    long long t1 = clock64();
    /* out: 41 rows in the order of `in` */
    if (V == 0 || V == 4 || V >= 8) { if (lane < 41) for (int j = 0; j < D; j++) out[lane * D + j] = T[j]; }
    else {
        if (lane < 16) for (int j = 0; j < D; j++) { out[(25 + lane) * D + j] = T[j]; out[lane * D + j] = B[j]; }
        else if (lane < 16 + NXR) for (int j = 0; j < D; j++) out[lane * D + j] = T[j];
    }
    if (lane == 0) cycles[0] = (t1 - t0) / reps;
    if (acc == 12345.678) out[0] = acc;
}

__global__ void probe(double *out, long long *cycles) {
    const int lane = threadIdx.x;
    double x = 1.0 + lane * 1e-3, y = 0.5;
    long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; i++) x = fma(x, y, 1.0);                         /* dependent fma */
    long long t1 = clock64();
    double z = x;
#pragma unroll
    for (int i = 0; i < 256; i++) z = fma(z, rdlane(z, i & 15), 1.0);         /* readlane + dependent fma */
    long long t2 = clock64();
    double w = z * 1e-30, b = 1e-3;
#pragma unroll
    for (int i = 0; i < 256; i++) fmac_bc<3, true>(w, w, b);                  /* dependent fmac_dpp (source = accumulator), with nop */
    long long t3 = clock64();
    double a0 = w, a1 = w + 1, a2 = w + 2, a3 = w + 3;
#pragma unroll
    for (int i = 0; i < 64; i++) { fmac_bc<3, false>(a0, b, y); fmac_bc<4, false>(a1, b, y); fmac_bc<5, false>(a2, b, y); fmac_bc<6, false>(a3, b, y); }   /* 4 independent fmac_dpp chains */
    long long t4 = clock64();
    double r = a0 + a1 + a2 + a3;
#pragma unroll
    for (int i = 0; i < 64; i++) r = pivot_rsqrt3(r + 2.0);
    long long t5 = clock64();
    double q = r;
#pragma unroll
    for (int i = 0; i < 256; i++) q = mov_bc<5, true>(q) + 1.0;               /* mov_dpp + add */
    long long t6 = clock64();
    out[lane] = q;
    if (lane == 0) { cycles[0] = (t1 - t0) / 256; cycles[1] = (t2 - t1) / 256; cycles[2] = (t3 - t2) / 256; cycles[3] = (t4 - t3) / 256; cycles[4] = (t5 - t4) / 64; cycles[5] = (t6 - t5) / 256; }
}

int main() {
    std::vector<double> h(42 * D, 0.0);
    for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) h[i * D + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
    for (int i = D; i < 25; i++) for (int j = 0; j < D; j++) h[i * D + j] = 0.1 * ((i * 7 + j * 3) % 11);
    for (int i = 0; i < D; i++) h[(25 + i) * D + i] = 1.0;
    double *din, *dout; long long *dc;
    hipMalloc(&din, sizeof(double) * 42 * D); hipMalloc(&dout, sizeof(double) * 64 * D); hipMalloc(&dc, 64);
    hipMemcpy(din, h.data(), sizeof(double) * 42 * D, hipMemcpyHostToDevice);
    long long c[8];
    std::vector<double> ref(41 * D), got(41 * D);
    for (int v = 0; v <= 11; v++) {
        hipMemset(dout, 0, sizeof(double) * 64 * D);
        for (int it = 0; it < 2; it++) {
            switch (v) {
                case 0: hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 1: hipLaunchKernelGGL(bench<1>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 2: hipLaunchKernelGGL(bench<2>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 3: hipLaunchKernelGGL(bench<3>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 4: hipLaunchKernelGGL(bench<4>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 5: hipLaunchKernelGGL(bench<5>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 6: hipLaunchKernelGGL(bench<6>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 7: hipLaunchKernelGGL(bench<7>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 8: hipLaunchKernelGGL(bench<8>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 9: hipLaunchKernelGGL(bench<9>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 10: hipLaunchKernelGGL(bench<10>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 11: hipLaunchKernelGGL(bench<11>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(c, dc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(got.data(), dout, sizeof(double) * 41 * D, hipMemcpyDeviceToHost);
        if (v == 0) ref = got;
        double err = 0;
        for (int i = 0; i < 41; i++) for (int j = 0; j < D; j++) {
            if (i < D && j > i) continue;                   /* above the diagonal of the factor: not defined */
            err = fmax(err, fabs(got[i * D + j] - ref[i * D + j]));
        }
        const char *names[] = {"readlane left-looking (production)", "dpp left-looking, nops", "dpp left-looking, no nops", "dpp right-looking, nops",
                               "readlane, two columns per step", "dpp two columns per step, nops", "dpp two columns per step, no nops", "dpp right-looking, no nops", "lds row broadcast, left-looking", "lds row broadcast, max-select pivot", "readlane, max-select pivot",
                               "4-col panels on VALU + rank-4 MFMA updates"};
        printf("variant %d %-40s %6lld cycles per 41x16 tall potrf   (max diff vs v0 %.2e)\n", v, names[v], c[0], err);
    }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, dc); hipDeviceSynchronize();
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, dc); hipDeviceSynchronize();
    hipMemcpy(c, dc, 48, hipMemcpyDeviceToHost);
    printf("dependent DP fma: %lld cyc | readlane+fma: %lld | dependent fmac_dpp (nop): %lld | 4 indep fmac_dpp chains (per instr): %lld | rsqrt3 chain: %lld | mov_dpp+add: %lld\n", c[0], c[1], c[2], c[3], c[4], c[5]);
    return 0;
}
