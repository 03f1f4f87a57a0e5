// Microbenchmark: single-wave tall Cholesky (25 x 16, f64) variants, cycles per factorization.
// Build: hipcc --offload-arch=gfx950 -O3 -o potrf_bench potrf_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define D 16
#define R 25

__device__ __forceinline__ double rdlane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pivot_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    double e = fma(-(p * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(p * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return p > 0.0 ? y : 0.0;
}

// V1: right-looking, readlane broadcasts
__device__ __forceinline__ void potrf_v1(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        const double pj = rdlane(T[j], j);
        const double finv = pivot_rsqrt(pj);
        T[j] *= finv;
#pragma unroll
        for (int c = j + 1; c < D; c++) { const double lc = rdlane(T[j], c); T[c] = fma(-T[j], lc, T[c]); }
    }
}

// V2: right-looking, column all-gather through LDS (1 write + 8 b128 reads per column)
__device__ __forceinline__ void potrf_v2(double (&T)[D], int lane, double *lds) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        if (lane < D) lds[lane] = T[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const double pj = lds[j];
        const double finv = pivot_rsqrt(pj);
        T[j] *= finv;
#pragma unroll
        for (int c = j + 1; c < D; c++) { const double lc = lds[c] * finv; T[c] = fma(-T[j], lc, T[c]); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// V3: left-looking with row broadcast through LDS: L kept in LDS row-major (stride 17)
__device__ __forceinline__ void potrf_v3(double (&T)[D], int lane, double *lds) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], lds[j * 17 + k], s);     // row j of L (broadcast read)
        // pivot = s of lane j
        if (lane == j) lds[16 * 17 + 0] = s;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const double pj = lds[16 * 17 + 0];
        const double finv = pivot_rsqrt(pj);
        s *= finv;
        T[j] = s;
        if (lane < D) lds[lane * 17 + j] = s;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// V4: left-looking, row broadcast by readlane (j readlanes per column, issued up front)
__device__ __forceinline__ void potrf_v4(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        const double finv = pivot_rsqrt(pj);
        T[j] = s * finv;
    }
}

// one-Newton-step reciprocal square root
__device__ __forceinline__ double pivot_rsqrt1(double p) {
    double y = __builtin_amdgcn_rsq(p);
    double e = fma(-(p * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return p > 0.0 ? y : 0.0;
}
// V6: left-looking, readlane, 4 partial accumulators per dot product
__device__ __forceinline__ void potrf_v6(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s0 = T[j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < j; k++) {
            const double l = rdlane(T[k], j);
            if ((k & 3) == 0) s0 = fma(-T[k], l, s0);
            if ((k & 3) == 1) s1 = fma(-T[k], l, s1);
            if ((k & 3) == 2) s2 = fma(-T[k], l, s2);
            if ((k & 3) == 3) s3 = fma(-T[k], l, s3);
        }
        const double s = (s0 + s1) + (s2 + s3);
        const double pj = rdlane(s, j);
        const double finv = pivot_rsqrt(pj);
        T[j] = s * finv;
    }
}
// V7: V4 with one Newton step
__device__ __forceinline__ void potrf_v7(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        const double finv = pivot_rsqrt1(pj);
        T[j] = s * finv;
    }
}
// V8: pivot lane computes its own pivot first (its dot product needs only its own registers), so the
// reciprocal square root chain overlaps the other rows' updates
__device__ __forceinline__ void potrf_v8(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        // pivot: p_j = T[j][j] - sum_k l_jk^2 needs lane j's own registers only -> broadcast squares
        double pj = rdlane(T[j], j);
#pragma unroll
        for (int k = 0; k < j; k++) { const double l = rdlane(T[k], j); pj = fma(-l, l, pj); }
        const double finv = pivot_rsqrt(pj);
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        T[j] = s * finv;
    }
}

// third-order one-step reciprocal square root (4 dependent ops after v_rsq)
__device__ __forceinline__ double pivot_rsqrt3(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y0), y0, 1.0);
    const double h = fma(0.375, e, 0.5);
    const double y = fma(y0 * e, h, y0);
    return p > 0.0 ? y : 0.0;
}
// reciprocal: v_rcp_f64 + one Newton step
__device__ __forceinline__ double pivot_rcp1(double p) {
    const double y0 = __builtin_amdgcn_rcp(p);
    const double e = fma(-p, y0, 1.0);
    const double y = fma(y0, e, y0);
    return p > 0.0 ? y : 0.0;
}
// V9: V4 with the one-step rsqrt
__device__ __forceinline__ void potrf_v9(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        const double finv = pivot_rsqrt3(pj);
        T[j] = s * finv;
    }
}
// V10: right-looking (V1) with the one-step rsqrt
__device__ __forceinline__ void potrf_v10(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        const double pj = rdlane(T[j], j);
        const double finv = pivot_rsqrt3(pj);
        T[j] *= finv;
#pragma unroll
        for (int c = j + 1; c < D; c++) { const double lc = rdlane(T[j], c); T[c] = fma(-T[j], lc, T[c]); }
    }
}
// V11: right-looking, unscaled columns (LDL'-style recurrence): only a reciprocal (rcp + 1 Newton) sits on the
// pivot-to-pivot chain, the rsqrt scaling of the columns happens off the critical path
__device__ __forceinline__ void potrf_v11(double (&T)[D], int lane) {
    double finv[D];
#pragma unroll
    for (int j = 0; j < D; j++) {
        const double pj = rdlane(T[j], j);
        const double dinv = pivot_rcp1(pj);
#pragma unroll
        for (int c = j + 1; c < D; c++) { const double lc = rdlane(T[j], c) * dinv; T[c] = fma(-T[j], lc, T[c]); }
        finv[j] = pivot_rsqrt3(pj);
    }
#pragma unroll
    for (int j = 0; j < D; j++) T[j] *= finv[j];
}
// V12: as V11 but the scaling of column j is issued right after its trailing updates (no finv array)
__device__ __forceinline__ void potrf_v12(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        const double pj = rdlane(T[j], j);
        const double dinv = pivot_rcp1(pj);
        const double tj = T[j];
#pragma unroll
        for (int c = j + 1; c < D; c++) { const double lc = rdlane(tj, c) * dinv; T[c] = fma(-tj, lc, T[c]); }
        T[j] = tj * pivot_rsqrt3(pj);
    }
}

// V13: left-looking; row j of the factor comes from LDS for the columns finished at least two steps ago
// (each finished column is stored once, row-major with stride 18, and read back as a broadcast, two
// entries per ds_read_b128), the newest column by readlane.  Reads for step j are issued during step j-1.
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ void potrf_v13(double (&T)[D], int lane, double *lds_generic) {
    lds_f64 *lds = (lds_f64 *)lds_generic;
    constexpr int LS = 18;
    double row[D], rown[D];
#pragma unroll
    for (int k = 0; k < D; k++) { row[k] = 0.0; rown[k] = 0.0; }
#pragma unroll
    for (int j = 0; j < D; j++) {
        /* prefetch row j+1 (columns 0 .. j-1 are in LDS; column j is not finished yet) */
        if (j + 1 < D) {
#pragma unroll
            for (int k = 0; k < j; k++) rown[k] = lds[(j + 1) * LS + k];
        }
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j - 1; k++) s = fma(-T[k], row[k], s);
        if (j >= 1) s = fma(-T[j - 1], rdlane(T[j - 1], j), s);
        const double pj = rdlane(s, j);
        const double finv = pivot_rsqrt3(pj);
        T[j] = s * finv;
        lds[lane * LS + j] = T[j];
#pragma unroll
        for (int k = 0; k < D; k++) row[k] = rown[k];
    }
}

// V5: right-looking with ds_bpermute broadcast (__shfl)
__device__ __forceinline__ void potrf_v5(double (&T)[D], int lane) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        const double pj = __shfl(T[j], j, 64);
        const double finv = pivot_rsqrt(pj);
        T[j] *= finv;
#pragma unroll
        for (int c = j + 1; c < D; c++) { const double lc = __shfl(T[j], c, 64); T[c] = fma(-T[j], lc, T[c]); }
    }
}

template <int V>
__global__ void bench(const double *in, double *out, long long *cycles, int reps) {
    __shared__ double lds[64 * 18 + 8];
    const int lane = threadIdx.x;
    double T0[D], T[D];
#pragma unroll
    for (int j = 0; j < D; j++) T0[j] = in[(lane < R ? lane : 0) * D + j];
    double acc = 0.0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < D; j++) T[j] = T0[j] + acc * 1e-300;
        if (V == 1) potrf_v1(T, lane);
        if (V == 2) potrf_v2(T, lane, lds);
        if (V == 3) potrf_v3(T, lane, lds);
        if (V == 4) potrf_v4(T, lane);
        if (V == 5) potrf_v5(T, lane);
        if (V == 6) potrf_v6(T, lane);
        if (V == 7) potrf_v7(T, lane);
        if (V == 8) potrf_v8(T, lane);
        if (V == 9) potrf_v9(T, lane);
        if (V == 10) potrf_v10(T, lane);
        if (V == 11) potrf_v11(T, lane);
        if (V == 12) potrf_v12(T, lane);
        if (V == 13) potrf_v13(T, lane, lds);
#pragma unroll
        for (int j = 0; j < D; j++) acc += T[j];
    }
    long long t1 = clock64();
#pragma unroll
    for (int j = 0; j < D; j++) out[lane * D + j] = T[j];
    if (lane == 0) { cycles[0] = (t1 - t0) / reps; }
    if (acc == 12345.678) out[0] = acc;
}

// pure latency probes
__global__ void probe(double *out, long long *cycles) {
    const int lane = threadIdx.x;
    double x = 1.0 + lane * 1e-3, y = 0.5;
    long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; i++) x = fma(x, y, 1.0);          // dependent DP FMA chain
    long long t1 = clock64();
    double z = x;
#pragma unroll
    for (int i = 0; i < 256; i++) { z = fma(z, rdlane(z, i & 15), 1.0); }   // readlane + dependent FMA
    long long t2 = clock64();
    double w = z;
#pragma unroll
    for (int i = 0; i < 256; i++) { w = fma(w, __shfl(w, i & 15, 64), 1.0); }   // bpermute + dependent FMA
    long long t3 = clock64();
    double a0 = w, a1 = w + 1, a2 = w + 2, a3 = w + 3;
#pragma unroll
    for (int i = 0; i < 64; i++) { a0 = fma(a0, y, 1.0); a1 = fma(a1, y, 1.0); a2 = fma(a2, y, 1.0); a3 = fma(a3, y, 1.0); }  // 4 independent chains
    long long t4 = clock64();
    double r = a0 + a1 + a2 + a3;
#pragma unroll
    for (int i = 0; i < 64; i++) r = pivot_rsqrt(r + 2.0);
    long long t5 = clock64();
    out[lane] = r;
    if (lane == 0) { cycles[0] = (t1 - t0) / 256; cycles[1] = (t2 - t1) / 256; cycles[2] = (t3 - t2) / 256; cycles[3] = (t4 - t3) / 256; cycles[4] = (t5 - t4) / 64; }
}

__global__ void rsq_acc(double *out) {
    const int l = threadIdx.x;
    const double p = 0.37 + 1.731 * l + 1e-3 * l * l;
    out[l] = __builtin_amdgcn_rsq(p); out[64 + l] = pivot_rsqrt1(p); out[128 + l] = pivot_rsqrt(p); out[192 + l] = p;
}

int main() {
    std::vector<double> h(R * D, 0.0);
    // SPD 16x16 + extra rows
    for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) h[i * D + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
    for (int i = D; i < R; i++) for (int j = 0; j < D; j++) h[i * D + j] = 0.1 * ((i * 7 + j * 3) % 11);
    double *din, *dout; long long *dc;
    hipMalloc(&din, sizeof(double) * R * D); hipMalloc(&dout, sizeof(double) * 64 * D); hipMalloc(&dc, 64);
    hipMemcpy(din, h.data(), sizeof(double) * R * D, hipMemcpyHostToDevice);
    long long c[8];
    std::vector<double> ref(64 * D), got(64 * D);
    for (int v = 1; v <= 13; v++) {
        for (int it = 0; it < 2; it++) {
            switch (v) {
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
                case 12: hipLaunchKernelGGL(bench<12>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
                case 13: hipLaunchKernelGGL(bench<13>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(c, dc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(got.data(), dout, sizeof(double) * 64 * D, hipMemcpyDeviceToHost);
        if (v == 1) ref = got;
        double err = 0;
        for (int i = 0; i < R; i++) for (int j = 0; j <= (i < D ? i : D - 1); j++) err = fmax(err, fabs(got[i * D + j] - ref[i * D + j]));
        printf("variant %d: %lld cycles per 25x16 tall potrf   (max diff vs v1 %.2e)\n", v, c[0], err);
    }
    { hipLaunchKernelGGL(rsq_acc, dim3(1), dim3(64), 0, 0, dout); hipDeviceSynchronize(); std::vector<double> r(256); hipMemcpy(r.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
      double e0 = 0, e1 = 0, e2 = 0; for (int i = 0; i < 64; i++) { double t = 1.0 / sqrt(r[192 + i]); e0 = fmax(e0, fabs(r[i] / t - 1)); e1 = fmax(e1, fabs(r[64 + i] / t - 1)); e2 = fmax(e2, fabs(r[128 + i] / t - 1)); }
      printf("rsq relative error: raw %.2e, 1 Newton %.2e, 2 Newton %.2e\n", e0, e1, e2); }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, dc); hipDeviceSynchronize();
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, dc); hipDeviceSynchronize();
    hipMemcpy(c, dc, 40, hipMemcpyDeviceToHost);
    printf("dependent DP fma: %lld cyc | readlane+fma: %lld | bpermute+fma: %lld | 4 indep fma chains (per fma): %lld | pivot_rsqrt chain: %lld\n", c[0], c[1], c[2], c[3], c[4]);
    return 0;
}
