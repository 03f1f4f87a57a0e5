// Instruction latency / issue probes for one wavefront on gfx950 (dependent f64 fma, mul, readlane + fma, rsq, mul + max).
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_lat probe_lat.hip ; the clock reads are tied to the data flow (s_memtime in asm volatile with the value as operand).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ double rdlane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
#define TICK(t, x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x) :: "memory")
__global__ void probe(double *out, long long *cyc, const double *in) {
    const int lane = threadIdx.x;
    double y = in[0], one = in[1];
    double x = in[2] + lane;
    unsigned long long t[16];
    int n = 0;
    TICK(t[n++], x);
#pragma unroll
    for (int i = 0; i < 128; i++) x = fma(x, y, one);                          /* dependent fma */
    TICK(t[n++], x);
#pragma unroll
    for (int i = 0; i < 128; i++) x = x * y;                                   /* dependent mul */
    TICK(t[n++], x);
#pragma unroll
    for (int i = 0; i < 128; i++) x = fma(x, rdlane(x, i & 15), one);          /* readlane pair + dependent fma */
    TICK(t[n++], x);
#pragma unroll
    for (int i = 0; i < 64; i++) x = __builtin_amdgcn_rsq(x) + one;            /* dependent rsq + add */
    TICK(t[n++], x);
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = x + i;
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = fma(a[q], y, one);                  /* 8 independent chains: 128 fma */
    x = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));     /* + 7 adds (3 deep) */
    TICK(t[n++], x);
#pragma unroll
    for (int i = 0; i < 64; i++) x = __builtin_fmax(x * y, one);               /* mul + max */
    TICK(t[n++], x);
#pragma unroll
    for (int i = 0; i < 64; i++) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 3); x = x + __hiloint2double(0x3ff00000, lo & 1); }  /* one readlane + int and + add */
    TICK(t[n++], x);
    out[lane] = x;
    if (lane == 0) for (int i = 0; i + 1 < n; i++) cyc[i] = t[i + 1] - t[i];
}
int main() {
    double *dout, *din; long long *dc; long long c[16];
    double hin[3] = {0.5, 1.0, 1.0};
    CK(hipMalloc(&dout, 64 * 8)); CK(hipMalloc(&dc, 128)); CK(hipMalloc(&din, 24)); CK(hipMemcpy(din, hin, 24, hipMemcpyHostToDevice));
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, dc, din); CK(hipGetLastError()); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(c, dc, 80, hipMemcpyDeviceToHost));
    printf("dependent fma: %.2f cyc\ndependent mul: %.2f\nreadlane pair + dep fma: %.2f\ndep rsq+add: %.2f\n128 fma in 8 chains + 7 adds: %.2f per fma\nmul+max: %.2f\nreadlane+and+add: %.2f\n",
        c[0] / 128.0, c[1] / 128.0, c[2] / 128.0, c[3] / 64.0, c[4] / 128.0, c[5] / 64.0, c[6] / 64.0);
    return 0;
}
