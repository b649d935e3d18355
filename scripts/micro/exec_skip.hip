// Issue rate of fp64 VALU work from one wave, and how it scales with waves per SIMD
// and with the number of active lanes.  Cycles are shader-clock cycles (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
template <int CHAINS>
__global__ void k(double* out, long long* cyc, int active, int n)
{
    if ((int)(threadIdx.x & 63) >= active) return;
    double a[CHAINS];
    for (int j = 0; j < CHAINS; ++j) a[j] = threadIdx.x * 1e-3 + j;
    double m = 1.0000001, c = 1e-9;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) a[j] = fma(a[j], m, c);
    }
    long long t1 = clock64();
    double s = 0; for (int j = 0; j < CHAINS; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int CHAINS>
int run(double* d, long long* c, int active, int blocks, int threads, int n)
{
    k<CHAINS><<<blocks, threads>>>(d, c, active, 100);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); k<CHAINS><<<blocks, threads>>>(d, c, active, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    static long long hb[1 << 16]; int nw = blocks * (threads / 64); CK(hipMemcpy(hb, c, 8 * nw, hipMemcpyDeviceToHost));
    long long h = 0; double mean = 0; for (int i = 0; i < nw; ++i) { if (hb[i] > h) h = hb[i]; mean += hb[i]; } mean /= nw;
    printf("chains %d active %2d blocks %5d x %3d thr: %.3f ms, cycles per fma instr: slowest wave %.2f mean %.2f; slowest-wave cycles / kernel time = %.2f GHz\n",
           CHAINS, active, blocks, threads, ms, (double)h / ((double)CHAINS * n), mean / ((double)CHAINS * n), h / (ms * 1e6));
    return 0;
}
int main()
{
    double* d; long long* c; CK(hipMalloc(&d, 1 << 26)); CK(hipMalloc(&c, 1 << 20));
    int n = 100000;
    run<8>(d, c, 64, 256, 64, n);
    run<8>(d, c, 64, 1024, 64, n);
    run<8>(d, c, 64, 2048, 64, n);
    run<8>(d, c, 64, 4096, 64, n);
    run<8>(d, c, 64, 1024, 256, n);
    run<8>(d, c, 64, 2048, 256, n);
    run<8>(d, c, 16, 256, 64, n);
    run<8>(d, c, 16, 1024, 64, n);
    run<1>(d, c, 64, 1024, 64, n);
    run<2>(d, c, 64, 1024, 64, n);
    run<4>(d, c, 64, 1024, 64, n);
    run<16>(d, c, 64, 1024, 64, n);
    return 0;
}
