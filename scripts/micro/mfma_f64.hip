// v_mfma_f64_16x16x4_f64: issue rate from one wave, and whether the result is bit-for-bit the
// k-ordered fma chain D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_num(const double* A, const double* B, const double* Cin, double* D)
{
    // A [16][4] row-major, B [4][16], C/D [16][16]
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = Cin[((l >> 4) + 4 * r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void k_rate(double* out, long long* cyc, int n)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
    long long t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC>
int rate(double* d, long long* c, int blocks, int threads, int n)
{
    k_rate<NACC><<<blocks, threads>>>(d, c, 100);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); k_rate<NACC><<<blocks, threads>>>(d, c, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    static long long hb[1 << 16]; int nw = blocks * (threads / 64); CK(hipMemcpy(hb, c, 8 * nw, hipMemcpyDeviceToHost));
    long long h = 0; for (int i = 0; i < nw; ++i) if (hb[i] > h) h = hb[i];
    double flop = 2.0 * 16 * 16 * 4 * (double)NACC * n * nw;
    printf("acc %d, %5d waves (%d per block): %.3f ms, %.1f cycles per mfma (slowest wave), %.2f TFLOP/s\n", NACC, nw,
           threads / 64, ms, (double)h / ((double)NACC * n), flop / (ms * 1e-3) / 1e12);
    return 0;
}

int main()
{
    double hA[64], hB[64], hC[256], hD[256];
    srand(5);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 40 - 20); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    for (auto& v : hC) v = rnd();
    double *A, *B, *C, *D; long long* cy;
    CK(hipMalloc(&A, 512)); CK(hipMalloc(&B, 512)); CK(hipMalloc(&C, 2048)); CK(hipMalloc(&D, 1 << 24)); CK(hipMalloc(&cy, 1 << 20));
    CK(hipMemcpy(A, hA, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(C, hC, 2048, hipMemcpyHostToDevice));
    k_num<<<1, 64>>>(A, B, C, D);
    CK(hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost));
    int bad_fwd = 0, bad_rev = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double f = hC[i * 16 + j], r = hC[i * 16 + j];
            for (int k = 0; k < 4; ++k) f = std::fma(hA[i * 4 + k], hB[k * 16 + j], f);
            for (int k = 3; k >= 0; --k) r = std::fma(hA[i * 4 + k], hB[k * 16 + j], r);
            bad_fwd += (f != hD[i * 16 + j]);
            bad_rev += (r != hD[i * 16 + j]);
        }
    printf("numerics: mismatches vs k-ascending fma chain %d / 256, vs k-descending %d / 256\n", bad_fwd, bad_rev);
    int n = 20000;
    rate<1>(D, cy, 1024, 64, n);
    rate<2>(D, cy, 1024, 64, n);
    rate<4>(D, cy, 1024, 64, n);
    rate<4>(D, cy, 256, 64, n);
    rate<4>(D, cy, 1024, 256, n / 2);
    rate<4>(D, cy, 2048, 256, n / 4);
    return 0;
}
