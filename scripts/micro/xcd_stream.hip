// Memory path per XCD: every workgroup streams a private slice of a large buffer (read, then write) and
// follows a chain of dependent loads through a region of its own, timing itself with the 100 MHz real-time
// counter and recording the XCD it ran on (HW_REG_XCC_ID).  Printed per XCD: mean and slowest workgroup of
// each test.  Answers whether some XCDs see a slower HBM path than others when the whole chip streams.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/xcd_stream.hip -o scripts/micro/xcd_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

struct Rec {
    unsigned long long t0, t1;
    unsigned xcc, pad;
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// each workgroup reads `per` double2 from its own slice
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ buf, size_t per, double* sink, Rec* rec)
{
    const double2* p = buf + (size_t)blockIdx.x * per;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    for (size_t i = threadIdx.x; i < per; i += 256) {
        double2 v = p[i];
        s += v.x + v.y;
    }
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0)
        rec[blockIdx.x] = Rec{t0, t1, xcc_id(), 0};
}

__global__ __launch_bounds__(256) void k_write(double2* __restrict__ buf, size_t per, Rec* rec)
{
    double2* p = buf + (size_t)blockIdx.x * per;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (size_t i = threadIdx.x; i < per; i += 256)
        p[i] = double2{(double)i, 1.0};
    __threadfence();
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0)
        rec[blockIdx.x] = Rec{t0, t1, xcc_id(), 0};
}

// one lane per workgroup follows `hops` dependent loads through a ring laid out in the workgroup's own slice
__global__ __launch_bounds__(64) void k_chase(const unsigned* __restrict__ ring, size_t per, int hops, unsigned* sink, Rec* rec)
{
    const unsigned* p = ring + (size_t)blockIdx.x * per;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned j = 0;
    if (threadIdx.x == 0)
        for (int h = 0; h < hops; ++h)
            j = p[j];
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        sink[blockIdx.x] = j;
        rec[blockIdx.x] = Rec{t0, t1, xcc_id(), 0};
    }
}

static void report(const char* name, const std::vector<Rec>& r, double bytes_per_block, double hops)
{
    double sum[8] = {0}, mx[8] = {0};
    int cnt[8] = {0};
    unsigned long long lo = ~0ull, hi = 0;
    for (const Rec& x : r) {
        double us = (x.t1 - x.t0) / 100.0;
        int c = x.xcc & 7;
        sum[c] += us;
        if (us > mx[c])
            mx[c] = us;
        cnt[c]++;
        if (x.t0 < lo)
            lo = x.t0;
        if (x.t1 > hi)
            hi = x.t1;
    }
    printf("%s: whole launch %.1f us", name, (hi - lo) / 100.0);
    if (bytes_per_block > 0)
        printf(" = %.2f TB/s", bytes_per_block * r.size() / ((hi - lo) / 100.0 * 1e-6) / 1e12);
    printf("\n");
    for (int c = 0; c < 8; ++c) {
        if (!cnt[c])
            continue;
        printf("  XCD %d: %4d workgroups, mean %.1f us, slowest %.1f us", c, cnt[c], sum[c] / cnt[c], mx[c]);
        if (bytes_per_block > 0)
            printf(", mean rate per workgroup %.2f GB/s", bytes_per_block / (sum[c] / cnt[c] * 1e-6) / 1e9);
        if (hops > 0)
            printf(", %.0f ns per dependent load", sum[c] / cnt[c] * 1e3 / hops);
        printf("\n");
    }
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048;          // 8 workgroups per CU
    const size_t per = (size_t)(argc > 2 ? atoi(argv[2]) : 4) << 16; // double2 per workgroup: 4 MiB
    double2* buf;
    double* sink;
    Rec* rec;
    CK(hipMalloc(&buf, blocks * per * sizeof(double2)));
    CK(hipMalloc(&sink, (size_t)blocks * 256 * 8));
    CK(hipMalloc(&rec, blocks * sizeof(Rec)));
    CK(hipMemset(buf, 0, blocks * per * sizeof(double2)));
    std::vector<Rec> h(blocks);
    for (int rep = 0; rep < 2; ++rep) {
        k_read<<<blocks, 256>>>(buf, per, sink, rec);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), rec, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
    report("read stream ", h, (double)per * 16, 0);
    for (int rep = 0; rep < 2; ++rep) {
        k_write<<<blocks, 256>>>(buf, per, rec);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), rec, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
    report("write stream", h, (double)per * 16, 0);
    // dependent loads: a ring with a stride of 4 KiB + 64 B through each workgroup's slice (every hop a new page row)
    {
        const size_t words = per * 4; // unsigned per slice
        std::vector<unsigned> ring(words, 0);
        const size_t stride = (4096 + 64) / 4;
        size_t n = words / stride;
        for (size_t q = 0; q < n; ++q)
            ring[q * stride] = (unsigned)(((q + 1) % n) * stride);
        for (int b = 0; b < blocks; ++b)
            CK(hipMemcpy((unsigned*)buf + (size_t)b * words, ring.data(), words * 4, hipMemcpyHostToDevice));
        const int hops = (int)n; // one lap: every hop a line nobody has loaded since the ring was written
        k_chase<<<blocks, 64>>>((const unsigned*)buf, words, hops, (unsigned*)sink, rec);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), rec, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
        report("dependent loads (all workgroups at once)", h, 0, hops);
        // the same with one workgroup per XCD only (unloaded latency)
        k_chase<<<8, 64>>>((const unsigned*)buf, words, hops, (unsigned*)sink, rec);
        CK(hipDeviceSynchronize());
        std::vector<Rec> h8(8);
        CK(hipMemcpy(h8.data(), rec, 8 * sizeof(Rec), hipMemcpyDeviceToHost));
        report("dependent loads (8 workgroups)", h8, 0, hops);
    }
    return 0;
}
