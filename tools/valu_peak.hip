// valu_peak.hip -- measures the integer-VALU issue peak of the GPU it runs on
// ("peak to divide by ... to be measured on the GPU box").
//
// Three instruction mixes, each a long unrolled stream over 8 independent
// register chains so that dependent-issue latency never limits it:
//   add   : v_add_u32
//   max   : v_max_i32
//   cell  : the linear-gap cell body of kernels.hip
//           (v_add_u32, v_cmp_gt_i32, v_addc_co_u32, v_max_i32, v_cmp_gt_i32, v_addc_co_u32, v_max_i32)
// Output: one JSON line with lane-ops/s for each mix.  A "lane-op" is one
// 32-bit operation in one lane (a wave64 instruction = 64 lane-ops).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;

template <int MIX>
__global__ void __launch_bounds__(256) peak_kernel(int *out, int seed)
{
    int a[8], b[8];
    unsigned c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x + k; b[k] = seed * 3 + k; c[k] = k; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MIX == 0) {
                asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0"
                             : "+v"(a[k]), "+v"(b[k]));
            } else if (MIX == 1) {
                asm volatile("v_max_i32 %0, %0, %1\n\tv_max_i32 %1, %1, %0\n\tv_max_i32 %0, %0, %1\n\tv_max_i32 %1, %1, %0"
                             : "+v"(a[k]), "+v"(b[k]));
            } else {
                // 7 instructions, the dependency shape of one DP cell
                asm volatile(
                    "v_add_u32 %0, %0, %1\n\t"
                    "v_cmp_gt_i32 vcc, %1, %0\n\t"
                    "v_addc_co_u32 %2, vcc, %2, %2, vcc\n\t"
                    "v_max_i32 %1, %1, %0\n\t"
                    "v_cmp_gt_i32 vcc, %0, %1\n\t"
                    "v_addc_co_u32 %2, vcc, %2, %2, vcc\n\t"
                    "v_max_i32 %0, %0, %1"
                    : "+v"(a[k]), "+v"(b[k]), "+v"(c[k])
                    :
                    : "vcc");
            }
        }
    }
    int r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += a[k] + b[k] + (int)c[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MIX>
double run(int *d_out, int blocks, int per_iter_instr)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(peak_kernel<MIX>, dim3(blocks), dim3(256), 0, 0, d_out, 1);
    CHECK(hipDeviceSynchronize());
    double best = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(peak_kernel<MIX>, dim3(blocks), dim3(256), 0, 0, d_out, rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = (double)blocks * 256 * ITERS * 8 * per_iter_instr;
        const double rate = ops / (ms * 1e-3);
        if (rate > best) best = rate;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    int *d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(int)));
    const double add = run<0>(d_out, blocks, 4);
    const double mx = run<1>(d_out, blocks, 4);
    const double cell = run<2>(d_out, blocks, 7);
    const double spec = (double)prop.multiProcessorCount * 4 * 32 * (prop.clockRate * 1e3);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"spec_lane_ops_per_s\": %.4g, "
           "\"add_lane_ops_per_s\": %.4g, \"max_lane_ops_per_s\": %.4g, \"cell_mix_lane_ops_per_s\": %.4g, "
           "\"cell_mix_gcups_at_7_ops\": %.1f}\n",
           prop.name, prop.multiProcessorCount, prop.clockRate / 1000, spec, add, mx, cell, cell / 7 / 1e9);
    return 0;
}
