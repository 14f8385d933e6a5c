// lone_wave.hip -- what ONE wavefront per SIMD can issue (gfx950): independent chains of v_max3_i32 / v_add_u32 written
// in C (no inline asm, so no boundary pads), 1 / 2 / 4 waves per SIMD.  Prints cycles per wave-instruction per WAVE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 4096;
template <int CH, int OP>
__global__ void __launch_bounds__(256) k(int *out, int seed, unsigned long long *clk)
{
    int a[CH], b[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { a[c] = seed + threadIdx.x + c; b[c] = seed * 3 + c; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) a[c] = max(max(a[c], b[c]), it);          // v_max3_i32
            if (OP == 1) a[c] = a[c] + b[c] + 0;                    // v_add_u32
            if (OP == 2) a[c] = __builtin_amdgcn_alignbit(a[c], b[c], 2);
            if (OP == 3) a[c] = __builtin_amdgcn_update_dpp(b[c], a[c], 0x138, 0xf, 0xf, false) + it;  // dpp + add
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int r = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) r += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int CH, int OP>
void run(int *d_out, unsigned long long *d_clk, int cus, int wps, const char *name)
{
    hipLaunchKernelGGL((k<CH, OP>), dim3(cus * wps), dim3(256), 0, 0, d_out, 1, d_clk);
    CHECK(hipDeviceSynchronize());
    unsigned long long clk;
    CHECK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
    const int per = (OP == 3) ? 2 : 1;
    printf("%-14s chains=%2d waves/SIMD=%d  %.2f cycles per instruction per wave (dependent distance = %d instr)\n", name, CH, wps,
           (double)clk / ((double)ITERS * CH * per), CH * per);
}
int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int *d_out; unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 8 * 256 * sizeof(int)));
    CHECK(hipMalloc(&d_clk, 16));
    const int cus = prop.multiProcessorCount;
    for (int wps : {1, 2, 4}) {
        run<1, 0>(d_out, d_clk, cus, wps, "v_max3_i32"); run<2, 0>(d_out, d_clk, cus, wps, "v_max3_i32"); run<4, 0>(d_out, d_clk, cus, wps, "v_max3_i32"); run<8, 0>(d_out, d_clk, cus, wps, "v_max3_i32");
        run<1, 1>(d_out, d_clk, cus, wps, "v_add_u32"); run<2, 1>(d_out, d_clk, cus, wps, "v_add_u32"); run<4, 1>(d_out, d_clk, cus, wps, "v_add_u32"); run<8, 1>(d_out, d_clk, cus, wps, "v_add_u32");
        run<1, 2>(d_out, d_clk, cus, wps, "v_alignbit"); run<4, 2>(d_out, d_clk, cus, wps, "v_alignbit");
        run<1, 3>(d_out, d_clk, cus, wps, "dpp+add"); run<4, 3>(d_out, d_clk, cus, wps, "dpp+add");
    }
    return 0;
}
