// hbm_write.hip -- what the chip takes in stores: a grid of wavefronts writing 16 bytes per lane (1 KB per wavefront and
// store, the unit of the bit-parallel kernels' plane stores) with plain, non-temporal and write-through stores, the
// wavefronts' streams laid out as the DP kernels lay theirs (every wavefront walks its own 384 KB region, ten streams
// of it side by side).  Prints GB/s per flavour.  Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_write.hip -o tools/bin/hbm_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// region per wavefront: STREAMS streams of GROUPS uint4 slots per lane; slot (s, g) of lane l lies at ((s * GROUPS + g) * 64 + l) * 16 bytes
template <int FLAVOUR>
__global__ void __launch_bounds__(256) write_kernel(uint4 *out, int streams, int groups, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4 *base = out + wave * (size_t)streams * groups * 64 + lane;
    uint4 v = make_uint4(seed + lane, seed * 3u, seed ^ (uint32_t)wave, 7u);
    for (int g = 0; g < groups; ++g) {
        for (int s = 0; s < streams; ++s) {
            uint4 *p = base + ((size_t)s * groups + g) * 64;
            v.x += v.y;
            if (FLAVOUR == 0) {
                *p = v;
            } else if (FLAVOUR == 1) {
                __builtin_nontemporal_store(v.x, &p->x);
                __builtin_nontemporal_store(v.y, &p->y);
                __builtin_nontemporal_store(v.z, &p->z);
                __builtin_nontemporal_store(v.w, &p->w);
            } else {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 vv = {v.x, v.y, v.z, v.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(vv) : "memory");
            }
        }
    }
}

int main()
{
    const int waves = 15625 * 1, streams = 10, groups = 38;  // 1M pairs: 15 625 bundles, ten plane streams of 38 column groups
    const size_t bytes = (size_t)waves * streams * groups * 64 * 16;
    uint4 *buf;
    CHECK(hipMalloc(&buf, bytes + (1 << 20)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[3] = {"plain", "non-temporal", "write-through (sc0 sc1)"};
    for (int f = 0; f < 3; ++f) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            const dim3 grid((waves + 3) / 4), block(256);
            if (f == 0) hipLaunchKernelGGL(write_kernel<0>, grid, block, 0, 0, buf, streams, groups, rep);
            if (f == 1) hipLaunchKernelGGL(write_kernel<1>, grid, block, 0, 0, buf, streams, groups, rep);
            if (f == 2) hipLaunchKernelGGL(write_kernel<2>, grid, block, 0, 0, buf, streams, groups, rep);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-26s %.3f ms for %.2f GB = %.0f GB/s\n", names[f], best, bytes / 1e9, bytes / 1e6 / best);
    }
    CHECK(hipFree(buf));
    return 0;
}
