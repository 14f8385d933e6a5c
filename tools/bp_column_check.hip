// Checks ba_bp_column (bio_amd/csrc/bp_column.h, explicit v_bitop3_b32 truth tables) on the GPU against the plain
// and/or/xor statement of the same recurrence compiled for the host, on random thermometer planes, four columns chained.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../bio_amd/csrc/bp_column.h"
#define HD __host__ __device__
static HD inline uint32_t alignbit_ref(uint32_t hi, uint32_t lo, uint32_t s)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31));
#endif
}
template <int WW>
HD inline void column_ref(uint32_t (&V1)[WW], uint32_t (&V2)[WW], uint32_t (&V3)[WW], const uint32_t (&M)[WW], uint32_t (&oP)[WW])
{
    uint32_t carry3 = 0, carry2 = 0, p1prev = 0;
#pragma unroll
    for (int w = 0; w < WW; ++w) {
        const uint32_t m = M[w], v1 = V1[w], v2 = V2[w], v3 = V3[w];
        const uint32_t X0 = ~v1, nV2 = ~v2, nV3 = ~v3;
        const uint32_t C3 = m & X0;
        uint32_t co;
        const uint32_t s3 = __builtin_addc(X0, C3, carry3, &co);
        carry3 = co;
        const uint32_t Q3 = s3 ^ X0 ^ C3;
        const uint32_t X1 = v1 & nV2;
        const uint32_t A2 = (m & nV2) | (X1 & Q3);
        const uint32_t a2 = A2 | X0;
        const uint32_t s2 = __builtin_addc(a2, A2, carry2, &co);
        carry2 = co;
        const uint32_t Q2 = s2 ^ a2 ^ A2;
        const uint32_t X2 = v2 & nV3;
        const uint32_t P1 = (m & nV3) | X0 | (X1 & Q2) | (X2 & Q3);
        const uint32_t Q1 = alignbit_ref(P1, p1prev, 31);
        p1prev = P1;
        oP[w] = P1;
        const uint32_t U3 = m | v3, U2 = m | v2;
        const uint32_t nQ1 = ~Q1, nQ2 = ~Q2, nQ3 = ~Q3;
        V3[w] = nQ1 & U3;
        V2[w] = (nQ1 & U2) | (nQ2 & U3);
        V1[w] = nQ1 | (nQ2 & U2) | (nQ3 & U3);
    }
}
constexpr int WW = 4;
constexpr int NC = 4;  // columns chained, as in the kernel's column group
// in: per case V1,V2,V3 [WW] then NC x M[WW]; out: per column V1 [WW], P [WW]; then final V2, V3
constexpr int IN_W = (3 + NC) * WW, OUT_W = (2 * NC + 2) * WW;
template <bool BARRIER, bool REF>
HD inline void chain(const uint32_t *in, uint32_t *out)
{
    uint32_t V1[WW], V2[WW], V3[WW], M[NC][WW], accV[NC][WW], accP[NC][WW];
    for (int w = 0; w < WW; ++w) {
        V1[w] = in[0 * WW + w];
        V2[w] = in[1 * WW + w];
        V3[w] = in[2 * WW + w];
        for (int c = 0; c < NC; ++c) M[c][w] = in[(3 + c) * WW + w];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (REF) column_ref<WW>(V1, V2, V3, M[c], accP[c]); else ba_bp_column<WW>(V1, V2, V3, M[c], accP[c]);
#pragma unroll
        for (int w = 0; w < WW; ++w) accV[c][w] = V1[w];
#ifdef __HIP_DEVICE_COMPILE__
        if (BARRIER) {
#pragma unroll
            for (int w = 0; w < WW; ++w) asm volatile("" : "+v"(V1[w]), "+v"(V2[w]), "+v"(V3[w]));
        }
#endif
    }
    for (int c = 0; c < NC; ++c)
        for (int w = 0; w < WW; ++w) {
            out[(2 * c + 0) * WW + w] = accV[c][w];
            out[(2 * c + 1) * WW + w] = accP[c][w];
        }
    for (int w = 0; w < WW; ++w) {
        out[(2 * NC + 0) * WW + w] = V2[w];
        out[(2 * NC + 1) * WW + w] = V3[w];
    }
}
template <bool BARRIER>
__global__ void k(const uint32_t *in, uint32_t *out, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    chain<BARRIER, false>(in + (size_t)t * IN_W, out + (size_t)t * OUT_W);
}
int main()
{
    const int n = 1 << 16;
    std::vector<uint32_t> in((size_t)n * IN_W), out((size_t)n * OUT_W), ref((size_t)n * OUT_W);
    srand(5);
    auto r32 = [] { return ((uint32_t)rand() << 17) ^ ((uint32_t)rand() << 4) ^ (uint32_t)rand(); };
    for (int t = 0; t < n; ++t)
        for (int w = 0; w < WW; ++w) {
            uint32_t a = r32() & r32() & ((t & 1) ? r32() : ~0u), b = a & r32(), c = b & r32();
            if (t % 7 == 0) a = b = c = 0;
            uint32_t *q = &in[(size_t)t * IN_W];
            q[0 * WW + w] = a;
            q[1 * WW + w] = b;
            q[2 * WW + w] = c;
            // four symbols: disjoint match vectors, as the kernel's
            uint32_t m0 = r32(), m1 = r32() & ~m0, m2 = r32() & ~m0 & ~m1, m3 = ~(m0 | m1 | m2);
            const uint32_t mm[4] = {m0, m1, m2, m3};
            for (int cc = 0; cc < NC; ++cc) q[(3 + cc) * WW + w] = mm[(t >> (2 * cc)) & 3];
        }
    uint32_t *di, *dout;
    (void)hipMalloc(&di, in.size() * 4);
    (void)hipMalloc(&dout, out.size() * 4);
    (void)hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    for (int t = 0; t < n; ++t) chain<false, true>(&in[(size_t)t * IN_W], &ref[(size_t)t * OUT_W]);
    int total = 0;
    {   // the merged forms against the plain statement, both on the host
        std::vector<uint32_t> h(OUT_W);
        int bad = 0;
        for (int t = 0; t < n; ++t) {
            chain<false, false>(&in[(size_t)t * IN_W], h.data());
            for (int i = 0; i < OUT_W; ++i) bad += h[i] != ref[(size_t)t * OUT_W + i];
        }
        printf("host, explicit truth tables vs plain statement: %d differing words\n", bad);
        total += bad;
    }
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) hipLaunchKernelGGL(k<false>, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
        else hipLaunchKernelGGL(k<true>, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
        (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < out.size(); ++i)
            if (out[i] != ref[i] && bad++ < 6)
                printf("pass %d case %zu word %zu gpu=%08x host=%08x\n", pass, i / OUT_W, i % OUT_W, out[i], ref[i]);
        printf("pass %d (%s): %d cases, %d differing words\n", pass, pass ? "value barrier between columns" : "plain", n, bad);
        total += bad;
    }
    return total != 0;
}
