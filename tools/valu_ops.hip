// valu_ops.hip -- per-instruction VALU throughput on the GPU it runs on.
// 16 independent register chains per lane, 8 waves per SIMD; each chain applies
// ONE instruction repeatedly, so the issue rate of that instruction is what is
// measured.  Prints cycles per wave64 instruction per SIMD assuming the
// in-kernel clock measured with s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2048;
constexpr int CH = 16;

#define OPK(str) asm volatile(str : "+v"(a[k]) : "v"(b[k]), "v"(c[k]) : "vcc")
#define OPK2(str) asm volatile(str : "+v"(a[k]), "+v"(c[k]) : "v"(b[k]) : "vcc")  // %0 = a, %1 = c (both read and written), %2 = b
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) op_kernel(int *out, int seed, unsigned long long *clk)
{
    int a[CH], b[CH], c[CH];
    v2f p2[CH], q2[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) { a[k] = seed + threadIdx.x + k; b[k] = seed * 3 + k; c[k] = seed ^ k; p2[k] = {(float)(seed + k), 1.0f}; q2[k] = {0.5f, (float)k}; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (OP == 0) OPK("v_add_u32 %0, %0, %1");
            if (OP == 1) OPK("v_max_i32 %0, %0, %1");
            if (OP == 2) OPK("v_max3_i32 %0, %0, %1, %2");
            if (OP == 3) OPK("v_cmp_gt_i32 vcc, %0, %1");
            if (OP == 4) OPK("v_addc_co_u32 %0, vcc, %0, %1, vcc");
            if (OP == 5) OPK("v_cndmask_b32 %0, %0, %1, vcc");
            if (OP == 6) OPK("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD");
            if (OP == 7) OPK("v_pk_add_i16 %0, %0, %1");
            if (OP == 8) OPK("v_pk_max_i16 %0, %0, %1");
            if (OP == 9) OPK("v_max_f32 %0, %0, %1");
            if (OP == 10) OPK("v_add_f32 %0, %0, %1");
            if (OP == 11) OPK("v_max3_f32 %0, %0, %1, %2");
            if (OP == 12) OPK("v_cmp_gt_f32 vcc, %0, %1");
            if (OP == 13) OPK("v_lshl_add_u32 %0, %0, 1, %1");
            if (OP == 14) OPK("v_and_b32 %0, %0, %1");
            if (OP == 15) OPK("v_max_u32 %0, %0, %1");
            if (OP == 16) OPK("v_sub_u32 %0, %0, %1");
            if (OP == 17) OPK("v_add3_u32 %0, %0, %1, %2");
            if (OP == 18) OPK("v_max_i16 %0, %0, %1");
            if (OP == 19) OPK("v_bfe_i32 %0, %0, %1, 8");
            if (OP == 20) OPK("v_mov_b32 %0, %1");
            if (OP == 21) OPK("v_alignbit_b32 %0, %0, %1, 31");
            if (OP == 22) OPK("v_fma_f32 %0, %0, %1, %2");
            if (OP == 23) OPK("v_pk_max_f16 %0, %0, %1");
            if (OP == 24) OPK("v_subb_co_u32 %0, vcc, %0, %1, vcc");
            if (OP == 25) OPK("v_med3_i32 %0, %0, %1, %2");
            if (OP == 26) OPK("v_pk_sub_i16 %0, %0, %1");
            if (OP == 27) OPK("v_perm_b32 %0, %0, %1, %2");
            if (OP == 28) OPK("v_max_f16 %0, %0, %1");
            if (OP == 29) OPK("v_cmp_gt_i32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc");
            if (OP == 30) OPK("v_xor_b32 %0, %0, %1");
            if (OP == 31) OPK("v_min_i32 %0, %0, %1");
            if (OP == 32) OPK("v_dot4_i32_i8 %0, %1, %2, %0");
            if (OP == 33) OPK("v_dot4c_i32_i8 %0, %1, %2");
            if (OP == 34) OPK("v_sub_co_u32 %0, vcc, %0, %1");
            if (OP == 35) OPK("v_add_u16 %0, %0, %1");
            if (OP == 36) OPK("v_sub_u16 %0, %0, %1");
            if (OP == 37) OPK("v_max_u16 %0, %0, %1");
            if (OP == 38) OPK("v_min_i16 %0, %0, %1");
            if (OP == 39) OPK("v_lshlrev_b32 %0, 1, %0");
            if (OP == 40) OPK("v_lshrrev_b32 %0, 1, %0");
            if (OP == 41) OPK("v_ashrrev_i32 %0, 1, %0");
            if (OP == 42) OPK("v_or_b32 %0, %0, %1");
            if (OP == 43) OPK("v_lshl_or_b32 %0, %0, 1, %1");
            if (OP == 44) OPK("v_and_or_b32 %0, %0, %1, %2");
            if (OP == 45) OPK("v_bfi_b32 %0, %0, %1, %2");
            if (OP == 46) OPK("v_sub_u32_e64 %0, %0, %1 clamp");
            if (OP == 47) OPK("v_mad_u32_u24 %0, %0, %1, %2");
            if (OP == 48) OPK("v_mul_u32_u24 %0, %0, %1");
            if (OP == 49) OPK("v_mad_u16 %0, %0, %1, %2");
            if (OP == 50) OPK("v_alignbyte_b32 %0, %0, %1, 1");
            if (OP == 51) OPK("v_add_u16_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD");
            if (OP == 52) OPK("v_max_i16_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD");
            if (OP == 53) OPK("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf");
            if (OP == 54) OPK("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf");
            if (OP == 55) OPK("v_bcnt_u32_b32 %0, %1, %0");
            if (OP == 56) OPK("v_sad_u8 %0, %0, %1, %2");
            if (OP == 57) OPK("v_pk_mad_u16 %0, %0, %1, %2");
            if (OP == 58) OPK("v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]");
            if (OP == 59) OPK("v_cvt_f32_i32 %0, %0");
            if (OP == 60) OPK("v_min3_i32 %0, %0, %1, %2");
            if (OP == 61) OPK("v_add_co_u32 %0, vcc, %0, %1");
            if (OP == 62) OPK("v_sub_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2");
            if (OP == 63) OPK("v_mad_i32_i24 %0, %0, %1, %2");
            // round 2, float-pattern decision bits of the affine cell (kernels_affine.hip)
            if (OP == 64) OPK("v_sub_f32_e64 %0, %0, %1 clamp");
            if (OP == 65) OPK("v_fmac_f32 %0, 2.0, %1");
            if (OP == 66) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1] clamp" : "+v"(p2[k]) : "v"(q2[k]));
            if (OP == 67) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[k]) : "v"(q2[k]));
            if (OP == 68) OPK2("v_add_u32_sdwa %0, sext(%2), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\tv_fmac_f32 %1, 2.0, %2");
            if (OP == 69) OPK2("v_bfe_i32 %1, %2, 8, 8\n\tv_add_u32 %0, %0, %1\n\tv_fmac_f32 %1, 2.0, %2");
            if (OP == 70) OPK2("v_add_u32_sdwa %0, sext(%2), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\tv_max_i32 %1, %1, %2");
            if (OP == 71) OPK2("v_sub_f32_e64 %1, %0, %2 clamp\n\tv_fmac_f32 %0, 2.0, %1");
            // round 3
            if (OP == 72) OPK("v_cvt_f32_ubyte1 %0, %1");
            if (OP == 73) OPK("v_sub_f16_e64 %0, %0, %1 clamp");
            if (OP == 74) OPK("v_fma_f16 %0, %0, %1, %2");
            if (OP == 75) OPK("v_madmk_f16 %0, %0, 0x4000, %1");
            if (OP == 76) OPK("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]");
            if (OP == 77) OPK("v_fma_f16 %0, %0, %1, %2 op_sel:[0,1,0,0]");
            if (OP == 78) OPK("v_cvt_f32_ubyte1 %0, %1\n\tv_add_f32 %0, %0, %2");
            if (OP == 79) OPK("v_max_f16 %0, %0, %1\n\tv_sub_f16_e64 %0, %0, %2 clamp");
            if (OP == 80) OPK("v_dot4c_i32_i8 %0, %1, %2\n");
            if (OP == 81) OPK("v_pk_fma_f16 %0, %0, %1, %2");
            if (OP == 82) OPK("v_add_f16 %0, %0, %1");
            if (OP == 83) OPK("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36");
            if (OP == 84) OPK("v_not_b32 %0, %1");
            if (OP == 85) OPK("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x30");
            if (OP == 86) OPK("v_ffbl_b32 %0, %1");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int r = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) r += a[k] + c[k] + (int)p2[k].x + (int)p2[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static const char *NAMES[] = {"v_add_u32", "v_max_i32", "v_max3_i32", "v_cmp_gt_i32", "v_addc_co_u32", "v_cndmask_b32",
    "v_add_u32_sdwa(sext byte)", "v_pk_add_i16", "v_pk_max_i16", "v_max_f32", "v_add_f32", "v_max3_f32", "v_cmp_gt_f32",
    "v_lshl_add_u32", "v_and_b32", "v_max_u32", "v_sub_u32", "v_add3_u32", "v_max_i16", "v_bfe_i32", "v_mov_b32",
    "v_alignbit_b32", "v_fma_f32", "v_pk_max_f16", "v_subb_co_u32", "v_med3_i32", "v_pk_sub_i16", "v_perm_b32", "v_max_f16",
    "v_cmp_gt_i32+v_addc_co_u32 (pair)", "v_xor_b32", "v_min_i32",
    "v_dot4_i32_i8", "v_dot4c_i32_i8", "v_sub_co_u32", "v_add_u16", "v_sub_u16", "v_max_u16", "v_min_i16", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_or_b32", "v_lshl_or_b32", "v_and_or_b32", "v_bfi_b32", "v_sub_u32 clamp", "v_mad_u32_u24", "v_mul_u32_u24", "v_mad_u16", "v_alignbyte_b32", "v_add_u16_sdwa(sext byte)", "v_max_i16_sdwa(word1)", "v_add_u32 dpp wave_shr", "v_mov_b32 dpp wave_shr", "v_bcnt_u32_b32", "v_sad_u8", "v_pk_mad_u16", "v_pk_lshrrev_b16", "v_cvt_f32_i32", "v_min3_i32", "v_add_co_u32", "v_sub_u32 sdwa sext byte", "v_mad_i32_i24",
    "v_sub_f32 clamp (e64)", "v_fmac_f32 x2.0", "v_pk_add_f32 neg clamp", "v_pk_fma_f32", "v_add_u32_sdwa + v_fmac_f32 (pair)", "v_bfe_i32 + v_add_u32 + v_fmac_f32 (triple)", "v_add_u32_sdwa + v_max_i32 (pair)", "v_sub_f32 clamp + v_fmac_f32 (pair)",
    "v_cvt_f32_ubyte1", "v_sub_f16 clamp (e64)", "v_fma_f16", "v_madmk_f16", "v_fma_mix_f32", "v_fma_f16 op_sel hi", "v_cvt_f32_ubyte1 + v_add_f32 (pair)", "v_max_f16 + v_sub_f16 clamp (pair)", "v_dot4c_i32_i8 again", "v_pk_fma_f16", "v_add_f16", "v_bitop3_b32 (3 sources)", "v_not_b32", "v_bitop3_b32 (2 sources)", "v_ffbl_b32"};

template <int OP>
void run(int *d_out, unsigned long long *d_clk, int cus, int waves_per_simd)
{
    const int blocks = cus * waves_per_simd;  // blocks of 4 waves: one wave per SIMD each
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(op_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1, d_clk);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(op_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, rep, d_clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned long long clk[2];
    CHECK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
    const double ghz = (double)clk[0] / (double)clk[1] * 0.1;  // s_memrealtime ticks at 100 MHz
    const int ninstr = (OP == 29 || OP == 68 || OP == 70 || OP == 71 || OP == 78 || OP == 79) ? 2 : OP == 69 ? 3 : 1;
    const double winstr_per_simd = (double)waves_per_simd * ITERS * CH * ninstr;  // wave-instructions per SIMD
    const double cyc = best * 1e-3 * ghz * 1e9 / winstr_per_simd;
    const double lane_ops = (double)blocks * 256 * ITERS * CH * ninstr / (best * 1e-3);
    printf("%-36s waves/SIMD=%d  %.3f ms  clk=%.2f GHz  %.2f cyc/wave-instr/SIMD  %.1f T lane-op/s\n",
           NAMES[OP], waves_per_simd, best, ghz, cyc, lane_ops / 1e12);
}

template <int OP>
void run_all(int *d_out, unsigned long long *d_clk, int cus)
{
    run<OP>(d_out, d_clk, cus, 8);
    if constexpr (OP + 1 < 87) run_all<OP + 1>(d_out, d_clk, cus);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int *d_out;
    unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 8 * 256 * sizeof(int)));
    CHECK(hipMalloc(&d_clk, 16));
    if (getenv("VALU_OPS_FROM")) {  // only the newer cases
        run_all<83>(d_out, d_clk, prop.multiProcessorCount);
        return 0;
    }
    run_all<0>(d_out, d_clk, prop.multiProcessorCount);
    // occupancy sweep for two representative ops
    for (int w : {1, 2, 4}) { run<0>(d_out, d_clk, prop.multiProcessorCount, w); run<1>(d_out, d_clk, prop.multiProcessorCount, w); }
    return 0;
}
