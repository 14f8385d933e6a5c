// kernels.hip -- gfx950 (MI355X) kernels of the batched pairwise aligner.
//
// Inter-sequence mapping: one wavefront = one bundle of up to 64 equal-shape
// pairs, one pair per lane.  Integer max-plus recurrences on the VALU; no MFMA.
//
// Recurrences and tie priorities follow the reference (file:line relative to
// the reference tree):
//   global  : align.go:97-122      up (gap in s2) > left (gap in s1) > diagonal, strict '>'
//   local   : align.go:179-208     skip-prefix(0) first, then up > left > diagonal
//   fitting : align_pair.go:157-166, 218-246   top row free, left column -i*indel
//   overlap : align_pair.go:186-195            top row -j*indel, left column free
// End cells: align.go:123-124 / 214-221, align_pair.go:123-126, 138-152, 169-182.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

// ----------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------

// bits = (bits << 1) | (x > y), as v_cmp_gt_i32 + v_addc_co_u32: two VALU
// instructions per decision bit (hipcc's own lowering of the same C is three).
__device__ __forceinline__ void ba_cmp_pack(uint32_t &bits, int x, int y)
{
    asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(bits)
        : "v"(x), "v"(y)
        : "vcc");
}

__device__ __forceinline__ int ba_max3(int a, int b, int c) { return max(max(a, b), c); }

// signed byte k (0..3) of a dword; hipcc folds this into the SDWA operand of the add that uses it
__device__ __forceinline__ int ba_sbyte(uint32_t w, int k) { return (int)(int8_t)(w >> (8 * k)); }

// ----------------------------------------------------------------------------
// pack: caller bytes -> dense codes, lane-interleaved words
//   word k of lane l of a bundle lives at  w_off + k*64 + l  (4 codes per word)
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ba_pack_kernel(BaPackParams P)
{
    const uint32_t b_idx = P.bundle_begin + blockIdx.x;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const uint32_t R = P.rows_per_strip;
    const uint32_t strips = (B.n + R - 1) / R;
    const uint32_t nw1 = strips * R / 4;
    const uint32_t nw2 = (B.m + 3) / 4;
    // s1 is padded at the FRONT up to a whole number of strips: the last strip
    // then always ends on the matrix's last row (see ba_dp_linear_kernel)
    const uint32_t front = strips * R - B.n;
    const uint32_t padword = P.pad_code * 0x01010101u;
    for (uint32_t idx = threadIdx.x; idx < (nw1 + nw2) * 64; idx += blockDim.x) {
        const bool side2 = idx >= nw1 * 64;
        const uint32_t rel = side2 ? idx - nw1 * 64 : idx;
        const uint32_t k = rel >> 6, l = rel & 63;
        uint32_t word = side2 ? 0u : padword;
        if (l < B.count) {
            const uint32_t p = P.perm[B.first + l];
            word = 0;
            if (side2) {
                const uint8_t *src = P.s2 + P.s2_off[p];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t pos = 4 * k + q;
                    const uint32_t c = pos < B.m ? (uint32_t)P.code2[src[pos]] : 0u;
                    word |= c << (8 * q);
                }
            } else {
                const uint8_t *src = P.s1 + P.s1_off[p];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t t = 4 * k + q;
                    const uint32_t c = t >= front ? (uint32_t)P.code1[src[t - front]] : P.pad_code;
                    word |= c << (8 * q);
                }
            }
        }
        if (side2)
            P.s2w[B.s2w_off + rel] = word;
        else
            P.s1w[B.s1w_off + rel] = word;
    }
}

// ----------------------------------------------------------------------------
// linear-gap DP, one strip of R rows swept over all columns
// ----------------------------------------------------------------------------
//
// Non-local modes run in a biased score space  H'(i,j) = H(i,j) + g*(i+j):
// both gap moves then cost nothing and the diagonal costs w + 2g (folded into
// the int8 profile), so a cell is  max(up', left', diag' + w'')  with no
// subtraction.  Every candidate of one cell carries the same bias, so all
// comparisons -- and therefore the directions -- are those of the reference.
// Local mode needs the true zero level and runs unbiased.
//
// Direction bits per cell: c1 = (left > up), c2 = (diag > max(up, left)).
//   c2 -> diagonal; else c1 -> left; else up      (priority up > left > diagonal)
// A local cell whose score is 0 chose skip-prefix (align.go:186-187): the
// traceback knows the running score and needs no third bit.
template <int MODE, int R>
__device__ __forceinline__ void ba_linear_strip(
    int (&h)[R], int (&bk)[R], uint32_t &b1, uint32_t &b2,
    int32_t *__restrict__ hb, const uint32_t *__restrict__ s2w, uint32_t *__restrict__ dirs,
    const uint8_t *prof_lane, int m, int g, int key_shift)
{
    constexpr int HALVES = R / 16;
    int topdiag = hb[0];
    hb[0] = h[R - 1];
    const int km = (1 << key_shift) - 1;
    uint32_t sw = 0;
    for (int j = 1; j <= m; ++j) {
        const int jj = j - 1;
        if ((jj & 3) == 0) sw = s2w[(size_t)(jj >> 2) * 64];
        const uint32_t b = (sw >> ((jj & 3) * 8)) & 0xffu;
        const uint4 *pp = reinterpret_cast<const uint4 *>(prof_lane + (size_t)b * (HALVES * 64 * 16));
        uint32_t pw[R / 4];
#pragma unroll
        for (int hf = 0; hf < HALVES; ++hf) {
            const uint4 v = pp[hf * 64];
            pw[hf * 4 + 0] = v.x;
            pw[hf * 4 + 1] = v.y;
            pw[hf * 4 + 2] = v.z;
            pw[hf * 4 + 3] = v.w;
        }
        const int top = hb[(size_t)j * 64];
        int up = top, diag = topdiag;
        topdiag = top;
        const int cj = km - j;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int left = h[r];
            const int d = diag + ba_sbyte(pw[r >> 2], r & 3);
            ba_cmp_pack(b1, left, up);
            int mx = max(up, left);
            if (MODE == BA_LOCAL) mx -= g;
            ba_cmp_pack(b2, d, mx);
            int hn;
            if (MODE == BA_LOCAL) {
                hn = ba_max3(mx, d, 0);
                bk[r] = max(bk[r], (hn << key_shift) + cj);
            } else {
                hn = max(mx, d);
            }
            diag = left;
            h[r] = hn;
            up = hn;
        }
        hb[(size_t)j * 64] = h[R - 1];
        if (R == 32) {
            reinterpret_cast<uint2 *>(dirs)[(size_t)jj * 64] = make_uint2(b1, b2);
        } else {
            dirs[(size_t)jj * 64] = (b1 & 0xffffu) | (b2 << 16);
        }
    }
}

template <int MODE, int R>
__global__ void __launch_bounds__(256) ba_dp_linear_kernel(BaDpParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int HALVES = R / 16;
    constexpr int DIRW = (R == 32) ? 2 : 1;  // direction dwords per (strip, column, lane)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;

    // the transposed int8 score table, shared by the block
    const int wt_words = P.nb4 * P.na_pad;
    for (int i = threadIdx.x; i < wt_words; i += blockDim.x) lds[i] = P.wt[i];
    __syncthreads();

    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const int n = (int)B.n, m = (int)B.m;
    const int g = P.gap;
    const int strips = (n + R - 1) / R;

    const int wt_aligned = (wt_words + 3) & ~3;
    uint8_t *prof = reinterpret_cast<uint8_t *>(lds + wt_aligned) + (size_t)wave * P.nb * (HALVES * 64 * 16);
    uint8_t *prof_lane = prof + lane * 16;

    int32_t *hb = P.hb + B.hb_off + lane;
    uint32_t *dir = P.dir + B.dir_off + (size_t)lane * DIRW;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;

    // row 0 (align.go:92-95, 175-177; align_pair.go:110-113, 158, 187), biased
    for (int j = 0; j <= m; ++j) hb[(size_t)j * 64] = (MODE == BA_FITTING) ? g * j : 0;

    int best_sc = 0, best_i = 0, best_j = 0;
    uint32_t b1 = 0, b2 = 0;
    const int km = (1 << P.key_shift) - 1;

    // Strips are aligned to the BOTTOM of the matrix: strip s covers matrix rows
    // i0+1 .. i0+R with i0 = n - (strips-s)*R, so the first strip may start with
    // rows i <= 0 that do not exist.  Those rows carry the padding code (score
    // -128 against everything) and column-0 value 0; in the biased space (and,
    // for local mode, under the clamp at 0) they reproduce row 0 exactly, so row
    // 1 sees the right border and the last strip always ends on row n.
    for (int s = 0; s < strips; ++s) {
        const int i0 = n - (strips - s) * R;
        // codes of the strip's rows
        uint32_t aw[R / 4];
#pragma unroll
        for (int q = 0; q < R / 4; ++q) aw[q] = s1w[(size_t)(s * (R / 4) + q) * 64];
        // strip profile: prof[b][half][lane][16 rows] = score(row r, column symbol b), int8
        for (int c = 0; c < P.nb4; ++c) {
            const uint32_t *wrow = lds + c * P.na_pad;
#pragma unroll
            for (int hf = 0; hf < HALVES; ++hf) {
                uint32_t o[4][4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const uint32_t wq = aw[hf * 4 + q4];
                    const uint32_t t0 = wrow[wq & 0xffu];
                    const uint32_t t1 = wrow[(wq >> 8) & 0xffu];
                    const uint32_t t2 = wrow[(wq >> 16) & 0xffu];
                    const uint32_t t3 = wrow[wq >> 24];
                    // 4x4 byte transpose: o[bb] = {t0.bb, t1.bb, t2.bb, t3.bb}
                    const uint32_t lo01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u);  // t0.0 t1.0 t0.1 t1.1
                    const uint32_t hi01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);  // t0.2 t1.2 t0.3 t1.3
                    const uint32_t lo23 = __builtin_amdgcn_perm(t3, t2, 0x05010400u);
                    const uint32_t hi23 = __builtin_amdgcn_perm(t3, t2, 0x07030602u);
                    o[0][q4] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
                    o[1][q4] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
                    o[2][q4] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
                    o[3][q4] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
                }
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int b = 4 * c + bb;
                    if (b < P.nb)
                        *reinterpret_cast<uint4 *>(prof_lane + (size_t)(b * HALVES + hf) * (64 * 16)) =
                            make_uint4(o[bb][0], o[bb][1], o[bb][2], o[bb][3]);
                }
            }
        }
        // column 0 (align.go:99-100, 181; align_pair.go:113, 131, 159, 188), biased
        int h[R], bk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            h[r] = (MODE == BA_OVERLAP) ? g * max(i0 + r + 1, 0) : 0;
            bk[r] = km;
        }
        uint32_t *dirs = dir + (size_t)s * m * (64 * DIRW);
        ba_linear_strip<MODE, R>(h, bk, b1, b2, hb, s2w, dirs, prof_lane, m, g, P.key_shift);
        if (MODE == BA_LOCAL) {
            // first strict maximum in row-major order (align.go:216-221): rows in
            // ascending order, strict '>' on the score; within a row the key keeps
            // the smallest column.  Rows i <= 0 score 0 everywhere and never win.
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int sc = bk[r] >> P.key_shift;
                if (sc > best_sc) {
                    best_sc = sc;
                    best_i = i0 + r + 1;
                    best_j = km - (bk[r] & km);
                }
            }
        }
    }
    if (MODE == BA_LOCAL && lane < (int)B.count) {
        BaLocalBest bb;
        bb.score = best_sc;
        bb.i = best_i;
        bb.j = best_j;
        bb.pad = 0;
        P.best[B.first + lane] = bb;
    }
}

// ----------------------------------------------------------------------------
// traceback, linear gap: one lane walks one pair's path back from its end cell
// and writes both traces back to front into the pair's output slot.
// (align.go:126-143, 224-248; align_pair.go:197-216)
// ----------------------------------------------------------------------------
template <int MODE, int R>
__global__ void __launch_bounds__(256) ba_trace_linear_kernel(BaTraceParams P)
{
    constexpr int DIRW = (R == 32) ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * (blockDim.x >> 6) + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    if (lane >= (int)B.count) return;
    const uint32_t p = P.perm[B.first + lane];
    const int n = (int)B.n, m = (int)B.m;
    const int64_t g = P.gap;
    const int32_t *hb = P.hb + B.hb_off + lane;
    const uint32_t *dir = P.dir + B.dir_off + (size_t)lane * DIRW;
    const int front = ((n + R - 1) / R) * R - n;
    const uint64_t o1 = P.s1_off[p], o2 = P.s2_off[p];
    const uint8_t *s1 = P.s1 + o1;
    const uint8_t *s2 = P.s2 + o2;

    // end cell and score
    int i, j;
    int64_t score;
    if (MODE == BA_GLOBAL) {
        i = n;
        j = m;
        score = (int64_t)hb[(size_t)m * 64] - g * (n + m);
    } else if (MODE == BA_LOCAL) {
        const BaLocalBest bb = P.best[B.first + lane];
        i = bb.i;
        j = bb.j;
        score = bb.score;
    } else {
        // leftmost strict maximum of the bottom row, scan from column 0 (align_pair.go:169-182)
        i = n;
        j = 0;
        score = (int64_t)hb[0] - g * n;
        for (int jj = 1; jj <= m; ++jj) {
            const int64_t v = (int64_t)hb[(size_t)jj * 64] - g * (n + jj);
            if (v > score) {
                score = v;
                j = jj;
            }
        }
    }
    P.scores[p] = score;
    if (P.score_only) return;

    const uint64_t slot = o1 + o2;
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    uint8_t *t1 = P.t1 + slot;
    uint8_t *t2 = P.t2 + slot;
    uint64_t k = cap;
    const uint8_t gs = P.gap_symbol;
    const bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    const bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    int64_t cur = score;  // running score of the current cell (local mode's stop test)
    while (i > 0 || j > 0) {
        int mv;  // 0 up, 1 left, 2 diagonal
        if (i == 0) {
            if (top_free) break;
            mv = 1;
        } else if (j == 0) {
            if (left_free) break;
            mv = 0;
        } else {
            if (MODE == BA_LOCAL && cur == 0) break;
            const int t = i - 1 + front;  // row in the front-padded strip grid
            const int s = t / R, r = t % R;
            const size_t w = ((size_t)s * m + (j - 1)) * (64 * DIRW);
            uint32_t c1, c2;
            if (R == 32) {
                const uint2 d = *reinterpret_cast<const uint2 *>(dir + w);
                c1 = (d.x >> (31 - r)) & 1u;
                c2 = (d.y >> (31 - r)) & 1u;
            } else {
                const uint32_t d = dir[w];
                c1 = (d >> (15 - r)) & 1u;
                c2 = (d >> (31 - r)) & 1u;
            }
            mv = c2 ? 2 : (c1 ? 1 : 0);
        }
        --k;
        if (mv == 0) {
            t1[k] = s1[i - 1];
            t2[k] = gs;
            --i;
            cur += g;
        } else if (mv == 1) {
            t1[k] = gs;
            t2[k] = s2[j - 1];
            --j;
            cur += g;
        } else {
            const uint8_t x = s1[i - 1], y = s2[j - 1];
            t1[k] = x;
            t2[k] = y;
            if (MODE == BA_LOCAL) cur -= P.table[(int)P.code1[x] * P.nb + (int)P.code2[y]];
            --i;
            --j;
        }
    }
    P.t_off[p] = slot + k;
    P.t_len[p] = cap - k;
}

// ----------------------------------------------------------------------------
// host-side launchers (called from api.cpp)
// ----------------------------------------------------------------------------
extern "C" size_t ba_dp_linear_lds_bytes(int nb, int nb4, int na_pad, int R, int waves)
{
    const int wt_words = nb4 * na_pad;
    const int wt_aligned = (wt_words + 3) & ~3;
    return (size_t)wt_aligned * 4 + (size_t)waves * nb * (R / 16) * 64 * 16;
}

extern "C" hipError_t ba_launch_pack(const BaPackParams &P, hipStream_t st)
{
    const uint32_t nb = P.bundle_end - P.bundle_begin;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(ba_pack_kernel, dim3(nb), dim3(256), 0, st, P);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_dp_linear_mode(const BaDpParams &P, int R, int waves, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + waves - 1) / waves;
    const size_t lds = ba_dp_linear_lds_bytes(P.nb, P.nb4, P.na_pad, R, waves);
    if (R == 32)
        hipLaunchKernelGGL((ba_dp_linear_kernel<MODE, 32>), dim3(blocks), dim3(waves * 64), lds, st, P);
    else
        hipLaunchKernelGGL((ba_dp_linear_kernel<MODE, 16>), dim3(blocks), dim3(waves * 64), lds, st, P);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_dp_linear(const BaDpParams &P, int mode, int R, int waves, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_dp_linear_mode<BA_GLOBAL>(P, R, waves, st);
    case BA_LOCAL: return launch_dp_linear_mode<BA_LOCAL>(P, R, waves, st);
    case BA_FITTING: return launch_dp_linear_mode<BA_FITTING>(P, R, waves, st);
    case BA_OVERLAP: return launch_dp_linear_mode<BA_OVERLAP>(P, R, waves, st);
    }
    return hipErrorInvalidValue;
}

template <int MODE>
static hipError_t launch_trace_linear_mode(const BaTraceParams &P, int R, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + 3) / 4;
    if (R == 32)
        hipLaunchKernelGGL((ba_trace_linear_kernel<MODE, 32>), dim3(blocks), dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL((ba_trace_linear_kernel<MODE, 16>), dim3(blocks), dim3(256), 0, st, P);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_trace_linear(const BaTraceParams &P, int mode, int R, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_trace_linear_mode<BA_GLOBAL>(P, R, st);
    case BA_LOCAL: return launch_trace_linear_mode<BA_LOCAL>(P, R, st);
    case BA_FITTING: return launch_trace_linear_mode<BA_FITTING>(P, R, st);
    case BA_OVERLAP: return launch_trace_linear_mode<BA_OVERLAP>(P, R, st);
    }
    return hipErrorInvalidValue;
}
