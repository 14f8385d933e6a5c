// kernels.hip -- gfx950 (MI355X) kernels of the batched pairwise aligner, linear gap.
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
//
// Working score space.  Non-local modes run biased, Hs(i,j) = H(i,j) + g*(i+j):
// both gap moves then cost nothing and the diagonal costs w + 2g (folded into
// the int8 profile), so a cell is max3(up, left, diag + w'') -- two VALU
// instructions.  All candidates of one cell carry the same bias, so every
// comparison, and therefore every traceback decision, is the reference's.
// Local mode needs the true zero level and runs unbiased.
//
// Traceback matrix.  The forward pass does not decide directions at all (the
// compare + pack pair per decision bit costs more than the recurrence itself on
// gfx950, where v_cmp/v_addc/v_max issue at half the v_add rate -- see
// profiles/r01_valu_ops_microbench.txt).  It stores the K low bits of every
// cell's working score instead ("residues", one v_alignbit_b32 per cell).
// Adjacent cells differ by a bounded amount: in the biased space the vertical
// and horizontal steps satisfy 0 <= Hs(i,j) - Hs(i-1,j) <= max(w) + 2g (local,
// unbiased: -g .. max(w)+g), so with 2^K > max(w) + 2g a neighbour's exact score
// follows from the current cell's exact score and the neighbour's residue.  The
// traceback kernel re-evaluates the reference's candidate order on those exact
// scores, only along the path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"

// ----------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------
__device__ __forceinline__ int ba_max3(int a, int b, int c) { return max(max(a, b), c); }

// signed byte k (0..3) of a dword; hipcc folds this into the SDWA operand of the add that uses it
__device__ __forceinline__ int ba_sbyte(uint32_t w, int k) { return (int)(int8_t)(w >> (8 * k)); }

// acc = (x << (32-K)) | (acc >> K): shifts the K low bits of x in at the top (one v_alignbit_b32)
template <int K>
__device__ __forceinline__ uint32_t ba_push_residue(uint32_t acc, int x)
{
    return __builtin_amdgcn_alignbit((uint32_t)x, acc, K);
}

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
// linear-gap DP: one column of one strip (R rows in VGPRs)
// ----------------------------------------------------------------------------
template <int MODE, int R, int K>
__device__ __forceinline__ void ba_linear_column(int (&h)[R], int (&rmax)[R], uint32_t (&acc)[R * K / 32],
                                                 const uint32_t (&pw)[R / 4], int top, int &topdiag, int g)
{
    constexpr int F = 32 / K;  // residues per word
    int up = top, diag = topdiag;
    topdiag = top;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int left = h[r];
        const int d = diag + ba_sbyte(pw[r >> 2], r & 3);
        int hn;
        if (MODE == BA_LOCAL) {
            hn = ba_max3(max(up, left) - g, d, 0);
            rmax[r] = max(rmax[r], hn);
        } else {
            hn = ba_max3(up, left, d);
        }
        acc[r / F] = ba_push_residue<K>(acc[r / F], hn);
        diag = left;
        h[r] = hn;
        up = hn;
    }
}

// Layout of the residue words of one bundle (dwords):
//   [strip][column group cg = (j-1)/4][lane][column c = (j-1)%4][word w]      WPC = R*K/32 words per column
// so a lane's 4 columns x WPC words are one contiguous 16*WPC-byte chunk and a
// wavefront stores a column group as 64 such chunks back to back.
template <int MODE, int R, int K>
__global__ void __launch_bounds__(256) ba_dp_linear_kernel(BaDpParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int HALVES = R / 16;
    constexpr int WPC = R * K / 32;
    constexpr int CHW = 4 * WPC;  // dwords per lane per column group
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
    const int cgs = (m + 3) / 4;

    const int wt_aligned = (wt_words + 3) & ~3;
    uint8_t *prof = reinterpret_cast<uint8_t *>(lds + wt_aligned) + (size_t)wave * P.nb * (HALVES * 64 * 16);
    uint8_t *prof_lane = prof + lane * 16;

    int32_t *hb = P.hb + B.hb_off + lane;
    uint32_t *dir = P.dir + B.dir_off + (size_t)lane * CHW;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;

    // row 0 (align.go:92-95, 175-177; align_pair.go:110-113, 158, 187) in the working space
    for (int j = 0; j <= m; ++j) hb[(size_t)j * 64] = (MODE == BA_FITTING) ? g * j : 0;

    int best_sc = 0, best_i = 0;

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
        // column 0 (align.go:99-100, 181; align_pair.go:113, 131, 159, 188) in the working space
        int h[R], rmax[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            h[r] = (MODE == BA_OVERLAP) ? g * max(i0 + r + 1, 0) : 0;
            rmax[r] = 0;
        }
        int topdiag = hb[0];
        hb[0] = h[R - 1];
        uint32_t *dirs = dir + (size_t)s * cgs * (64 * CHW);
        uint32_t acc[4][WPC];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int w = 0; w < WPC; ++w) acc[c][w] = 0;
        // one group of 4 columns (one packed s2 word): FULL groups run without any
        // per-column test so that the strip stays in the same registers all along
        auto column_group = [&](int cg, auto full) {
            constexpr bool FULL = decltype(full)::value;
            const uint32_t sw = s2w[(size_t)cg * 64];
            const int jbase = cg * 4 + 1;
            int tops[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) tops[c] = (FULL || jbase + c <= m) ? hb[(size_t)(jbase + c) * 64] : 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (FULL || jbase + c <= m) {
                    const uint32_t b = (sw >> (c * 8)) & 0xffu;
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
                    ba_linear_column<MODE, R, K>(h, rmax, acc[c], pw, tops[c], topdiag, g);
                    hb[(size_t)(jbase + c) * 64] = h[R - 1];
                }
            }
            uint4 *dst = reinterpret_cast<uint4 *>(dirs + (size_t)cg * (64 * CHW));
#pragma unroll
            for (int q = 0; q < CHW / 4; ++q) {
                const int e = q * 4;
                dst[q] = make_uint4(acc[e / WPC][e % WPC], acc[(e + 1) / WPC][(e + 1) % WPC],
                                    acc[(e + 2) / WPC][(e + 2) % WPC], acc[(e + 3) / WPC][(e + 3) % WPC]);
            }
        };
        const int full_groups = m / 4;
        for (int cg = 0; cg < full_groups; ++cg) column_group(cg, std::true_type{});
        if (full_groups < cgs) column_group(full_groups, std::false_type{});
        if (MODE == BA_LOCAL) {
            // first row (ascending) that holds the strict maximum (align.go:216-221);
            // the traceback kernel finds the first column of that row.  Rows i <= 0
            // score 0 everywhere and never win.
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (rmax[r] > best_sc) {
                    best_sc = rmax[r];
                    best_i = i0 + r + 1;
                }
            }
        }
    }
    if (MODE == BA_LOCAL && lane < (int)B.count) {
        BaLocalBest bb;
        bb.score = best_sc;
        bb.i = best_i;
        bb.j = 0;
        bb.pad = 0;
        P.best[B.first + lane] = bb;
    }
}

// ----------------------------------------------------------------------------
// traceback, linear gap: one lane walks one pair's path back from its end cell
// and writes both traces back to front into the pair's output slot
// (align.go:126-143, 224-248; align_pair.go:197-216).
// ----------------------------------------------------------------------------
//
// Residue chunks are staged through a per-lane 2x2 window in LDS (strip parity
// x column-group parity), because the four cells one step needs -- (i,j),
// (i-1,j), (i,j-1), (i-1,j-1) -- can straddle a strip and a column-group
// boundary at once.  LDS words are lane-interleaved ([word][lane]), so any
// per-lane index is bank-conflict free.
template <int R, int K>
struct BaResidueWindow {
    static constexpr int WPC = R * K / 32;
    static constexpr int CHW = 4 * WPC;
    static constexpr int F = 32 / K;
    static constexpr int WORDS = 4 * CHW + 4;  // 4 chunk slots + 4 tags, per lane
    uint32_t *lds;        // this wave's window, already offset by the lane
    const uint32_t *dir;  // bundle's residue words, already offset by lane*CHW
    int cgs, front;

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int t = 0; t < 4; ++t) lds[(4 * CHW + t) * 64] = 0xffffffffu;
    }
    // K low bits of the working score of cell (i, j), i >= 1, j >= 1
    __device__ __forceinline__ uint32_t get(int i, int j)
    {
        const int t = i - 1 + front;
        const int s = t / R, r = t % R;
        const int cg = (j - 1) >> 2, c = (j - 1) & 3;
        const int slot = ((s & 1) << 1) | (cg & 1);
        const uint32_t tag = (uint32_t)(s * cgs + cg);
        if (lds[(4 * CHW + slot) * 64] != tag) {
            const uint4 *src = reinterpret_cast<const uint4 *>(dir + ((size_t)s * cgs + cg) * (64 * CHW));
#pragma unroll
            for (int q = 0; q < CHW / 4; ++q) {
                const uint4 v = src[q];
                lds[(slot * CHW + q * 4 + 0) * 64] = v.x;
                lds[(slot * CHW + q * 4 + 1) * 64] = v.y;
                lds[(slot * CHW + q * 4 + 2) * 64] = v.z;
                lds[(slot * CHW + q * 4 + 3) * 64] = v.w;
            }
            lds[(4 * CHW + slot) * 64] = tag;
        }
        const uint32_t word = lds[(slot * CHW + c * WPC + r / F) * 64];
        return (word >> (K * (r % F))) & ((1u << K) - 1u);
    }
};

// byte `idx` of a byte array through a one-dword register cache (aligned loads)
struct BaByteCache {
    const uint8_t *base;
    uint64_t tag;
    uint32_t word;
    __device__ __forceinline__ void init(const uint8_t *p)
    {
        base = p;
        tag = ~0ull;
        word = 0;
    }
    __device__ __forceinline__ uint32_t get(int idx)
    {
        const uint64_t a = (uint64_t)(base + idx);
        const uint64_t al = a & ~3ull;
        if (al != tag) {
            word = *reinterpret_cast<const uint32_t *>(al);
            tag = al;
        }
        return (word >> ((a & 3) * 8)) & 0xffu;
    }
};

// Writes bytes back to front.  Bytes are collected per aligned dword and go out
// as one dword store, except in the two dwords that can be shared with a
// neighbouring slot (the one holding the slot's end and the one holding the
// trace's first byte): those are written byte by byte.
struct BaBackWriter {
    uint64_t end;    // absolute address one past the slot
    uint64_t a;      // absolute address of the last byte written (== end when empty)
    uint32_t word;   // bytes collected for the aligned dword that contains a
    __device__ __forceinline__ void init(uint8_t *buf, uint64_t end_index)
    {
        end = (uint64_t)(buf + end_index);
        a = end;
        word = 0;
    }
    __device__ __forceinline__ void put(uint32_t byte)
    {
        --a;
        word |= byte << ((uint32_t)(a & 3) * 8);
        if ((a & 3) == 0) {
            if (a + 4 <= end) {
                *reinterpret_cast<uint32_t *>(a) = word;
            } else {  // the dword that holds the slot's end
                for (uint64_t q = a; q < end; ++q) *reinterpret_cast<uint8_t *>(q) = (uint8_t)(word >> ((q & 3) * 8));
            }
            word = 0;
        }
    }
    __device__ __forceinline__ void flush()
    {
        if (a & 3) {
            const uint64_t al = a & ~3ull;
            for (uint64_t q = a; q < al + 4 && q < end; ++q)
                *reinterpret_cast<uint8_t *>(q) = (uint8_t)(word >> ((q & 3) * 8));
        }
    }
};

template <int MODE, int R, int K>
__global__ void __launch_bounds__(256) ba_trace_linear_kernel(BaTraceParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    using Win = BaResidueWindow<R, K>;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    // score table (as the caller gave it) and the byte -> code maps, shared by the block
    const int tbl_words = (P.na * P.nb + 3) & ~3;
    int32_t *tbl = reinterpret_cast<int32_t *>(lds);
    uint8_t *code1 = reinterpret_cast<uint8_t *>(lds + tbl_words);
    uint8_t *code2 = code1 + 256;
    for (int i = threadIdx.x; i < P.na * P.nb; i += blockDim.x) tbl[i] = P.table[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        code1[i] = P.code1[i];
        code2[i] = P.code2[i];
    }
    __syncthreads();
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    if (lane >= (int)B.count) return;
    const uint32_t p = P.perm[B.first + lane];
    const int n = (int)B.n, m = (int)B.m;
    const int g = P.gap;
    const int32_t *hb = P.hb + B.hb_off + lane;
    const uint64_t o1 = P.s1_off[p], o2 = P.s2_off[p];

    Win win;
    win.lds = lds + tbl_words + 128 + (size_t)wave * (Win::WORDS * 64) + lane;
    win.dir = P.dir + B.dir_off + (size_t)lane * Win::CHW;
    win.cgs = (m + 3) / 4;
    win.front = ((n + R - 1) / R) * R - n;
    win.init();

    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    constexpr uint32_t MASK = (1u << K) - 1u;
    const int off = (MODE == BA_LOCAL) ? g : 0;  // lower bound of an adjacent step is -off

    // end cell, its working-space score `cur`, and the reported score
    int i, j, cur;
    int64_t score;
    if (MODE == BA_GLOBAL) {
        i = n;
        j = m;
        cur = hb[(size_t)m * 64];
        score = (int64_t)cur - (int64_t)g * (n + m);
    } else if (MODE == BA_LOCAL) {
        const BaLocalBest bb = P.best[B.first + lane];
        i = bb.i;
        j = 0;
        cur = 0;
        score = bb.score;
        if (bb.score > 0) {
            // first column of row i that reaches the maximum: rebuild the row's exact
            // scores left to right from their residues (H(i,0) = 0)
            int v = 0;
            for (int jj = 1; jj <= m; ++jj) {
                const uint32_t rho = win.get(i, jj);
                v = v + (int)(((uint32_t)(rho - (uint32_t)v + (uint32_t)off)) & MASK) - off;
                if (v == bb.score) {
                    j = jj;
                    break;
                }
            }
            cur = bb.score;
        } else {
            i = 0;
        }
    } else {
        // leftmost strict maximum of the bottom row, scan from column 0 (align_pair.go:169-182)
        i = n;
        j = 0;
        cur = hb[0];
        score = (int64_t)cur - (int64_t)g * n;
        for (int jj = 1; jj <= m; ++jj) {
            const int hv = hb[(size_t)jj * 64];
            const int64_t v = (int64_t)hv - (int64_t)g * (n + jj);
            if (v > score) {
                score = v;
                j = jj;
                cur = hv;
            }
        }
    }
    P.scores[p] = score;
    if (P.score_only) return;

    const uint64_t slot = o1 + o2;
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    BaByteCache c1, c2;
    c1.init(P.s1 + o1);
    c2.init(P.s2 + o2);
    BaBackWriter w1, w2;
    w1.init(P.t1, slot + cap);
    w2.init(P.t2, slot + cap);
    const uint32_t gs = P.gap_symbol;
    const int wbias = (MODE == BA_LOCAL) ? 0 : 2 * g;

    while (i > 0 || j > 0) {
        int mv;  // 0 up, 1 left, 2 diagonal
        int nxt = 0;
        uint32_t x = 0, y = 0;
        if (i == 0) {
            if (top_free) break;
            mv = 1;
            nxt = 0;  // global / overlap: Hs(0, j-1) = 0
        } else if (j == 0) {
            if (left_free) break;
            mv = 0;
            nxt = 0;  // global / fitting: Hs(i-1, 0) = 0
        } else {
            if (MODE == BA_LOCAL && cur == 0) break;
            // exact neighbours from residues (borders by formula)
            int up, left, diag;
            if (i == 1)
                up = (MODE == BA_FITTING) ? g * j : 0;
            else
                up = cur - ((int)(((uint32_t)cur - win.get(i - 1, j) + (uint32_t)off) & MASK) - off);
            if (j == 1)
                left = (MODE == BA_OVERLAP) ? g * i : 0;
            else
                left = cur - ((int)(((uint32_t)cur - win.get(i, j - 1) + (uint32_t)off) & MASK) - off);
            if (i == 1)
                diag = (MODE == BA_FITTING) ? g * (j - 1) : 0;
            else if (j == 1)
                diag = (MODE == BA_OVERLAP) ? g * (i - 1) : 0;
            else  // left neighbour of (i-1, j)
                diag = up - ((int)(((uint32_t)up - win.get(i - 1, j - 1) + (uint32_t)off) & MASK) - off);
            x = c1.get(i - 1);
            y = c2.get(j - 1);
            const int w = tbl[(int)code1[x] * P.nb + (int)code2[y]] + wbias;
            const int d = diag + w;
            const int mx = (MODE == BA_LOCAL) ? max(up, left) - g : max(up, left);
            if (d > mx) {
                mv = 2;
                nxt = diag;
            } else if (left > up) {
                mv = 1;
                nxt = left;
            } else {
                mv = 0;
                nxt = up;
            }
        }
        if (mv == 0) {
            if (j == 0) x = c1.get(i - 1);
            w1.put(x);
            w2.put(gs);
            --i;
        } else if (mv == 1) {
            if (i == 0) y = c2.get(j - 1);
            w1.put(gs);
            w2.put(y);
            --j;
        } else {
            w1.put(x);
            w2.put(y);
            --i;
            --j;
        }
        cur = nxt;
    }
    w1.flush();
    w2.flush();
    const uint64_t len = w1.end - w1.a;
    P.t_off[p] = slot + cap - len;
    P.t_len[p] = len;
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

template <int MODE, int R, int K>
static hipError_t launch_dp(const BaDpParams &P, int waves, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + waves - 1) / waves;
    const size_t lds = ba_dp_linear_lds_bytes(P.nb, P.nb4, P.na_pad, R, waves);
    hipLaunchKernelGGL((ba_dp_linear_kernel<MODE, R, K>), dim3(blocks), dim3(waves * 64), lds, st, P);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_dp_mode(const BaDpParams &P, int R, int K, int waves, hipStream_t st)
{
    if (R == 32 && K == 2) return launch_dp<MODE, 32, 2>(P, waves, st);
    if (R == 32 && K == 4) return launch_dp<MODE, 32, 4>(P, waves, st);
    if (R == 16 && K == 2) return launch_dp<MODE, 16, 2>(P, waves, st);
    if (R == 16 && K == 4) return launch_dp<MODE, 16, 4>(P, waves, st);
    if (R == 16 && K == 8) return launch_dp<MODE, 16, 8>(P, waves, st);
    return hipErrorInvalidValue;
}

extern "C" hipError_t ba_launch_dp_linear(const BaDpParams &P, int mode, int R, int K, int waves, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_dp_mode<BA_GLOBAL>(P, R, K, waves, st);
    case BA_LOCAL: return launch_dp_mode<BA_LOCAL>(P, R, K, waves, st);
    case BA_FITTING: return launch_dp_mode<BA_FITTING>(P, R, K, waves, st);
    case BA_OVERLAP: return launch_dp_mode<BA_OVERLAP>(P, R, K, waves, st);
    }
    return hipErrorInvalidValue;
}

extern "C" size_t ba_trace_linear_lds_bytes(int na, int nb, int R, int K, int waves)
{
    const int tbl = (na * nb + 3) & ~3;
    const int words = 4 * (4 * (R * K / 32)) + 4;
    return ((size_t)tbl + 128 + (size_t)waves * words * 64) * 4;
}

template <int MODE, int R, int K>
static hipError_t launch_trace(const BaTraceParams &P, hipStream_t st)
{
    const int waves = 4;
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + waves - 1) / waves;
    const size_t lds = ba_trace_linear_lds_bytes(P.na, P.nb, R, K, waves);
    hipLaunchKernelGGL((ba_trace_linear_kernel<MODE, R, K>), dim3(blocks), dim3(waves * 64), lds, st, P);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_trace_mode(const BaTraceParams &P, int R, int K, hipStream_t st)
{
    if (R == 32 && K == 2) return launch_trace<MODE, 32, 2>(P, st);
    if (R == 32 && K == 4) return launch_trace<MODE, 32, 4>(P, st);
    if (R == 16 && K == 2) return launch_trace<MODE, 16, 2>(P, st);
    if (R == 16 && K == 4) return launch_trace<MODE, 16, 4>(P, st);
    if (R == 16 && K == 8) return launch_trace<MODE, 16, 8>(P, st);
    return hipErrorInvalidValue;
}

extern "C" hipError_t ba_launch_trace_linear(const BaTraceParams &P, int mode, int R, int K, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_trace_mode<BA_GLOBAL>(P, R, K, st);
    case BA_LOCAL: return launch_trace_mode<BA_LOCAL>(P, R, K, st);
    case BA_FITTING: return launch_trace_mode<BA_FITTING>(P, R, K, st);
    case BA_OVERLAP: return launch_trace_mode<BA_OVERLAP>(P, R, K, st);
    }
    return hipErrorInvalidValue;
}
