// kernels_wide.hip -- the WIDE path of the batched aligner: inputs the fast kernels' packed encodings cannot hold.
//
// The fast kernels (kernels.hip, kernels_affine.hip) keep substitution scores as int8 profile bytes, residues in at
// most 8 bits, affine scores as float patterns within +-2^22.  The reference takes any `int` score and penalty
// (align.go:76, align_pair.go:3) and any float64 penalty (align_affine.go:12); what does not fit those encodings runs
// here instead: 32-bit scores, the substitution score of a cell looked up in the caller's int32 table, one decision
// BYTE per cell.  Same rules as the fast kernels, restated plainly:
//   linear  (align.go:97-122, 186-202; align_pair.go:109-195): candidates [0, local only] -> up -> left -> diagonal,
//           a later one replaces only on strict '>';
//   affine  (align_affine.go:44-103 in integers, borders and end cells of the four modes on the middle level as the
//           oracle's bo_align_affine states them): E / F extension wins ties, H = first maximum of
//           [0, local only] -> F (close the gap in s2) -> diagonal -> E (close the gap in s1).
// One lane = one pair (the planner's bundles and packed sequence words are the fast path's), the row above in a
// lane-interleaved buffer in HBM, cells in row-major order -- so the first maximum in row-major order (align.go:216-221)
// is simply the first strict improvement.  Slow by design (a handful of loads per cell); correct for every input the
// 32-bit score range holds, which the planner checks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "device_helpers.h"

#define BA_WIDE_NEG (-(1 << 30))  // minus infinity of the E / F levels (align_affine.go:36-37, 45-46, 63-64)

// decision byte of a cell.  bits 0-1: where the middle level (or, linear gaps, the cell) came from
enum { BA_WD_STOP = 0, BA_WD_UP = 1, BA_WD_DIAG = 2, BA_WD_LEFT = 3 };  // affine: UP = closes a gap in s2 (F), LEFT = in s1 (E)
#define BA_WD_OPEN_E 4u  // affine: E of this cell opened from the middle level (else: extended)
#define BA_WD_OPEN_F 8u  // affine: F likewise

// Layout of a bundle's workspace: decision bytes [i-1][j-1][lane] (dir words, n*m*16 dwords); rows in hb:
// H[0..m][lane], then (affine) F[0..m][lane].
template <int MODE, bool AFF>
__global__ void __launch_bounds__(256) ba_dp_wide_kernel(BaDpParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * (blockDim.x >> 6) + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const int n = (int)B.n, m = (int)B.m;
    const int R = P.wide_rows;  // rows per strip of the packed s1 words (front padding)
    const int front = ((n + R - 1) / R) * R - n;
    const int go = AFF ? P.gap_open : P.gap, ge = AFF ? P.gap_ext : P.gap;
    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    int32_t *hrow = P.hb + B.hb_off + lane;                      // hrow[j * 64]
    int32_t *frow = hrow + (size_t)(m + 1) * 64;                 // affine
    uint8_t *dir = reinterpret_cast<uint8_t *>(P.dir + B.dir_off) + lane;  // dir[((i-1) * m + (j-1)) * 64]
    const int32_t *tbl = P.table;
    const int nb = P.nb;
    auto border_h = [&](int i, int j) -> int {  // the middle level (linear: the matrix) on row 0 / column 0
        if (i == 0 && j == 0) return 0;
        if (i == 0) return top_free ? 0 : (AFF ? -go - (j - 1) * ge : -j * ge);
        return left_free ? 0 : (AFF ? -go - (i - 1) * ge : -i * ge);
    };
    // row 0
    for (int j = 0; j <= m; ++j) {
        hrow[(size_t)j * 64] = border_h(0, j);
        if (AFF) frow[(size_t)j * 64] = BA_WIDE_NEG;
    }
    int best_sc = 0, best_i = 0, best_j = 0;
    if (MODE == BA_GLOBAL) {
        best_i = n;
        best_j = m;
        best_sc = n == 0 ? border_h(0, m) : 0;  // (n >= 1: set after the last row)
    }
    if ((MODE == BA_FITTING || MODE == BA_OVERLAP) && n == 0) {
        // the bottom row is row 0: leftmost strict maximum from column 0 (align_pair.go:169-182)
        best_sc = border_h(0, 0);
        for (int j = 1; j <= m; ++j)
            if (border_h(0, j) > best_sc) {
                best_sc = border_h(0, j);
                best_j = j;
            }
    }
    for (int i = 1; i <= n; ++i) {
        const int t = i - 1 + front;
        const uint32_t a = P.code1[(s1w[(size_t)(t >> 2) * 64] >> ((t & 3) * 8)) & 0xffu];
        const int32_t *trow = tbl + (size_t)a * nb;
        int diag = hrow[0];            // H(i-1, 0)
        int left = border_h(i, 0);     // H(i, 0)
        int e = BA_WIDE_NEG;           // E(i, 0) = -inf (align_affine.go:45-46)
        hrow[0] = left;
        if ((MODE == BA_FITTING || MODE == BA_OVERLAP) && i == n) {
            best_sc = left;
            best_i = n;
            best_j = 0;
        }
        uint32_t yw = 0;
        for (int j = 1; j <= m; ++j) {
            if (((j - 1) & 3) == 0) yw = s2w[(size_t)((j - 1) >> 2) * 64];
            const uint32_t b = P.code2[(yw >> (((j - 1) & 3) * 8)) & 0xffu];
            const int w = trow[b];
            const int up = hrow[(size_t)j * 64];  // H(i-1, j)
            int h;
            uint32_t d;
            if (AFF) {
                // E(i,j) = max(E(i,j-1) - ext, H(i,j-1) - open), extension wins ties (align_affine.go:51-57); F likewise (67-75)
                const int e_ext = e == BA_WIDE_NEG ? BA_WIDE_NEG : e - ge, e_open = left - go;
                const bool oe = e_open > e_ext;
                e = oe ? e_open : e_ext;
                const int fo = frow[(size_t)j * 64];
                const int f_ext = fo == BA_WIDE_NEG ? BA_WIDE_NEG : fo - ge, f_open = up - go;
                const bool of = f_open > f_ext;
                const int f = of ? f_open : f_ext;
                frow[(size_t)j * 64] = f;
                d = BA_WD_UP;
                h = f;  // close the gap in s2 (align_affine.go:82-83)
                if (MODE == BA_LOCAL) {
                    d = BA_WD_STOP;
                    h = 0;
                    if (f > h) {
                        h = f;
                        d = BA_WD_UP;
                    }
                }
                if (diag + w > h) {  // (87-94)
                    h = diag + w;
                    d = BA_WD_DIAG;
                }
                if (e > h) {  // close the gap in s1 (97-100)
                    h = e;
                    d = BA_WD_LEFT;
                }
                d |= (oe ? BA_WD_OPEN_E : 0u) | (of ? BA_WD_OPEN_F : 0u);
            } else {
                // align.go:105-116 / 186-202, align_pair.go:117-121: [0] -> up -> left -> diagonal, strict '>'
                h = up - ge;
                d = BA_WD_UP;
                if (MODE == BA_LOCAL) {
                    d = BA_WD_STOP;
                    h = 0;
                    if (up - ge > h) {
                        h = up - ge;
                        d = BA_WD_UP;
                    }
                }
                if (left - ge > h) {
                    h = left - ge;
                    d = BA_WD_LEFT;
                }
                if (diag + w > h) {
                    h = diag + w;
                    d = BA_WD_DIAG;
                }
            }
            dir[((size_t)(i - 1) * m + (j - 1)) * 64] = (uint8_t)d;
            hrow[(size_t)j * 64] = h;
            diag = up;
            left = h;
            if (MODE == BA_LOCAL) {
                if (h > best_sc) {  // first strict maximum in row-major order (align.go:216-221)
                    best_sc = h;
                    best_i = i;
                    best_j = j;
                }
            } else if ((MODE == BA_FITTING || MODE == BA_OVERLAP) && i == n) {
                if (h > best_sc) {  // leftmost strict maximum of the bottom row (align_pair.go:169-182)
                    best_sc = h;
                    best_j = j;
                }
            }
        }
        if (MODE == BA_GLOBAL && i == n) best_sc = left;  // H(n, m) (m == 0: the border)
    }
    if (lane < (int)B.count) {
        BaLocalBest bb;
        bb.score = best_sc;  // true score of the end cell
        bb.i = best_i;
        bb.j = best_j;
        bb.pad = 0;
        P.best[B.first + lane] = bb;
    }
}

// The traceback of the wide path: one lane walks one pair's path over the decision bytes exactly as the reference's
// trace() does (align.go:126-143, 224-248; align_pair.go:197-216; affine: the oracle's canonical walk over (level, node))
// and writes both traces back to front into the pair's slot.
template <int MODE, bool AFF>
__global__ void __launch_bounds__(256) ba_trace_wide_kernel(BaTraceParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * (blockDim.x >> 6) + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const bool live = lane < (int)B.count;
    const uint32_t p = live ? P.perm[B.first + lane] : 0u;
    const int n = (int)B.n, m = (int)B.m;
    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    const uint8_t *dir = reinterpret_cast<const uint8_t *>(P.dir + B.dir_off) + lane;
    const uint64_t o1 = live ? P.s1_off[p] : 0, o2 = live ? P.s2_off[p] : 0;
    int i = 0, j = 0;
    if (live) {
        const BaLocalBest bb = P.best[B.first + lane];
        i = bb.i;
        j = bb.j;
        P.scores[p] = (int64_t)bb.score;
        if (P.score_only && P.t_off) {  // BIOALIGN_END_CELLS
            P.t_off[p] = (uint64_t)i;
            P.t_len[p] = (uint64_t)j;
        }
    }
    if (P.score_only || !live) return;
    const uint64_t slot = o1 + o2, cap = (uint64_t)n + (uint64_t)m;
    const uint8_t *s1 = P.s1 + o1, *s2 = P.s2 + o2;
    BaBackWriter w1, w2;
    w1.init(P.t1, slot + cap);
    w2.init(P.t2, slot + cap);
    const uint32_t gs = P.gap_symbol;
    auto cell = [&](int ii, int jj) -> uint32_t { return dir[((size_t)(ii - 1) * m + (jj - 1)) * 64]; };
    enum { LV_MM = 0, LV_G1 = 1, LV_G2 = 2 };
    int level = LV_MM;
    while (i > 0 || j > 0) {
        if (!AFF) {
            uint32_t d;
            if (i == 0) {
                if (top_free) break;
                d = BA_WD_LEFT;  // top border of global / overlap (align.go:92-95, align_pair.go:110-113)
            } else if (j == 0) {
                if (left_free) break;
                d = BA_WD_UP;
            } else {
                d = cell(i, j) & 3u;
            }
            if (d == BA_WD_STOP) break;
            w1.put(d == BA_WD_LEFT ? gs : s1[i - 1]);
            w2.put(d == BA_WD_UP ? gs : s2[j - 1]);
            i -= (d != BA_WD_LEFT);
            j -= (d != BA_WD_UP);
            continue;
        }
        if (level == LV_MM) {
            if (i == 0) {
                if (top_free) break;
                level = LV_G1;  // top border: F = -inf, no diagonal
            } else if (j == 0) {
                if (left_free) break;
                level = LV_G2;
            } else {
                const uint32_t d = cell(i, j) & 3u;
                if (d == BA_WD_STOP) break;
                if (d == BA_WD_DIAG) {
                    w1.put(s1[i - 1]);
                    w2.put(s2[j - 1]);
                    --i;
                    --j;
                    continue;
                }
                level = d == BA_WD_UP ? LV_G2 : LV_G1;  // close a gap: switch level at the same node
            }
        }
        if (level == LV_G1) {  // gap in s1: emit (gap, s2[j-1]); j >= 1 on this level
            const bool opened = i == 0 ? (j == 1) : (j == 1 ? true : (cell(i, j) & BA_WD_OPEN_E) != 0);
            w1.put(gs);
            w2.put(s2[j - 1]);
            level = opened ? LV_MM : LV_G1;
            --j;
        } else {  // gap in s2: emit (s1[i-1], gap); i >= 1 on this level
            const bool opened = j == 0 ? (i == 1) : (i == 1 ? true : (cell(i, j) & BA_WD_OPEN_F) != 0);
            w1.put(s1[i - 1]);
            w2.put(gs);
            level = opened ? LV_MM : LV_G2;
            --i;
        }
    }
    w1.finish();
    w2.finish();
    const uint64_t len = (uint64_t)w1.length();
    P.t_off[p] = slot + cap - len;
    P.t_len[p] = len;
}

template <int MODE>
static hipError_t launch_wide(const BaDpParams &P, int affine, hipStream_t st)
{
    const uint32_t nb = P.bundle_end - P.bundle_begin, blocks = (nb + 3) / 4;
    if (affine)
        hipLaunchKernelGGL((ba_dp_wide_kernel<MODE, true>), dim3(blocks), dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL((ba_dp_wide_kernel<MODE, false>), dim3(blocks), dim3(256), 0, st, P);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_dp_wide(const BaDpParams &P, int mode, int affine, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    if (!P.table || P.wide_rows <= 0) return hipErrorInvalidValue;
    switch (mode) {
    case BA_GLOBAL: return launch_wide<BA_GLOBAL>(P, affine, st);
    case BA_LOCAL: return launch_wide<BA_LOCAL>(P, affine, st);
    case BA_FITTING: return launch_wide<BA_FITTING>(P, affine, st);
    case BA_OVERLAP: return launch_wide<BA_OVERLAP>(P, affine, st);
    }
    return hipErrorInvalidValue;
}

template <int MODE>
static hipError_t launch_trace_wide(const BaTraceParams &P, int affine, hipStream_t st)
{
    const uint32_t nb = P.bundle_end - P.bundle_begin, blocks = (nb + 3) / 4;
    if (affine)
        hipLaunchKernelGGL((ba_trace_wide_kernel<MODE, true>), dim3(blocks), dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL((ba_trace_wide_kernel<MODE, false>), dim3(blocks), dim3(256), 0, st, P);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_trace_wide(const BaTraceParams &P, int mode, int affine, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_trace_wide<BA_GLOBAL>(P, affine, st);
    case BA_LOCAL: return launch_trace_wide<BA_LOCAL>(P, affine, st);
    case BA_FITTING: return launch_trace_wide<BA_FITTING>(P, affine, st);
    case BA_OVERLAP: return launch_trace_wide<BA_OVERLAP>(P, affine, st);
    }
    return hipErrorInvalidValue;
}
