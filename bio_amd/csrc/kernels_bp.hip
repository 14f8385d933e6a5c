// kernels_bp.hip -- gfx950 (MI355X): BIT-PARALLEL alignment (global, local, fitting, overlap) for unit scoring.
//
// Scope: AlignGlobal / AlignLocal / AlignPair (align.go:76-249, align_pair.go:3-246) when every substitution score is
// +1 or -1 and the gap penalty is 1 -- BASELINE.json's DNA configuration -- with at most four column symbols and s1 of
// at most 256 symbols.  Everything else stays on the general kernels (kernels.hip).  Same boundary, same outputs.
//
// One lane = one pair, as in kernels.hip; but a lane holds a whole COLUMN of the matrix as bit vectors over the rows
// (bit t of word t/32 = row t of the front-padded rows) and a column costs about 20 logic instructions per 32 rows
// instead of 3.4 (mostly half-rate) instructions per cell.
//
// Recurrence.  With dV(i,j) = S(i,j) - S(i-1,j) and dH(i,j) = S(i,j) - S(i,j-1), match +1, mismatch -1, gap -1:
// both differences lie in {-1, 0, 1, 2}.  Shifted by the gap, x = dV(i,j-1) + 1, g = dH(i-1,j) + 1 and
// w' = w + 2 (3 on a match, 1 on a mismatch) are in 0..3 and the cell is
//     Z = max(w', x, g),   dV(i,j) + 1 = Z - g,   dH(i,j) + 1 = Z - x
// (align.go:105-116 relative to S(i-1,j-1)).  The horizontal difference runs down the column as
//     h_i = max(c_i, h_{i-1} - x_i),   c_i = max(w'_i - x_i, 0),
// i.e. in thermometer planes H_t = [h >= t]:
//     H_3[i] = C_3[i] | (X0[i] & H_3[i-1])
//     H_2[i] = C_2[i] | (X1[i] & H_3[i-1]) | (X0[i] & H_2[i-1])
//     H_1[i] = (M & ~V3) | X0 | (X1 & H_2[i-1]) | (X2 & H_3[i-1])          (no run: x = 0 already gives h >= 1)
// and a run "P[i] = G[i] | (R[i] & P[i-1])" is the carry chain of one addition (a = G | R, b = G: the carry INTO bit
// i is P[i-1]; sum ^ a ^ b recovers all of them, and the carry out of a word is the carry into the next one --
// v_addc_co_u32).  The new vertical planes follow from u = max(w', x) and g without any chain.
//
// Traceback matrix: two bits per cell, as before -- but they are the reference's DECISIONS, not residues: plane 0 =
// [dV' >= 1] (0: the cell came from above, align.go:105-108 first candidate), plane 1 = [dH' >= 1] (0, and plane 0
// set: from the left; both set: diagonal).  Layout per bundle (dwords), the general kernel's with (word, plane) in
// the place of its row blocks:   [word w][plane][column group cg = (j-1)/4][lane][column (j-1)%4].
//
// Checked against a plain DP in tests/test_bp_recurrence.py (CPU, the same formulas on Python integers) and against
// the oracle on the GPU (tests/test_gpu_bitparallel.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "device_helpers.h"
#include "bp_column.h"

__device__ __forceinline__ void ba_lds_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#define BA_BP_MAX_WORDS 8   // rows <= 256
#define BA_BP_SCORE_BITS 8  // local mode: bit-sliced scores, so min(len(s1), len(s2)) <= 255

// ----------------------------------------------------------------------------
// DP
// ----------------------------------------------------------------------------
// match vectors of the lane's s1 against every column symbol: peq[(b * WW + w) * 64]
template <int WW>
__device__ __forceinline__ void ba_bp_build_peq(const uint32_t *s1w, const uint8_t *mm, int front, uint32_t *peq)
{
#pragma unroll
    for (int w = 0; w < WW; ++w) {
        uint32_t pl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t word = s1w[(size_t)(w * 8 + q) * 64];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t mv = mm[(word >> (8 * k)) & 0xffu];
#pragma unroll
                for (int b = 0; b < 4; ++b) pl[b] = __builtin_amdgcn_alignbit(mv >> b, pl[b], 1);
            }
        }
        // rows above the matrix match nothing
        const uint32_t keep = (w == 0) ? (0xffffffffu << front) : 0xffffffffu;
#pragma unroll
        // (relaxed workgroup-scope atomic stores stay single ds_write_b32: hipcc pairs plain stores into ds_write2st64_b32
        // and overwrites a data register in the next instruction -- the pattern tools/scan_ds_write2.py looks for)
        for (int b = 0; b < 4; ++b) ba_lds_store(peq + (b * WW + w) * 64, pl[b] & keep);
    }
}

// GLOBAL mode.  The DP stores no traceback matrix: every CKPT columns it leaves the three planes of the column state
// (a CHECKPOINT: 3 bits per row), and the lane's match vectors once; the walk recomputes the columns between two
// checkpoints in registers with the same column function and decides its moves from them (ba_bp_walk_bundle).  The
// kernel was bound by its 2 bits per cell of stores (5.6 GB per launch at 1M x 150 x 150); a column costs a fifth of
// the time those bytes took.  Layout per bundle in the direction-word buffer (dwords, lane-interleaved):
//     [4 symbols][WW words][64 lanes] match vectors, then [block][3 planes][WW words][64 lanes] checkpoints,
// block b = the state BEFORE column b * CKPT + 1.
// MODE: global, fitting (align_pair.go:157-166: top row free, left column -i) or overlap (align_pair.go:186-195: top
// row -j, left column free).  The borders are the column-0 state (x = 0 where the left column costs a gap per row,
// x = 1 where it is free) and the top border's difference handed to row 1 (ba_bp_column's top1); fitting / overlap end
// in the leftmost strict maximum of the bottom row (align_pair.go:169-182), followed along it by its differences.
template <int WW, int CKPT, int MODE>
__device__ __forceinline__ void ba_bp_bundle(const BaDpParams &P, const BaBundle &B, uint32_t *peq, const uint8_t *mm,
                                             const uint8_t *code2, int lane)
{
    static_assert(CKPT == 4 || CKPT == 8, "checkpoints every 4 or 8 columns");
    static_assert(MODE == BA_GLOBAL || MODE == BA_FITTING || MODE == BA_OVERLAP, "local mode: ba_bp_local_bundle");
    const int n = (int)B.n, m = (int)B.m;
    const int front = WW * 32 - n;  // 0..31
    const int cgs = (m + 3) / 4;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    uint32_t *ck = P.dir + B.dir_off + lane;
    uint32_t *cs = ck + 4 * WW * 64;

    ba_bp_build_peq<WW>(s1w, mm, front, peq);
    if (m > 0 && !P.no_dir) {  // (no columns: the buffer has no room and the walk no use for them)
#pragma unroll
        for (int k = 0; k < 4 * WW; ++k) ck[k * 64] = peq[k * 64];
    }

    // column 0: dV = gap in every row (x = 0); the rows above the matrix hold x = 1 for ever and hand h = 0 -- the top
    // border's dH = gap (align.go:92-95) -- down to row 1
    uint32_t V1[WW], V2[WW], V3[WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) {
        V1[w] = MODE == BA_OVERLAP ? 0xffffffffu : 0u;  // overlap: H(i, 0) = 0, dV = 0
        V2[w] = V3[w] = 0u;
    }
    if (MODE != BA_OVERLAP) V1[0] = (1u << front) - 1u;
    const uint32_t top1 = MODE == BA_FITTING ? (1u << front) : 0u;  // fitting: H(0, j) = 0, dH = 0
    // bottom row: S(n, 0), then one difference per column
    int cur = MODE == BA_FITTING ? -n : 0, end_v = cur, end_j = 0;

    auto column_group = [&](int cg, uint32_t sw, auto full) {
        constexpr bool FULL = decltype(full)::value;
        if ((cg * 4) % CKPT == 0 && !P.no_dir) {
            uint32_t *c0 = cs + (size_t)((cg * 4) / CKPT) * (3 * WW * 64);
#pragma unroll
            for (int w = 0; w < WW; ++w) {
                c0[(0 * WW + w) * 64] = V1[w];
                c0[(1 * WW + w) * 64] = V2[w];
                c0[(2 * WW + w) * 64] = V3[w];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (FULL || cg * 4 + c < m) {
                const uint32_t b = (uint32_t)code2[(sw >> (8 * c)) & 0xffu] & 3u;
                const uint32_t *pq = peq + b * (WW * 64);
                uint32_t M[WW], oP[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) M[w] = pq[w * 64];
                const uint32_t hb = ba_bp_column<WW>(V1, V2, V3, M, oP, top1);
                if (MODE != BA_GLOBAL) {
                    cur += (int)hb - 1;
                    const bool up = cur > end_v;
                    end_v = up ? cur : end_v;
                    end_j = up ? cg * 4 + c + 1 : end_j;
                }
            }
        }
    };
    if (cgs > 0) {
        const int full_groups = m / 4;
        uint32_t nxt = s2w[0];
        asm volatile("" : "+v"(nxt));  // (waited for here, not in every iteration: see kernels.hip's sweep)
        for (int cg = 0; cg < full_groups; ++cg) {
            const uint32_t cur = nxt;
            nxt = s2w[(size_t)min(cg + 1, cgs - 1) * 64];
            column_group(cg, cur, std::true_type{});
        }
        if (full_groups < cgs) column_group(full_groups, nxt, std::false_type{});
    }
    // S(n, m) = S(0, m) + sum of dV down the last column = -m + sum (v - 1); the padded rows hold v = 1
    int sum = 0;
#pragma unroll
    for (int w = 0; w < WW; ++w) sum += __popc(V1[w]) + __popc(V2[w]) + __popc(V3[w]);
    if (lane < (int)B.count) {
        BaLocalBest bb;
        bb.score = MODE == BA_GLOBAL ? sum - front - n - m : end_v;
        bb.i = n;
        bb.j = MODE == BA_GLOBAL ? m : end_j;
        bb.pad = 0;
        P.best[B.first + lane] = bb;
    }
}

// LOCAL mode (align.go:152-249): bp_column.h's ba_bp_local_column; the end cell is tracked column by column.
// NODIR: score-only plans -- nothing is stored for a walk.  (A compile-time switch: as a run-time test inside the column
// group it cost the five-word instance 28 more spilled registers, 1.57 -> 2.23 ms per 1M pairs.)
template <int WW, bool NODIR>
__device__ __forceinline__ void ba_bp_local_bundle(const BaDpParams &P, const BaBundle &B, uint32_t *peq, const uint8_t *mm,
                                                   const uint8_t *code2, int lane)
{
    constexpr int NB = BA_BP_SCORE_BITS;
    const int n = (int)B.n, m = (int)B.m;
    const int front = WW * 32 - n;  // 0..31
    const int cgs = (m + 3) / 4;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    uint32_t *dir = P.dir + B.dir_off + (size_t)lane * 4;
    ba_bp_build_peq<WW>(s1w, mm, front, peq);

    // column 0: H = 0 in every row, so dV = 0 (x = 1)
    uint32_t V1[WW], V2[WW], V3[WW], NZ[WW], H[NB][WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) {
        V1[w] = 0xffffffffu;
        V2[w] = V3[w] = NZ[w] = 0u;
#pragma unroll
        for (int k = 0; k < NB; ++k) H[k][w] = 0u;
    }
    // the maximum so far and its first cell in row-major order (align.go:214-221), row counted over the padded rows
    int best = 0, best_t = 0, best_j = 0;

    uint32_t accA[4][WW], accB[4][WW];
    auto column_group = [&](int cg, uint32_t sw, auto full) {
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (FULL || cg * 4 + c < m) {
                const uint32_t b = (uint32_t)code2[(sw >> (8 * c)) & 0xffu] & 3u;
                const uint32_t *pq = peq + b * (WW * 64);
                uint32_t M[WW], S[NB], T[NB], gt[WW], eq[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) M[w] = pq[w * 64];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    S[k] = 0u - (((uint32_t)best >> k) & 1u);
                    T[k] = 0u - (((uint32_t)(best + 1) >> k) & 1u);
                }
                ba_bp_local_column<WW, NB>(V1, V2, V3, H, NZ, M, S, T, accA[c], accB[c], gt, eq);
                // Within a column the maximum grows by at most 1.  A row above it (gt): the new maximum, first cell =
                // the lowest such row of this column.  Otherwise a row AT it (eq) in a higher row than the cell held
                // so far comes first in row-major order.  (best = 0: nothing is held; every empty cell "equals" it.)
                uint32_t og = gt[0];
#pragma unroll
                for (int w = 1; w < WW; ++w) og |= gt[w];
                const bool grew = og != 0u;
                int r = 0x7fffffff;
#pragma unroll
                for (int w = WW - 1; w >= 0; --w) {
                    const uint32_t cand = grew ? gt[w] : eq[w];
                    if (cand) r = 32 * w + (int)__builtin_ctz(cand);
                }
                const bool take = grew || (best > 0 && r < best_t);
                best += grew ? 1 : 0;
                best_t = take ? r : best_t;
                best_j = take ? cg * 4 + c + 1 : best_j;
            } else {
#pragma unroll
                for (int w = 0; w < WW; ++w) accA[c][w] = accB[c][w] = 0u;
            }
        }
        if (NODIR) return;  // (the end cell is tracked above, nothing is walked)
#pragma unroll
        for (int w = 0; w < WW; ++w) {
            *reinterpret_cast<uint4 *>(dir + ((size_t)(w * 2 + 0) * cgs + cg) * 256) =
                make_uint4(accA[0][w], accA[1][w], accA[2][w], accA[3][w]);
            *reinterpret_cast<uint4 *>(dir + ((size_t)(w * 2 + 1) * cgs + cg) * 256) =
                make_uint4(accB[0][w], accB[1][w], accB[2][w], accB[3][w]);
        }
    };
    if (cgs > 0) {
        const int full_groups = m / 4;
        uint32_t nxt = s2w[0];
        asm volatile("" : "+v"(nxt));
        for (int cg = 0; cg < full_groups; ++cg) {
            const uint32_t cur = nxt;
            nxt = s2w[(size_t)min(cg + 1, cgs - 1) * 64];
            column_group(cg, cur, std::true_type{});
        }
        if (full_groups < cgs) column_group(full_groups, nxt, std::false_type{});
    }
    if (lane < (int)B.count) {
        BaLocalBest bb;
        bb.score = best;
        bb.i = best > 0 ? best_t - front + 1 : 0;
        bb.j = best > 0 ? best_j : 0;
        bb.pad = 0;
        P.best[B.first + lane] = bb;
    }
}

// WMAX: the longest rows (in words) this instance takes -- its registers are those of its largest bundle shape
// (local, five words: 168 VGPRs = three wavefronts per SIMD, with one 8-byte spill per column group)
template <int MODE, int CKPT, int WMAX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE != BA_LOCAL ? 4 : WMAX <= 3 ? 4 : WMAX <= 5 ? 3 : 2)))
ba_dp_bp_kernel(BaDpParams P, int max_words)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;
    // byte of s1 -> which column codes it matches (bit b: score(code1[byte], b) > 0); byte of s2 -> column code
    uint8_t *mm = reinterpret_cast<uint8_t *>(lds);
    uint8_t *code2 = mm + 256;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        const int a = P.code1[i];
        uint32_t v = 0;
        if (a < P.pad_code)
            for (int b = 0; b < P.nb && b < 4; ++b) v |= (P.table[a * P.nb + b] > 0 ? 1u : 0u) << b;
        mm[i] = (uint8_t)v;
        code2[i] = P.code2[i];
    }
    __syncthreads();
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    uint32_t *peq = lds + 128 + (size_t)wave * (4 * max_words * 64) + lane;
    const int words = ((int)B.n + 31) / 32;
    if (words == 0) {  // no rows: global S(0, m) = -m, the walk goes left along the top border; other modes: the empty alignment
        if (lane < (int)B.count) {
            BaLocalBest bb;
            bb.score = MODE == BA_GLOBAL ? -(int)B.m : 0;  // (the bottom row is the top row: its leftmost maximum is column 0)
            bb.i = 0;
            bb.j = MODE == BA_GLOBAL ? (int)B.m : 0;
            bb.pad = 0;
            P.best[B.first + lane] = bb;
        }
        return;
    }
#define BA_BP_CASE(W)                                                                              \
    case W:                                                                                        \
        if constexpr (W > WMAX) break;                                                             \
        else if constexpr (MODE == BA_LOCAL) {                                                     \
            if (P.no_dir) ba_bp_local_bundle<W, true>(P, B, peq, mm, code2, lane);                 \
            else ba_bp_local_bundle<W, false>(P, B, peq, mm, code2, lane);                         \
        }                                                                                          \
        else if constexpr (W <= 5 || CKPT == 4) ba_bp_bundle<W, CKPT, MODE>(P, B, peq, mm, code2, lane); \
        break;
    switch (words) {
        BA_BP_CASE(1)
        BA_BP_CASE(2)
        BA_BP_CASE(3)
        BA_BP_CASE(4)
        BA_BP_CASE(5)
        BA_BP_CASE(6)
        BA_BP_CASE(7)
        BA_BP_CASE(8)
    default: break;  // (the planner never sends longer rows here)
    }
#undef BA_BP_CASE
}

// ----------------------------------------------------------------------------
// traceback: one lane walks one pair's path back from (n, m) and records its moves (BaMoveRec, expanded by
// kernels.hip's ba_expand kernels).  Row-synchronous like kernels.hip's walk: the wavefront goes up word by word
// (32 rows); per word a lane stages a band of BG column groups of both planes around its column in LDS
// ([plane][column][lane]: any per-lane index is conflict free) and every step reads the two bits of its own cell.
// ----------------------------------------------------------------------------
#ifndef BA_BP_BG
#define BA_BP_BG 10  // column groups per band: 40 columns, so that a diagonal through a word's 32 rows needs one load (4 / 6 / 8 / 10 groups: 0.73 / 0.69 / 0.78 / 0.61 ms)
#endif

#ifndef BA_BP_MOVES
#define BA_BP_MOVES 4  // moves between two votes on "is any lane still in this word" (1 / 4 / 8: 0.60 / 0.57 / 0.57 ms)
#endif
template <bool LOCAL>
__global__ void __launch_bounds__(256) ba_trace_bp_kernel(BaTraceParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int BG = BA_BP_BG, BW = BG * 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const bool live = lane < (int)B.count;
    const uint32_t p = live ? P.perm[B.first + lane] : 0u;
    const int n = (int)B.n, m = (int)B.m;
    const uint32_t *dir = P.dir + B.dir_off + (size_t)lane * 4;
    const uint64_t o1 = live ? P.s1_off[p] : 0, o2 = live ? P.s2_off[p] : 0;
    const int cgs = (m + 3) / 4;
    const int front = ((n + 31) / 32) * 32 - n;
    uint32_t *band = lds + (size_t)wave * (2 * BW * 64) + lane;  // band[(plane * BW + column) * 64]

    int i = 0, j = 0;
    if (live) {
        const BaLocalBest bb = P.best[B.first + lane];
        i = bb.i;  // global: (n, m), align.go:123-124; local: the first maximum in row-major order, (0, 0) if there is none
        j = bb.j;
        P.scores[p] = (int64_t)bb.score;
        if (P.score_only && P.t_off) {  // BIOALIGN_END_CELLS
            P.t_off[p] = (uint64_t)i;
            P.t_len[p] = (uint64_t)j;
        }
    }
    if (P.score_only) return;

    const uint64_t slot = o1 + o2;
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    const BaRecs recs = ba_recs(P.rec, B, lane);
    // the record being filled: `sh` / 2 moves so far, 2 bits each; the cell its first move left follows from the cell
    // the walk is in now and the moves themselves, so nothing is kept per move
    uint32_t opw = 0, sh = 0, nrec = 0;
    auto flush = [&]() {
        const uint32_t count = sh >> 1;
        const uint32_t valid = count >= 16 ? 0xffffffffu : ((1u << sh) - 1u);
        const uint32_t lo = opw & 0x55555555u, hi = (opw >> 1) & 0x55555555u;
        const uint32_t n_left = __popc(lo & ~hi & valid), n_up = __popc(~lo & ~hi & 0x55555555u & valid);
        BaMoveRec rr;
        rr.ops = opw;
        rr.i = (uint32_t)i + count - n_left;  // every move but a left one came down a row
        rr.j = (uint32_t)j + count - n_up;
        rr.count = count;
        recs[nrec++] = rr;
        opw = 0;
        sh = 0;
    };

    int col0 = 0;  // first column (0-based) of the staged band
    // stage the band of word `w` that ends at the column group of column index `col` (0-based)
    auto load_band = [&](int w, int col) {
        const int cg_hi = col >> 2;
        uint4 qv[BG], qp[BG];
#pragma unroll
        for (int k = 0; k < BG; ++k) {
            const int cg = max(cg_hi - (BG - 1) + k, 0);  // (groups left of the matrix are clamped and never looked at)
            qv[k] = *reinterpret_cast<const uint4 *>(dir + ((size_t)(w * 2 + 0) * cgs + cg) * 256);
            qp[k] = *reinterpret_cast<const uint4 *>(dir + ((size_t)(w * 2 + 1) * cgs + cg) * 256);
        }
#pragma unroll
        for (int k = 0; k < BG; ++k) {
            ba_lds_store(band + (k * 4 + 0) * 64, qv[k].x);
            ba_lds_store(band + (k * 4 + 1) * 64, qv[k].y);
            ba_lds_store(band + (k * 4 + 2) * 64, qv[k].z);
            ba_lds_store(band + (k * 4 + 3) * 64, qv[k].w);
            ba_lds_store(band + (BW + k * 4 + 0) * 64, qp[k].x);
            ba_lds_store(band + (BW + k * 4 + 1) * 64, qp[k].y);
            ba_lds_store(band + (BW + k * 4 + 2) * 64, qp[k].z);
            ba_lds_store(band + (BW + k * 4 + 3) * 64, qp[k].w);
        }
        return (cg_hi - (BG - 1)) * 4;
    };

    // Bundles are equal-shape: every lane starts in row n, so the wavefront goes up WORD by word together (one band
    // load per word and lane, all lanes' loads in flight at once); inside a word every lane takes one move per round
    // until all have left it upwards.  The round is branch-free but for the record that fills up every 16 moves:
    // `row` is the lane's row inside the word, `idx` its column inside the band.
    bool done = false;  // local: the walk has reached a cell with H = 0 (or a border, where H = 0 too)
    for (int w = (n + front - 1) >> 5; w >= 0 && n > 0; --w) {
        const int lo_i = max(32 * w + 1 - front, 1);  // first row of the word; a lane is in it while i >= lo_i
        uint32_t row = (uint32_t)(i - 1 + front) & 31u;  // the lane's row inside the word
        if (i >= lo_i && j >= 1) col0 = load_band(w, j - 1);
        int idx = j - 1 - col0;
        while (true) {
            const bool act = i >= lo_i && !done;
            if (!__any(act)) break;
            if (__any(act && j >= 1 && idx < 0)) {  // (a lane has walked out of its band on the left)
                if (act && j >= 1 && idx < 0) {
                    col0 = load_band(w, j - 1);
                    idx = j - 1 - col0;
                }
            }
            // BA_BP_MOVES moves per vote: a lane that leaves the word or its band inside the group sits the rest of it out
#pragma unroll
            for (int u = 0; u < BA_BP_MOVES; ++u) {
                if (i >= lo_i && !done && !(j >= 1 && idx < 0)) {
                    const int ix = max(idx, 0);  // (j == 0: nothing to look up)
                    const uint32_t a = band[ix * 64], b = band[(BW + ix) * 64];
                    // the cell's two bits as integers: (0,0) up, (1,0) left, (1,1) diagonal, (0,1) stop; the move and
                    // the steps it takes are arithmetic on them (as fast as the compare-and-select form it replaced:
                    // the walk is not bound by its instruction count, profiles/r03_notes.md)
                    uint32_t a1 = (a >> row) & 1u;
                    const uint32_t b1 = (b >> row) & 1u;
                    // align.go:105-116 / 186-202: up unless left is strictly better, left unless the diagonal is
                    uint32_t stop = 0u;
                    if (LOCAL) stop = (b1 & ~a1) | (uint32_t)(j == 0);  // (0, 1): H = 0; column 0: H = 0
                    else if (j == 0) a1 = 0u;                             // global: the left border's moves are forced (align.go:99-100)
                    const uint32_t ab = a1 & b1;
                    const uint32_t keep = stop ^ 1u;
                    const uint32_t mv = a1 + ab;  // BA_MV_UP = 0, BA_MV_LEFT = 1, BA_MV_DIAG = 2
                    done = done || stop != 0u;
                    opw |= (mv & (0u - keep)) << sh;
                    sh += 2u * keep;
                    const int di = (int)((1u - a1 + ab) & keep), dj = (int)(a1 & keep);
                    i -= di;
                    j -= dj;
                    idx -= dj;
                    row -= (uint32_t)di;
                    if (sh == 32) flush();
                }
            }
        }
    }
    // row 0: global walks left along the top border; local stops there (H = 0)
    while (!LOCAL && __any(j > 0)) {
        if (j > 0) {
            opw |= (uint32_t)BA_MV_LEFT << sh;
            sh += 2;
            --j;
            if (sh == 32) flush();
        }
    }
    if (!live) return;
    const uint32_t kq = nrec * 16 + (sh >> 1);
    if (sh) flush();
    // where the trace ends inside the pair's slot (as kernels.hip's walk places it)
    const uint64_t len = kq;
    // (128- or 64-byte aligned where the slot has the room -- the expansion's four threads of a pair then store one whole
    // 64-byte sector per round instead of parts of two: 0.31 -> 0.26 ms per 1M pairs -- else 16-byte aligned, else
    // wherever the slot ends)
    const uint64_t mis128 = ((uint64_t)P.t1 + slot + cap) & 127u, mis64 = mis128 & 63u, mis = mis64 & 15u;
    const uint64_t end = (mis128 + len <= cap) ? slot + cap - mis128 : (mis64 + len <= cap) ? slot + cap - mis64 : (mis + len <= cap) ? slot + cap - mis : slot + cap;
    P.t_off[p] = end - len;
    P.t_len[p] = len;
    BaMoveRec hd;
    hd.ops = (uint32_t)len;
    hd.i = (uint32_t)end;
    hd.j = (uint32_t)(end >> 32);
    hd.count = p;
    recs[-1] = hd;
}

// ----------------------------------------------------------------------------
// traceback, GLOBAL mode: recompute and walk.  The wavefront goes over the blocks of CKPT columns from the last to
// the first, all lanes together (every global path crosses every column): a block's columns are recomputed from
// its checkpoint into registers -- two planes per column, [dV' >= 1] and [dH' >= 1] -- and then walked right to
// left.  In a column a lane first takes its up moves (a run of zero bits of the first plane below its row: one
// count-leading-zeros), then one left or diagonal move into the column before.  Nothing but the checkpoints
// (3 bits per row every CKPT columns) and the match vectors is read.
// ----------------------------------------------------------------------------
template <int WW>
__device__ __forceinline__ uint32_t ba_bp_pick(const uint32_t (&a)[WW], int w)
{
    // A tree of two-way choices on the bits of w, each ONE v_bitop3_b32 (m ? x : y) with the bit spread over the word
    // by v_bfe_i32: 3 + (WW - 1) instructions.  (Not a chain of selects: hipcc turns `w == k ? a[k] : v` chains into an
    // indexed load from scratch memory and the block's planes then live there; a mask per word, (w == k) ? ~0 : 0 into
    // an OR, was 3 WW instructions.)
    static_assert(WW >= 1 && WW <= 8, "up to eight words of rows");
    constexpr uint32_t MUX = (BA_TA & BA_TB) | (~BA_TA & BA_TC) & 0xffu;
    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe(w, 0, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe(w, 1, 1),
                   m2 = (uint32_t)__builtin_amdgcn_sbfe(w, 2, 1);
    uint32_t t[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < (WW + 1) / 2; ++k) t[k] = 2 * k + 1 < WW ? ba_bitop3<MUX>(m0, a[2 * k + 1], a[2 * k]) : a[2 * k];
    if (WW <= 2) return t[0];
    uint32_t u[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < (WW + 3) / 4; ++k) u[k] = 4 * k + 2 < WW ? ba_bitop3<MUX>(m1, t[2 * k + 1], t[2 * k]) : t[2 * k];
    if (WW <= 4) return u[0];
    return ba_bitop3<MUX>(m2, u[1], u[0]);
}

// jstart: the column of the lane's end cell (global: m; fitting / overlap: where the DP kernel found the bottom row's
// maximum) -- a lane joins the sweep when the wavefront reaches that column.
template <int WW, int CKPT, int MODE>
__device__ __forceinline__ uint32_t ba_bp_walk_bundle(const BaTraceParams &P, const BaBundle &B, uint32_t *peq,
                                                      const uint8_t *code2, int lane, bool live, int jstart)
{
    const int n = (int)B.n, m = (int)B.m;
    const int front = WW * 32 - n;
    const int cgs = (m + 3) / 4;
    int jmax = jstart;
    if (MODE != BA_GLOBAL) {  // (global: every lane ends in column m)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) jmax = max(jmax, __shfl_xor(jmax, d));
    }
    const int nblk = ((MODE == BA_GLOBAL ? m : __builtin_amdgcn_readfirstlane(jmax)) + CKPT - 1) / CKPT;  // blocks right of every lane's end cell are not looked at
    const uint32_t top1 = MODE == BA_FITTING ? (1u << front) : 0u;
    const uint32_t *ck = P.dir + B.dir_off + lane;
    const uint32_t *cs = ck + 4 * WW * 64;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    if (m > 0) {
#pragma unroll
        for (int k = 0; k < 4 * WW; ++k) ba_lds_store(peq + k * 64, ck[k * 64]);
    }
    int i = live ? n : 0;  // the row the lane is in; its column is the wavefront's once it has joined
    int jc = jstart;       // ... and the column of the cell it is in (for the records)
    bool fin = false;      // fitting: the lane has reached the free top row
    const BaRecs recs = ba_recs(P.rec, B, lane);
    uint32_t opw = 0, sh = 0, nrec = 0;
    // closes the record: `ci`, `cj` is the cell its last move arrived in
    auto flush = [&](int ci, int cj) __attribute__((always_inline)) {
        const uint32_t count = sh >> 1;
        const uint32_t valid = count >= 16 ? 0xffffffffu : ((1u << sh) - 1u);
        const uint32_t lo = opw & 0x55555555u, hi = (opw >> 1) & 0x55555555u;
        const uint32_t n_left = __popc(lo & ~hi & valid), n_up = __popc(~lo & ~hi & 0x55555555u & valid);
        BaMoveRec rr;
        rr.ops = opw;
        rr.i = (uint32_t)ci + count - n_left;
        rr.j = (uint32_t)cj + count - n_up;
        rr.count = count;
        if (live) recs[nrec] = rr;
        ++nrec;
        opw = 0;
        sh = 0;
    };
    auto ups = [&](int k, int cj) __attribute__((always_inline)) {  // k up moves in column cj (code 0: only the position advances)
        while (k > 0) {
            const int take = min(k, (int)((32u - sh) >> 1));
            sh += 2u * (uint32_t)take;
            i -= take;
            k -= take;
            if (sh == 32) flush(i, cj);
        }
    };
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int j0 = blk * CKPT, cols = min(CKPT, m - j0);
        uint32_t V1[WW], V2[WW], V3[WW];
        const uint32_t *c0 = cs + (size_t)blk * (3 * WW * 64);
#pragma unroll
        for (int w = 0; w < WW; ++w) {
            V1[w] = c0[(0 * WW + w) * 64];
            V2[w] = c0[(1 * WW + w) * 64];
            V3[w] = c0[(2 * WW + w) * 64];
        }
        uint32_t sw[CKPT / 4];
#pragma unroll
        for (int q = 0; q < CKPT / 4; ++q) sw[q] = s2w[(size_t)min(j0 / 4 + q, cgs - 1) * 64];
        uint32_t oV[CKPT][WW], oP[CKPT][WW];
#pragma unroll
        for (int c = 0; c < CKPT; ++c) {
            if (c < cols) {
                const uint32_t b = (uint32_t)code2[(sw[c / 4] >> (8 * (c % 4))) & 0xffu] & 3u;
                const uint32_t *pq = peq + b * (WW * 64);
                uint32_t M[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) M[w] = pq[w * 64];
                ba_bp_column<WW>(V1, V2, V3, M, oP[c], top1);
#pragma unroll
                for (int w = 0; w < WW; ++w) oV[c][w] = V1[w];
            }
        }
        // (one instance per column, spelled out: left as a loop around the run loop below, hipcc keeps it rolled
        // and the block's planes, indexed by the loop counter, go to scratch memory)
        auto walk_column = [&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            if (c < cols) {
                const int j = j0 + c + 1;
                if (MODE == BA_GLOBAL || (j <= jstart && !fin)) {  // (global: every lane walks every column)
                    // up while the cell came from above (align.go:105-108: the first candidate wins ties)
                    while (i > 0) {
                        const int t = i - 1 + front;
                        const uint32_t f = (uint32_t)t & 31u;
                        const uint32_t x = ba_bp_pick<WW>(oV[c], t >> 5) << (31u - f);  // my row's bit on top, the rows above it below
                        const int run = x ? __clz(x) : (int)f + 1;
                        if (run == 0) break;
                        ups(run, j);
                        if (x) break;  // (else the run goes on in the word above)
                    }
                    if (i == 0 && MODE == BA_FITTING) {  // free top row: the alignment starts here (align_pair.go:158)
                        fin = true;
                    } else {
                        if (i == 0) {  // top border: left (align.go:92-95)
                            opw |= (uint32_t)BA_MV_LEFT << sh;
                        } else {
                            const int t = i - 1 + front;
                            const uint32_t pw = ba_bp_pick<WW>(oP[c], t >> 5);
                            const uint32_t diag = (pw >> ((uint32_t)t & 31u)) & 1u;  // left unless the diagonal is strictly better
                            opw |= (diag ? (uint32_t)BA_MV_DIAG : (uint32_t)BA_MV_LEFT) << sh;
                            i -= (int)diag;
                        }
                        jc = j - 1;
                        sh += 2;
                        if (sh == 32) flush(i, jc);
                    }
                }
            }
        };
        if constexpr (CKPT == 8) {
            walk_column(std::integral_constant<int, 7>{});
            walk_column(std::integral_constant<int, 6>{});
            walk_column(std::integral_constant<int, 5>{});
            walk_column(std::integral_constant<int, 4>{});
        }
        walk_column(std::integral_constant<int, 3>{});
        walk_column(std::integral_constant<int, 2>{});
        walk_column(std::integral_constant<int, 1>{});
        walk_column(std::integral_constant<int, 0>{});
    }
    if (MODE != BA_OVERLAP && !fin) ups(i, 0);  // left border: up (align.go:99-100); overlap: the left column is free, the alignment starts there
    const uint32_t len = nrec * 16 + (sh >> 1);
    if (sh) flush(i, jc);
    return len;
}

template <int CKPT, int MODE>
__global__ void __launch_bounds__(256) ba_trace_bp_ckpt_kernel(BaTraceParams P, int max_words)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;
    uint8_t *code2 = reinterpret_cast<uint8_t *>(lds) + 256;  // (same places as in the DP kernel)
    for (int k = threadIdx.x; k < 256; k += blockDim.x) code2[k] = P.code2[k];
    __syncthreads();
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const bool live = lane < (int)B.count;
    const uint32_t p = live ? P.perm[B.first + lane] : 0u;
    const int n = (int)B.n, m = (int)B.m;
    int jstart = 0;
    if (live) {
        const BaLocalBest bb = P.best[B.first + lane];
        P.scores[p] = (int64_t)bb.score;
        jstart = bb.j;
        if (P.score_only && P.t_off) {  // BIOALIGN_END_CELLS: (n, m), align.go:123-124; (n, leftmost maximum), align_pair.go:169-182
            P.t_off[p] = (uint64_t)bb.i;
            P.t_len[p] = (uint64_t)bb.j;
        }
    }
    if (P.score_only) return;
    uint32_t *peq = lds + 128 + (size_t)wave * (4 * max_words * 64) + lane;
    const int words = (n + 31) / 32;
    uint32_t len = 0;
#define BA_BP_CASE(W)                                                                                          \
    case W:                                                                                                    \
        if constexpr (W <= 5 || CKPT == 4) len = ba_bp_walk_bundle<W, CKPT, MODE>(P, B, peq, code2, lane, live, jstart); \
        break;
    switch (words) {
    case 0: {  // no rows: global, m left moves along the top border; fitting / overlap: the empty alignment
        const BaRecs recs = ba_recs(P.rec, B, lane);
        for (int k = 0; MODE == BA_GLOBAL && k < m; k += 16) {
            const int count = min(16, m - k);
            BaMoveRec rr;
            rr.ops = 0x55555555u & (count >= 16 ? 0xffffffffu : ((1u << (2 * count)) - 1u));
            rr.i = 0;
            rr.j = (uint32_t)(m - k);
            rr.count = (uint32_t)count;
            if (live) recs[k / 16] = rr;
        }
        len = MODE == BA_GLOBAL ? (uint32_t)m : 0u;
        break;
    }
        BA_BP_CASE(1)
        BA_BP_CASE(2)
        BA_BP_CASE(3)
        BA_BP_CASE(4)
        BA_BP_CASE(5)
        BA_BP_CASE(6)
        BA_BP_CASE(7)
        BA_BP_CASE(8)
    default: break;
    }
#undef BA_BP_CASE
    if (!live) return;
    // where the trace ends inside the pair's slot (as kernels.hip's walk places it)
    const uint64_t slot = P.s1_off[p] + P.s2_off[p];
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    // (128- or 64-byte aligned where the slot has the room -- the expansion's four threads of a pair then store one whole
    // 64-byte sector per round instead of parts of two: 0.31 -> 0.26 ms per 1M pairs -- else 16-byte aligned, else
    // wherever the slot ends)
    const uint64_t mis128 = ((uint64_t)P.t1 + slot + cap) & 127u, mis64 = mis128 & 63u, mis = mis64 & 15u;
    const uint64_t end = (mis128 + len <= cap) ? slot + cap - mis128 : (mis64 + len <= cap) ? slot + cap - mis64 : (mis + len <= cap) ? slot + cap - mis : slot + cap;
    P.t_off[p] = end - len;
    P.t_len[p] = len;
    BaMoveRec hd;
    hd.ops = len;
    hd.i = (uint32_t)end;
    hd.j = (uint32_t)(end >> 32);
    hd.count = p;
    ba_recs(P.rec, B, lane)[-1] = hd;
}

// ----------------------------------------------------------------------------
// host-side launchers (called from api.cpp)
// ----------------------------------------------------------------------------
extern "C" int ba_bp_max_rows(void) { return BA_BP_MAX_WORDS * 32; }
extern "C" int ba_bp_max_local_score(void) { return (1 << BA_BP_SCORE_BITS) - 1; }

extern "C" size_t ba_dp_bp_lds_bytes(int max_words, int waves) { return 512 + (size_t)waves * 4 * max_words * 64 * 4; }

// checkpoints every 8 columns where the rows fit five words (the walk holds a block's planes in registers)
static inline int ba_bp_ckpt(int max_words) { return max_words <= 5 ? 8 : 4; }

extern "C" hipError_t ba_launch_dp_bp(const BaDpParams &P, int mode, int max_words, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    if (nbundles == 0) return hipSuccess;
    if (max_words < 1 || max_words > BA_BP_MAX_WORDS || P.nb > 4 || !P.table) return hipErrorInvalidValue;
    const int waves = 4;
    const dim3 grid((nbundles + waves - 1) / waves), block(waves * 64);
    const size_t lds = ba_dp_bp_lds_bytes(max_words, waves);
#define BA_BP_LAUNCH_DP(M)                                                                                   \
    if (mode == M && ba_bp_ckpt(max_words) == 8)                                                             \
        hipLaunchKernelGGL((ba_dp_bp_kernel<M, 8, 5>), grid, block, lds, st, P, max_words);                  \
    else if (mode == M)                                                                                      \
        hipLaunchKernelGGL((ba_dp_bp_kernel<M, 4, 8>), grid, block, lds, st, P, max_words);
    BA_BP_LAUNCH_DP(BA_GLOBAL)
    else BA_BP_LAUNCH_DP(BA_FITTING)
    else BA_BP_LAUNCH_DP(BA_OVERLAP)
#undef BA_BP_LAUNCH_DP
    else if (mode == BA_LOCAL && max_words <= 3)
        hipLaunchKernelGGL((ba_dp_bp_kernel<BA_LOCAL, 4, 3>), grid, block, lds, st, P, max_words);
    else if (mode == BA_LOCAL && max_words <= 5)
        hipLaunchKernelGGL((ba_dp_bp_kernel<BA_LOCAL, 4, 5>), grid, block, lds, st, P, max_words);
    else if (mode == BA_LOCAL)
        hipLaunchKernelGGL((ba_dp_bp_kernel<BA_LOCAL, 4, 8>), grid, block, lds, st, P, max_words);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// the walk only (scores, move records, lengths); the expansion is ba_launch_trace_linear's phase 2
extern "C" hipError_t ba_launch_trace_bp(const BaTraceParams &P, int mode, int max_words, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    if (nbundles == 0) return hipSuccess;
    if (!P.score_only && !P.rec) return hipErrorInvalidValue;
    if (max_words < 1 || max_words > BA_BP_MAX_WORDS) return hipErrorInvalidValue;
    const int waves = 4;
    const dim3 grid((nbundles + waves - 1) / waves), block(waves * 64);
    if (mode == BA_GLOBAL || mode == BA_FITTING || mode == BA_OVERLAP) {
        const size_t lds = ba_dp_bp_lds_bytes(max_words, waves);
#define BA_BP_LAUNCH_WALK(M)                                                                                 \
    if (mode == M && ba_bp_ckpt(max_words) == 8)                                                             \
        hipLaunchKernelGGL((ba_trace_bp_ckpt_kernel<8, M>), grid, block, lds, st, P, max_words);             \
    else if (mode == M)                                                                                      \
        hipLaunchKernelGGL((ba_trace_bp_ckpt_kernel<4, M>), grid, block, lds, st, P, max_words);
        BA_BP_LAUNCH_WALK(BA_GLOBAL)
        else BA_BP_LAUNCH_WALK(BA_FITTING)
        else BA_BP_LAUNCH_WALK(BA_OVERLAP)
#undef BA_BP_LAUNCH_WALK
    } else if (mode == BA_LOCAL) {
        hipLaunchKernelGGL(ba_trace_bp_kernel<true>, grid, block, (size_t)waves * 2 * BA_BP_BG * 4 * 64 * 4, st, P);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
