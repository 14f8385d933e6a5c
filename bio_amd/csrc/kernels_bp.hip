// kernels_bp.hip -- gfx950 (MI355X): BIT-PARALLEL global alignment for unit scoring.
//
// Scope: AlignGlobal (align.go:76-143) when every substitution score is +1 or -1 and the gap penalty is 1 --
// BASELINE.json's DNA configuration -- with at most four column symbols and s1 of at most 256 symbols.  Everything
// else stays on the general kernels (kernels.hip).  Same boundary, same outputs, same workspace layout sizes.
//
// One lane = one pair, as in kernels.hip; but a lane holds a whole COLUMN of the matrix as bit vectors over the rows
// (bit t of word t/32 = row t of the front-padded rows) and a column costs about 40 logic instructions per 32 rows
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
        for (int b = 0; b < 4; ++b) peq[(b * WW + w) * 64] = pl[b] & keep;
    }
}

template <int WW>
__device__ __forceinline__ void ba_bp_bundle(const BaDpParams &P, const BaBundle &B, uint32_t *peq, const uint8_t *mm,
                                             const uint8_t *code2, int lane)
{
    const int n = (int)B.n, m = (int)B.m;
    const int front = WW * 32 - n;  // 0..31
    const int cgs = (m + 3) / 4;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    uint32_t *dir = P.dir + B.dir_off + (size_t)lane * 4;

    ba_bp_build_peq<WW>(s1w, mm, front, peq);

    // column 0: dV = gap in every row (x = 0); the rows above the matrix hold x = 1 for ever and hand h = 0 -- the top
    // border's dH = gap (align.go:92-95) -- down to row 1
    uint32_t V1[WW], V2[WW], V3[WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) V1[w] = V2[w] = V3[w] = 0u;
    V1[0] = (1u << front) - 1u;

    uint32_t accV[4][WW], accP[4][WW];
    auto column_group = [&](int cg, uint32_t sw, auto full) {
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (FULL || cg * 4 + c < m) {
                const uint32_t b = (uint32_t)code2[(sw >> (8 * c)) & 0xffu] & 3u;
                const uint32_t *pq = peq + b * (WW * 64);
                uint32_t M[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) M[w] = pq[w * 64];
                ba_bp_column<WW>(V1, V2, V3, M, accP[c]);
#pragma unroll
                for (int w = 0; w < WW; ++w) accV[c][w] = V1[w];
            } else {
#pragma unroll
                for (int w = 0; w < WW; ++w) accV[c][w] = accP[c][w] = 0u;
            }
        }
#pragma unroll
        for (int w = 0; w < WW; ++w) {
            *reinterpret_cast<uint4 *>(dir + ((size_t)(w * 2 + 0) * cgs + cg) * 256) =
                make_uint4(accV[0][w], accV[1][w], accV[2][w], accV[3][w]);
            *reinterpret_cast<uint4 *>(dir + ((size_t)(w * 2 + 1) * cgs + cg) * 256) =
                make_uint4(accP[0][w], accP[1][w], accP[2][w], accP[3][w]);
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
        bb.score = sum - front - n - m;
        bb.i = n;
        bb.j = m;
        bb.pad = 0;
        P.best[B.first + lane] = bb;
    }
}

// LOCAL mode (align.go:152-249): bp_column.h's ba_bp_local_column; the end cell is tracked column by column.
template <int WW>
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
    uint32_t V1[WW], V2[WW], V3[WW], Zr[WW], H[NB][WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) {
        V1[w] = Zr[w] = 0xffffffffu;
        V2[w] = V3[w] = 0u;
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
                ba_bp_local_column<WW, NB>(V1, V2, V3, H, Zr, M, S, T, accA[c], accB[c], gt, eq);
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

template <int MODE>
__global__ void __launch_bounds__(256) ba_dp_bp_kernel(BaDpParams P, int max_words)
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
    if (words == 0) {  // no rows: global S(0, m) = -m, the walk goes left along the top border; local: the empty alignment
        if (lane < (int)B.count) {
            BaLocalBest bb;
            bb.score = MODE == BA_LOCAL ? 0 : -(int)B.m;
            bb.i = 0;
            bb.j = MODE == BA_LOCAL ? 0 : (int)B.m;
            bb.pad = 0;
            P.best[B.first + lane] = bb;
        }
        return;
    }
#define BA_BP_CASE(W)                                                                   \
    case W:                                                                             \
        if (MODE == BA_LOCAL) ba_bp_local_bundle<W>(P, B, peq, mm, code2, lane);        \
        else ba_bp_bundle<W>(P, B, peq, mm, code2, lane);                               \
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
#define BA_BP_BG 6  // column groups per band (24 columns: a diagonal through 32 rows reloads once)

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
    BaMoveRec *recs = reinterpret_cast<BaMoveRec *>(P.rec) + B.rec_off + (size_t)lane * B.rec_cap + 1;
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
            band[(k * 4 + 0) * 64] = qv[k].x;
            band[(k * 4 + 1) * 64] = qv[k].y;
            band[(k * 4 + 2) * 64] = qv[k].z;
            band[(k * 4 + 3) * 64] = qv[k].w;
            band[(BW + k * 4 + 0) * 64] = qp[k].x;
            band[(BW + k * 4 + 1) * 64] = qp[k].y;
            band[(BW + k * 4 + 2) * 64] = qp[k].z;
            band[(BW + k * 4 + 3) * 64] = qp[k].w;
        }
        return (cg_hi - (BG - 1)) * 4;
    };

    // Bundles are equal-shape: every lane starts in row n, so the wavefront goes up WORD by word together (one band
    // load per word and lane, all lanes' loads in flight at once); inside a word every lane takes one move per round
    // until all have left it upwards.  The round is branch-free but for the record that fills up every 16 moves:
    // `bit` is the lane's row inside the word, `idx` its column inside the band.
    bool done = false;  // local: the walk has reached a cell with H = 0 (or a border, where H = 0 too)
    for (int w = (n + front - 1) >> 5; w >= 0 && n > 0; --w) {
        const int lo_i = max(32 * w + 1 - front, 1);  // first row of the word; a lane is in it while i >= lo_i
        uint32_t bit = 1u << ((uint32_t)(i - 1 + front) & 31u);
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
            if (act) {
                const int ix = max(idx, 0);  // (j == 0: nothing to look up)
                const uint32_t a = band[ix * 64], b = band[(BW + ix) * 64];
                const bool abit = (a & bit) != 0u, bbit = (b & bit) != 0u;
                // align.go:105-116 / 186-202: up unless left is strictly better, left unless the diagonal is
                uint32_t mv = !abit ? BA_MV_UP : !bbit ? BA_MV_LEFT : BA_MV_DIAG;
                bool stop = false;
                if (LOCAL) stop = j == 0 || (!abit && bbit);  // (0, 1): H = 0; column 0: H = 0
                else if (j == 0) mv = BA_MV_UP;               // global: the left border's moves are forced (align.go:99-100)
                if (stop) {
                    done = true;
                } else {
                    opw |= mv << sh;
                    sh += 2;
                    const int di = mv != BA_MV_LEFT, dj = mv != BA_MV_UP;
                    i -= di;
                    j -= dj;
                    idx -= dj;
                    bit >>= di;
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
    const uint64_t mis = ((uint64_t)P.t1 + slot + cap) & 15u;
    const uint64_t end = (mis + len <= cap) ? slot + cap - mis : slot + cap;
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
// host-side launchers (called from api.cpp)
// ----------------------------------------------------------------------------
extern "C" int ba_bp_max_rows(void) { return BA_BP_MAX_WORDS * 32; }
extern "C" int ba_bp_max_local_score(void) { return (1 << BA_BP_SCORE_BITS) - 1; }

extern "C" size_t ba_dp_bp_lds_bytes(int max_words, int waves) { return 512 + (size_t)waves * 4 * max_words * 64 * 4; }

extern "C" hipError_t ba_launch_dp_bp(const BaDpParams &P, int mode, int max_words, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    if (nbundles == 0) return hipSuccess;
    if (max_words < 1 || max_words > BA_BP_MAX_WORDS || P.nb > 4 || !P.table) return hipErrorInvalidValue;
    const int waves = 4;
    const dim3 grid((nbundles + waves - 1) / waves), block(waves * 64);
    const size_t lds = ba_dp_bp_lds_bytes(max_words, waves);
    if (mode == BA_GLOBAL)
        hipLaunchKernelGGL(ba_dp_bp_kernel<BA_GLOBAL>, grid, block, lds, st, P, max_words);
    else if (mode == BA_LOCAL)
        hipLaunchKernelGGL(ba_dp_bp_kernel<BA_LOCAL>, grid, block, lds, st, P, max_words);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// the walk only (scores, move records, lengths); the expansion is ba_launch_trace_linear's phase 2
extern "C" hipError_t ba_launch_trace_bp(const BaTraceParams &P, int mode, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    if (nbundles == 0) return hipSuccess;
    if (!P.score_only && !P.rec) return hipErrorInvalidValue;
    const int waves = 4;
    const dim3 grid((nbundles + waves - 1) / waves), block(waves * 64);
    const size_t lds = (size_t)waves * 2 * BA_BP_BG * 4 * 64 * 4;
    if (mode == BA_GLOBAL)
        hipLaunchKernelGGL(ba_trace_bp_kernel<false>, grid, block, lds, st, P);
    else if (mode == BA_LOCAL)
        hipLaunchKernelGGL(ba_trace_bp_kernel<true>, grid, block, lds, st, P);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
