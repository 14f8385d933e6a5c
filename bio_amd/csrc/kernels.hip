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
#include "device_helpers.h"

// ----------------------------------------------------------------------------
// pack: caller bytes -> lane-interleaved words of RAW bytes (4 symbols per word)
//   word k of lane l of a bundle lives at  w_off + k*64 + l
// s1 is padded at the FRONT up to a whole number of strips (the DP kernel gives
// those rows the padding code); s2 is padded at the back of its last word.
// One lane = one pair: a lane walks its own sequence with 4-byte loads (the
// consecutive words of a pair share cache lines) and the wavefront's stores are
// full 256-byte lines.
// ----------------------------------------------------------------------------
// Bytes src[pos .. pos+3] of a sequence of `len` > 0 bytes that starts `so` bytes
// into the 4-byte-aligned array `ab`, from ALIGNED dword loads only (unaligned
// dword loads run at a fraction of the aligned rate on gfx950: 0.37 TB/s measured
// for this kernel).  Positions outside [0, len) come back as arbitrary bytes (no
// consumer looks at them); the two loads are clamped to the dwords that hold the
// sequence's first and last byte, so nothing outside the sequence's own dwords is
// ever touched.
__device__ __forceinline__ uint32_t ba_load4(const uint8_t *ab, int64_t so, int pos, int len)
{
    const int64_t first = so & ~(int64_t)3, last = (so + len - 1) & ~(int64_t)3;
    const int64_t ad = so + pos;
    const int64_t al = ad & ~(int64_t)3;
    const int64_t p0 = min(max(al, first), last), p1 = min(max(al + 4, first), last);
    const uint32_t lo = *reinterpret_cast<const uint32_t *>(ab + p0);
    const uint32_t hi = *reinterpret_cast<const uint32_t *>(ab + p1);
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(ad & 3) * 8);
}

// One wavefront transposes one bundle through an LDS tile of 64 pairs x 256
// bytes: reading, the whole wavefront takes 256 consecutive bytes of ONE pair
// per instruction (coalesced); writing, lane l emits its own pair's words, so a
// store instruction is one full 256-byte line of the interleaved layout.
// Tile rows are 65 words apart: both phases are bank-conflict free.
template <bool FRONT_PADDED>
__device__ __forceinline__ void ba_pack_side(const uint8_t *base, uint64_t mine, int len, int shift, int nwords,
                                             int count, uint32_t *tile, uint32_t *out, int lane)
{
    // `mine` is this lane's pair's byte offset into `base`.
    // output word k holds source bytes 4k-shift .. 4k-shift+3 (zero outside [0, len))
    const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
    const int mis = (int)((uint64_t)base & 3);
    const uint8_t *ab = base - mis;
    if (len == 0) {  // bundle-uniform: nothing to read (s2 has no words; s1's are all padding)
        for (int k = 0; k < nwords; ++k) out[(size_t)k * 64] = 0u;
        return;
    }
    // The bundle's sequences back to back in the caller's array (a batch of one shape in its own order: BASELINE's
    // configurations): the wavefront stages the whole region with coalesced dword loads -- one round trip instead of
    // one per 16 pairs -- and every lane cuts its own words out of it (two aligned LDS reads and an alignbit per word).
    {
        const uint32_t nlo = (uint32_t)__shfl_down((int)lo, 1), nhi = (uint32_t)__shfl_down((int)hi, 1);
        const uint64_t next = ((uint64_t)nhi << 32) | nlo;
        const bool follows = lane + 1 >= count || next == mine + (uint64_t)len;
        const uint64_t total = (uint64_t)count * (uint64_t)len;
        if (__all(follows) && total + 8 <= 64 * 65 * 4) {
            const uint64_t first = (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, 0) << 32) |
                                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)lo, 0)) + (uint64_t)mis;
            const uint64_t a0 = first & ~(uint64_t)3;
            const int sh0 = (int)(first & 3);
            const int nd = (sh0 + (int)total + 3) / 4;  // dwords that hold the region: nothing outside them is touched
            for (int d = lane; d < nd; d += 64) tile[d] = *reinterpret_cast<const uint32_t *>(ab + a0 + 4 * (uint64_t)d);
            // (LDS operations of one wavefront complete in order: no barrier)
            for (int k = 0; k < nwords; ++k) {
                const int bb = lane * len + 4 * k - shift + sh0;  // first byte of the word, relative to a0 (may lie outside the region)
                const int i0 = min(max(bb >> 2, 0), nd - 1), i1 = min(max((bb >> 2) + 1, 0), nd - 1);
                const uint32_t w0 = tile[i0], w1 = tile[i1];
                out[(size_t)k * 64] = lane < count ? __builtin_amdgcn_alignbit(w1, w0, (uint32_t)(bb & 3) * 8) : 0u;
            }
            return;
        }
    }
    for (int k0 = 0; k0 < nwords; k0 += 64) {
        const int kw = min(64, nwords - k0);
        // 16 pairs per round, so that 32 independent loads are in flight per lane
        for (int q0 = 0; q0 < count; q0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = min(q0 + u, count - 1);
                // (readlane returns int: go through uint32_t, or the low half sign-extends over the high one)
                const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, q) << 32) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)lo, q);
                // unconditional (addresses are clamped into the sequence): no branch, so the
                // loads of a round issue back to back and one wait covers them all
                v[u] = ba_load4(ab, (int64_t)off + mis, 4 * (k0 + lane) - shift, len);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (q0 + u < count && lane < kw) tile[(q0 + u) * 65 + lane] = v[u];
        }
        // each lane reads back only its own row, after its own wavefront wrote it: LDS
        // operations of one wavefront complete in order, no barrier needed
        for (int k = 0; k < kw; ++k) out[(size_t)(k0 + k) * 64] = lane < count ? tile[lane * 65 + k] : 0u;
    }
}

__global__ void __launch_bounds__(256) ba_pack_kernel(BaPackParams P)
{
    __shared__ uint32_t tiles[4][64 * 65];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * 4 + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const int R = P.rows_per_strip;
    const int n = (int)B.n, m = (int)B.m;
    const int strips = (n + R - 1) / R;
    const int front = strips * R - n;
    const bool live = lane < (int)B.count;
    const uint32_t p = live ? P.perm[B.first + lane] : 0u;
    const uint64_t a1 = live ? P.s1_off[p] : 0, a2 = live ? P.s2_off[p] : 0;
    ba_pack_side<true>(P.s1, a1, n, front, strips * R / 4, (int)B.count, tiles[wave], P.s1w + B.s1w_off + lane, lane);
    ba_pack_side<false>(P.s2, a2, m, 0, (m + 3) / 4, (int)B.count, tiles[wave], P.s2w + B.s2w_off + lane, lane);
}

// ----------------------------------------------------------------------------
// linear-gap DP: one column of one strip (R rows in VGPRs)
// ----------------------------------------------------------------------------
// R0: first row of the strip that is computed.  The first strip of a pair starts with `front`
// rows that lie above the matrix; they only reproduce row 0, which `top` / `topdiag` already
// are, so the first R0 <= front of them are skipped (their residue fields are never read, and
// the fields of the rows that are computed land where they always do: v_alignbit fills a
// word from the top).
template <int MODE, int R, int K, int R0>
__device__ __forceinline__ void ba_linear_column(int (&h)[R], int (&rmax)[R / 2], uint32_t (&acc)[R * K / 32],
                                                 const uint32_t (&pw)[R / 4], int top, int &topdiag, int g)
{
    constexpr int F = 32 / K;  // residues per word
    static_assert(R0 % 2 == 0 && R0 < R, "R0 must be even and leave rows to compute");
    int up = top, diag = topdiag;
    topdiag = top;
#pragma unroll
    for (int r = R0; r < R; ++r) {
        const int left = h[r];
        const int d = diag + ba_sbyte(pw[r >> 2], r & 3);
        int hn;
        if (MODE == BA_LOCAL) {
            // max(0, max(up, left) - g, diag + w) with w + g in the profile: one max3 and one
            // saturating unsigned subtract (v_sub_u32 clamp); max3 >= up >= 0, so unsigned is exact
            hn = (int)__builtin_elementwise_sub_sat((uint32_t)ba_max3_asm(up, left, d), (uint32_t)g);
            // running maximum per PAIR of rows: one v_max3_i32 per two cells
            if (r & 1) rmax[r / 2] = ba_max3_asm(rmax[r / 2], up, hn);
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
//   [strip][row block w][column group cg = (j-1)/4][lane][column c = (j-1)%4]
// A word holds F = 32/K consecutive rows of one column (a "row block"); a strip
// has WPC = R/F row blocks.  The 4 columns of one row block are one aligned uint4
// per lane -- the unit the traceback fetches -- and the 64 lanes' uint4s of one
// (row block, column group) are one contiguous KB: a store of the DP kernel and a
// load of the walk each touch whole cache lines and nothing else (with the row
// blocks of a strip interleaved per lane the walk fetched twice what it used).
template <int MODE, int R, int K>
__global__ void __launch_bounds__(256) ba_dp_linear_kernel(BaDpParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int HALVES = R / 16;
    constexpr int WPC = R * K / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;

    // the transposed int8 score table and the byte -> code maps, shared by the block
    const int wt_words = P.nb4 * P.na_pad;
    const int wt_aligned = (wt_words + 3) & ~3;
    for (int i = threadIdx.x; i < wt_words; i += blockDim.x) lds[i] = P.wt[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x)
        lds[wt_aligned + i] = i < 64 ? reinterpret_cast<const uint32_t *>(P.code1)[i]
                                     : reinterpret_cast<const uint32_t *>(P.code2)[i - 64];
    const uint8_t *code1 = reinterpret_cast<const uint8_t *>(lds + wt_aligned);
    const uint8_t *code2 = code1 + 256;
    __syncthreads();

    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const int n = (int)B.n, m = (int)B.m;
    const int g = P.gap;
    const int strips = (n + R - 1) / R;
    const int cgs = (m + 3) / 4;
    const int front = strips * R - n;

    uint8_t *prof = reinterpret_cast<uint8_t *>(lds + wt_aligned + 128) + (size_t)wave * P.nb * (HALVES * 64 * 16);
    uint8_t *prof_lane = prof + lane * 16;

    // boundary deltas: the row a strip hands to the next one, as K-bit horizontal
    // differences (F columns per word, lane-interleaved, two buffers by strip parity)
    constexpr int F = 32 / K;
    constexpr uint32_t MASK = (1u << K) - 1u;
    constexpr uint32_t REP = 0xffffffffu / MASK;  // every K-bit field = 1
    const int nbw = (m + F - 1) / F;
    uint32_t *bd = reinterpret_cast<uint32_t *>(P.hb) + B.hb_off + lane;
    uint32_t *dir = P.dir + B.dir_off + (size_t)lane * 4;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    const int off = (MODE == BA_LOCAL) ? g : 0;  // a horizontal step is never below -off

    int best_sc = 0, best_i = 0, best_j = 0;  // local: maximum and its row pair; others: end cell
    if (MODE == BA_GLOBAL) best_j = m;        // (strips == 0: row 0 is the bottom row, Hs(0, .) handled by the trace kernel)

    // Strips are aligned to the BOTTOM of the matrix: strip s covers matrix rows
    // i0+1 .. i0+R with i0 = n - (strips-s)*R, so the first strip may start with
    // rows i <= 0 that do not exist.  Those rows carry the padding code (score
    // -128 against everything) and column-0 value 0; in the biased space (and,
    // for local mode, under the clamp at 0) they reproduce row 0 exactly, so row
    // 1 sees the right border and the last strip always ends on row n.
    for (int s = 0; s < strips; ++s) {
        const int i0 = n - (strips - s) * R;
        const bool last_strip = (s == strips - 1);
        const uint32_t *bd_rd = bd + (size_t)((s + 1) & 1) * nbw * 64;
        uint32_t *bd_wr = bd + (size_t)(s & 1) * nbw * 64;
        // row above the first strip = row 0 (align.go:92-95, 175-177; align_pair.go:110-113, 158, 187):
        // constant horizontal step in the working space
        const uint32_t bd_row0 = (uint32_t)(((MODE == BA_FITTING) ? g : 0) + off) * REP;
        // codes of the strip's rows
        uint32_t aw[R / 4];
#pragma unroll
        for (int q = 0; q < R / 4; ++q) aw[q] = s1w[(size_t)(s * (R / 4) + q) * 64];
        ba_build_strip_profile<R>(lds, P.nb, P.nb4, P.na_pad, code1, P.pad_code, aw, s, front, prof_lane);
        // column 0 (align.go:99-100, 181; align_pair.go:113, 131, 159, 188) in the working space
        int h[R], rmax[R / 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            h[r] = (MODE == BA_OVERLAP) ? g * max(i0 + r + 1, 0) : 0;
            rmax[r / 2] = 0;
        }
        int topdiag = (MODE == BA_OVERLAP) ? g * max(i0, 0) : 0;  // Hs(i0, 0)
        uint32_t *dirs = dir + (size_t)s * WPC * cgs * 256;  // + (w * cgs + cg) * 256
        uint32_t acc[4][WPC];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int w = 0; w < WPC; ++w) acc[c][w] = 0;
        uint32_t accb = 0;  // boundary deltas of this strip's last row, being collected
        // bottom row of the matrix: leftmost strict maximum of the UNBIASED scores, scan from
        // column 0 (align_pair.go:169-182); unbiased = Hs(n, j) - g*j up to a constant
        int end_v = h[R - 1], end_j = 0, end_cur = h[R - 1];
        // What a column group needs from memory: its packed s2 word and the word of boundary
        // deltas above it.  They are fetched one group AHEAD (the loads for group cg+1 are
        // issued before group cg is computed), so their latency never sits on the critical path.
        struct GroupIn {
            uint32_t sw, bw;
        };
        auto fetch_group = [&](int cg) {
            GroupIn gi;
            gi.sw = s2w[(size_t)cg * 64];
            gi.bw = bd_rd[(size_t)((cg * 4) / F) * 64];
            return gi;
        };
        // one group of 4 columns (one packed s2 word): FULL groups run without any
        // per-column test so that the strip stays in the same registers all along
        auto column_group = [&](int cg, const GroupIn &in, auto full, auto first_row) {
            constexpr bool FULL = decltype(full)::value;
            constexpr int R0 = decltype(first_row)::value;
            const int jbase = cg * 4 + 1;
            const uint32_t bw = (s == 0) ? bd_row0 : in.bw;
            const int sh0 = ((cg * 4) % F) * K;  // uniform
            uint32_t bcode[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bcode[c] = code2[(in.sw >> (c * 8)) & 0xffu];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (FULL || jbase + c <= m) {
                    const uint32_t b = bcode[c];
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
                    // score above this column = score above the previous one + the stored step
                    const int top = topdiag + (int)((bw >> (sh0 + c * K)) & MASK) - off;
                    const int old_last = h[R - 1];
                    ba_linear_column<MODE, R, K, R0>(h, rmax, acc[c], pw, top, topdiag, g);
                    accb = __builtin_amdgcn_alignbit((uint32_t)(h[R - 1] - old_last + off), accb, K);
                    if ((MODE == BA_FITTING || MODE == BA_OVERLAP) && last_strip) {
                        const int v = h[R - 1] - g * (jbase + c);
                        if (v > end_v) {
                            end_v = v;
                            end_j = jbase + c;
                            end_cur = h[R - 1];
                        }
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < WPC; ++w)
                *reinterpret_cast<uint4 *>(dirs + ((size_t)w * cgs + cg) * 256) =
                    make_uint4(acc[0][w], acc[1][w], acc[2][w], acc[3][w]);
            if (FULL && !last_strip && ((cg + 1) * 4) % F == 0) bd_wr[(size_t)((cg * 4) / F) * 64] = accb;
        };
        const int full_groups = m / 4;
        auto sweep = [&](auto first_row) {
            GroupIn nxt = fetch_group(0);
            // (the first group's loads are waited for HERE: left pending into the loop, the compiler's counter state at
            // the loop head asks for `vmcnt(1)` in every iteration, which also waits for the previous group's stores)
            asm volatile("" : "+v"(nxt.sw), "+v"(nxt.bw));
            for (int cg = 0; cg < full_groups; ++cg) {
                const GroupIn cur = nxt;
                nxt = fetch_group(min(cg + 1, cgs - 1));
                column_group(cg, cur, std::true_type{}, first_row);
            }
            if (full_groups < cgs) column_group(full_groups, nxt, std::false_type{}, first_row);
        };
        if (cgs > 0) {
            // the rows of the first strip that lie above the matrix are skipped in steps of 8
            // (global and local only: with the end-cell scan of fitting / overlap in the loop the extra
            // instances cost the kernel its occupancy)
            constexpr bool TRIM = (MODE == BA_GLOBAL || MODE == BA_LOCAL);
            const int skip = (TRIM && s == 0) ? (front & ~7) : 0;
            if (!TRIM || skip == 0) sweep(std::integral_constant<int, 0>{});
            else if (skip == 8) sweep(std::integral_constant<int, (TRIM ? 8 : 0)>{});
            else if (R > 16 && skip == 16) sweep(std::integral_constant<int, (TRIM && R > 16 ? 16 : 0)>{});
            else sweep(std::integral_constant<int, (TRIM && R > 16 ? 24 : 0)>{});
            // the last, partial word of boundary deltas: its m % F fields sit at the top
            if (!last_strip && (m % F) != 0) bd_wr[(size_t)(m / F) * 64] = accb >> ((F - m % F) * K);
        }
        if (last_strip && MODE != BA_LOCAL) {
            if (MODE == BA_GLOBAL) {
                best_sc = h[R - 1];  // Hs(n, m)
                best_i = n;
                best_j = m;
            } else {
                best_sc = end_cur;
                best_i = n;
                best_j = end_j;
            }
        }
        if (MODE == BA_LOCAL) {
            // first PAIR of rows (ascending) that holds the strict maximum (align.go:216-221);
            // the traceback kernel finds the first cell of that pair in row-major order.
            // Rows i <= 0 score 0 everywhere and never win.
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                if (rmax[r / 2] > best_sc) {
                    best_sc = rmax[r / 2];
                    best_i = i0 + r + 1;  // first row of the pair (may be <= 0 only with a zero score)
                }
            }
        }
    }
    // end cell for the traceback kernel.  local: (maximum, first row of its row pair, -);
    // other modes: (working-space score of the end cell, i, j)
    if (lane < (int)B.count) {
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
// and writes both traces back to front into the pair's output slot
// (align.go:126-143, 224-248; align_pair.go:197-216).
// ----------------------------------------------------------------------------
//
// The wavefront walks the matrix ROW-SYNCHRONOUSLY: all lanes are in the same
// row i (bundles are equal-shape, so a row is the same strip / row block / bit
// position in every lane and those indices live in SGPRs); each lane takes its
// left moves inside the row, then its up or diagonal move into row i-1.  Per
// row block (F rows) a lane stages a band of BG column groups of residue words
// around its current column in LDS: the band of the current row block and,
// while the row is the first of its block, the band of the block above.  LDS
// words are lane-interleaved ([word][lane]): any per-lane index is conflict free.

template <int R, int K>
struct BaTraceGeom {
    static constexpr int F = 32 / K;           // rows per residue word
    static constexpr int WPC = R / F;          // row blocks per strip
    static constexpr int BG = F / 4 + 2;       // column groups per staged band
    static constexpr int BW = BG * 4;          // band width in columns
    static constexpr int WORDS = 2 * BW;       // LDS words per lane (two bands)
};

// (BaMoveRec, the record of 16 moves the walk writes, is in device_helpers.h: the bit-parallel walk writes them too)

// The walk only records its moves (BaMoveRec per 16 moves); ba_expand_kernel turns them into
// bytes with one thread per record -- no symbol load, no byte shuffling and no scattered store
// on the walk's critical path (a walk that wrote the bytes itself took 1.7 - 2 ms per mode).
template <int MODE, int R, int K>
__global__ void __launch_bounds__(256) ba_trace_linear_kernel(BaTraceParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    using G = BaTraceGeom<R, K>;
    constexpr int F = G::F, BG = G::BG, BW = G::BW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: the bundle record and all it implies live in SGPRs
    const int waves = blockDim.x >> 6;
    const uint32_t b_idx = P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const bool live = lane < (int)B.count;
    const uint32_t p = live ? P.perm[B.first + lane] : 0u;
    const int n = (int)B.n, m = (int)B.m;
    const int g = P.gap;
    const uint32_t *dir = P.dir + B.dir_off + (size_t)lane * 4;
    const uint64_t o1 = live ? P.s1_off[p] : 0, o2 = live ? P.s2_off[p] : 0;
    const int cgs = (m + 3) / 4;
    const int front = ((n + R - 1) / R) * R - n;
    uint32_t *band = lds + (size_t)wave * (G::WORDS * 64) + lane;  // band[word * 64]

    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    constexpr uint32_t MASK = (1u << K) - 1u;
    const int off = (MODE == BA_LOCAL) ? g : 0;  // an adjacent step is never below -off
    // exact neighbour score from the current exact score and the neighbour's residue
    auto neighbour = [&](int c, uint32_t rho) { return c - ((int)(((uint32_t)c - rho + (uint32_t)off) & MASK) - off); };
    // residue words (4 columns) of row block `rb` (counted over the front-padded rows), column group cg
    auto block_words = [&](int rb, int cg) {
        return *reinterpret_cast<const uint4 *>(dir + ((size_t)rb * cgs + cg) * 256);  // rb = strip * WPC + w
    };

    // end cell, its working-space score `cur`, and the reported score
    int i = 0, j = 0, cur = 0;
    int64_t score = 0;
    if (live) {
        const BaLocalBest bb = P.best[B.first + lane];
        if (MODE == BA_LOCAL) {
            score = bb.score;
            if (bb.score > 0) {
                // The DP kernel names a pair of rows (bb.i, bb.i+1) that holds the maximum.
                // First cell in row-major order: rebuild each row's exact scores left to
                // right from their residues (H(i,0) = 0) until one equals the maximum.
                for (int row = max(bb.i, 1); row <= min(bb.i + 1, n) && j == 0; ++row) {
                    const int t = row - 1 + front;
                    const int rb = t / F, sh = (t % F) * K;
                    int v = 0;
                    // eight column groups per round: the loads of a round are all in flight together
                    for (int cg0 = 0; cg0 < cgs && j == 0; cg0 += 8) {
                        uint4 q[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) q[u] = block_words(rb, min(cg0 + u, cgs - 1));
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int cg = cg0 + u;
                            const uint32_t wd[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const uint32_t rho = (wd[c] >> sh) & MASK;
                                v = v + (int)((rho - (uint32_t)v + (uint32_t)off) & MASK) - off;
                                if (j == 0 && cg * 4 + c < m && v == bb.score) j = cg * 4 + c + 1;
                            }
                        }
                    }
                    i = row;
                }
                cur = bb.score;
            }
        } else {
            // end cell found by the DP kernel: (n, m) for global (align.go:123-124), the
            // leftmost strict maximum of the bottom row for fitting / overlap
            // (align_pair.go:169-182); bb.score is its working-space score Hs = H + g*(i+j)
            i = bb.i;
            j = bb.j;
            cur = bb.score;
            score = (int64_t)cur - (int64_t)g * (i + j);
        }
        P.scores[p] = score;
        if (P.score_only && P.t_off) {  // BIOALIGN_END_CELLS: where the alignment ends, without a traceback
            P.t_off[p] = (uint64_t)i;
            P.t_len[p] = (uint64_t)j;
        }
    }
    if (P.score_only) return;

    const uint64_t slot = o1 + o2;
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    // the record being filled
    // (record 0 of a pair is a header for the expansion: trace length and end, so that it need not chase
    // perm -> t_len / t_off first)
    const BaRecs recs = ba_recs(P.rec, B, lane);
    uint32_t opw = 0, kq = 0, rec_i = 0, rec_j = 0;
    auto record = [&](uint32_t mv) {  // the move leaves cell (i, j)
        if ((kq & 15u) == 0) {
            rec_i = (uint32_t)i;
            rec_j = (uint32_t)j;
        }
        opw |= mv << ((kq & 15u) * 2);
        ++kq;
        if ((kq & 15u) == 0) {
            BaMoveRec rr;
            rr.ops = opw;
            rr.i = rec_i;
            rr.j = rec_j;
            rr.count = 16;
            recs[(kq >> 4) - 1] = rr;
            opw = 0;
        }
    };
    const int b_top = (MODE == BA_FITTING) ? g : 0;   // Hs(0, j) = b_top * j
    const int b_left = (MODE == BA_OVERLAP) ? g : 0;  // Hs(i, 0) = b_left * i

    bool done = !live || (i == 0 && j == 0);
    // band bookkeeping, per lane: which row block each of the two LDS bands holds and its first column
    int tag0 = -2, tag1 = -2, col00 = 0, col01 = 0;  // (group indices start at -1)

    // stage the band of row block rb that ends at the column group of column index `col` (0-based)
    auto load_band = [&](int which, int rb, int col) {
        const int cg_hi = col >> 2;
        uint4 q[BG];
#pragma unroll
        for (int k = 0; k < BG; ++k) {
            // unconditional (column groups left of the matrix are clamped and never looked at): the loads
            // of a band issue back to back and share one wait
            const int cg = max(cg_hi - (BG - 1) + k, 0);
            q[k] = block_words(rb, cg);
        }
        uint32_t *dst = band + (size_t)which * BW * 64;
#pragma unroll
        for (int k = 0; k < BG; ++k) {
            dst[(k * 4 + 0) * 64] = q[k].x;
            dst[(k * 4 + 1) * 64] = q[k].y;
            dst[(k * 4 + 2) * 64] = q[k].z;
            dst[(k * 4 + 3) * 64] = q[k].w;
        }
        return (cg_hi - (BG - 1)) * 4;  // first column of the band
    };

    // rows n .. 1.  The wavefront moves GROUP by group of F rows, group index in an SGPR; inside a group
    // every lane takes one move per round, whatever row it is in, until all lanes have left the group
    // upwards (or are finished).  Rounds per group = the longest stretch of path any lane has inside it.
    // A move out of row t (counted over the front-padded rows) reads rows t and t-1; group g holds the
    // rows t with (t-1) / F == g, so that row t-1 always lies in row block g -- whose band every lane
    // stages together when the wavefront enters the group (one wait per group for all lanes) -- and row t
    // does too, except for the group's bottom row, which is row 0 of block g+1: the band of the
    // previous group, still staged.
    constexpr int LOGF = (F == 16) ? 4 : (F == 8) ? 3 : 2;
    static_assert((1 << LOGF) == F, "F is 4, 8 or 16");
    int i_top = i;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) i_top = max(i_top, __shfl_xor(i_top, d));
    i_top = __builtin_amdgcn_readfirstlane(i_top);  // uniform: the group loop's bookkeeping runs on the scalar unit
    const int g_last = (front - 1) >> LOGF;         // the group that holds row 1 (-1 when there is no padding: row 0 is never read)
    for (int grp = (i_top >= 1) ? ((i_top - 2 + front) >> LOGF) : g_last - 1; grp >= g_last; --grp) {
        const int sel = grp & 1;        // uniform: LDS band of row block grp; the other one holds block grp+1
        const int rbm = max(grp, 0);    // (group -1: its band is never looked at)
        if (!done && i >= 1 && j >= 1 && ((i - 2 + front) >> LOGF) == grp) {
            const int c0 = load_band(sel, rbm, j - 1);
            if (sel) { tag1 = grp; col01 = c0; } else { tag0 = grp; col00 = c0; }
        }
        while (true) {
            const int t = i - 1 + front;  // my row, counted over the front-padded rows
            const bool act = !done && i >= 1 && ((t - 1) >> LOGF) == grp;
            if (!__any(act)) break;
            if (act) {
                const int f = t & (F - 1), fu = (t - 1) & (F - 1);
                if (j == 0) {
                    if (left_free) {
                        done = true;
                    } else {  // left border of global / fitting: forced up move, Hs(i-1, 0) = 0
                        record(BA_MV_UP);
                        --i;
                        cur = 0;
                    }
                } else if (MODE == BA_LOCAL && cur == 0) {
                    done = true;
                } else {
                    const int cl = max(j - 2, 0);  // leftmost column index (0-based) this step reads
                    int tagm = sel ? tag1 : tag0, colm = sel ? col01 : col00;
                    if (tagm != grp || cl < colm) {
                        colm = load_band(sel, rbm, j - 1);
                        tagm = grp;
                        if (sel) { tag1 = tagm; col01 = colm; } else { tag0 = tagm; col00 = colm; }
                    }
                    const uint32_t *bm = band + (size_t)sel * BW * 64;
                    const uint32_t A = bm[(j - 1 - colm) * 64];
                    const uint32_t Bw = bm[(cl - colm) * 64];
                    const uint32_t r_up = (A >> (fu * K)) & MASK;
                    const uint32_t r_diag = (Bw >> (fu * K)) & MASK;
                    uint32_t r_left;
                    if (f == 0) {  // my row is row 0 of row block grp+1
                        int taga = sel ? tag0 : tag1, cola = sel ? col00 : col01;
                        if (taga != grp + 1 || cl < cola) {
                            cola = load_band(sel ^ 1, grp + 1, j - 1);
                            taga = grp + 1;
                            if (sel) { tag0 = taga; col00 = cola; } else { tag1 = taga; col01 = cola; }
                        }
                        r_left = band[((size_t)(sel ^ 1) * BW + (cl - cola)) * 64] & MASK;
                    } else {
                        r_left = (Bw >> (f * K)) & MASK;
                    }
                    int mv;  // 0 up, 1 left, 2 diagonal
                    if (MODE == BA_LOCAL) {
                        // cur = max(0, up - g, left - g, diag + w) > 0 here, skip-prefix first, then up > left >
                        // diagonal (align.go:186-202): the cell came through the diagonal iff it is above both
                        // up - g and left - g.  Exact neighbours follow from the residues; no score table needed.
                        const int up = (i == 1) ? 0 : neighbour(cur, r_up);
                        const int left = (j == 1) ? 0 : neighbour(cur, r_left);
                        const int diag = (i == 1 || j == 1) ? 0 : neighbour(up, r_diag);
                        const bool from_up = cur == up - g, from_left = cur == left - g;
                        mv = (!from_up && !from_left) ? 2 : (from_left && !from_up) ? 1 : 0;
                        cur = mv == 2 ? diag : mv == 1 ? left : up;
                    } else {
                        // Biased space: cur = max3(up, left, diag + w) with 0 <= cur - up, cur - left < 2^K, so
                        //   diagonal <=> cur > up and cur > left;  left <=> cur == left and cur > up;  else up
                        // (align.go:105-116) -- and "cur > x" is "residues differ".  No score, no table, no
                        // symbol on the decision path; `cur` only carries the current cell's residue.
                        const uint32_t rc = (uint32_t)cur & MASK;
                        const uint32_t ru = (i == 1) ? (uint32_t)(b_top * j) & MASK : r_up;
                        const uint32_t rl = (j == 1) ? (uint32_t)(b_left * i) & MASK : r_left;
                        const uint32_t rd = (i == 1) ? (uint32_t)(b_top * (j - 1)) & MASK
                                                     : (j == 1) ? (uint32_t)(b_left * (i - 1)) & MASK : r_diag;
                        const bool above_up = rc != ru, above_left = rc != rl;
                        mv = (above_up && above_left) ? 2 : (above_up && !above_left) ? 1 : 0;
                        cur = (int)(mv == 2 ? rd : mv == 1 ? rl : ru);
                    }
                    record((uint32_t)mv);
                    i -= (mv != 1);
                    j -= (mv != 0);
                }
            }
        }
    }
    // row 0: global / overlap walk left along the top border; local / fitting stop there
    if (!done && !top_free) {
        while (j > 0) {
            record(BA_MV_LEFT);
            --j;
        }
    }
    if (!live) return;
    if (kq & 15u) {
        BaMoveRec rr;
        rr.ops = opw;
        rr.i = rec_i;
        rr.j = rec_j;
        rr.count = kq & 15u;
        recs[kq >> 4] = rr;
    }
    // Where the trace ends inside the pair's slot: at the slot's end, or -- whenever the slot
    // has the room -- at the 16-byte boundary just below it, so that every full record of
    // the expansion is one aligned 16-byte store.
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

// Expansion of the walk's records into the two gapped byte strings (align.go:126-143, 224-248;
// align_pair.go:197-216).  One block per bundle, 16 threads per pair, one record (16 trace
// columns) per thread and round: the symbols a record needs are two short runs of s1 and s2.
#ifndef BA_EXPAND_TPP
#define BA_EXPAND_TPP 4  // threads per pair (2 / 4 / 8 / 16: 0.36 / 0.32 / 0.35 / 0.51 ms per 1M pairs of 150 bp: more, smaller blocks in flight per CU)
#endif
__global__ void __launch_bounds__(64 * BA_EXPAND_TPP) ba_expand_kernel(BaTraceParams P)
{
    // per thread: the (at most 16, unaligned) bytes of s1 and of s2 its record consumes, as 5 aligned dwords each
    __shared__ uint32_t stage[64 * BA_EXPAND_TPP * 10];
    const uint32_t b_idx = P.bundle_begin + blockIdx.x;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const uint32_t q = threadIdx.x / BA_EXPAND_TPP, g16 = threadIdx.x % BA_EXPAND_TPP;
    if (q >= B.count) return;
    const BaRecs recs = ba_recs(P.rec, B, q);
    const BaMoveRec hd = recs[-1];
    const uint32_t p = hd.count;
    const uint64_t len = hd.ops;
    const bool dense = P.dense_end != nullptr;
    const uint64_t end = dense ? P.dense_end[p] : ((uint64_t)hd.i | ((uint64_t)hd.j << 32));
    const uint8_t *s1 = P.s1 + P.s1_off[p];
    const uint8_t *s2 = P.s2 + P.s2_off[p];
    const uint32_t nrec = (uint32_t)((len + 15) / 16);
    const uint32_t gs = P.gap_symbol;
    const bool al1 = (((uint64_t)P.t1 + end) & 15u) == 0, al2 = (((uint64_t)P.t2 + end) & 15u) == 0;
    uint32_t *st1 = stage + threadIdx.x * 5, *st2 = stage + 64 * BA_EXPAND_TPP * 5 + threadIdx.x * 5;  // stride 5: no bank conflicts
    // stages src[lo .. lo+15] (clamped to the sequence's own dwords); returns the staged bytes, index = position - lo + shift
    auto stage_in = [&](uint32_t *dst, const uint8_t *src, uint32_t lo, uint32_t seq_len) -> const uint8_t * {
        const uint64_t a = (uint64_t)(src + lo), a0 = a & ~(uint64_t)3;
        const uint64_t last = ((uint64_t)src + seq_len - 1) & ~(uint64_t)3;
        uint32_t v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = *reinterpret_cast<const uint32_t *>(min(a0 + 4 * k, last));
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[k] = v[k];
        return reinterpret_cast<const uint8_t *>(dst) + (a - a0);
    };
    for (uint32_t r = g16; r < nrec; r += BA_EXPAND_TPP) {
        const BaMoveRec rr = recs[r];
        const uint32_t valid = rr.count >= 16 ? 0xffffffffu : ((1u << (2 * rr.count)) - 1u);
        const uint32_t lo = rr.ops & 0x55555555u, hi = (rr.ops >> 1) & 0x55555555u;
        const uint32_t is_left = lo & ~hi & valid, is_up = ~lo & ~hi & 0x55555555u & valid;
        const uint32_t c1 = rr.count - __popc(is_left), c2 = rr.count - __popc(is_up);  // symbols consumed
        // the record consumes s1[i - c1 .. i - 1] and s2[j - c2 .. j - 1] (0-based), from the top down
        const uint8_t *b1 = reinterpret_cast<const uint8_t *>(st1), *b2 = reinterpret_cast<const uint8_t *>(st2);
        if (c1) b1 = stage_in(st1, s1, rr.i - c1, B.n);
        if (c2) b2 = stage_in(st2, s2, rr.j - c2, B.m);
        uint32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        // codes beyond rr.count are 0 (= up): they only ever produce bytes that are not stored
        int u1 = (int)c1, u2 = (int)c2;  // symbols of the staged runs not yet used, taken from the top down
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t mv = (rr.ops >> (2 * k)) & 3u;
            u1 -= (mv != BA_MV_LEFT);
            u2 -= (mv != BA_MV_UP);
            const uint32_t xs = b1[max(u1, 0)], ys = b2[max(u2, 0)];
            const uint32_t x = mv == BA_MV_LEFT ? gs : xs;
            const uint32_t y = mv == BA_MV_UP ? gs : ys;
            // code k is trace column end - 16r - 1 - k: byte 15 - k of the record's 16-byte chunk
            a[(15 - k) >> 2] |= x << (((15 - k) & 3) * 8);
            b[(15 - k) >> 2] |= y << (((15 - k) & 3) * 8);
        }
        const uint64_t chunk = end - 16ull * (r + 1);
        if (rr.count == 16 && (al1 || dense)) {
            // (dense layout: 16-byte stores at whatever alignment the trace's end has -- global memory takes them)
            const uint4 v = make_uint4(a[0], a[1], a[2], a[3]);
            __builtin_memcpy(P.t1 + chunk, &v, 16);
        } else {
            for (uint32_t k = 0; k < rr.count; ++k) P.t1[chunk + 15 - k] = (uint8_t)(a[(15 - k) >> 2] >> (((15 - k) & 3) * 8));
        }
        if (rr.count == 16 && (al2 || dense)) {
            const uint4 v = make_uint4(b[0], b[1], b[2], b[3]);
            __builtin_memcpy(P.t2 + chunk, &v, 16);
        } else {
            for (uint32_t k = 0; k < rr.count; ++k) P.t2[chunk + 15 - k] = (uint8_t)(b[(15 - k) >> 2] >> (((15 - k) & 3) * 8));
        }
    }
}

// The expansion for short pairs (the common case; the kernel above stays for bundles whose sequences do not fit the
// LDS).  One block per bundle, eight threads per pair as above -- so that the records of a pair are read, and its
// chunks of the traces written, as 128 contiguous bytes -- but everything whose address follows from the bundle
// record alone is asked for at once: the bundle's packed sequences (the pack kernel's lane-interleaved words:
// coalesced, no chase through perm -> s1_off -> bytes), the pairs' headers and each thread's first records.  The
// sequences are staged in LDS one pair per row with an odd row stride, so that a symbol is one ds_read_u8 at
// `row + index` and threads -- different rows, or the same row sixteen symbols apart -- hardly ever meet in a bank
// (the per-thread staging of the kernel above conflicted on a fifth of its reads, and its blocks went through four
// dependent round trips to memory).
#ifndef BA_EXPAND_PRE
#define BA_EXPAND_PRE 3  // records per thread loaded ahead (3 * 4 * 16 trace columns: pairs of 96 + 96 in one go)
#endif
extern "C" size_t ba_expand_bundle_lds_bytes(int max_n, int max_m, int R)
{
    const size_t w1 = (size_t)((max_n + R - 1) / R) * R / 4, w2 = (size_t)(max_m + 3) / 4;
    return 64 * ((w1 | 1) + (w2 | 1)) * 4 + 16;  // (+16: a record that consumes no symbol still looks at the byte behind the run)
}

__global__ void __launch_bounds__(64 * BA_EXPAND_TPP) ba_expand_bundle_kernel(BaTraceParams P)
{
    extern __shared__ __align__(16) uint32_t xs[];
    const uint32_t b_idx = P.bundle_begin + blockIdx.x;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const uint32_t q = threadIdx.x / BA_EXPAND_TPP, g16 = threadIdx.x % BA_EXPAND_TPP;
    const int R = P.rows_per_strip;
    const int n = (int)B.n, m = (int)B.m;
    const int strips = (n + R - 1) / R;
    const int front = strips * R - n;
    const int words1 = strips * R / 4, words2 = (m + 3) / 4;
    const int stride1 = words1 | 1, stride2 = words2 | 1;  // dwords per pair, odd
    const bool live = q < B.count;
    const BaRecs recs = ba_recs(P.rec, B, q);
    BaMoveRec hd;
    hd.ops = hd.i = hd.j = hd.count = 0;
    BaMoveRec pre[BA_EXPAND_PRE];
    if (live) {
        hd = recs[-1];
#pragma unroll
        for (int k = 0; k < BA_EXPAND_PRE; ++k)
            pre[k] = recs[min((uint32_t)(g16 + k * BA_EXPAND_TPP), B.rec_cap - 2)];
    }
    {
        const uint32_t *s1w = P.s1w + B.s1w_off, *s2w = P.s2w + B.s2w_off;
        uint32_t *x1 = xs, *x2 = xs + 64 * stride1;
        for (int idx = threadIdx.x; idx < words1 * 64; idx += 64 * BA_EXPAND_TPP) x1[(idx & 63) * stride1 + (idx >> 6)] = s1w[idx];
        for (int idx = threadIdx.x; idx < words2 * 64; idx += 64 * BA_EXPAND_TPP) x2[(idx & 63) * stride2 + (idx >> 6)] = s2w[idx];
    }
    __syncthreads();
    if (!live) return;
    const uint32_t p = hd.count;
    const uint64_t len = hd.ops;
    const bool dense = P.dense_end != nullptr;
    const uint64_t end = dense ? P.dense_end[p] : ((uint64_t)hd.i | ((uint64_t)hd.j << 32));
    const uint32_t nrec = (uint32_t)((len + 15) / 16);
    const uint32_t gs = P.gap_symbol;
    const bool al1 = (((uint64_t)P.t1 + end) & 15u) == 0, al2 = (((uint64_t)P.t2 + end) & 15u) == 0;
    const uint8_t *row1 = reinterpret_cast<const uint8_t *>(xs + q * stride1) + front;  // row1[x] = s1[x]
    const uint8_t *row2 = reinterpret_cast<const uint8_t *>(xs + 64 * stride1 + q * stride2);
    auto expand = [&](const BaMoveRec &rr, uint32_t r) {
        const uint32_t valid = rr.count >= 16 ? 0xffffffffu : ((1u << (2 * rr.count)) - 1u);
        const uint32_t lo = rr.ops & 0x55555555u, hi = (rr.ops >> 1) & 0x55555555u;
        const uint32_t is_left = lo & ~hi & valid, is_up = ~lo & ~hi & 0x55555555u & valid;
        const uint32_t c1 = rr.count - __popc(is_left), c2 = rr.count - __popc(is_up);  // symbols consumed
        // the record consumes s1[i - c1 .. i - 1] and s2[j - c2 .. j - 1] (0-based), from the top down
        const uint8_t *b1 = row1 + ((int)rr.i - (int)c1), *b2 = row2 + ((int)rr.j - (int)c2);
        uint32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        // codes beyond rr.count are 0 (= up): they only ever produce bytes that are not stored
        int u1 = (int)c1, u2 = (int)c2;  // symbols of the runs not yet used, taken from the top down
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t mv = (rr.ops >> (2 * k)) & 3u;
            u1 -= (mv != BA_MV_LEFT);
            u2 -= (mv != BA_MV_UP);
            const uint32_t xq = b1[max(u1, 0)], yq = b2[max(u2, 0)];
            const uint32_t x = mv == BA_MV_LEFT ? gs : xq;
            const uint32_t y = mv == BA_MV_UP ? gs : yq;
            // code k is trace column end - 16r - 1 - k: byte 15 - k of the record's 16-byte chunk
            a[(15 - k) >> 2] |= x << (((15 - k) & 3) * 8);
            b[(15 - k) >> 2] |= y << (((15 - k) & 3) * 8);
        }
        const uint64_t chunk = end - 16ull * (r + 1);
        // A full record is one 16-byte store.  The last record of a trace holds rr.count < 16 moves, the top rr.count bytes of
        // its chunk: whole dwords first, then at most three single bytes (a byte per move took 19 of the 25 store
        // instructions of a wavefront and half of the kernel's time: 1 byte per lane and instruction).
        auto put = [&](uint8_t *t, const uint32_t (&v)[4], bool aligned) {
            if (rr.count == 16 && (aligned || dense)) {
                // (dense layout: 16-byte stores at whatever alignment the trace's end has -- global memory takes them)
                const uint4 q4 = make_uint4(v[0], v[1], v[2], v[3]);
                __builtin_memcpy(t + chunk, &q4, 16);
            } else if (aligned || dense) {
                const uint32_t nd = rr.count >> 2, rem = rr.count & 3u;
#pragma unroll
                for (int d = 3; d >= 0; --d)
                    if ((uint32_t)(3 - d) < nd) __builtin_memcpy(t + chunk + 4 * d, &v[d], 4);
                if (rem) {
                    const uint32_t d = 3u - nd;  // the dword the remaining moves lie in, from its top byte down
                    const uint32_t word = d == 3 ? v[3] : d == 2 ? v[2] : d == 1 ? v[1] : v[0];
                    for (uint32_t k = 0; k < rem; ++k) t[chunk + 4 * d + 3 - k] = (uint8_t)(word >> ((3 - k) * 8));
                }
            } else {
                for (uint32_t k = 0; k < rr.count; ++k) {
                    const uint32_t d = (15 - k) >> 2;
                    const uint32_t word = d == 3 ? v[3] : d == 2 ? v[2] : d == 1 ? v[1] : v[0];
                    t[chunk + 15 - k] = (uint8_t)(word >> (((15 - k) & 3) * 8));
                }
            }
        };
        put(P.t1, a, al1);
        put(P.t2, b, al2);
    };
#pragma unroll
    for (int k = 0; k < BA_EXPAND_PRE; ++k) {
        const uint32_t r = g16 + (uint32_t)k * BA_EXPAND_TPP;
        if (r < nrec) expand(pre[k], r);
    }
    for (uint32_t r = g16 + BA_EXPAND_PRE * BA_EXPAND_TPP; r < nrec; r += BA_EXPAND_TPP) expand(recs[r], r);
}

// ----------------------------------------------------------------------------
// host-side launchers (called from api.cpp)
// ----------------------------------------------------------------------------
extern "C" size_t ba_dp_linear_lds_bytes(int nb, int nb4, int na_pad, int R, int waves)
{
    const int wt_words = nb4 * na_pad;
    const int wt_aligned = (wt_words + 3) & ~3;
    return (size_t)(wt_aligned + 128) * 4 + (size_t)waves * nb * (R / 16) * 64 * 16;
}

extern "C" hipError_t ba_launch_pack(const BaPackParams &P, hipStream_t st)
{
    const uint32_t nb = P.bundle_end - P.bundle_begin;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(ba_pack_kernel, dim3((nb + 3) / 4), dim3(256), 0, st, P);
    return hipGetLastError();
}

template <int MODE, int R, int K>
static hipError_t launch_dp(const BaDpParams &P, int waves, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + waves - 1) / waves;
    const size_t lds = ba_dp_linear_lds_bytes(P.nb, P.nb4, P.na_pad, R, waves);
    if (lds > 64 * 1024) {  // large alphabets: more than 64 KB of dynamic LDS has to be asked for
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_dp_linear_kernel<MODE, R, K>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ae != hipSuccess) return ae;
    }
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
    (void)na;
    (void)nb;
    const int F = 32 / K;
    const int words = 2 * (F / 4 + 2) * 4;  // BaTraceGeom::WORDS
    return ((size_t)waves * words * 64) * 4;
}

// Dense trace layout: exclusive / inclusive prefix sums of the trace lengths in pair order.
//   ba_len_partial: block sums over 2048 lengths;  (one-block scan of the sums: kernels_fasta.hip's fa_scan1)
//   ba_len_apply:   t_off[p] = sum of the lengths before p, dense_end[p] = t_off[p] + len[p]
__global__ void __launch_bounds__(256) ba_len_partial_kernel(const uint64_t *len, uint64_t n, uint64_t *part)
{
    __shared__ uint64_t sa[4];
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    uint64_t a = 0;
    for (int k = 0; k < 8; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) a += len[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3];
}

__global__ void __launch_bounds__(256) ba_len_apply_kernel(const uint64_t *len, uint64_t n, const uint64_t *part,
                                                           uint64_t *t_off, uint64_t *dense_end)
{
    __shared__ uint64_t wa[4];
    __shared__ uint64_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    if (threadIdx.x == 0) carry = part[blockIdx.x];  // scanned: the lengths before this block
    __syncthreads();
    for (int k = 0; k < 8; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        const uint64_t v = i < n ? len[i] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wa[wave] = inc;
        __syncthreads();
        uint64_t pre = carry, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) pre += wa[w];
            tot += wa[w];
        }
        if (i < n) {
            t_off[i] = pre + inc - v;
            dense_end[i] = pre + inc;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
}

extern "C" hipError_t ba_launch_len_partial(const uint64_t *len, uint64_t n, uint64_t *part, hipStream_t st)
{
    hipLaunchKernelGGL(ba_len_partial_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st, len, n, part);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_len_apply(const uint64_t *len, uint64_t n, const uint64_t *part, uint64_t *t_off,
                                          uint64_t *dense_end, hipStream_t st)
{
    hipLaunchKernelGGL(ba_len_apply_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st, len, n, part, t_off, dense_end);
    return hipGetLastError();
}

// Compact trace output (BIOALIGN_TRACE_OPS): per pair a record of 8 + 4 * ceil(len / 16) bytes
// -- the end cell (i, j) as two uint32, then the walk's 2-bit move codes, 16 to a dword, first move = last column
// of the alignment -- laid back to back in pair order.  ba_ops_sizes: record sizes from the trace lengths (the
// prefix sums are ba_len_partial / fa_scan1 / ba_len_apply); ba_ops_emit: the records themselves.
__global__ void __launch_bounds__(256) ba_ops_sizes_kernel(const uint64_t *t_len, uint64_t n, uint64_t *sizes)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) sizes[i] = 8 + 4 * ((t_len[i] + 15) / 16);
}

__global__ void __launch_bounds__(64 * BA_EXPAND_TPP) ba_ops_emit_kernel(BaTraceParams P)
{
    const uint32_t b_idx = P.bundle_begin + blockIdx.x;
    if (b_idx >= P.bundle_end) return;
    const BaBundle B = P.bundles[b_idx];
    const uint32_t q = threadIdx.x / BA_EXPAND_TPP, g16 = threadIdx.x % BA_EXPAND_TPP;
    if (q >= B.count) return;
    const BaRecs recs = ba_recs(P.rec, B, q);
    const BaMoveRec hd = recs[-1];
    const uint32_t p = hd.count;
    const uint32_t nrec = (uint32_t)(((uint64_t)hd.ops + 15) / 16);
    uint32_t *out = reinterpret_cast<uint32_t *>(P.t1 + P.t_off[p]);  // record offsets are multiples of 4
    if (g16 == 0) {
        out[0] = nrec ? recs[0].i : 0u;  // the cell the first move leaves: the end cell of the alignment
        out[1] = nrec ? recs[0].j : 0u;
    }
    for (uint32_t r = g16; r < nrec; r += BA_EXPAND_TPP) out[2 + r] = recs[r].ops;
}

extern "C" hipError_t ba_launch_ops_sizes(const uint64_t *t_len, uint64_t n, uint64_t *sizes, hipStream_t st)
{
    hipLaunchKernelGGL(ba_ops_sizes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t_len, n, sizes);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_ops_emit(const BaTraceParams &P, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    if (nbundles == 0) return hipSuccess;
    hipLaunchKernelGGL(ba_ops_emit_kernel, dim3(nbundles), dim3(64 * BA_EXPAND_TPP), 0, st, P);
    return hipGetLastError();
}

enum { BA_TRACE_WALK = 1, BA_TRACE_EXPAND = 2 };

template <int MODE, int R, int K>
static hipError_t launch_trace(const BaTraceParams &P, int phases, hipStream_t st)
{
    const int waves = 4;
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + waves - 1) / waves;
    const size_t lds = ba_trace_linear_lds_bytes(P.na, P.nb, R, K, waves);
    if (!P.score_only && !P.rec) return hipErrorInvalidValue;  // the walk records its moves for the expansion
    hipError_t e = hipSuccess;
    if (phases & BA_TRACE_WALK) {
        hipLaunchKernelGGL((ba_trace_linear_kernel<MODE, R, K>), dim3(blocks), dim3(waves * 64), lds, st, P);
        e = hipGetLastError();
    }
    if (e != hipSuccess || P.score_only || !(phases & BA_TRACE_EXPAND)) return e;
    // short pairs: the bundle's sequences staged in LDS (8 blocks per CU at 20 KB); otherwise thread by thread
    const size_t xl = ba_expand_bundle_lds_bytes(P.max_n, P.max_m, R);
    if (P.max_n > 0 && xl <= 40 * 1024 && P.s1w && P.s2w)
        hipLaunchKernelGGL(ba_expand_bundle_kernel, dim3(nbundles), dim3(64 * BA_EXPAND_TPP), xl, st, P);
    else
        hipLaunchKernelGGL(ba_expand_kernel, dim3(nbundles), dim3(64 * BA_EXPAND_TPP), 0, st, P);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_trace_mode(const BaTraceParams &P, int R, int K, int phases, hipStream_t st)
{
    if (R == 32 && K == 2) return launch_trace<MODE, 32, 2>(P, phases, st);
    if (R == 32 && K == 4) return launch_trace<MODE, 32, 4>(P, phases, st);
    if (R == 16 && K == 2) return launch_trace<MODE, 16, 2>(P, phases, st);
    if (R == 16 && K == 4) return launch_trace<MODE, 16, 4>(P, phases, st);
    if (R == 16 && K == 8) return launch_trace<MODE, 16, 8>(P, phases, st);
    return hipErrorInvalidValue;
}

// phases: 1 = the walk (scores, move records, lengths), 2 = the expansion into bytes, 3 = both
extern "C" hipError_t ba_launch_trace_linear(const BaTraceParams &P, int mode, int R, int K, int phases, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_trace_mode<BA_GLOBAL>(P, R, K, phases, st);
    case BA_LOCAL: return launch_trace_mode<BA_LOCAL>(P, R, K, phases, st);
    case BA_FITTING: return launch_trace_mode<BA_FITTING>(P, R, K, phases, st);
    case BA_OVERLAP: return launch_trace_mode<BA_OVERLAP>(P, R, K, phases, st);
    }
    return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------
// All pairs of one list (bioalign_all_pairs): where the two sequences of pair c0 + k lie in the list.
// Pair p = i(i-1)/2 + j, j < i, row by row (or every ordered pair i != j); q1[k] = off[i] (row sequence),
// q2[k] = off[j] (column sequence).
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ba_pair_offsets_kernel(const uint64_t *__restrict__ off, uint64_t c0, uint64_t n,
                                                              uint64_t n_seqs, int ordered, uint64_t *__restrict__ q1,
                                                              uint64_t *__restrict__ q2)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint64_t p = c0 + k;
    if (ordered) {  // every (i, j), i != j, row by row: p = i (n_seqs - 1) + r, j = r skipping i
        const uint64_t i = p / (n_seqs - 1), r = p % (n_seqs - 1);
        q1[k] = off[i];
        q2[k] = off[r < i ? r : r + 1];
        return;
    }
    // the row from the square root, then exact in integers (p < 2^62: the double is off by at most a few rows)
    uint64_t r = (uint64_t)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
    if (r < 1) r = 1;
    for (int t = 0; t < 8 && r > 1 && r * (r - 1) / 2 > p; ++t) --r;
    for (int t = 0; t < 8 && (r + 1) * r / 2 <= p; ++t) ++r;
    q1[k] = off[r];
    q2[k] = off[p - r * (r - 1) / 2];
}

extern "C" hipError_t ba_launch_pair_offsets(const uint64_t *off, uint64_t c0, uint64_t n, uint64_t n_seqs, int ordered,
                                             uint64_t *q1, uint64_t *q2, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ba_pair_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, off, c0, n, n_seqs,
                       ordered, q1, q2);
    return hipGetLastError();
}

// out[k] = k: the pair order of a batch whose pairs all have one shape (nothing to sort)
__global__ void __launch_bounds__(256) ba_iota_kernel(uint32_t *__restrict__ out, uint64_t n)
{
    const uint64_t k = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (k + 4 <= n) {
        *reinterpret_cast<uint4 *>(out + k) = make_uint4((uint32_t)k, (uint32_t)k + 1, (uint32_t)k + 2, (uint32_t)k + 3);
    } else {
        for (uint64_t q = k; q < n; ++q) out[q] = (uint32_t)q;
    }
}

extern "C" hipError_t ba_launch_iota(uint32_t *out, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ba_iota_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, out, n);
    return hipGetLastError();
}
