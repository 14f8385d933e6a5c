// kernels_affine.hip -- gfx950 kernels of the batched pairwise aligner, affine gaps
// (three-level Gotoh), four modes.
//
// Recurrences (reference align_affine.go:44-103, evaluated in integers; borders and end
// cells of the four modes follow align_pair.go:109-195 on the middle level, as the
// oracle's bo_align_affine restates them):
//   E(i,j) = max(E(i,j-1) - ext, H(i,j-1) - open)      gap in s1, extension wins ties (51-57)
//   F(i,j) = max(F(i-1,j) - ext, H(i-1,j) - open)      gap in s2, extension wins ties (67-75)
//   H(i,j) = first maximum of [0 (local only)], F, H(i-1,j-1) + w, E      strict '>' (82-100)
// so a gap of length L costs open + (L-1)*ext.
//
// Working score space.  Every level runs biased by ext*(i+j): X' = X + ext*(i+j).  Both
// extension moves then cost nothing, opening costs c = open - ext and the diagonal costs
// w + 2*ext:
//   E' = max(E'(left), H'(left) - c)     F' = max(F'(up), H'(up) - c)
//   H' = max3(E', F', H'(diag) + w + 2*ext)
// All candidates of one decision carry the same bias, so every comparison -- and every
// back-pointer -- is the reference's.  A strip keeps hq = H' - c and E' per row in VGPRs
// (the profile byte holds w + 2*ext + c so that the diagonal add works on hq directly).
// The true zero of local mode is ext*(i+j) in this space: wave-uniform, an SGPR operand.
//
// Traceback matrix: four decision bits per cell, each a comparison a > b of two scores that
// can only be a >= b:
//   nF = [F' != H']   nD = [diag + w != H']   oE = [E' opened from H]   oF = [F' opened from H]
// On gfx950 a sign push (v_sub_u32 + v_alignbit_b32) costs three issue units, the alignbit being
// half-rate.  The scores therefore live as the BIT PATTERNS OF FLOATS: x is kept as
// 0x4B400000 + x, which read as an f32 is exactly 2^23 + 2^22 + x for |x| < 2^22 (the planner
// proves that range).  Integer adds and maxima work on the patterns unchanged, and
//   bit = v_sub_f32(a, b) clamp      (1.0 if a > b, 0.0 if a == b: the clamp saturates at 1)
//   plane = v_fma_f32(plane, 2.0, bit)
// are two full-rate instructions.  A plane starts at 2^7, so after the 16 rows of a column its
// pattern is 0x4B000000 | the 16 bits, which one v_perm_b32 moves into the decision word.
// Middle-level back-pointer = G2 (close the gap in s2) if !nF, else diagonal if !nD, else
// G1 -- exactly the candidate order above.  (Local: a cell whose true score is 0 is a
// skip-prefix cell whatever its bits say; the traceback tracks the true score.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "device_helpers.h"

#define BA_FBIAS 0x4B400000  // score x is held as BA_FBIAS + x: the f32 with this pattern is 2^23 + 2^22 + x
#define BA_NEG (BA_FBIAS - (1 << 22))  // minus infinity of the E / F levels (align_affine.go:36-37, 45-46, 63-64): below every score the planner admits
#ifndef BA_AFF_WPE
#define BA_AFF_WPE 3  // wavefronts per SIMD the register allocator is asked to leave room for (32-row strips, not local: 168 VGPRs)
#endif

// Signed byte k of w through v_bfe_i32, kept out of the compiler's hands: it would fold the extraction into
// v_add_u32_sdwa, and among the full-rate float instructions of this cell an SDWA add stalls the pipeline far beyond
// its own half-rate slot (config 4: 2562 -> 2923 GCUPS with one instruction more per cell).
__device__ __forceinline__ int ba_sbyte_bfe(uint32_t w, int k)
{
    int r = __builtin_amdgcn_sbfe(w, 8 * k, 8);
    asm("" : "+v"(r));
    return r;
}
// 1.0f if a > b, 0.0f if a == b (a >= b; both are score patterns, see the header)
__device__ __forceinline__ float ba_gt_bit(int a, int b)
{
    float r;
    asm("v_sub_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 16 decision bits gathered in a float plane (pattern 0x4B000000 | bits) go into the bottom of a decision word
__device__ __forceinline__ uint32_t ba_push16(uint32_t acc, float plane)
{
    return __builtin_amdgcn_perm(acc, __float_as_uint(plane), 0x05040100u);
}

// One column of one strip.  `top_hq`/`f` enter as hq and F' of the row above the strip.
// Local mode clamps H' from below at its zero level ext*(i+j), a running value that steps
// by ext per row, and folds the best-cell search into a subtract, a v_lshl_add and half a
// v_max3 per cell:  key = (true score << key_shift) | (first row of its pair ? 1 : 0) << jbits | (jmax - j),
// so the largest key of a row pair is its best cell, first in row-major order.
// (klow0 / klow1: the low key bits for the even / odd rows of this column.)
// NIB: pw holds 4-bit scores (row r = nibble r); the diagonal is hq + nibble + kappa (v_bfe_u32 + v_add3_u32).
// FKEY (local): the key is built as a float -- v_sub_f32 + v_fma_f32, full-rate, instead of a subtract and a half-rate
// v_lshl_add -- when (score + 1) << key_shift stays below 2^24 (the planner decides); keys are non-negative floats, whose
// patterns order like integers.
template <int MODE, int R, bool NIB, bool FKEY>
__device__ __forceinline__ void ba_affine_column(int (&hq)[R], int (&e)[R], uint32_t (&acc)[4], int (&rkey)[R / 2],
                                                 const uint32_t (&pw)[NIB ? R / 8 : R / 4], int top_hq, int &f,
                                                 int &topdiag, int c, int ge, int fl, int key_shift, int klow0,
                                                 int klow1, int kappa)
{
    int up = top_hq, diag = topdiag;
    topdiag = top_hq;
    int kprev = 0;
    float pl[4];
    const float kscale = (float)(1 << key_shift), klow0f = (float)klow0, klow1f = (float)klow1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int eo = e[r];
        // Local: the zero level of row r, ext*(i+j), enters through E' -- en = max(E', zero) never decides anything it
        // should not: a gap level is only looked at by a traceback that came from a cell above its zero level, and
        // E' of every cell of that gap is then above the (lower or equal) zero levels to its left.
        if (MODE == BA_LOCAL) fl += ge;
        const int en = MODE == BA_LOCAL ? ba_max3(eo, hq[r], fl) : max(eo, hq[r]);
        const int fn = max(f, up);
        const int d = NIB ? diag + (int)__builtin_amdgcn_ubfe(pw[NIB ? r >> 3 : 0], 4 * (r & 7), 4) + kappa
                          : diag + ba_sbyte_bfe(pw[NIB ? 0 : r >> 2], r & 3);
        const int H = ba_max3(en, fn, d);
        const float b0 = ba_gt_bit(H, fn), b1 = ba_gt_bit(H, d), b2 = ba_gt_bit(en, eo), b3 = ba_gt_bit(fn, f);
        if ((r & 15) == 0) {  // 2^7 * 2 + bit
            pl[0] = b0 + 256.0f;
            pl[1] = b1 + 256.0f;
            pl[2] = b2 + 256.0f;
            pl[3] = b3 + 256.0f;
        } else {
            pl[0] = __builtin_fmaf(pl[0], 2.0f, b0);
            pl[1] = __builtin_fmaf(pl[1], 2.0f, b1);
            pl[2] = __builtin_fmaf(pl[2], 2.0f, b2);
            pl[3] = __builtin_fmaf(pl[3], 2.0f, b3);
        }
        if ((r & 15) == 15) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = ba_push16(acc[k], pl[k]);
        }
        if (MODE == BA_LOCAL) {
            int key;
            if (FKEY) {
                // (volatile: left to itself the scheduler gathers these at the end of the column and runs out of registers)
                float sc;
                asm volatile("v_subrev_f32 %0, %1, %2" : "=v"(sc) : "s"(fl), "v"(H));  // (the zero level is wave-uniform: an SGPR)
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(key) : "v"(sc), "v"(kscale), "v"((r & 1) ? klow1f : klow0f));
            } else
                key = (int)(((uint32_t)(H - fl) << key_shift) + (uint32_t)((r & 1) ? klow1 : klow0));
            if (r & 1)
                rkey[r / 2] = ba_max3(rkey[r / 2], kprev, key);
            else
                kprev = key;
        }
        const int hn = H - c;
        diag = hq[r];
        hq[r] = hn;
        e[r] = en;
        f = fn;
        up = hn;
    }
}

// Layout of the decision words of one bundle (dwords):
//   [strip][column word cw = (j-1) / CPW][lane][plane 0..3]       CPW = 32 / R columns per word
// bit 31 - ((j-1) % CPW * R + r) of a word belongs to row r of the strip.  The four planes
// of one (strip, column word, lane) are one aligned uint4.
//
// Two launch shapes:
//   PIPE == false : one wavefront per bundle, strips one after the other (large batches:
//                   the bundles alone fill the chip);
//   PIPE == true  : one BLOCK per bundle; wave w takes strips w, w+W, w+2W, ... and the W
//                   strips in flight form a pipeline over the columns: a strip may compute
//                   a group of 4 columns once the strip above has handed over that group's
//                   boundary row (written to global memory, announced through a progress
//                   word in LDS).  W times as many wavefronts for batches of few, long pairs.
//
// NIB: 4-bit strip profile (scores w - min(w), top-free modes, R == 16): an alphabet of 20 column symbols
// takes 10 KB of LDS per wavefront instead of 20, which doubles the wavefronts a CU holds.
template <int MODE, int R, bool PIPE, bool NIB, bool FKEY = false>
__global__ void __launch_bounds__(PIPE ? (R == 16 ? 1024 : (MODE == BA_LOCAL ? 512 : 768)) : 256)
    __attribute__((amdgpu_waves_per_eu(R == 16 ? 4 : (MODE == BA_LOCAL ? 2 : BA_AFF_WPE)))) ba_dp_affine_kernel(BaDpParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int HALVES = R / 16;
    constexpr int PB = NIB ? 8 : 16;  // profile bytes per lane per (column symbol, 16 rows)
    constexpr int CPW = 32 / R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;

    const int wt_words = P.nb4 * P.na_pad;
    const int wt_aligned = (wt_words + 3) & ~3;
    for (int i = threadIdx.x; i < wt_words; i += blockDim.x) lds[i] = P.wt[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x)
        lds[wt_aligned + i] = i < 64 ? reinterpret_cast<const uint32_t *>(P.code1)[i]
                                     : reinterpret_cast<const uint32_t *>(P.code2)[i - 64];
    const uint8_t *code1 = reinterpret_cast<const uint8_t *>(lds + wt_aligned);
    const uint8_t *code2 = code1 + 256;
    // PIPE: progress words, one per wave: (round << 16) | column groups handed over so far.  They are read and written
    // with relaxed workgroup-scope atomics, which stay ds_read_b32 / ds_write_b32: a volatile access through the generic
    // pointer became a flat_load, and a flat operation returns out of order, so that every poll drained the wave's
    // whole queue of stores (s_waitcnt vmcnt(0) lgkmcnt(0)).
    uint32_t *prog = lds + wt_aligned + 128 + (size_t)waves * P.nb * (HALVES * 64 * (PB / 4));
    if (PIPE && threadIdx.x < 16) prog[threadIdx.x] = 0;
    __syncthreads();

    // PIPE: the block is persistent -- it takes bundles blockIdx.x, blockIdx.x + gridDim.x, ... and its waves run the
    // strips of that whole sequence round robin, so the pipeline never drains between bundles: strip 0 of the next
    // bundle (which waits for nobody) starts while the last strips of the current one are still on their way.
    uint32_t b_idx = PIPE ? P.bundle_begin + blockIdx.x : P.bundle_begin + blockIdx.x * waves + wave;
    if (b_idx >= P.bundle_end) return;  // (PIPE: the whole block leaves together)
    const uint32_t b_step = PIPE ? gridDim.x : 0u;
    const int ge = P.gap_ext, c = P.gap_open - P.gap_ext;
    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);

    uint8_t *prof = reinterpret_cast<uint8_t *>(lds + wt_aligned + 128) + (size_t)wave * P.nb * (HALVES * 64 * PB);
    uint8_t *prof_lane = prof + lane * PB;
    const int kappa = P.kappa;

    // the bundle in hand (wave-uniform)
    int n, m, strips, cgs, cws, front, jbits, key_shift, jmax;
    uint32_t b_first, b_count;
    int2 *bnd;          // the row a strip hands to the next one: (hq, F') per column, two buffers by strip parity
    BaLocalBest *cand;  // PIPE, local: the best cell of every strip, [strip][lane] (behind the two boundary buffers)
    uint4 *dir;
    const uint32_t *s1w, *s2w;
    auto open_bundle = [&]() {
        const BaBundle B = P.bundles[b_idx];
        n = (int)B.n;
        m = (int)B.m;
        strips = (n + R - 1) / R;
        cgs = (m + 3) / 4;
        cws = (m + CPW - 1) / CPW;
        front = strips * R - n;
        b_first = B.first;
        b_count = B.count;
        bnd = reinterpret_cast<int2 *>(P.hb + B.hb_off) + lane;
        cand = reinterpret_cast<BaLocalBest *>(P.hb + B.hb_off + (size_t)4 * m * 64) + lane;
        dir = reinterpret_cast<uint4 *>(P.dir + B.dir_off) + lane;
        s1w = P.s1w + B.s1w_off + lane;
        s2w = P.s2w + B.s2w_off + lane;
        // local: position key (see ba_affine_column); jbits covers 1..m
        jbits = 32 - __builtin_clz((uint32_t)(m | 1));
        key_shift = jbits + 1;
        jmax = (1 << jbits) - 1;
    };
    open_bundle();

    // H'(i, 0) and H'(0, j): align_pair.go:110-113, 130-131, 158-159, 187-188 on the middle level
    // (all scores as float patterns: + BA_FBIAS)
    auto col0 = [&](int i) { return BA_FBIAS + (left_free ? ge * max(i, 0) : (i >= 1 ? -c : 0)); };
    auto row0 = [&](int j) { return BA_FBIAS + (top_free ? ge * j : (j >= 1 ? -c : 0)); };

    int best_sc = 0, best_i = 0, best_j = 0;
    auto border_best = [&]() {  // strips == 0: the end cell lies on the top border; global: (0, m)
        best_sc = best_i = best_j = 0;
        if (MODE == BA_GLOBAL) {
            best_j = m;
            best_sc = m >= 1 ? -P.gap_open - (m - 1) * ge : 0;
        }
    };
    auto write_best = [&]() {
        if (lane < (int)b_count) {
            BaLocalBest bb;
            bb.score = best_sc;  // true score of the end cell
            bb.i = best_i;
            bb.j = best_j;
            bb.pad = 0;
            P.best[b_first + lane] = bb;
        }
    };
    border_best();

    const int pred = (wave + waves - 1) % waves;  // PIPE: the wave that runs the strip above mine
    uint32_t avail = 0;                           // PIPE: last progress word seen from it
    int s = PIPE ? wave : 0;  // my next strip of the bundle in hand
    for (int round = 0;; ++round, s += PIPE ? waves : 1) {
        if (PIPE) {
            bool out = false;
            while (s >= strips) {  // on to the next bundle(s); every wave passes every bundle of the block
                if (strips == 0 && wave == 0) {
                    border_best();
                    write_best();
                }
                s -= strips;
                b_idx += b_step;
                if (b_idx >= P.bundle_end) {
                    out = true;
                    break;
                }
                open_bundle();
            }
            if (out) break;
        } else if (s >= strips)
            break;
        // progress word of the strip above once it has handed over `groups` column groups
        const uint32_t pred_base = (uint32_t)(wave > 0 ? round : round - 1) << 16;
        auto wait_for = [&](int groups) {
            if (!PIPE || s == 0) return;
            const uint32_t need = pred_base + (uint32_t)groups;
            if (avail >= need) return;
            // bounded: every wave leaves this loop even if something upstream went wrong
            for (int spin = 0; spin < (1 << 26); ++spin) {
                avail = __hip_atomic_load(prog + pred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (avail >= need) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        // The boundary stores of the groups announced have been acknowledged before the word says so.  Vector-memory
        // operations of a wave complete in issue order (MI355X_MICROARCH.md, s_waitcnt), so a group is announced ONE GROUP
        // LATE behind `s_waitcnt vmcnt(4)`: the four boundary stores of the group just computed are younger than
        // everything the announced groups stored, and nothing is drained.  The last announcement of a strip waits for all.
        auto publish = [&](int groups, bool final) {
            if (!PIPE) return;
            if (final)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __hip_atomic_store(prog + wave, ((uint32_t)round << 16) + (uint32_t)groups, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        const int i0 = n - (strips - s) * R;
        const bool last_strip = (s == strips - 1);
        const int2 *bnd_rd = bnd + (size_t)((s + 1) & 1) * m * 64;
        int2 *bnd_wr = bnd + (size_t)(s & 1) * m * 64;
        uint32_t aw[R / 4];
#pragma unroll
        for (int q = 0; q < R / 4; ++q) aw[q] = s1w[(size_t)(s * (R / 4) + q) * 64];
        ba_build_strip_profile<R, NIB>(lds, P.nb, P.nb4, P.na_pad, code1, P.pad_code, aw, s, front, prof_lane);

        int hq[R], e[R], rkey[R / 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            hq[r] = col0(i0 + r + 1) - c;
            e[r] = BA_NEG;  // align_affine.go:45-46
            rkey[r / 2] = 0;
        }
        int topdiag = col0(i0) - c;
        // fitting / overlap: leftmost strict maximum of the strip's bottom row in true scores, found on the way (the
        // last strip's is the end cell, align_pair.go:169-182; true score = H'(n, j) - ext*j up to a constant).  A scan
        // of the stored row afterwards was a chain of 2000 dependent loads: a third of a millisecond per bundle.
        int ev = col0(n) - c, ej = 0;
        uint4 *dirs = dir + (size_t)s * cws * 64;
        uint32_t acc[4] = {0, 0, 0, 0};
        // fitting / overlap hand the last strip's bottom row over as well: the end-cell scan reads it back
        const bool write_bnd = true;  // (the last strip too: fitting / overlap scan it for the end cell; a conditional store costs registers)

        struct GroupIn {
            uint32_t sw;
            int2 b[4];
        };
        auto fetch_group = [&](int cg) {
            GroupIn gi;
            gi.sw = s2w[(size_t)cg * 64];
            if (s > 0) {
                wait_for(cg + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) gi.b[q] = bnd_rd[(size_t)min(cg * 4 + q, m - 1) * 64];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) gi.b[q] = make_int2(0, 0);
            }
            return gi;
        };
        // What a column group needs from LDS is looked up one group ahead as well: the dense codes of its four column
        // symbols and the profile words of its first column; inside the group the next column's profile words are in
        // flight while the current column is computed (a wave alone on its SIMD otherwise waits out every ds_read).
        constexpr int PWN = NIB ? R / 8 : R / 4;
        constexpr bool STAGE = PIPE && (R == 32 || NIB);  // (where the registers allow it: not one wavefront per bundle, not 16-row strips with a byte profile)
        struct Staged {
            uint32_t bcode[4];
            uint32_t pw[PWN];
        };
        auto load_pw = [&](uint32_t bc, uint32_t (&pw)[PWN]) {
            if (NIB) {
                const uint2 *pp = reinterpret_cast<const uint2 *>(prof_lane + (size_t)bc * (HALVES * 64 * 8));
#pragma unroll
                for (int hf = 0; hf < HALVES; ++hf) {
                    const uint2 v = pp[hf * 64];
                    pw[NIB ? hf * 2 + 0 : 0] = v.x;
                    pw[NIB ? hf * 2 + 1 : 0] = v.y;
                }
            } else {
                const uint4 *pp = reinterpret_cast<const uint4 *>(prof_lane + (size_t)bc * (HALVES * 64 * 16));
#pragma unroll
                for (int hf = 0; hf < HALVES; ++hf) {
                    const uint4 v = pp[hf * 64];
                    pw[NIB ? 0 : hf * 4 + 0] = v.x;
                    pw[NIB ? 0 : hf * 4 + 1] = v.y;
                    pw[NIB ? 0 : hf * 4 + 2] = v.z;
                    pw[NIB ? 0 : hf * 4 + 3] = v.w;
                }
            }
        };
        auto stage = [&](const GroupIn &gi) {
            Staged st;
#pragma unroll
            for (int q = 0; q < 4; ++q) st.bcode[q] = code2[(gi.sw >> (q * 8)) & 0xffu];
            load_pw(st.bcode[0], st.pw);
            return st;
        };
        auto column_group = [&](int cg, const GroupIn &in, const Staged &st, auto full) {
            constexpr bool FULL = decltype(full)::value;
            const int jbase = cg * 4 + 1;
            uint32_t pw[PWN];
#pragma unroll
            for (int k = 0; k < PWN; ++k) pw[k] = st.pw[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (FULL || jbase + q <= m) {
                    const int j = jbase + q;
                    uint32_t pwn[PWN];
                    if (!STAGE && q > 0) load_pw(st.bcode[q], pw);
                    if (STAGE && q < 3 && (FULL || jbase + q + 1 <= m)) load_pw(st.bcode[q < 3 ? q + 1 : 3], pwn);
                    // the row above the strip: row 0 (F = -inf, align_affine.go:63-64) or the previous strip's last row
                    const int top_hq = (s == 0) ? row0(j) - c : in.b[q].x;
                    int f = (s == 0) ? BA_NEG : in.b[q].y;
                    const int fl0 = BA_FBIAS + ge * (i0 + j);  // local: zero level of the row above the strip (wave-uniform: the rows' levels are an s_add each)
                    ba_affine_column<MODE, R, NIB, FKEY>(hq, e, acc, rkey, pw, top_hq, f, topdiag, c, ge, fl0, key_shift,
                                                   (1 << jbits) + (jmax - j), jmax - j, kappa);
                    if ((j - 1) % CPW == CPW - 1)
                        dirs[(size_t)((j - 1) / CPW) * 64] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                    if (write_bnd) bnd_wr[(size_t)(j - 1) * 64] = make_int2(hq[R - 1], f);
                    if (MODE == BA_FITTING || MODE == BA_OVERLAP) {
                        const int v = hq[R - 1] - ge * j;
                        ej = v > ev ? j : ej;
                        ev = max(ev, v);
                    }
                    if (STAGE && q < 3 && (FULL || jbase + q + 1 <= m)) {
#pragma unroll
                        for (int k = 0; k < PWN; ++k) pw[k] = pwn[k];
                    }
                }
            }
        };
        const int full_groups = m / 4;
        if (cgs > 0) {
            GroupIn nxt = fetch_group(0);
            // (the first group's loads are waited for HERE: left pending into the loop, the compiler's counter state at the
            // loop head asks for `vmcnt(5)` in every iteration, which also waits for all of the previous group's stores)
            asm volatile("" : "+v"(nxt.sw), "+v"(nxt.b[0].x), "+v"(nxt.b[1].x), "+v"(nxt.b[2].x), "+v"(nxt.b[3].x));
            asm volatile("" : "+v"(nxt.b[0].y), "+v"(nxt.b[1].y), "+v"(nxt.b[2].y), "+v"(nxt.b[3].y));
            Staged nst;
            if (STAGE) nst = stage(nxt);
            for (int cg = 0; cg < full_groups; ++cg) {
                const GroupIn cur = nxt;
                const Staged cst = STAGE ? nst : stage(cur);
                nxt = fetch_group(min(cg + 1, cgs - 1));
                column_group(cg, cur, cst, std::true_type{});
                if (STAGE) nst = stage(nxt);
                publish(cg, false);  // (groups 0 .. cg - 1; this group's four stores are the younger ones)
            }
            if (full_groups < cgs) column_group(full_groups, nxt, STAGE ? nst : stage(nxt), std::false_type{});
        }
        if (MODE == BA_LOCAL) {
            // first strict maximum in row-major order (align.go:216-221 on the middle level).
            // Row pairs that lie wholly above the matrix (first strip) carry no scores.
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int k = FKEY ? (int)__int_as_float(rkey[r / 2]) : rkey[r / 2];
                const int sc = k >> key_shift;
                if (i0 + r + 2 >= 1 && sc > best_sc) {
                    best_sc = sc;
                    best_i = i0 + r + 1 + (((k >> jbits) & 1) ? 0 : 1);
                    best_j = jmax - (k & jmax);
                }
            }
            if (PIPE && !last_strip) {
                BaLocalBest own;
                own.score = best_sc;
                own.i = best_i;
                own.j = best_j;
                own.pad = 0;
                cand[(size_t)s * 64] = own;
            }
        }
        if (cgs > 0) {
            publish(cgs, true);
            // a half-filled last word (R == 16, odd m): its column sits in the top half like any first column
            if (CPW == 2 && (m & 1))
                dirs[(size_t)(m / 2) * 64] = make_uint4(acc[0] << 16, acc[1] << 16, acc[2] << 16, acc[3] << 16);
        }

        if (last_strip && MODE != BA_LOCAL) {
            int ecur = hq[R - 1];  // global: the last cell (align_pair.go:123-126)
            if (MODE == BA_FITTING || MODE == BA_OVERLAP) {
                ecur = ev + ge * ej;
            } else
                ej = m;
            // true score of the end cell: H = hq + c - ext*(i+j)
            best_sc = ecur - BA_FBIAS + c - ge * (n + ej);
            best_i = n;
            best_j = ej;
            if (PIPE) write_best();  // (the wave that ran the last strip knows the end cell)
        }
        if (MODE == BA_LOCAL && PIPE) {
            // The strips of a bundle are spread over the waves: every strip leaves its own first best cell in `cand`
            // (stored before the strip's last announcement, which waits for it), and the wave of the last strip --
            // every other strip of the bundle has made its last announcement by then -- takes the first maximum over
            // the strips in order, which is the first in row-major order (align.go:216-221 on the middle level).
            if (last_strip) {
                int bsc = 0, bi = 0, bj = 0;
                auto take = [&](const BaLocalBest &x) {
                    if (x.score > bsc) {
                        bsc = x.score;
                        bi = x.i;
                        bj = x.j;
                    }
                };
                if (cgs > 0)  // (no columns: no announcements to order the reads behind, and nothing to find)
                    for (int t = 0; t < s; ++t) take(cand[(size_t)t * 64]);
                BaLocalBest own;
                own.score = best_sc;
                own.i = best_i;
                own.j = best_j;
                take(own);
                best_sc = bsc;
                best_i = bi;
                best_j = bj;
                write_best();
            }
            best_sc = best_i = best_j = 0;
        }
    }
    if (!PIPE) write_best();
}

// ----------------------------------------------------------------------------
// traceback, affine: one lane walks one pair's path over (level, node) exactly as the
// oracle's canonical traceback does, reading the decision bits, and writes both traces
// back to front into the pair's output slot.
// ----------------------------------------------------------------------------
template <int MODE, int R>
__global__ void __launch_bounds__(256) ba_trace_affine_kernel(BaTraceParams P)
{
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int CPW = 32 / R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: the bundle record and all it implies live in SGPRs
    const int waves = blockDim.x >> 6;
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
    const bool live = lane < (int)B.count;
    const uint32_t p = live ? P.perm[B.first + lane] : 0u;
    const int n = (int)B.n, m = (int)B.m;
    const int go = P.gap_open, ge = P.gap_ext;
    const uint4 *dir = reinterpret_cast<const uint4 *>(P.dir + B.dir_off) + lane;
    const uint64_t o1 = live ? P.s1_off[p] : 0, o2 = live ? P.s2_off[p] : 0;
    const int cws = (m + CPW - 1) / CPW;
    const int front = ((n + R - 1) / R) * R - n;
    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);

    int i = 0, j = 0, V = 0;
    if (live) {
        const BaLocalBest bb = P.best[B.first + lane];
        i = bb.i;
        j = bb.j;
        V = bb.score;
        P.scores[p] = (int64_t)bb.score;
        if (P.score_only && P.t_off) {  // BIOALIGN_END_CELLS
            P.t_off[p] = (uint64_t)i;
            P.t_len[p] = (uint64_t)j;
        }
    }
    if (P.score_only) return;

    const uint64_t slot = o1 + o2;
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    const uint32_t *s1w = P.s1w + B.s1w_off + lane;
    const uint32_t *s2w = P.s2w + B.s2w_off + lane;
    int k1_tag = -1, k2_tag = -1;
    uint32_t x_word = 0, y_word = 0;
    auto sym1 = [&](int i1) {  // s1[i1-1]
        const int t = i1 - 1 + front;
        if ((t >> 2) != k1_tag) {
            k1_tag = t >> 2;
            x_word = s1w[(size_t)k1_tag * 64];
        }
        return (x_word >> ((t & 3) * 8)) & 0xffu;
    };
    auto sym2 = [&](int j1) {  // s2[j1-1]
        const int cc = j1 - 1;
        if ((cc >> 2) != k2_tag) {
            k2_tag = cc >> 2;
            y_word = s2w[(size_t)k2_tag * 64];
        }
        return (y_word >> ((cc & 3) * 8)) & 0xffu;
    };
    BaBackWriter w1, w2;
    w1.init(P.t1, slot + cap);
    w2.init(P.t2, slot + cap);
    const uint32_t gs = P.gap_symbol;

    // the decision word of (strip, column word) last fetched by this lane
    int tag = -1;
    uint4 q = make_uint4(0, 0, 0, 0);
    uint32_t sh = 0;
    auto fetch = [&](int ii, int jj) {  // 1 <= ii <= n, 1 <= jj <= m
        const int t = ii - 1 + front;
        const int s = t / R, r = t % R;
        const int cw = (jj - 1) / CPW;
        const int id = s * cws + cw;
        if (id != tag) {
            tag = id;
            q = dir[(size_t)id * 64];
        }
        sh = 31u - (uint32_t)(((jj - 1) % CPW) * R + r);
    };

    enum { LV_MM = 0, LV_G1 = 1, LV_G2 = 2 };
    int level = LV_MM;
    bool done = !live || (i == 0 && j == 0);
    // (every round takes a lane one row or one column closer to the top left corner: n + m rounds at most; the bound
    // only matters if the decision words were ever wrong -- the walk then ends instead of running off the matrix)
    for (int guard = n + m + 2; guard > 0 && __any(!done); --guard) {
        if (!done) {
            if (level == LV_MM) {
                // back-pointer of the middle level (oracle: bmm)
                if (i == 0) {
                    if (top_free || j == 0) done = true; else level = LV_G1;   // top border: sg2 = -inf, no diagonal
                } else if (j == 0) {
                    if (left_free) done = true; else level = LV_G2;            // left border
                } else if (MODE == BA_LOCAL && V == 0) {
                    done = true;                                               // skip-prefix cell
                } else {
                    fetch(i, j);
                    if (((q.x >> sh) & 1u) == 0) {
                        level = LV_G2;
                    } else if (((q.y >> sh) & 1u) == 0) {
                        const uint32_t x = sym1(i), y = sym2(j);
                        w1.put(x);
                        w2.put(y);
                        if (MODE == BA_LOCAL) V -= tbl[(int)code1[x] * P.nb + (int)code2[y]];
                        --i;
                        --j;
                    } else {
                        level = LV_G1;
                    }
                }
            }
            if (!done && level == LV_G1) {
                // gap in s1: emit (gap, s2[j-1]); j >= 1 on this level
                bool opened;
                if (i == 0) {
                    opened = (j == 1);  // top border (open >= ext): only the first step opens
                } else if (j == 1) {
                    opened = true;      // E(i, 0) = -inf
                } else {
                    fetch(i, j);
                    opened = ((q.z >> sh) & 1u) != 0;
                }
                w1.put(gs);
                w2.put(sym2(j));
                if (MODE == BA_LOCAL) V += opened ? go : ge;
                level = opened ? LV_MM : LV_G1;
                --j;
            } else if (!done && level == LV_G2) {
                // gap in s2: emit (s1[i-1], gap); i >= 1 on this level
                bool opened;
                if (j == 0) {
                    opened = (i == 1);
                } else if (i == 1) {
                    opened = true;      // F(0, j) = -inf
                } else {
                    fetch(i, j);
                    opened = ((q.w >> sh) & 1u) != 0;
                }
                w1.put(sym1(i));
                w2.put(gs);
                if (MODE == BA_LOCAL) V += opened ? go : ge;
                level = opened ? LV_MM : LV_G2;
                --i;
            }
            if (level == LV_MM && i == 0 && j == 0) done = true;
        }
    }
    if (live) {
        w1.finish();
        w2.finish();
        const uint64_t len = (uint64_t)w1.length();
        P.t_off[p] = slot + cap - len;
        P.t_len[p] = len;
    }
}

// ----------------------------------------------------------------------------
// host-side launchers (called from api.cpp)
// ----------------------------------------------------------------------------
extern "C" size_t ba_dp_linear_lds_bytes(int nb, int nb4, int na_pad, int R, int waves);

// profile + tables as the linear kernel, plus the progress words and the best-cell exchange of PIPE
// (nib: 4-bit profile, half the bytes)
extern "C" size_t ba_dp_affine_lds_bytes(int nb, int nb4, int na_pad, int R, int waves, int nib)
{
    const int wt_aligned = (nb4 * na_pad + 3) & ~3;
    return (size_t)(wt_aligned + 128) * 4 + (size_t)waves * nb * (R / 16) * 64 * (nib ? 8 : 16) +
           16 * 4;
}

template <int MODE, int R>
static hipError_t launch_dp_aff(const BaDpParams &P, int waves, int pipe, hipStream_t st)
{
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const size_t lds = ba_dp_affine_lds_bytes(P.nb, P.nb4, P.na_pad, R, waves, P.nib);
    dim3 grid(pipe ? nbundles : (nbundles + waves - 1) / waves), block(waves * 64);
    int cus = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    // more than 64 KB of dynamic LDS (large alphabets) has to be asked for, kernel by kernel
    auto allow = [&](auto kernel) {
        return lds > 64 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                               : hipSuccess;
    };
    // local mode has a second instantiation whose best-cell keys are built as floats (FKEY)
    auto go = [&](auto pipe_c, auto nib_c) {
        constexpr bool PIPE = decltype(pipe_c)::value, NIB = decltype(nib_c)::value;
        hipError_t ae;
        // PIPE: persistent blocks, as many as the chip holds at once; each takes every grid.x-th bundle
        auto resident = [&](auto kernel) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)block.x, lds) != hipSuccess || per_cu < 1)
                per_cu = 1;
            // ... but not so few blocks that the dispatcher has nothing left to balance with: about `per_block`
            // bundles each (the pipeline drains once per block, not once per bundle)
            const uint64_t resident_blocks = (uint64_t)per_cu * (uint64_t)cus;
            const uint64_t per_block = (uint64_t)std::max(1, P.pipe_bundles);
            if (PIPE) grid.x = (unsigned)std::max<uint64_t>(std::min<uint64_t>(nbundles, resident_blocks), (nbundles + per_block - 1) / per_block);
        };
        if (MODE == BA_LOCAL && P.fkey) {
            if ((ae = allow(&ba_dp_affine_kernel<MODE, R, PIPE, NIB, MODE == BA_LOCAL>)) != hipSuccess) return ae;
            resident(&ba_dp_affine_kernel<MODE, R, PIPE, NIB, MODE == BA_LOCAL>);
            hipLaunchKernelGGL((ba_dp_affine_kernel<MODE, R, PIPE, NIB, MODE == BA_LOCAL>), grid, block, lds, st, P);
        } else {
            if ((ae = allow(&ba_dp_affine_kernel<MODE, R, PIPE, NIB>)) != hipSuccess) return ae;
            resident(&ba_dp_affine_kernel<MODE, R, PIPE, NIB>);
            hipLaunchKernelGGL((ba_dp_affine_kernel<MODE, R, PIPE, NIB>), grid, block, lds, st, P);
        }
        return hipGetLastError();
    };
    if (P.nib) {
        // the 4-bit profile exists for the modes with a free top row and 16-row strips (api.cpp picks it)
        if constexpr (R == 16 && (MODE == BA_LOCAL || MODE == BA_FITTING))
            return pipe ? go(std::true_type{}, std::true_type{}) : go(std::false_type{}, std::true_type{});
        return hipErrorInvalidValue;
    }
    return pipe ? go(std::true_type{}, std::false_type{}) : go(std::false_type{}, std::false_type{});
}

template <int MODE>
static hipError_t launch_dp_aff_mode(const BaDpParams &P, int R, int waves, int pipe, hipStream_t st)
{
    if (R == 32) return launch_dp_aff<MODE, 32>(P, waves, pipe, st);
    if (R == 16) return launch_dp_aff<MODE, 16>(P, waves, pipe, st);
    return hipErrorInvalidValue;
}

// waves: wavefronts per block (pipe: 2..8, strips in flight per bundle; else 1..4 bundles per block)
extern "C" hipError_t ba_launch_dp_affine(const BaDpParams &P, int mode, int R, int waves, int pipe, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    if (waves < 1 || waves > (pipe ? (R == 16 ? 16 : (mode == BA_LOCAL ? 8 : 12)) : 4)) return hipErrorInvalidValue;
    switch (mode) {
    case BA_GLOBAL: return launch_dp_aff_mode<BA_GLOBAL>(P, R, waves, pipe, st);
    case BA_LOCAL: return launch_dp_aff_mode<BA_LOCAL>(P, R, waves, pipe, st);
    case BA_FITTING: return launch_dp_aff_mode<BA_FITTING>(P, R, waves, pipe, st);
    case BA_OVERLAP: return launch_dp_aff_mode<BA_OVERLAP>(P, R, waves, pipe, st);
    }
    return hipErrorInvalidValue;
}

template <int MODE, int R>
static hipError_t launch_trace_aff(const BaTraceParams &P, hipStream_t st)
{
    const int waves = 4;
    const uint32_t nbundles = P.bundle_end - P.bundle_begin;
    const uint32_t blocks = (nbundles + waves - 1) / waves;
    const size_t lds = ((size_t)((P.na * P.nb + 3) & ~3) + 128) * 4;
    if (lds > 64 * 1024) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_trace_affine_kernel<MODE, R>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ae != hipSuccess) return ae;
    }
    hipLaunchKernelGGL((ba_trace_affine_kernel<MODE, R>), dim3(blocks), dim3(waves * 64), lds, st, P);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_trace_aff_mode(const BaTraceParams &P, int R, hipStream_t st)
{
    if (R == 32) return launch_trace_aff<MODE, 32>(P, st);
    if (R == 16) return launch_trace_aff<MODE, 16>(P, st);
    return hipErrorInvalidValue;
}

extern "C" hipError_t ba_launch_trace_affine(const BaTraceParams &P, int mode, int R, hipStream_t st)
{
    if (P.bundle_end == P.bundle_begin) return hipSuccess;
    switch (mode) {
    case BA_GLOBAL: return launch_trace_aff_mode<BA_GLOBAL>(P, R, st);
    case BA_LOCAL: return launch_trace_aff_mode<BA_LOCAL>(P, R, st);
    case BA_FITTING: return launch_trace_aff_mode<BA_FITTING>(P, R, st);
    case BA_OVERLAP: return launch_trace_aff_mode<BA_OVERLAP>(P, R, st);
    }
    return hipErrorInvalidValue;
}
