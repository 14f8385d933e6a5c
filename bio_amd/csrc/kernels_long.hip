// kernels_long.hip -- long mode: ONE pair aligned by many wavefronts (intra-sequence,
// anti-diagonal wavefront), linear gap, global / local / fitting / overlap.
//
// Same recurrences, tie priorities and working score space as kernels.hip
// (Hs = H + g*(i+j); cell = max3(up, left, diag + w + 2g); K-bit residues of every cell as
// the traceback matrix; reference align.go:97-143, align_pair.go:109-246), different
// mapping of the matrix onto the machine:
//
//   * a STRIP is 64*T consecutive rows; lane l of the strip's wavefront owns rows
//     l*T .. l*T+T-1 and keeps their scores in VGPRs;
//   * the wavefront sweeps the strip left to right with an anti-diagonal skew: at step t
//     lane l works on column t - l.  What lane l needs from the row above its first row
//     is what lane l-1 computed one step earlier: one DPP wavefront shift per step
//     (v_mov_b32_dpp wave_shr:1), no LDS.  The column symbols travel down the lanes the
//     same way;
//   * strips are a pipeline over the columns: strip s reads the bottom row of strip s-1
//     from HBM/L2 in blocks of 64 columns, announced by a per-strip progress word.  Both
//     go through agent-scope atomic stores/loads (they bypass the non-coherent cache
//     levels), so no cache flush is needed.  A strip depends only on the strip with the
//     next lower block index; blocks are dispatched in index order, so the wave a strip
//     waits for is always resident or finished.  Every wait is bounded anyway.
//   * residue words are stored by (strip, step, lane): one coalesced 256-byte line per
//     store instruction; the traceback undoes the skew (step = column + lane).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "device_helpers.h"

#define BA_LONG_FINAL 0x7fffffffu   // progress word: every column has been handed over
#define BA_LONG_POISON 0x80000000u  // progress word, top bit: a wait ran out upstream; do not wait for this strip
#ifndef BA_LONG_W
#define BA_LONG_W 4                 // strips (wavefronts) per workgroup: one per SIMD of a CU
#endif
#define BA_LONG_RB 32                // depth, in 64-column blocks, of the ring a strip hands its bottom row over in

// lane l <- lane l-1; lane 0 keeps `first`
__device__ __forceinline__ int ba_wave_shr1(int first, int src)
{
    return __builtin_amdgcn_update_dpp(first, src, 0x138, 0xf, 0xf, false);
}

// Waits until every store this wave has issued has been acknowledged, and keeps the compiler
// from moving memory operations across it.
__device__ __forceinline__ void ba_stores_done()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // inline asm: the waitcnt pass cannot drop it (MI355X_MICROARCH.md, compiler hazard)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// One dword to LDS, never merged with a neighbour into a ds_write2_b32.  Seen on gfx950 (MI355X, ROCm 7.2): after
//   ds_write2_b32 vA, vD0, vD1 ...   /   v_alignbit_b32 vD1, ...
// back to back, LDS received the NEW vD1 -- the store hands its three source dwords over at two cycles each
// (MI355X_MICROARCH.md, LDS) and nothing holds the next VALU write back; hipcc pads only stores of more than 64 bits
// of data.  The merged form appeared (54 times) once the steps were unrolled; an atomic store stays a ds_write_b32,
// whose one data dword is gone before a following instruction can write it.  tools/scan_ds_write2.py looks for the
// pattern in a hipcc -S dump.
__device__ __forceinline__ void ba_lds_store(int32_t *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int T, int K>
struct BaLongGeom {
    static constexpr int RS = 64 * T;          // rows per strip
    static constexpr int CPD = 32 / (T * K);   // steps per residue word
    static_assert(CPD >= 1 && CPD * T * K == 32, "T * K must divide 32");
};

// What one strip (wavefront) keeps in LDS for itself, dwords.
struct BaLongWaveLds {
    uint32_t ohw[128];   // column symbols as byte offsets into a lane's strip profile (4 * code), window of two 64-column
                         // blocks [previous | current]: lane l at step k of a block reads slot 64 + k - l
    int32_t dummy[128];  // where the lanes other than 63 put their (not yet final) bottom-row values
    int32_t dummy_f[128];  // affine: the same for F'
};

#define BA_LONG_RING (BA_LONG_RB * 64)  // columns a hand-over ring holds
#ifndef BA_WALK_TW
#define BA_WALK_TW 128                  // residue words (x 64 lanes) the walk stages per tile
#endif

// One step of one lane: T cells of one column.  `pw` holds the substitution scores of the lane's T rows against this
// column's symbol, one signed byte per row (from the lane's strip profile in LDS): the byte is an SDWA operand of the add.
// CHECK: the lane may be outside its column range (start-up and drain of the skew) and must then keep its state.
// LOCAL: true scores (no bias), H = max(0, max(up, left) - g, diag + w) with w + g in the profile: one max3 and one
// saturating unsigned subtract, as kernels.hip does it; `lmax` collects the largest score the lane has seen.
template <int T, int K, bool CHECK, bool LOCAL>
__device__ __forceinline__ void ba_long_step(int (&h)[T], uint32_t &acc, int top, int diag0, uint32_t pw, bool active, int g,
                                             int &lmax)
{
    int up = top, diag = diag0;
#pragma unroll
    for (int r = 0; r < T; ++r) {
        const int left = h[r];
        const int d = diag + ba_sbyte(pw, r);
        int hn = ba_max3(up, left, d);
        if (LOCAL) hn = (int)__builtin_elementwise_sub_sat((uint32_t)hn, (uint32_t)g);  // (max3 >= up >= 0)
        if (CHECK) hn = active ? hn : left;
        acc = __builtin_amdgcn_alignbit((uint32_t)hn, acc, K);
        diag = left;
        h[r] = hn;
        up = hn;
    }
    if (LOCAL) {
        if (T == 1) lmax = max(lmax, h[0]);
#pragma unroll
        for (int r = 0; r + 1 < T; r += 2) lmax = ba_max3(lmax, h[r], h[r + 1]);
    }
}

#define BA_LONG_NEG (-(1 << 28))  // minus infinity of the E / F levels (align_affine.go:36-37, 45-46, 63-64)

// push the sign bit of t into the bottom of a bit field
__device__ __forceinline__ uint32_t ba_long_push_sign(uint32_t acc, int t)
{
    return __builtin_amdgcn_alignbit(acc, (uint32_t)t, 31);
}

// Affine gaps (three-level Gotoh, align_affine.go:44-103), the cell of kernels_affine.hip on the long-mode mapping:
// every level biased by ext*(i+j), a lane keeps hq = H' - c (c = open - ext) and E' of its T rows; F' travels down the
// column -- from the lane above through a second DPP shift, from the strip above through a second ring.  Four decision
// bits per cell (nF, nD, oE, oF, see kernels_affine.hip), pushed as one nibble: the K = 4 layout of the residues.
// Local: the zero level of a cell is ext*(i+j) (`fl`, one add per row); the best cell is kept per ROW as (largest true
// score, step it was first reached at): strict '>' keeps the first column, and rows are compared afterwards.
template <int T, bool CHECK, int MODE>
__device__ __forceinline__ void ba_long_step_aff(int (&hq)[T], int (&e)[T], uint32_t &acc, int top, int ftop, int diag0,
                                                 uint32_t pw, bool active, int c, int ge, int fl, int (&rmax)[T],
                                                 int (&rstep)[T], int t, int &fout)
{
    int up = top, diag = diag0, f = ftop;
#pragma unroll
    for (int r = 0; r < T; ++r) {
        const int eo = e[r], left = hq[r];
        const int en = max(eo, left);
        const int fn = max(f, up);
        const int d = diag + ba_sbyte(pw, r);
        int H = ba_max3(en, fn, d);
        if (MODE == BA_LOCAL) {
            fl += ge;
            H = max(H, fl);
        }
        uint32_t cb = 0;
        cb = ba_long_push_sign(cb, fn - H);   // nF
        cb = ba_long_push_sign(cb, d - H);    // nD
        cb = ba_long_push_sign(cb, eo - en);  // oE
        cb = ba_long_push_sign(cb, f - fn);   // oF
        acc = __builtin_amdgcn_alignbit(cb, acc, 4);
        int hn = H - c, e_new = en, f_new = fn;
        if (CHECK && !active) {
            hn = left;
            e_new = eo;
            f_new = f;
        }
        if (MODE == BA_LOCAL) {
            const int ht = H - fl;
            const bool better = (!CHECK || active) && ht > rmax[r];
            rmax[r] = better ? ht : rmax[r];
            rstep[r] = better ? t : rstep[r];
        }
        diag = left;
        hq[r] = hn;
        e[r] = e_new;
        f = f_new;
        up = hn;
    }
    fout = f;
}

// The DP of one pair.  A workgroup is BA_LONG_W consecutive strips, one wavefront each (one per SIMD of a CU), plus one
// helper wavefront; compute wave w of workgroup S runs strip s = S*W + w - pad (pad: phantom waves above the matrix in
// the first workgroup, so that the last strip is the last wave of the last workgroup).
//
// Hand-over inside the workgroup: ring r (LDS, BA_LONG_RING columns, slot = column mod size) holds the bottom row of
// compute wave r-1 and is read by compute wave r.  A column of a strip's bottom row is final once lane 63 has passed
// it, so lane 63 alone writes the ring, one column per step (the other lanes' stores go to a scratch row), and the
// producer publishes a column count every 16 steps; the reader needs column t at its step t, so a strip runs
// 63 + 16 steps behind the one above plus the time the counter takes.  Both counters (columns final, columns the
// reader is done with) live in LDS.
//
// Hand-over between workgroups, through HBM/L2: the last compute wave stores its final columns 16 at a time as
// 8-byte granules {score, tag} -- one 128-byte line per store, agent scope (sc1), fire and forget; the tag is the
// launch's epoch and the buffer is zeroed before the launch.  The helper wave of the workgroup below polls the
// granules themselves (sc1 loads, up to four blocks in flight), copies the valid prefix into ring 0 (ring `pad` in
// the first workgroup) and publishes it: no progress word, no store drain, one memory round trip per poll
// (MI355X_MICROARCH.md, hand-off price list: data-tagged granules).  For strip 0 the helper writes the top border's
// closed form instead.
//
// Every wait is bounded; a wait that runs out raises a flag in LDS that ends every other wait of the workgroup, the
// rest of the workgroup's bottom row goes out with a poisoned tag, which fails the workgroups below at once, and the
// host reports the run as failed.  (The cross-workgroup wait assumes what the status word guards: a workgroup's
// predecessor is resident or finished, which holds while workgroups are dispatched in index order.)
//
// MULTI: the grid holds the workgroups of several column stripes of one pair (P0.multi / P0.order); the workgroups
// are ordered by the time they can start, which keeps every workgroup behind the two it waits for -- the one above it
// in its own stripe and the one at its height in the stripe to its left.
//
// AFF: affine gaps (ba_long_step_aff): a hand-over carries two values per column, hq and F' -- a second LDS ring, and
// in the HBM granule F' travels as an 8-bit offset from hq (F' <= hq + c always; anything below hq behaves like hq - 1),
// next to a 24-bit tag.
// NORES: no residue words are stored (score only; the first pass of the linear-space traceback).
template <int MODE, int T, int K, bool MULTI, bool AFF, bool NORES = false>
__global__ void __launch_bounds__(64 * (BA_LONG_W + 1)) ba_long_dp_kernel(BaLongParams P0)
{
    static_assert(!AFF || (K == 4 && !MULTI), "affine: four decision bits per cell, unstriped");
    using G = BaLongGeom<T, K>;
    constexpr int RS = G::RS, CPD = G::CPD, W = BA_LONG_W, RING = BA_LONG_RING;
    extern __shared__ __align__(16) uint32_t dyn_lds[];  // strip profiles: [wave][column symbol][lane] dwords, then padding
    __shared__ uint32_t code2s[64];  // byte -> code of the column side, packed
    __shared__ BaLongWaveLds wl[W];
    __shared__ int32_t rings[W + 1][RING];
    __shared__ int32_t rings_f[AFF ? W + 1 : 1][AFF ? RING : 1];  // affine: F' of the same columns
    __shared__ uint32_t prog_s[W + 1];   // columns of ring r that are final (all columns below this number)
    __shared__ uint32_t cons_s[W + 1];   // columns of ring r its reader is done with
    __shared__ uint32_t fail_s;          // a wait has run out somewhere in this workgroup
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    BaLongParams Pm;
    int S_;
    if (MULTI) {
        const uint2 o = P0.order[blockIdx.x];
        Pm = P0.multi[o.x];
        S_ = (int)o.y;
    } else {
        S_ = (int)(blockIdx.x + P0.wg_begin);
    }
    const BaLongParams &P = MULTI ? Pm : P0;
    const int S = S_;
    const int n = (int)P.n, m = (int)P.m, g = P.gap;
    const int strips = (int)P.strips;
    const int pad = (int)P.nsuper * W - strips;
    if (threadIdx.x < 64) code2s[threadIdx.x] = reinterpret_cast<const uint32_t *>(P.code2)[threadIdx.x];
    if (threadIdx.x <= W) {
        prog_s[threadIdx.x] = 0;
        cons_s[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) fail_s = 0;
    __syncthreads();  // (the only barrier: waves leave at different times below)

    const int total_steps = m + 63;
    const int nblk = (total_steps + 63) / 64;
    const int first_ring = (S == 0) ? pad : 0;  // the ring the helper fills: the first real strip of the workgroup reads it
    const int b_top = (MODE == BA_FITTING) ? g : 0;   // Hs(0, j) = b_top * j
    const int b_left = (MODE == BA_OVERLAP) ? g : 0;  // Hs(i, 0) = b_left * i
    const int col0 = (int)P.col0;  // column stripes: my column 0 is column col0 of the matrix
    const uint32_t epoch = P.epoch;
    // affine: H'(i, 0) and H'(0, j) (align_pair.go:110-113, 130-131, 158-159, 187-188 on the middle level, kernels_affine.hip)
    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    const int ge = P.gap_ext, cc = P.gap - P.gap_ext;
    auto aff_col0 = [&](int i) { return left_free ? ge * max(i, 0) : (i >= 1 ? -cc : 0); };
    auto aff_row0 = [&](int j) { return top_free ? ge * j : (j >= 1 ? -cc : 0); };
    // granule: {value, tag}; affine: {hq, tag << 8 | (F' - hq clamped to -1 .. 254) + 1}
    constexpr uint32_t POISON = AFF ? 0x800000u : BA_LONG_POISON;
    auto tag_of = [](unsigned long long gr) { return AFF ? (uint32_t)(gr >> 40) : (uint32_t)(gr >> 32); };
    auto mk_gran = [](uint32_t tag, int v, int fv) -> unsigned long long {
        if (AFF) {
            const uint32_t delta = (uint32_t)(min(max(fv - v, -1), 254) + 1);
            return ((unsigned long long)((tag << 8) | delta) << 32) | (uint32_t)v;
        }
        return ((unsigned long long)tag << 32) | (uint32_t)v;
    };

    auto ld = [](uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto st = [](uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // a bounded wait on a counter in LDS: until *word >= need (or the workgroup has failed); returns what it saw
    auto wait_lds = [&](uint32_t *word, uint32_t need) -> uint32_t {
        uint32_t v = ld(word);
        if (v >= need) return v;
        int spin = 0;
        for (; spin < (1 << 22); ++spin) {
            v = ld(word);
            if (v >= need || ld(&fail_s)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (spin == (1 << 22)) st(&fail_s, 1u);
        return v >= need ? v : need;  // (a failed run goes on without waiting)
    };
    // a counter behind the LDS writes it announces: one wavefront's LDS operations are carried out in the order they
    // are issued, so the reader that sees the counter sees the writes; the fence only pins the compiler's order
    auto publish = [&](uint32_t *word, uint32_t v) {
        asm volatile("" ::: "memory");
        if (lane == 0) st(word, v);
    };
    unsigned long long *bnd_out = reinterpret_cast<unsigned long long *>(P.bnd) + (size_t)S * P.bnd_stride;

    // ---------------- helper: the row above the workgroup's first strip -> ring first_ring ----------------
    if (w == W) {
        const unsigned long long *bnd_in = reinterpret_cast<const unsigned long long *>(P.bnd) + (size_t)(S > 0 ? S - 1 : 0) * P.bnd_stride;
        int32_t *ring = rings[first_ring];
        int32_t *ring_f = rings_f[AFF ? first_ring : 0];
        int done = 0;  // columns handed to the ring so far: a multiple of 16 until the end
        int idle = 0;
        while (done < m) {
            // room in the ring: the reader must be done with the columns these slots held
            const int b0 = done / 64;  // the block the next column lies in; up to four blocks from here per round
            uint32_t cons = ld(&cons_s[first_ring]);
            if ((uint32_t)(b0 * 64 + 64) > cons + RING) cons = wait_lds(&cons_s[first_ring], (uint32_t)(b0 * 64 + 64 - RING));
            const int room_blocks = (int)((cons + RING) / 64) - b0;
            const int cnt = min(min(4, room_blocks), (m + 63) / 64 - b0);
            unsigned long long gv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = (b0 + q) * 64 + lane;
                if (S == 0)  // the top border in closed form (affine: row 0 of the middle level, F' = -inf)
                    gv[q] = AFF ? mk_gran(epoch, aff_row0(col + 1) - cc, BA_LONG_NEG) : mk_gran(epoch, b_top * (col0 + col + 1), 0);
                else
                    gv[q] = (q < cnt && col < m) ? __hip_atomic_load(bnd_in + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : mk_gran(epoch, 0, 0);
            }
            // valid prefix, in columns from b0 * 64
            int prefix = 0;
            bool open = true, poisoned = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t tag = tag_of(gv[q]);
                poisoned = poisoned || (q < cnt && __any(tag == (epoch | POISON)));
                const unsigned long long ok = __ballot(tag == epoch);
                if (q < cnt) ring[((b0 + q) * 64 + lane) & (RING - 1)] = (int32_t)(uint32_t)gv[q];
                if (AFF && q < cnt)
                    ring_f[((b0 + q) * 64 + lane) & (RING - 1)] = (int32_t)(uint32_t)gv[q] + (int)((uint32_t)(gv[q] >> 32) & 0xffu) - 1;
                if (q < cnt && open) {
                    const int lead = ok == ~0ull ? 64 : (int)__builtin_ctzll(~ok);
                    prefix += lead;
                    open = lead == 64;
                }
            }
            if (poisoned) {
                st(&fail_s, 1u);
                break;
            }
            int upto = min(b0 * 64 + prefix, m);
            if (upto < m) upto &= ~15;  // whole sub-blocks of 16 columns
            if (upto > done) {
                done = upto;
                publish(&prog_s[first_ring], (uint32_t)done);
                idle = 0;
            } else {
                if (++idle > (1 << 22)) {
                    st(&fail_s, 1u);
                    break;
                }
                if (ld(&fail_s)) break;
                // back off while nothing arrives: a workgroup far behind the front polls for milliseconds
                for (int z = 0; z < min(idle, 4); ++z) __builtin_amdgcn_s_sleep(8);
            }
        }
        if (ld(&fail_s)) publish(&prog_s[first_ring], 0x7fffffffu);  // nobody waits for this ring any more
        return;
    }

    // ---------------- compute waves ----------------
    const int s = S * W + w - pad;
    if (s < 0) return;  // phantom strip above the matrix
    const uint8_t *code2 = reinterpret_cast<const uint8_t *>(code2s);
    BaLongWaveLds &L = wl[w];
    const int32_t *ring_in = rings[w];
    int32_t *ring_out = rings[w + 1];
    const int32_t *ring_f_in = rings_f[AFF ? w : 0];
    int32_t *ring_f_out = rings_f[AFF ? w + 1 : 0];
    uint32_t *prog_in = &prog_s[w], *cons_in = &cons_s[w], *prog_out = &prog_s[w + 1], *cons_out = &cons_s[w + 1];
    const bool last_wave = (w == W - 1);  // my bottom row goes to the workgroup below (and to the traceback)

    // bottom-aligned strips: strip s covers rows i0+1 .. i0+RS; rows <= 0 (first strip only)
    // carry the padding code and reproduce row 0 (see kernels.hip)
    const int i0 = n - (strips - s) * RS;
    const int32_t *left_col = P.left_col;  // column stripes: handed over by the stripe to the left
    auto first_col = [&](int i) {
        if (AFF) return aff_col0(i) - cc;  // hq = H' - c
        if (MULTI && P.left_gran) return 0;  // (comes through the granules below)
        return left_col ? left_col[max(i, 0)] : b_left * max(i, 0);
    };
    int h[T];
    // strip profile: prof[b][lane] = the scores of my T rows against column symbol b, one signed byte per row
    // (score + 2g; rows above the matrix carry the padding code: -128 against everything)
    // (one row more than there are column symbols: row nb scores -128 against every row -- the columns left of the matrix
    // that a lane passes before its first real one, see `steady` below)
    uint32_t *prof = dyn_lds + (size_t)w * (P.nb + 1) * 64 + lane;
    int left_top = first_col(i0);  // Hs(i0, 0): the diagonal of my first row's first column
    {
        int code[T];
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = i0 + lane * T + r + 1;
            code[r] = i >= 1 ? (int)P.code1[P.s1[i - 1]] : P.pad_code;
            h[r] = first_col(i);
        }
        if (MULTI && P.left_gran) {
            // my rows of the left neighbour's last column, as soon as its strip at my height has finished (bounded)
            const unsigned long long *lg = P.left_gran;
            bool bad = false;
            for (int spin = 0;; ++spin) {
                bool ok = true;
                const unsigned long long g0 = __hip_atomic_load(lg + max(i0, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (uint32_t)(g0 >> 32) == epoch;
                bad = bad || (uint32_t)(g0 >> 32) == (epoch | BA_LONG_POISON);
                left_top = (int)(uint32_t)g0;
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    const int i = i0 + lane * T + r + 1;
                    const unsigned long long gr = __hip_atomic_load(lg + max(i, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (uint32_t)(gr >> 32) == epoch;
                    bad = bad || (uint32_t)(gr >> 32) == (epoch | BA_LONG_POISON);
                    h[r] = (int)(uint32_t)gr;
                }
                if (__all(ok)) break;
                if (__any(bad) || spin >= (1 << 22) || ld(&fail_s)) {
                    st(&fail_s, 1u);
                    break;
                }
                for (int z = 0; z < min(spin + 1, 4); ++z) __builtin_amdgcn_s_sleep(8);
            }
            // the plain copy the walk reads (P.left_col is this stripe's own buffer)
            int32_t *lp = const_cast<int32_t *>(P.left_col);
#pragma unroll
            for (int r = 0; r < T; ++r) {
                const int i = i0 + lane * T + r + 1;
                if (i >= 1) lp[i] = h[r];
            }
            if (s == 0 && lane == 0) lp[0] = left_top;
        }
        const int8_t *tab = reinterpret_cast<const int8_t *>(P.rowprof);  // [na + 1][nb] signed bytes
        for (int b = 0; b < P.nb; ++b) {
            uint32_t word = 0;
#pragma unroll
            for (int r = 0; r < T; ++r) word |= (uint32_t)(uint8_t)tab[code[r] * P.nb + b] << (8 * r);
            prof[b * 64] = word;
        }
        prof[P.nb * 64] = 0x80808080u;
        L.ohw[lane] = 256u * (uint32_t)P.nb;  // the 64 columns before column 1 (block 0 reads them as its previous block's symbols)
    }
    // (checkpointed runs: the first pass stores no residues; a row block's pass stores them in the block's own buffer)
    constexpr bool store_resid = !NORES;
    uint32_t *resid = P.resid + ((ptrdiff_t)s - (ptrdiff_t)P.first_strip) * (ptrdiff_t)P.words_per_strip * 64 + lane;

    // skew state: `bottom` = my last row after the previous step; the lane below me reads it
    int bottom = h[T - 1];
    // score above my first row in the column BEFORE my current one (the diagonal of my first row);
    // before the first step that is column 0 of the row above: my upper neighbour's initial bottom
    int topprev = ba_wave_shr1(left_top, bottom);
    uint32_t acc = 0;
    int lmax = 0;  // local: the largest score of my rows so far
    // affine: E' of my rows (align_affine.go:45-46: -inf in column 0), F' of my last row after the previous step, and
    // (local) the best cell of every row
    int e[T], rmax[T], rstep[T];
#pragma unroll
    for (int r = 0; r < T; ++r) {
        e[r] = BA_LONG_NEG;
        rmax[r] = 0;
        rstep[r] = 0;
    }
    int bottom_f = BA_LONG_NEG;
    const int fl_const = ge * (i0 + lane * T - lane + 1);  // zero level of the row above my first one at step t: this + ge * t

    const uint32_t *oh_me = L.ohw + 64 - lane;  // my window slot for step 0 of a block

    auto load_symbols = [&](int b) -> uint32_t {  // column b*64 + lane (0-based): where its scores lie in a lane's profile
        const int col = b * 64 + lane;
        return col < m ? 256u * (uint32_t)code2[P.s2[col]] : 0u;  // (byte offset of prof[code][0]; beyond the matrix: anything valid)
    };
    uint32_t ohnext = load_symbols(0);
    uint32_t pseen = 0;  // columns of the row above known to be final
    int sent = 0;        // last wave: columns of my bottom row sent to HBM so far (a multiple of 16 until the end)
    // last wave: columns [sent, upto) of ring_out are final; send them on, 16 granules (one 128-byte line) per store
    auto send = [&](int upto, uint32_t tag) {
        for (; sent + 16 <= upto || (upto == m && sent < m); sent += 16) {
            const int col = sent + (lane & 15);
            if (lane < 16 && col < m) {
                const int v = ring_out[col & (RING - 1)];
                const int fv = AFF ? ring_f_out[col & (RING - 1)] : 0;
                __hip_atomic_store(bnd_out + col, mk_gran(tag, v, fv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    for (int b = 0; b < nblk; ++b) {
        const int t0 = b * 64;
        // ---- inputs of this block ----
        L.ohw[64 + lane] = ohnext;
        if (b + 1 < nblk) ohnext = load_symbols(b + 1);
        const int32_t *tin_base = ring_in + (t0 & (RING - 1));
        const int32_t *tinf_base = ring_f_in + (AFF ? (t0 & (RING - 1)) : 0);
        uint32_t *resid_blk = resid + (size_t)(t0 / CPD) * 64;
        // All lanes inside their column range for the whole block?  Linear gaps: block 0 counts too.  Lane l spends its
        // first l steps left of the matrix, where nothing needs masking: those columns carry a symbol that scores -128
        // against every row, so a cell there is max(up, left) = left -- in the biased space a score never falls from one
        // row to the next (local: all of column 0 is 0) -- and the lane still holds its column-0 scores when column 1
        // comes, with the lane above's column-0 score as that cell's diagonal.  (Worth 0.05 of 8.15 ms at 100k x 100k: the
        // first steps of a strip are slow because it trails the strip above closely, not because of the masks,
        // profiles/r03_notes.md.)
        const bool steady = (AFF ? b >= 1 : true) && (t0 + 64 <= m);
        // lane 63 writes column t0 + k - 63 at step k: slots of block b-1 for k < 63, the first slot of block b for k = 63
        // (the other lanes write to the scratch row).  The slots held columns RING further left: their reader must be done.
        if (!last_wave && t0 + 1 > RING) wait_lds(cons_out, (uint32_t)(t0 + 1 - RING));
        int32_t *wp = (lane == 63) ? ring_out + ((t0 - 63) & (RING - 1)) : L.dummy + lane;
        int32_t *wp63 = (lane == 63) ? ring_out + (t0 & (RING - 1)) : L.dummy + lane;
        int32_t *wpf = (AFF && lane == 63) ? ring_f_out + ((t0 - 63) & (RING - 1)) : L.dummy_f + lane;
        int32_t *wpf63 = (AFF && lane == 63) ? ring_f_out + (t0 & (RING - 1)) : L.dummy_f + lane;

        constexpr int TN = AFF ? 32 : 16;  // affine: tin[16 + k] = F' of the row above
        uint32_t ohA[16], ohB[16], pwA[16], pwB[16];
        int tinA[TN], tinB[TN];
        // columns j*16 .. j*16+15 of the row above must be final (columns >= m never are: nobody reads them)
        auto need_cols = [&](int j) -> uint32_t { return (uint32_t)min(t0 + j * 16 + 16, m); };
        auto fetch = [&](uint32_t (&oh)[16], int (&tin)[TN], int j) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                oh[k] = oh_me[j * 16 + k];
                tin[k] = tin_base[j * 16 + k];
                if (AFF) tin[(AFF ? 16 : 0) + k] = tinf_base[j * 16 + k];
            }
        };
        // the scores of my rows against the 16 column symbols of a sub-block: one dword of my strip profile each
        auto scores = [&](uint32_t (&pw)[16], const uint32_t (&oh)[16]) {
#pragma unroll
            for (int k = 0; k < 16; ++k) pw[k] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(prof) + oh[k]);
        };
        auto compute = [&](const uint32_t (&pw)[16], const int (&tin)[TN], int j) {
            if (AFF) {
                auto steps = [&](auto check) {
                    constexpr bool CHECK = decltype(check)::value;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int t = t0 + j * 16 + k;
                        if (!CHECK || t < total_steps) {
                            const int top = ba_wave_shr1(tin[k], bottom);
                            const int ftop = ba_wave_shr1(tin[(AFF ? 16 : 0) + k], bottom_f);
                            const bool active = !CHECK || ((t >= lane) && (t - lane < m));
                            int fout;
                            ba_long_step_aff<T, CHECK, MODE>(h, e, acc, top, ftop, topprev, pw[k], active, cc, ge, fl_const + ge * t,
                                                             rmax, rstep, t, fout);
                            topprev = top;
                            bottom = h[T - 1];
                            bottom_f = fout;
                            ba_lds_store((j * 16 + k == 63) ? wp63 : wp + j * 16 + k, bottom);
                            ba_lds_store((j * 16 + k == 63) ? wpf63 : wpf + j * 16 + k, bottom_f);
                            if ((k + 1) % CPD == 0 && store_resid) resid_blk[(size_t)((j * 16 + k) / CPD) * 64] = acc;
                        }
                    }
                };
                if (steady)
                    steps(std::false_type{});
                else
                    steps(std::true_type{});
            } else if (steady) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int top = ba_wave_shr1(tin[k], bottom);
                    ba_long_step<T, K, false, MODE == BA_LOCAL>(h, acc, top, topprev, pw[k], true, g, lmax);
                    topprev = top;
                    bottom = h[T - 1];
                    ba_lds_store((j * 16 + k == 63) ? wp63 : wp + j * 16 + k, bottom);
                    if ((k + 1) % CPD == 0 && store_resid) resid_blk[(size_t)((j * 16 + k) / CPD) * 64] = acc;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int t = t0 + j * 16 + k;
                    if (t < total_steps) {
                        const int top = ba_wave_shr1(tin[k], bottom);
                        const bool active = (t >= lane) && (t - lane < m);
                        ba_long_step<T, K, true, MODE == BA_LOCAL>(h, acc, top, topprev, pw[k], active, g, lmax);
                        topprev = top;
                        bottom = h[T - 1];
                        ba_lds_store((j * 16 + k == 63) ? wp63 : wp + j * 16 + k, bottom);
                        if ((k + 1) % CPD == 0 && store_resid) resid_blk[(size_t)((j * 16 + k) / CPD) * 64] = acc;
                    }
                }
            }
            // after step t every column up to t - 63 of my bottom row is final
            const int fin = min(max(t0 + j * 16 + 15 - 62, 0), m);
            if (last_wave)
                send(fin, epoch);
            else
                publish(prog_out, (uint32_t)fin);
        };
        // sub-block j with its inputs in (pw, tin); the inputs of sub-block j+1 are fetched into (ohn, tinn) before the
        // steps when the row above is known to be far enough, after them (waiting) otherwise, and its scores after the
        // steps (their addresses are the symbols just fetched)
        auto stage = [&](const uint32_t (&pw)[16], const int (&tin)[TN], uint32_t (&ohn)[16], int (&tinn)[TN],
                         uint32_t (&pwn)[16], int j, bool more) {
            // (compiler barriers, not fences: what orders the ring reads behind the counter read is the LDS itself,
            // which carries out one wavefront's operations in the order they were issued)
            asm volatile("" ::: "memory");
            const uint32_t fresh = ld(prog_in);  // (in flight during the steps)
            const bool early = more && pseen >= need_cols(j + 1);
            if (early) fetch(ohn, tinn, j + 1);
            compute(pw, tin, j);
            pseen = max(pseen, fresh);
            if (more && !early) {
                if (pseen < need_cols(j + 1)) pseen = max(pseen, wait_lds(prog_in, need_cols(j + 1)));
                asm volatile("" ::: "memory");
                fetch(ohn, tinn, j + 1);
            }
            if (more) scores(pwn, ohn);
        };
        if (pseen < need_cols(0)) pseen = max(pseen, wait_lds(prog_in, need_cols(0)));
        asm volatile("" ::: "memory");
        if (b == 0 && P.stamps && lane == 0) P.stamps[2 * (size_t)s] = wall_clock64();
        fetch(ohA, tinA, 0);
        scores(pwA, ohA);
        stage(pwA, tinA, ohB, tinB, pwB, 0, true);
        stage(pwB, tinB, ohA, tinA, pwA, 1, true);
        stage(pwA, tinA, ohB, tinB, pwB, 2, true);
        stage(pwB, tinB, ohA, tinA, pwA, 3, false);
        // ---- end of the block ----
        publish(cons_in, (uint32_t)min(t0 + 64, m));  // (the values read are in registers or used up)
        {
            const uint32_t o = L.ohw[64 + lane];
            L.ohw[lane] = o;
        }
    }
    // last, partial residue word: its pushes sit at the top, bring them down to their places
    if (total_steps % CPD != 0 && store_resid)
        resid[(size_t)(total_steps / CPD) * 64] = acc >> ((CPD - total_steps % CPD) * T * K);
    // every column is final now
    if (P.stamps && lane == 0) P.stamps[2 * (size_t)s + 1] = wall_clock64();
    const bool failed = ld(&fail_s) != 0;
    if (last_wave) {
        send(m, failed ? (epoch | POISON) : epoch);
        if (failed) {  // nobody below may wait for what was sent with a good tag before the failure either
            for (int col = lane; col < m; col += 64)
                __hip_atomic_store(bnd_out + col, mk_gran(epoch | POISON, 0, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) atomicAdd(P.status, 1u);
        }
    } else {
        publish(prog_out, (uint32_t)m);
    }
    if (MODE == BA_LOCAL && !AFF) P.lane_max[(size_t)s * 64 + lane] = lmax;
    if (MODE == BA_LOCAL && AFF) {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            // (rows above the matrix score -ext*i against their zero level: they are not cells)
            if (i0 + lane * T + r + 1 < 1) rmax[r] = 0;
            P.lane_max[(((size_t)s * 64 + lane) * T + r) * 2 + 0] = rmax[r];
            P.lane_max[(((size_t)s * 64 + lane) * T + r) * 2 + 1] = rstep[r];
        }
    }
    // column stripes: my last column is the next stripe's column 0
    if (P.right_col) {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = i0 + lane * T + r + 1;
            if (i >= 1) P.right_col[i] = h[r];
        }
        if (s == 0 && lane == 0) P.right_col[0] = b_top * (col0 + m);
    }
    if (MULTI && P.right_gran) {
        const unsigned long long tag = (unsigned long long)(failed ? (epoch | BA_LONG_POISON) : epoch) << 32;
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = i0 + lane * T + r + 1;
            if (i >= 1) __hip_atomic_store(P.right_gran + i, tag | (uint32_t)h[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (s == 0 && lane == 0)
            __hip_atomic_store(P.right_gran, tag | (uint32_t)(b_top * (col0 + m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ----------------------------------------------------------------------------
// traceback, long mode, in two kernels.
//
// (1) The WALK.  In the working space a cell is cur = max3(up, left, diag + w) with the
// reference's order up > left > diagonal on ties (align.go:105-116), so
//     diagonal  <=>  cur > up and cur > left
//     left      <=>  cur == left and cur > up
//     up        <=>  otherwise
// and cur - up, cur - left lie in 0 .. max(w) + 2g < 2^K: both tests read off the K-bit
// residues alone, (rho(cur) - rho(up)) mod 2^K != 0.  The walk therefore needs no scores,
// no symbols and no score table: three residues per move (border cells have the residues
// of their closed-form scores).  One wavefront walks (all lanes the same walk: uniform
// control flow, LDS broadcasts) over residue tiles staged in LDS -- the tile of the
// current strip (64 residue words x 64 lanes = one contiguous 16 KB block of HBM) and, for
// the rows that touch the strip above, that strip's last lane -- and emits one 2-bit move
// code per step, 16 to a dword.
//
// (2) The EXPANSION of the move codes into the two gapped byte strings (align.go:126-143;
// align_pair.go:197-216) is a prefix sum: one block, 1024 threads.
// ----------------------------------------------------------------------------
// (BA_MV_UP / LEFT / DIAG: device_helpers.h)

// Local mode: the end cell is the first maximum in row-major order (align.go:216-221).  The DP left the largest
// score of every lane's T rows (lane_max, in row order); this kernel finds the maximum M and the first lane group that
// holds it, then rebuilds the exact scores of that group's rows, one row after the other, from the residues --
// H(i, 0) = 0 and a horizontal step is ((rho(j) - rho(j-1) + g) mod 2^K) - g -- as a block-wide prefix sum over the
// columns, until a row shows M: its first such column is the end cell.  One block; leaves (i, j, M) in P.walk.
template <int T, int K>
__global__ void __launch_bounds__(1024) ba_long_local_find_kernel(BaLongParams P)
{
    using G = BaLongGeom<T, K>;
    constexpr int RS = G::RS, CPD = G::CPD;
    constexpr uint32_t MASK = (1u << K) - 1u;
    __shared__ int red_v[16];
    __shared__ uint32_t red_k[16];
    __shared__ int wsum[16];
    __shared__ int found_s[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (int)P.n, m = (int)P.m, g = P.gap;
    const int strips = (int)P.strips;
    const int wps = (int)P.words_per_strip;
    // the maximum and the first lane group (strip * 64 + lane: row order) that holds it
    int best = 0;
    uint32_t bk = 0xffffffffu;
    for (uint32_t k = (uint32_t)tid; k < (uint32_t)strips * 64u; k += 1024u) {
        const int v = P.lane_max[k];
        if (v > best || (v == best && k < bk)) {
            best = v;
            bk = k;
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int ov = __shfl_xor(best, d);
        const uint32_t ok = (uint32_t)__shfl_xor((int)bk, d);
        if (ov > best || (ov == best && ok < bk)) {
            best = ov;
            bk = ok;
        }
    }
    if (lane == 0) {
        red_v[wave] = best;
        red_k[wave] = bk;
    }
    __syncthreads();
    best = red_v[0];
    bk = red_k[0];
    for (int q = 1; q < 16; ++q)
        if (red_v[q] > best || (red_v[q] == best && red_k[q] < bk)) {
            best = red_v[q];
            bk = red_k[q];
        }
    const int M = best;
    int fi = 0, fj = 0;
    if (M > 0 && !P.no_resid) {  // (score only: no residues were stored, and nobody asks where the maximum lies)
        const int s = (int)(bk >> 6), l = (int)(bk & 63u);
        const int i0 = n - (strips - s) * RS;
        const int chunk = (m + 1023) / 1024;
        const int c_lo = tid * chunk + 1, c_hi = min(c_lo + chunk - 1, m);  // my columns (1-based); empty if c_lo > m
        for (int r = 0; r < T && fi == 0; ++r) {
            const int i = i0 + l * T + r + 1;
            if (i < 1) continue;  // (uniform) a row above the matrix
            const uint32_t *row = P.resid + (size_t)(s - P.first_strip) * wps * 64 + l;  // (a recomputed block holds strips first_strip ..)
            auto rho = [&](int j) -> uint32_t {  // residue of H(i, j), j >= 0
                if (j == 0) return 0u;
                const int t = (j - 1) + l;
                return (row[(size_t)(t / CPD) * 64] >> (uint32_t)(((t % CPD) * T + r) * K)) & MASK;
            };
            int sum = 0;
            if (c_lo <= m) {
                uint32_t prev = rho(c_lo - 1);
                for (int j = c_lo; j <= c_hi; ++j) {
                    const uint32_t cur = rho(j);
                    sum += (int)((cur - prev + (uint32_t)g) & MASK) - g;
                    prev = cur;
                }
            }
            // exclusive prefix of the chunk sums over the block
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int a = __shfl_up(incl, d);
                if (lane >= d) incl += a;
            }
            __syncthreads();  // (wsum / found_s of the previous row have been read)
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = incl - sum;
            for (int q = 0; q < wave; ++q) base += wsum[q];
            int found = 0x7fffffff;
            if (c_lo <= m) {
                int v = base;
                uint32_t prev = rho(c_lo - 1);
                for (int j = c_lo; j <= c_hi; ++j) {
                    const uint32_t cur = rho(j);
                    v += (int)((cur - prev + (uint32_t)g) & MASK) - g;
                    prev = cur;
                    if (v == M && found == 0x7fffffff) found = j;
                }
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) found = min(found, __shfl_xor(found, d));
            if (lane == 0) found_s[wave] = found;
            __syncthreads();
            found = found_s[0];
            for (int q = 1; q < 16; ++q) found = min(found, found_s[q]);
            if (found != 0x7fffffff) {
                fi = i;
                fj = found;
            }
        }
    }
    if (tid == 0) {
        *P.score = (int64_t)M;
        P.walk[0] = (uint32_t)fi;
        P.walk[1] = (uint32_t)fj;
        P.walk[2] = 0;
        P.walk[3] = (uint32_t)M;
        P.walk[4] = (uint32_t)fi;
        P.walk[5] = (uint32_t)fj;
        P.walk[6] = bk;  // the lane group (strip * 64 + lane): the linear-space traceback recomputes the rows down to it first
        if (M > 0 && fi == 0 && !P.no_resid) atomicAdd(P.status, 1u);  // (cannot happen: some row of the group holds M)
    }
}

// CKPT: the walk of one recomputed row block of the linear-space traceback (first_strip, stop_row).
template <int MODE, int T, int K, bool CKPT = false>
__global__ void __launch_bounds__(256) ba_long_walk_kernel(BaLongParams P)
{
    using G = BaLongGeom<T, K>;
    constexpr int RS = G::RS, CPD = G::CPD;
    constexpr uint32_t MASK = (1u << K) - 1u;
    // Four wavefronts: wave 0 walks, waves 1-3 load residue words for it -- a tile of BA_WALK_TW words x 64 lanes of a
    // strip (a strip's whole passage fits one) with lane 63 of the strip above around the column where the walk will
    // cross into it.  One CU pulls cold lines at about 12 GB/s (a 32 KB tile: 3-4 us), so the tile of the NEXT strip is
    // fetched into a second buffer while the walk is still in the current one: where the walk will enter it follows
    // from its diagonal, and a walk that drifts out of the prediction falls back to a load it waits for.
    extern __shared__ __align__(16) uint32_t walk_lds[];
    constexpr int TW = BA_WALK_TW;
    uint32_t *buf0 = walk_lds, *buf1 = walk_lds + (TW * 64 + TW);  // two buffers: tile [TW][64], then above [TW]
    // mailbox: [0] sequence number of the request, [1] what to load (0: leave), [2..5] where from, [6] buffer, [7] done count
    int *mail = reinterpret_cast<int *>(walk_lds + 2 * (TW * 64 + TW));
    uint32_t *tile = buf0, *above = buf0 + TW * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)P.n, m = (int)P.m, g = P.gap;
    const int strips = (int)P.strips;
    const int front = strips * RS - n;
    const int wps = (int)P.words_per_strip;
    if (tid < 8) mail[tid] = 0;
    __syncthreads();  // (the only barrier)

    const int b_top = (MODE == BA_FITTING) ? g : 0;
    const int b_left = (MODE == BA_OVERLAP) ? g : 0;
    constexpr bool top_free = (MODE == BA_FITTING || MODE == BA_LOCAL);
    constexpr bool left_free = (MODE == BA_OVERLAP || MODE == BA_LOCAL);
    // local: true scores; a step between neighbours is never below -g (kernels.hip); other modes: biased space, never below 0
    const uint32_t off = (MODE == BA_LOCAL) ? (uint32_t)g : 0u;
    // the last strip is the last wave of the last workgroup; its bottom row lies in HBM as {score, tag} granules
    const int32_t *bottom_row = P.bnd + (size_t)(P.nsuper - 1) * P.bnd_stride * 2;
    // column stripes: columns are the stripe's own (1..m), column 0 belongs to the stripe on the left
    const int col0 = (int)P.col0;
    const int32_t *left_col = P.left_col;
    const bool first_stripe = (left_col == nullptr);
    // checkpointed runs: resid holds the strips from first_strip on; the walk stops at stop_row, the last row above them,
    // whose scores (hence residues) lie in the bottom row of the workgroup above the block
    const int first_strip = CKPT ? P.first_strip : 0, stop_row = CKPT ? P.stop_row : 0;
    const int32_t *row_above = P.bnd + (size_t)(P.wg_begin > 0 ? P.wg_begin - 1 : 0) * P.bnd_stride * 2;

    // end cell (align.go:123-124; align_pair.go:169-182): n >= 1 and m >= 1 here
    int i = n, j = m, cur = 0, bv = 0;
    if (MODE == BA_LOCAL) {
        // ba_long_local_find_kernel has found the end cell and written the score; a later block of the linear-space
        // traceback goes on from the cell the host names, with the score the block below left in walk[3]
        i = P.start_j >= 0 ? P.start_i : (int)P.walk[0];
        j = P.start_j >= 0 ? P.start_j : (int)P.walk[1];
        cur = bv = (int)P.walk[3];
    } else if (P.start_j >= 0) {
        // the walk comes in from the stripe on the right (or the host has chosen the end cell)
        i = P.start_i;
        j = P.start_j;
    } else if (MODE == BA_GLOBAL) {
        cur = bottom_row[2 * (m - 1)];
        bv = cur - g * (n + col0 + m);
    } else {
        // leftmost strict maximum of the true scores Hs(n, j) - g*j of the bottom row, from column 0
        // (column 0 of the matrix only: a later stripe's column 0 is the last column of its neighbour)
        int bj = 0, bc = b_left * n;
        bv = first_stripe ? b_left * n : INT32_MIN;
        for (int c0 = 0; c0 < m; c0 += 64) {
            const int col = c0 + lane;  // 0-based array index = column col+1
            const int hv = col < m ? bottom_row[2 * col] : 0;
            int v = col < m ? hv - g * (col0 + col + 1) : INT32_MIN;
            int jj = col + 1, cc = hv;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {  // wave argmax, smaller column wins ties
                const int ov = __shfl_xor(v, d), oj = __shfl_xor(jj, d), oc = __shfl_xor(cc, d);
                if (ov > v || (ov == v && oj < jj)) {
                    v = ov;
                    jj = oj;
                    cc = oc;
                }
            }
            if (v > bv) {
                bv = v;
                bj = jj;
                bc = cc;
            }
        }
        j = bj;
        cur = bc;
        bv = (bv == INT32_MIN) ? INT32_MIN : bv - g * n;  // true score H(n, j)
    }
    i = __builtin_amdgcn_readfirstlane(i);
    j = __builtin_amdgcn_readfirstlane(j);
    cur = __builtin_amdgcn_readfirstlane(cur);
    bv = __builtin_amdgcn_readfirstlane(bv);
    if (tid == 0) {
        if (P.start_j < 0 && MODE != BA_LOCAL) *P.score = (int64_t)cur - (int64_t)g * (i + col0 + j);
        P.walk[0] = (uint32_t)i;  // where the walk starts
        P.walk[1] = (uint32_t)j;
        P.walk[2] = 0;
        P.walk[3] = (uint32_t)bv;
        P.walk[4] = (uint32_t)i;
        P.walk[5] = (uint32_t)j;
    }
    if (P.score_only || P.scan_only) return;

    // residue tiles
    int tile_s = -1, tile_u = 0;   // strip and first word index of `tile`
    int ab_s = -1, ab_u = 0;       // strip and first word index of `above` (lane 63 of that strip)
    auto mload = [](int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (wave != 0) {
        // loaders: take one request after the other from the mailbox (the arguments are written before the sequence
        // number, the words before the done count: one wavefront's LDS operations are carried out in order)
        int seen = 0;
        for (;;) {
            int sq;
            while ((sq = mload(&mail[0])) == seen) __builtin_amdgcn_s_sleep(2);
            seen = sq;
            asm volatile("" ::: "memory");
            const int op = mail[1];
            if (op == 0) break;
            uint32_t *tb = mail[6] ? buf1 : buf0;
            if (op & 1) {
                const int sx = mail[2], tu = mail[3];
                const uint32_t *src = P.resid + ((size_t)(sx - first_strip) * wps + tu) * 64;
                const int words = min(TW, wps - tu);
                constexpr int PER = (TW + 2) / 3;
                uint32_t v[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) v[q] = src[(size_t)min((wave - 1) * PER + q, words - 1) * 64 + lane];
#pragma unroll
                for (int q = 0; q < PER; ++q)
                    if ((wave - 1) * PER + q < TW) tb[((wave - 1) * PER + q) * 64 + lane] = v[q];
            }
            if (op & 2) {
                const int sx = mail[4], au = mail[5];
                for (int x = tid - 64; x < TW; x += 192) {
                    const int uu = au + x;
                    tb[TW * 64 + x] = uu < wps ? P.resid[((size_t)(sx - first_strip) * wps + uu) * 64 + 63] : 0u;
                }
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&mail[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    uint32_t n_requests = 0, n_rounds = 0;  // (synchronous requests and rounds, reported in P.walk[6], [7]: BIOALIGN_TIMING=2 prints them)
    int posted = 0;  // requests posted so far
    int cbuf = 0;    // the buffer the walk reads
    bool pf_valid = false;  // the other buffer is being filled with (pf_s, pf_tu) and (pf_as, pf_au)
    int pf_s = -1, pf_tu = 0, pf_as = -1, pf_au = 0;
    auto post = [&](int op, int b, int sx, int tu, int as, int au) {
        ++posted;
        if (lane == 0) {
            mail[1] = op;
            mail[2] = sx;
            mail[3] = tu;
            mail[4] = as;
            mail[5] = au;
            mail[6] = b;
            asm volatile("" ::: "memory");
            __hip_atomic_store(&mail[0], posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto wait_all = [&]() {
        while (mload(&mail[7]) != 3 * posted) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    auto use_buffer = [&](int b) {
        cbuf = b;
        tile = b ? buf1 : buf0;
        above = tile + TW * 64;
    };
    auto clampw = [&](int u0) { return max(0, min(u0, wps - TW)); };
    auto window = [&](int u) { return clampw(u - (TW - 9)); };  // the walk moves towards smaller words
    // the column the walk reaches the last row of strip sx - 1 in if it keeps to its diagonal
    auto cross_col = [&](int sx) { return max(j - (i - (sx * RS - front)), 1); };
    auto entry_word = [&](int jc) { return ((jc - 1) + 63) / CPD; };  // lane 63 at column jc
    auto load_tile = [&](int sx, int u) {
        wait_all();
        if (pf_valid && pf_s == sx && u >= pf_tu && u < pf_tu + TW) {
            use_buffer(cbuf ^ 1);  // the words are there already
            tile_s = pf_s;
            tile_u = pf_tu;
            ab_s = pf_as;
            ab_u = pf_au;
        } else {
            ++n_requests;
            tile_s = sx;
            tile_u = window(u);
            int op = 1;
            if (sx - 1 >= first_strip && sx >= 1) {  // with lane 63 of the strip above, around where the walk will cross
                ab_s = sx - 1;
                ab_u = clampw(entry_word(cross_col(sx)) - TW / 2);
                op |= 2;
            }
            post(op, cbuf ^ 1, tile_s, tile_u, ab_s, ab_u);
            wait_all();
            use_buffer(cbuf ^ 1);
        }
        pf_valid = false;
        if (sx - 1 >= first_strip && sx >= 1) {
            // meanwhile the loaders fetch the tile of the strip above into the other buffer
            const int jc = cross_col(sx);
            pf_s = sx - 1;
            pf_tu = clampw(entry_word(jc) - (TW - 24));
            int op = 1;
            pf_as = -1;
            pf_au = 0;
            if (sx - 2 >= first_strip && sx >= 2) {
                pf_as = sx - 2;
                pf_au = clampw(entry_word(max(jc - RS, 1)) - TW / 2);
                op |= 2;
            }
            post(op, cbuf ^ 1, pf_s, pf_tu, pf_as, pf_au);
            pf_valid = true;
        }
    };
    auto load_above = [&](int sx, int u) {
        ++n_requests;
        wait_all();
        ab_s = sx;
        ab_u = window(u);
        post(2, cbuf, 0, 0, ab_s, ab_u);
        wait_all();
    };
    // residue of cell (ii, jj), 0 <= ii <= n, 0 <= jj <= m
    auto residue = [&](int ii, int jj) -> uint32_t {
        if (ii == 0) return (uint32_t)(b_top * (col0 + jj)) & MASK;  // Hs(0, j) = b_top * j
        if (jj == 0) return (uint32_t)(left_col ? left_col[ii] : b_left * ii) & MASK;  // Hs(i, 0) = b_left * i
        const int tt = ii - 1 + front;
        const int s = tt / RS, l = (tt % RS) / T, r = tt % T;
        if (CKPT && s < first_strip) return (uint32_t)row_above[2 * (jj - 1)] & MASK;  // (row stop_row: nothing above it is asked for)
        const int t = (jj - 1) + l;
        const int u = t / CPD;
        const uint32_t sh = (uint32_t)(((t % CPD) * T + r) * K);
        uint32_t word;
        if (s == tile_s || tile_s < 0 || l != 63) {
            if (s != tile_s || u < tile_u || u >= tile_u + TW) load_tile(s, u);
            word = tile[(u - tile_u) * 64 + l];
        } else {
            // a cell of the strip above the one the walk is in: always its last lane
            if (s != ab_s || u < ab_u || u >= ab_u + TW) load_above(s, u);
            word = above[u - ab_u];
        }
        // every lane read the same word: say so, and the whole walk runs on the scalar unit
        word = (uint32_t)__builtin_amdgcn_readfirstlane((int)word);
        return (word >> sh) & MASK;
    };

    // The same lookup without loading anything: per-lane cell, says whether the staged tiles hold it
    auto peek = [&](int ii, int jj, bool &ok) -> uint32_t {
        if (ii == 0) return (uint32_t)(b_top * (col0 + jj)) & MASK;
        if (jj == 0) return (uint32_t)(left_col ? left_col[ii] : b_left * ii) & MASK;
        const int tt = ii - 1 + front;
        const int s = tt / RS, l = (tt % RS) / T, r = tt % T;
        if (CKPT && s < first_strip) {
            if (ii != stop_row) {
                ok = false;
                return 0u;
            }
            return (uint32_t)row_above[2 * (jj - 1)] & MASK;
        }
        const int t = (jj - 1) + l;
        const int u = t / CPD;
        const uint32_t sh = (uint32_t)(((t % CPD) * T + r) * K);
        uint32_t word = 0;
        if (s == tile_s || l != 63) {
            const bool in = (s == tile_s) && u >= tile_u && u < tile_u + TW;
            ok = ok && in;
            word = tile[(in ? (u - tile_u) : 0) * 64 + l];
        } else {
            const bool in = (s == ab_s) && u >= ab_u && u < ab_u + TW;
            ok = ok && in;
            word = above[in ? (u - ab_u) : 0];
        }
        return (word >> sh) & MASK;
    };

    // move codes: 64-bit accumulator, flushed two dwords at a time
    uint32_t *ops = P.ops;
    uint64_t opacc = 0;
    uint32_t fill = 0;   // codes in opacc
    uint32_t k = 0;      // codes flushed so far
    auto emit_run = [&](uint32_t mv, uint32_t count) {
        const uint64_t pattern = mv == BA_MV_DIAG ? 0xAAAAAAAAAAAAAAAAull : mv == BA_MV_LEFT ? 0x5555555555555555ull : 0ull;
        while (count > 0) {
            const uint32_t c = min(count, 32u - fill);
            const uint64_t bits = c == 32 ? pattern : (pattern & ((1ull << (2 * c)) - 1ull));
            opacc |= bits << (2 * fill);
            fill += c;
            count -= c;
            if (fill == 32) {
                if (lane == 0) {
                    ops[k >> 4] = (uint32_t)opacc;
                    ops[(k >> 4) + 1] = (uint32_t)(opacc >> 32);
                }
                k += 32;
                opacc = 0;
                fill = 0;
            }
        }
    };
    auto emit = [&](uint32_t mv) { emit_run(mv, 1); };

    // Lane q looks at cell (i - q, j - q): the cells of a diagonal run are evaluated together, and
    // one round of LDS reads advances the walk by the whole run plus the move that ends it.
    while (i > stop_row && j > 0) {
        ++n_rounds;
        const int ii = i - lane, jj = j - lane;
        bool ok = ii > stop_row && jj >= 1;
        if (MODE != BA_LOCAL) {
            // The common round, with the index arithmetic of the three residues shared: cell (ii, jj) lies in the staged
            // tile, its left neighbour one step earlier in the same lane, its upper neighbour in the same word (same
            // lane, row above) or -- first row of a lane -- one step earlier in the lane above (`above` for lane 0).
            // Lanes this does not cover are left to the next round; when it does not cover the walk's own cell the
            // general code below takes the round.
            const int tt = ii - 1 + front;
            const int sx = tt / RS, rem = tt % RS, l = rem / T, r = rem % T;
            const int t = (jj - 1) + l, t1 = t - 1;
            const int u = t / CPD, f = t % CPD, u1 = t1 / CPD, f1 = t1 % CPD;
            const bool from_above = (r == 0 && l == 0);
            const int tu_ = from_above ? (jj - 1) + 63 : t1;  // the step the row above was at in this column
            const int uu_ = tu_ / CPD, fu_ = tu_ % CPD;
            const bool fast = ok && ii >= 2 && jj >= 2 && sx == tile_s && u1 >= tile_u && u < tile_u + TW &&
                              (!from_above || (sx - 1 >= first_strip && ab_s == sx - 1 && uu_ >= ab_u && uu_ < ab_u + TW));
            const uint64_t fmask = __ballot(fast);
            if (fmask & 1ull) {
                uint32_t wA = 0, w1 = 0, w2 = 0;
                if (fast) {
                    wA = tile[(u - tile_u) * 64 + l];
                    w1 = tile[(u1 - tile_u) * 64 + l];
                    w2 = r > 0 ? wA : from_above ? above[uu_ - ab_u] : tile[(uu_ - tile_u) * 64 + l - 1];
                }
                const uint32_t rA = (wA >> (uint32_t)((f * T + r) * K)) & MASK;
                const uint32_t rL = (w1 >> (uint32_t)((f1 * T + r) * K)) & MASK;
                const uint32_t rU = r > 0 ? (wA >> (uint32_t)((f * T + r - 1) * K)) & MASK
                                          : (w2 >> (uint32_t)((fu_ * T + T - 1) * K)) & MASK;
                const bool a_up = rA != rU, a_left = rA != rL;
                const uint64_t dmask = __ballot(fast && a_up && a_left);
                const uint32_t run = dmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~dmask);
                emit_run(BA_MV_DIAG, run);
                i -= (int)run;
                j -= (int)run;
                if (run < 64 && ((fmask >> run) & 1ull)) {
                    const uint64_t lmask = __ballot(!a_left && a_up);
                    if ((lmask >> run) & 1ull) {
                        emit(BA_MV_LEFT);
                        --j;
                    } else {
                        emit(BA_MV_UP);
                        --i;
                    }
                }
                continue;
            }
        }
        uint32_t rcq = 0, ruq = 0, rlq = 0, rdq = 0;
        bool have = false;
        if (MODE == BA_LOCAL) {
            // the common round of local mode: as above, plus the diagonal neighbour (one step before the upper one)
            const int tt = ii - 1 + front;
            const int sx = tt / RS, rem = tt % RS, l = rem / T, r = rem % T;
            const int t = (jj - 1) + l, t1 = t - 1;
            const int u = t / CPD, f = t % CPD, u1 = t1 / CPD, f1 = t1 % CPD;
            const bool from_above = (r == 0 && l == 0);
            const int tu_ = from_above ? (jj - 1) + 63 : t1;
            const int uu_ = tu_ / CPD, fu_ = tu_ % CPD;
            const int tx_ = tu_ - 1, ux_ = tx_ / CPD, fx_ = tx_ % CPD;
            const bool fast = ok && ii >= 2 && jj >= 2 && sx == tile_s && u1 >= tile_u && u < tile_u + TW &&
                              (!from_above || (ab_s == sx - 1 && ux_ >= ab_u && uu_ < ab_u + TW)) &&
                              (r > 0 || from_above || ux_ >= tile_u);
            if (__ballot(fast) & 1ull) {
                uint32_t wA = 0, w1 = 0, w2 = 0, w3 = 0;
                if (fast) {
                    wA = tile[(u - tile_u) * 64 + l];
                    w1 = tile[(u1 - tile_u) * 64 + l];
                    if (r == 0) {
                        w2 = from_above ? above[uu_ - ab_u] : tile[(uu_ - tile_u) * 64 + l - 1];
                        w3 = from_above ? above[ux_ - ab_u] : tile[(ux_ - tile_u) * 64 + l - 1];
                    }
                }
                rcq = (wA >> (uint32_t)((f * T + r) * K)) & MASK;
                rlq = (w1 >> (uint32_t)((f1 * T + r) * K)) & MASK;
                ruq = r > 0 ? (wA >> (uint32_t)((f * T + r - 1) * K)) & MASK : (w2 >> (uint32_t)((fu_ * T + T - 1) * K)) & MASK;
                rdq = r > 0 ? (w1 >> (uint32_t)((f1 * T + r - 1) * K)) & MASK : (w3 >> (uint32_t)((fx_ * T + T - 1) * K)) & MASK;
                ok = fast;
                have = true;
            }
        }
        if (!have && ok) {
            rcq = peek(ii, jj, ok);
            ruq = peek(ii - 1, jj, ok);
            rlq = peek(ii, jj - 1, ok);
            if (MODE == BA_LOCAL) rdq = peek(ii - 1, jj - 1, ok);
        }
        const uint64_t okmask = __ballot(ok);
        if (!(okmask & 1ull)) {
            // the walk's own cell is not staged: stage what it needs (uniform path), then look again
            (void)residue(i, j);
            (void)residue(i - 1, j);
            (void)residue(i, j - 1);
            if (MODE == BA_LOCAL) (void)residue(i - 1, j - 1);
            continue;
        }
        // local: cur = max(0, up - g, left - g, diag + w) > 0, up > left > diagonal (align.go:186-202): "cur > up - g" is
        // "the residues differ by something other than -g"; the other modes: off = 0
        const bool above_up = ((rcq - ruq + off) & MASK) != 0;    // cur > up (local: cur > up - g)
        const bool above_left = ((rcq - rlq + off) & MASK) != 0;  // cur > left (local: cur > left - g)
        const bool is_diag = ok && above_up && above_left;
        const uint32_t mvq = (!above_left && above_up) ? BA_MV_LEFT : BA_MV_UP;
        const uint64_t dmask = __ballot(is_diag);
        uint32_t run = dmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~dmask);
        if (MODE == BA_LOCAL) {
            // exact scores along the diagonal: H(ii, jj) - H(ii-1, jj-1) = (cur - up) + (up - diag), both steps known
            // from the residues; the walk ends at the first cell that scores 0 (align.go:224-226)
            const int d_up = (int)((rcq - ruq + off) & MASK) - g, d_left = (int)((rcq - rlq + off) & MASK) - g;
            const int dq = d_up + (int)((ruq - rdq + off) & MASK) - g;
            int incl = ok ? dq : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int a = __shfl_up(incl, d);
                if (lane >= d) incl += a;
            }
            const int hq = cur - (incl - (ok ? dq : 0));  // exact H of my cell, valid while every lane below me is a diagonal move
            const uint64_t zmask = __ballot(ok && hq == 0);
            const uint32_t stop = zmask ? (uint32_t)__builtin_ctzll(zmask) : 64u;
            if (zmask && stop <= run) {
                emit_run(BA_MV_DIAG, stop);
                i -= (int)stop;
                j -= (int)stop;
                cur = 0;
                break;
            }
            emit_run(BA_MV_DIAG, run);
            i -= (int)run;
            j -= (int)run;
            if (run > 0) cur -= __builtin_amdgcn_readlane(incl, (int)run - 1);
            if (run < 64 && ((okmask >> run) & 1ull)) {
                const uint32_t mv = (uint32_t)__builtin_amdgcn_readlane((int)mvq, (int)run);
                emit(mv);
                if (mv == BA_MV_LEFT) {
                    --j;
                    cur -= __builtin_amdgcn_readlane(d_left, (int)run);
                } else {
                    --i;
                    cur -= __builtin_amdgcn_readlane(d_up, (int)run);
                }
            }
            continue;
        }
        emit_run(BA_MV_DIAG, run);
        i -= (int)run;
        j -= (int)run;
        if (run < 64 && ((okmask >> run) & 1ull)) {
            // lane `run` sits on the cell the run ends at and knows its (non-diagonal) move
            const uint32_t mv = (uint32_t)__builtin_amdgcn_readlane((int)mvq, (int)run);
            emit(mv);
            if (mv == BA_MV_LEFT)
                --j;
            else
                --i;
        }
    }
    // borders (align.go:92-100; align_pair.go:110-113, 158-159, 187-188).  A stripe that is not the first one
    // hands the walk on at its column 0 (the left border is not its own); the top border is every stripe's.
    if (first_stripe && !left_free && i > 0 && j == 0) {
        emit_run(BA_MV_UP, (uint32_t)i);
        i = 0;
    }
    if (!top_free && i == 0 && j > 0) {
        emit_run(BA_MV_LEFT, (uint32_t)j);
        j = 0;
    }
    wait_all();
    post(0, 0, 0, 0, 0, 0);  // the loaders may leave
    if (lane == 0) {
        if (fill > 0) ops[k >> 4] = (uint32_t)opacc;
        if (fill > 16) ops[(k >> 4) + 1] = (uint32_t)(opacc >> 32);
        P.walk[2] = k + fill;  // trace length
        if (MODE == BA_LOCAL) P.walk[3] = (uint32_t)cur;  // the score of the cell the walk stopped in: 0 = the alignment's start
        P.walk[4] = (uint32_t)i;  // where the walk stopped: (0, .) or (., 0) -- the next stripe's (i, its m)
        P.walk[5] = (uint32_t)j;
        P.walk[6] = n_requests;
        P.walk[7] = n_rounds;
    }
}

// ----------------------------------------------------------------------------
// traceback, long mode, affine gaps: the walk of kernels_affine.hip's ba_trace_affine_kernel -- over (level, cell),
// reading the four decision bits -- done by one wavefront over the staged tiles of the linear walk.  On the middle
// level lane q looks at cell (i-q, j-q) and a run of diagonal moves is one round; on a gap level lane q looks at the
// q-th cell along the gap and the run ends at the first cell that opened the gap.  Local: the true score V is carried
// along (the substitution scores of a diagonal run as a prefix sum over the lanes) and the walk ends at the first
// middle-level cell with V = 0.
// ----------------------------------------------------------------------------

// affine local: (largest true score, first step) of every row -> first maximum in row-major order -> P.walk
template <int T>
__global__ void __launch_bounds__(1024) ba_long_affine_find_kernel(BaLongParams P)
{
    __shared__ int red_v[16];
    __shared__ uint32_t red_k[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t rows = (uint32_t)P.strips * 64u * (uint32_t)T;  // in row order: (strip, lane, r)
    int best = 0;
    uint32_t bk = 0xffffffffu;
    for (uint32_t k = (uint32_t)tid; k < rows; k += 1024u) {
        const int v = P.lane_max[(size_t)k * 2];
        if (v > best || (v == best && k < bk)) {
            best = v;
            bk = k;
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int ov = __shfl_xor(best, d);
        const uint32_t ok = (uint32_t)__shfl_xor((int)bk, d);
        if (ov > best || (ov == best && ok < bk)) {
            best = ov;
            bk = ok;
        }
    }
    if (lane == 0) {
        red_v[wave] = best;
        red_k[wave] = bk;
    }
    __syncthreads();
    if (tid == 0) {
        best = red_v[0];
        bk = red_k[0];
        for (int q = 1; q < 16; ++q)
            if (red_v[q] > best || (red_v[q] == best && red_k[q] < bk)) {
                best = red_v[q];
                bk = red_k[q];
            }
        int fi = 0, fj = 0;
        if (best > 0) {
            const int RS = 64 * T;
            const int s = (int)(bk / (uint32_t)RS), rem = (int)(bk % (uint32_t)RS), l = rem / T;
            const int step = P.lane_max[(size_t)bk * 2 + 1];
            fi = (int)P.n - ((int)P.strips - s) * RS + rem + 1;
            fj = step - l + 1;
        }
        *P.score = (int64_t)best;
        P.walk[0] = (uint32_t)fi;
        P.walk[1] = (uint32_t)fj;
        P.walk[2] = 0;
        P.walk[3] = (uint32_t)best;
        P.walk[4] = (uint32_t)fi;
        P.walk[5] = (uint32_t)fj;
    }
}

// CKPT: the walk of one recomputed row block of the linear-space traceback: the decision words of strips first_strip ..
// are there, the walk stops on reaching row stop_row -- on whatever level it is on -- and the next block goes on from
// (level, i, j) and, in local mode, the score V (walk[6], [4], [5], [3]).
template <int MODE, int T, bool CKPT = false>
__global__ void __launch_bounds__(256) ba_long_walk_affine_kernel(BaLongParams P)
{
    using G = BaLongGeom<T, 4>;
    constexpr int RS = G::RS, CPD = G::CPD;
    // wave 0 walks, waves 1-3 load decision words for it, the next strip's tile ahead of the walk (see ba_long_walk_kernel)
    extern __shared__ __align__(16) uint32_t walk_lds[];
    constexpr int TW = BA_WALK_TW;
    uint32_t *buf0 = walk_lds, *buf1 = walk_lds + (TW * 64 + TW);
    int *mail = reinterpret_cast<int *>(walk_lds + 2 * (TW * 64 + TW));
    uint32_t *tile = buf0, *above = buf0 + TW * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 8) mail[tid] = 0;
    __syncthreads();  // (the only barrier)
    const int n = (int)P.n, m = (int)P.m;
    const int go = P.gap, ge = P.gap_ext, cc = go - ge;
    const int strips = (int)P.strips;
    const int front = strips * RS - n;
    const int wps = (int)P.words_per_strip;
    constexpr bool top_free = (MODE == BA_LOCAL || MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_LOCAL || MODE == BA_OVERLAP);
    const int32_t *bottom_row = P.bnd + (size_t)(P.nsuper - 1) * P.bnd_stride * 2;  // granules: low dword = hq(n, j)
    const int first_strip = CKPT ? P.first_strip : 0, stop_row = CKPT ? P.stop_row : 0;
    enum { LV_MM = 0, LV_G1 = 1, LV_G2 = 2 };
    int level = LV_MM;

    // end cell: global (n, m) (align_pair.go:123-126); fitting / overlap the leftmost strict maximum of the bottom row's
    // true scores, from column 0 (align_pair.go:169-182); local: ba_long_affine_find_kernel has found it
    int i = n, j = m, V = 0;
    if (P.start_j >= 0) {
        // the host names the cell (linear-space traceback: where the block below stopped, on the level it was on)
        i = P.start_i;
        j = P.start_j;
        V = (int)P.walk[3];
        level = min((int)P.walk[6], (int)LV_G2);
    } else if (MODE == BA_LOCAL) {
        i = (int)P.walk[0];
        j = (int)P.walk[1];
        V = (int)P.walk[3];
    } else if (MODE == BA_GLOBAL) {
        V = bottom_row[2 * (m - 1)] + cc - ge * (n + m);
    } else {
        // H'(n, j) - ext*j orders the bottom row as the true scores do
        const int col0n = left_free ? ge * n : -cc;  // H'(n, 0), n >= 1
        int bj = 0, bv = col0n - cc;
        for (int c0 = 0; c0 < m; c0 += 64) {
            const int col = c0 + lane;
            int v = col < m ? bottom_row[2 * col] - ge * (col + 1) : INT32_MIN;
            int jj = col + 1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int ov = __shfl_xor(v, d), oj = __shfl_xor(jj, d);
                if (ov > v || (ov == v && oj < jj)) {
                    v = ov;
                    jj = oj;
                }
            }
            if (v > bv) {
                bv = v;
                bj = jj;
            }
        }
        j = bj;
        V = bv + cc - ge * n;
    }
    i = __builtin_amdgcn_readfirstlane(i);
    j = __builtin_amdgcn_readfirstlane(j);
    V = __builtin_amdgcn_readfirstlane(V);
    level = __builtin_amdgcn_readfirstlane(level);
    if (tid == 0) {
        if (MODE != BA_LOCAL && P.start_j < 0) *P.score = (int64_t)V;
        P.walk[0] = (uint32_t)i;
        P.walk[1] = (uint32_t)j;
        P.walk[2] = 0;
        P.walk[3] = (uint32_t)V;
        P.walk[4] = (uint32_t)i;
        P.walk[5] = (uint32_t)j;
    }
    if (P.score_only || P.scan_only) return;

    // staged decision words (the tiles of ba_long_walk_kernel)
    int tile_s = -1, tile_u = 0, ab_s = -1, ab_u = 0;
    auto mload = [](int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (wave != 0) {
        int seen = 0;
        for (;;) {
            int sq;
            while ((sq = mload(&mail[0])) == seen) __builtin_amdgcn_s_sleep(2);
            seen = sq;
            asm volatile("" ::: "memory");
            const int op = mail[1];
            if (op == 0) break;
            uint32_t *tb = mail[6] ? buf1 : buf0;
            if (op & 1) {
                const int sx = mail[2], tu = mail[3];
                const uint32_t *src = P.resid + ((size_t)(sx - first_strip) * wps + tu) * 64;
                const int words = min(TW, wps - tu);
                constexpr int PER = (TW + 2) / 3;
                uint32_t v[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) v[q] = src[(size_t)min((wave - 1) * PER + q, words - 1) * 64 + lane];
#pragma unroll
                for (int q = 0; q < PER; ++q)
                    if ((wave - 1) * PER + q < TW) tb[((wave - 1) * PER + q) * 64 + lane] = v[q];
            }
            if (op & 2) {
                const int sx = mail[4], au = mail[5];
                for (int x = tid - 64; x < TW; x += 192) {
                    const int uu = au + x;
                    tb[TW * 64 + x] = uu < wps ? P.resid[((size_t)(sx - first_strip) * wps + uu) * 64 + 63] : 0u;
                }
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&mail[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    int posted = 0, cbuf = 0;
    bool pf_valid = false;
    int pf_s = -1, pf_tu = 0, pf_as = -1, pf_au = 0;
    auto post = [&](int op, int b, int sx, int tu, int as, int au) {
        ++posted;
        if (lane == 0) {
            mail[1] = op;
            mail[2] = sx;
            mail[3] = tu;
            mail[4] = as;
            mail[5] = au;
            mail[6] = b;
            asm volatile("" ::: "memory");
            __hip_atomic_store(&mail[0], posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto wait_all = [&]() {
        while (mload(&mail[7]) != 3 * posted) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    auto use_buffer = [&](int b) {
        cbuf = b;
        tile = b ? buf1 : buf0;
        above = tile + TW * 64;
    };
    auto clampw = [&](int u0) { return max(0, min(u0, wps - TW)); };
    auto window = [&](int u) { return clampw(u - (TW - 9)); };
    auto cross_col = [&](int sx) { return max(j - (i - (sx * RS - front)), 1); };
    auto entry_word = [&](int jc) { return ((jc - 1) + 63) / CPD; };
    auto load_tile = [&](int sx, int u) {
        wait_all();
        if (pf_valid && pf_s == sx && u >= pf_tu && u < pf_tu + TW) {
            use_buffer(cbuf ^ 1);
            tile_s = pf_s;
            tile_u = pf_tu;
            ab_s = pf_as;
            ab_u = pf_au;
        } else {
            tile_s = sx;
            tile_u = window(u);
            int op = 1;
            if (sx - 1 >= first_strip && sx >= 1) {
                ab_s = sx - 1;
                ab_u = clampw(entry_word(cross_col(sx)) - TW / 2);
                op |= 2;
            }
            post(op, cbuf ^ 1, tile_s, tile_u, ab_s, ab_u);
            wait_all();
            use_buffer(cbuf ^ 1);
        }
        pf_valid = false;
        if (sx - 1 >= first_strip && sx >= 1) {
            const int jc = cross_col(sx);
            pf_s = sx - 1;
            pf_tu = clampw(entry_word(jc) - (TW - 24));
            int op = 1;
            pf_as = -1;
            pf_au = 0;
            if (sx - 2 >= first_strip && sx >= 2) {
                pf_as = sx - 2;
                pf_au = clampw(entry_word(max(jc - RS, 1)) - TW / 2);
                op |= 2;
            }
            post(op, cbuf ^ 1, pf_s, pf_tu, pf_as, pf_au);
            pf_valid = true;
        }
    };
    auto load_above = [&](int sx, int u) {
        wait_all();
        ab_s = sx;
        ab_u = window(u);
        post(2, cbuf, 0, 0, ab_s, ab_u);
        wait_all();
    };
    // stages what cell (ii, jj) needs, 1 <= ii <= n, 1 <= jj <= m (uniform)
    auto stage = [&](int ii, int jj) {
        const int tt = ii - 1 + front;
        const int s = tt / RS, l = (tt % RS) / T;
        const int u = ((jj - 1) + l) / CPD;
        if (s == tile_s || tile_s < 0 || l != 63) {
            if (s != tile_s || u < tile_u || u >= tile_u + TW) load_tile(s, u);
        } else if (s != ab_s || u < ab_u || u >= ab_u + TW) {
            load_above(s, u);
        }
    };
    // the four bits of cell (ii, jj) if they are staged (per lane): bit 3 nF, bit 2 nD, bit 1 oE, bit 0 oF
    auto peek = [&](int ii, int jj, bool &ok) -> uint32_t {
        const int tt = ii - 1 + front;
        const int s = tt / RS, l = (tt % RS) / T, r = tt % T;
        const int t = (jj - 1) + l;
        const int u = t / CPD;
        const uint32_t sh = (uint32_t)(((t % CPD) * T + r) * 4);
        uint32_t word = 0;
        if (s == tile_s || l != 63) {
            const bool in = (s == tile_s) && u >= tile_u && u < tile_u + TW;
            ok = ok && in;
            word = tile[(in ? (u - tile_u) : 0) * 64 + l];
        } else {
            const bool in = (s == ab_s) && u >= ab_u && u < ab_u + TW;
            ok = ok && in;
            word = above[in ? (u - ab_u) : 0];
        }
        return (word >> sh) & 15u;
    };

    uint32_t *ops = P.ops;
    uint64_t opacc = 0;
    uint32_t fill = 0, k = 0;
    auto emit_run = [&](uint32_t mv, uint32_t count) {
        const uint64_t pattern = mv == BA_MV_DIAG ? 0xAAAAAAAAAAAAAAAAull : mv == BA_MV_LEFT ? 0x5555555555555555ull : 0ull;
        while (count > 0) {
            const uint32_t c = min(count, 32u - fill);
            const uint64_t bits = c == 32 ? pattern : (pattern & ((1ull << (2 * c)) - 1ull));
            opacc |= bits << (2 * fill);
            fill += c;
            count -= c;
            if (fill == 32) {
                if (lane == 0) {
                    ops[k >> 4] = (uint32_t)opacc;
                    ops[(k >> 4) + 1] = (uint32_t)(opacc >> 32);
                }
                k += 32;
                opacc = 0;
                fill = 0;
            }
        }
    };

    // every round stages a tile or moves: a walk that is still going after this many rounds has lost its way (cannot
    // happen with decision bits the DP wrote; the status word makes the host report it)
    int64_t rounds_left = 4 * ((int64_t)n + m) + 1024;
    if (i < 0 || i > n || j < 0 || j > m) rounds_left = 0;
    while (!(level == LV_MM && i == 0 && j == 0)) {
        if (--rounds_left < 0) {
            if (lane == 0) atomicAdd(P.status, 1u);
            break;
        }
        if (CKPT && stop_row > 0 && i <= stop_row) break;  // the rows above belong to the next block
        if (level == LV_MM) {
            if (i == 0) {  // top border: sg2 = -inf, no diagonal
                if (top_free || j == 0) break;
                level = LV_G1;
                continue;
            }
            if (j == 0) {  // left border
                if (left_free) break;
                level = LV_G2;
                continue;
            }
            if (MODE == BA_LOCAL && V == 0) break;  // skip-prefix cell
            const int ii = i - lane, jj = j - lane;
            bool ok = ii > stop_row && jj >= 1;
            uint32_t nib = 0;
            if (ok) nib = peek(ii, jj, ok);
            const uint64_t okmask = __ballot(ok);
            if (!(okmask & 1ull)) {
                stage(i, j);
                continue;
            }
            const bool is_diag = ok && (nib & 8u) && !(nib & 4u);  // not from F, diagonal candidate equals H
            const uint64_t dmask = __ballot(is_diag);
            const uint32_t run = dmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~dmask);
            if (MODE == BA_LOCAL) {
                // true scores along the run: V drops by the substitution score of every diagonal move
                int wq = 0;
                if (ok) wq = P.table[(int)P.code1[P.s1[ii - 1]] * P.nb + (int)P.code2[P.s2[jj - 1]]];
                int incl = wq;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int a = __shfl_up(incl, d);
                    if (lane >= d) incl += a;
                }
                const int vq = V - (incl - wq);  // true score of my cell while every lane below me moves diagonally
                const uint64_t zmask = __ballot(ok && vq == 0);
                const uint32_t stop = zmask ? (uint32_t)__builtin_ctzll(zmask) : 64u;
                if (zmask && stop <= run) {
                    emit_run(BA_MV_DIAG, stop);
                    i -= (int)stop;
                    j -= (int)stop;
                    V = 0;
                    break;
                }
                if (run > 0) V -= __builtin_amdgcn_readlane(incl, (int)run - 1);
            }
            emit_run(BA_MV_DIAG, run);
            i -= (int)run;
            j -= (int)run;
            if (run < 64 && ((okmask >> run) & 1ull)) {
                // lane `run` sits on the cell the run ends at: closing a gap in s2 comes first, then the diagonal (taken
                // above), then a gap in s1 (align_affine.go:82-100)
                const uint32_t nb_ = (uint32_t)__builtin_amdgcn_readlane((int)nib, (int)run);
                level = (nb_ & 8u) ? LV_G1 : LV_G2;
                // the gap's first move is decided by the same cell's open bit, which is at hand: a gap of length 1 -- most
                // of them -- is taken here and the walk stays on the middle level (i, j >= 1 in this cell)
                if (level == LV_G2 && (i == 1 || (nb_ & 1u))) {
                    emit_run(BA_MV_UP, 1);
                    --i;
                    if (MODE == BA_LOCAL) V += go;
                    level = LV_MM;
                } else if (level == LV_G1 && (j == 1 || (nb_ & 2u))) {
                    emit_run(BA_MV_LEFT, 1);
                    --j;
                    if (MODE == BA_LOCAL) V += go;
                    level = LV_MM;
                }
            }
        } else if (level == LV_G1) {
            // gap in s1: (gap, s2[j-1]) per move, j >= 1; the gap was opened at column 1, on the top border only there,
            // else where the oE bit says so
            const int jj = j - lane;
            bool ok = jj >= 1;
            bool opened = false;
            if (ok) {
                if (jj == 1) opened = true;
                else if (i > 0) opened = (peek(i, jj, ok) & 2u) != 0;
            }
            const uint64_t okmask = __ballot(ok);
            if (!(okmask & 1ull)) {
                stage(i, j);
                continue;
            }
            const uint32_t usable = okmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okmask);  // staged prefix
            const uint64_t omask = __ballot(ok && opened) & (usable == 64 ? ~0ull : ((1ull << usable) - 1ull));
            if (omask) {
                const uint32_t first = (uint32_t)__builtin_ctzll(omask);
                emit_run(BA_MV_LEFT, first + 1);
                j -= (int)first + 1;
                if (MODE == BA_LOCAL) V += (int)first * ge + go;
                level = LV_MM;
            } else {
                emit_run(BA_MV_LEFT, usable);
                j -= (int)usable;
                if (MODE == BA_LOCAL) V += (int)usable * ge;
            }
        } else {
            // gap in s2: (s1[i-1], gap) per move, i >= 1
            const int ii = i - lane;
            bool ok = ii > stop_row;
            bool opened = false;
            if (ok) {
                if (ii == 1) opened = true;
                else if (j > 0) opened = (peek(ii, j, ok) & 1u) != 0;
            }
            const uint64_t okmask = __ballot(ok);
            if (!(okmask & 1ull)) {
                stage(i, j);
                continue;
            }
            const uint32_t usable = okmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okmask);
            const uint64_t omask = __ballot(ok && opened) & (usable == 64 ? ~0ull : ((1ull << usable) - 1ull));
            if (omask) {
                const uint32_t first = (uint32_t)__builtin_ctzll(omask);
                emit_run(BA_MV_UP, first + 1);
                i -= (int)first + 1;
                if (MODE == BA_LOCAL) V += (int)first * ge + go;
                level = LV_MM;
            } else {
                emit_run(BA_MV_UP, usable);
                i -= (int)usable;
                if (MODE == BA_LOCAL) V += (int)usable * ge;
            }
        }
    }
    wait_all();
    post(0, 0, 0, 0, 0, 0);  // the loaders may leave
    if (lane == 0) {
        if (fill > 0) ops[k >> 4] = (uint32_t)opacc;
        if (fill > 16) ops[(k >> 4) + 1] = (uint32_t)(opacc >> 32);
        P.walk[2] = k + fill;
        if (MODE == BA_LOCAL) P.walk[3] = (uint32_t)V;  // 0 on the middle level: the alignment's start
        P.walk[4] = (uint32_t)i;
        P.walk[5] = (uint32_t)j;
        P.walk[6] = (uint32_t)level;
    }
}

__global__ void __launch_bounds__(1024) ba_long_expand_kernel(BaLongParams P)
{
    __shared__ uint32_t wsum1[16], wsum2[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t L = P.walk[2];
    const uint64_t cap = P.trace_cap ? P.trace_cap : (uint64_t)P.n + (uint64_t)P.m;
    const uint32_t gs = P.gap_symbol;
    uint32_t i0 = P.walk[0], j0 = P.walk[1];  // row / column of the first move of the current round
    for (uint32_t base = 0; base < L; base += 16 * 1024) {
        const uint32_t first = base + 16u * tid;  // this thread's 16 moves
        const uint32_t w = first < L ? P.ops[first >> 4] : 0u;
        const uint32_t cnt = first < L ? min(16u, L - first) : 0u;
        const uint32_t valid = cnt == 16 ? 0xffffffffu : ((1u << (2 * cnt)) - 1u);
        const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;  // code 0 up, 1 left, 2 diagonal
        const uint32_t is_left = lo & ~hi & valid & 0x55555555u;
        const uint32_t is_up = ~lo & ~hi & valid & 0x55555555u;
        const uint32_t c1 = cnt - __popc(is_left);  // moves that consume a symbol of s1
        const uint32_t c2 = cnt - __popc(is_up);    // ... of s2
        // block-wide exclusive prefix sums of c1, c2
        uint32_t p1 = c1, p2 = c2;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t a = __shfl_up(p1, d), b = __shfl_up(p2, d);
            if (lane >= d) {
                p1 += a;
                p2 += b;
            }
        }
        if (lane == 63) {
            wsum1[wave] = p1;
            wsum2[wave] = p2;
        }
        __syncthreads();
        uint32_t off1 = 0, off2 = 0, tot1 = 0, tot2 = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wave) {
                off1 += wsum1[q];
                off2 += wsum2[q];
            }
            tot1 += wsum1[q];
            tot2 += wsum2[q];
        }
        uint32_t ii = i0 - (off1 + p1 - c1), jj = j0 - (off2 + p2 - c2);  // cell of my first move
        for (uint32_t q = 0; q < cnt; ++q) {
            const uint32_t mv = (w >> (2 * q)) & 3u;
            const uint64_t pos = cap - 1 - (uint64_t)P.trace_done - (uint64_t)(first + q);
            P.t1[pos] = mv == BA_MV_LEFT ? (uint8_t)gs : P.s1[ii - 1];
            P.t2[pos] = mv == BA_MV_UP ? (uint8_t)gs : P.s2[jj - 1];
            ii -= (mv != BA_MV_LEFT);
            jj -= (mv != BA_MV_UP);
        }
        i0 -= tot1;
        j0 -= tot2;
        __syncthreads();
    }
    if (tid == 0) {
        *P.t_off = cap - P.trace_done - L;
        *P.t_len = (uint64_t)P.trace_done + L;
    }
}

// ----------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------
template <int MODE, int T, int K>
static hipError_t launch_long(const BaLongParams &P, int phases, hipStream_t st)
{
    // 84 KB of (unused) dynamic LDS: at most one workgroup per CU -- each of its four strips has a SIMD to itself,
    // and it is the placement the sc1 hand-off was measured in (MI355X_MICROARCH.md, "Hand-offs measured with sc1
    // loads"); workgroups beyond the CU count queue up behind the ones they depend on
    // the strip profiles of the workgroup's waves (256 bytes per column symbol and wave), padded to 84 KB
    const size_t prof_lds = (size_t)BA_LONG_W * (size_t)(P.nb + 1) * 256;  // (+ the row of the columns left of the matrix)
    const size_t pad_lds = std::max<size_t>(prof_lds, 84 * 1024);
    hipError_t e = hipSuccess;
    if ((phases & BA_LONG_PHASE_DP) && P.multi) {
        // several stripes in one launch: wg_count = the workgroups of all of them (P.order)
        if constexpr (MODE == BA_LOCAL) {
            return hipErrorInvalidValue;  // (local mode runs unstriped)
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_dp_kernel<MODE, T, K, true, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds);
            hipLaunchKernelGGL((ba_long_dp_kernel<MODE, T, K, true, false>), dim3(P.wg_count), dim3(64 * (BA_LONG_W + 1)), pad_lds, st, P);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    } else if (phases & BA_LONG_PHASE_DP) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_dp_kernel<MODE, T, K, false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds);
        const uint32_t wgs = P.wg_count ? P.wg_count : P.nsuper - P.wg_begin;
        if (P.no_resid) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_dp_kernel<MODE, T, K, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds);
            hipLaunchKernelGGL((ba_long_dp_kernel<MODE, T, K, false, false, true>), dim3(wgs), dim3(64 * (BA_LONG_W + 1)), pad_lds, st, P);
        } else {
            hipLaunchKernelGGL((ba_long_dp_kernel<MODE, T, K, false, false>), dim3(wgs), dim3(64 * (BA_LONG_W + 1)), pad_lds, st, P);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (phases & BA_LONG_PHASE_WALK) {
        if constexpr (MODE == BA_LOCAL) {
            if (P.start_j < 0) {  // (a later block of the linear-space traceback: the end cell is known)
                hipLaunchKernelGGL((ba_long_local_find_kernel<T, K>), dim3(1), dim3(1024), 0, st, P);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
        }
        const size_t walk_lds = (size_t)(2 * (BA_WALK_TW * 64 + BA_WALK_TW) + 8) * 4;
        if (P.first_strip || P.stop_row) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_walk_kernel<MODE, T, K, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
            hipLaunchKernelGGL((ba_long_walk_kernel<MODE, T, K, true>), dim3(1), dim3(256), walk_lds, st, P);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_walk_kernel<MODE, T, K>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
            hipLaunchKernelGGL((ba_long_walk_kernel<MODE, T, K>), dim3(1), dim3(256), walk_lds, st, P);
        }
        if ((e = hipGetLastError()) != hipSuccess || P.score_only || P.scan_only) return e;
        hipLaunchKernelGGL(ba_long_expand_kernel, dim3(1), dim3(1024), 0, st, P);
        e = hipGetLastError();
    }
    return e;
}

template <int MODE>
static hipError_t launch_long_mode(const BaLongParams &P, int T, int K, int phases, hipStream_t st)
{
    if (T == 1 && K == 2) return launch_long<MODE, 1, 2>(P, phases, st);
    if (T == 1 && K == 4) return launch_long<MODE, 1, 4>(P, phases, st);
    if (T == 1 && K == 8) return launch_long<MODE, 1, 8>(P, phases, st);
    if (T == 2 && K == 2) return launch_long<MODE, 2, 2>(P, phases, st);
    if (T == 2 && K == 4) return launch_long<MODE, 2, 4>(P, phases, st);
    if (T == 2 && K == 8) return launch_long<MODE, 2, 8>(P, phases, st);
    if (T == 4 && K == 2) return launch_long<MODE, 4, 2>(P, phases, st);
    if (T == 4 && K == 4) return launch_long<MODE, 4, 4>(P, phases, st);
    if (T == 4 && K == 8) return launch_long<MODE, 4, 8>(P, phases, st);
    return hipErrorInvalidValue;
}

// affine gaps: DP (K = 4: four decision bits per cell), then (local) the best-cell search, the walk and the expansion
template <int MODE, int T>
static hipError_t launch_long_affine(const BaLongParams &P, int phases, hipStream_t st)
{
    // two hand-over rings per strip (hq and F') are 80 KB of static LDS: the padding that keeps a CU to one workgroup
    // shrinks accordingly
    const size_t prof_lds = (size_t)BA_LONG_W * (size_t)(P.nb + 1) * 256;
    const size_t pad_lds = std::max<size_t>(prof_lds, 44 * 1024);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_dp_kernel<MODE, T, 4, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds);
    if (e != hipSuccess) return e;
    if (phases & BA_LONG_PHASE_DP) {
        const uint32_t wgs = P.wg_count ? P.wg_count : P.nsuper - P.wg_begin;  // (a row block of the linear-space traceback, or all)
        if (P.no_resid) {  // score only: the decision words are not stored
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_dp_kernel<MODE, T, 4, false, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds)) != hipSuccess)
                return e;
            hipLaunchKernelGGL((ba_long_dp_kernel<MODE, T, 4, false, true, true>), dim3(wgs), dim3(64 * (BA_LONG_W + 1)), pad_lds, st, P);
        } else {
            hipLaunchKernelGGL((ba_long_dp_kernel<MODE, T, 4, false, true>), dim3(wgs), dim3(64 * (BA_LONG_W + 1)), pad_lds, st, P);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (phases & BA_LONG_PHASE_WALK) {
        if constexpr (MODE == BA_LOCAL) {
            if (P.start_j < 0) {  // (a block of the linear-space traceback: the end cell is known)
                hipLaunchKernelGGL((ba_long_affine_find_kernel<T>), dim3(1), dim3(1024), 0, st, P);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
        }
        const size_t walk_lds = (size_t)(2 * (BA_WALK_TW * 64 + BA_WALK_TW) + 8) * 4;
        if (P.first_strip || P.stop_row) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_walk_affine_kernel<MODE, T, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds)) != hipSuccess)
                return e;
            hipLaunchKernelGGL((ba_long_walk_affine_kernel<MODE, T, true>), dim3(1), dim3(256), walk_lds, st, P);
        } else {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_long_walk_affine_kernel<MODE, T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds)) != hipSuccess)
                return e;
            hipLaunchKernelGGL((ba_long_walk_affine_kernel<MODE, T>), dim3(1), dim3(256), walk_lds, st, P);
        }
        if ((e = hipGetLastError()) != hipSuccess || P.score_only || P.scan_only) return e;
        hipLaunchKernelGGL(ba_long_expand_kernel, dim3(1), dim3(1024), 0, st, P);
        e = hipGetLastError();
    }
    return e;
}

template <int MODE>
static hipError_t launch_long_affine_mode(const BaLongParams &P, int T, int phases, hipStream_t st)
{
    if (T == 2) return launch_long_affine<MODE, 2>(P, phases, st);
    if (T == 4) return launch_long_affine<MODE, 4>(P, phases, st);
    return hipErrorInvalidValue;
}

// affine: rows per lane.  A row's step is a chain of three dependent operations (max, max3, subtract) instead of one:
// shorter strips than the linear kernel's.  BIOALIGN_LONG_AFFINE_T=4 for experiments.
extern "C" int ba_long_affine_rows_per_lane(void)
{
    const char *e = getenv("BIOALIGN_LONG_AFFINE_T");
    return (e && atoi(e) == 4) ? 4 : 2;
}

extern "C" hipError_t ba_launch_long_affine(const BaLongParams &P, int mode, int T, int phases, hipStream_t st)
{
    if (P.n == 0 || P.m == 0 || P.strips == 0 || (uint64_t)P.nsuper * BA_LONG_W < P.strips) return hipErrorInvalidValue;
    if (P.multi || P.left_col || P.right_col || P.col0) return hipErrorInvalidValue;  // unstriped only
    if ((uint64_t)P.wg_begin + P.wg_count > P.nsuper || P.wg_begin >= P.nsuper) return hipErrorInvalidValue;
    if (P.gap < P.gap_ext || P.gap_ext < 0 || P.epoch == 0 || P.epoch >= 0x800000u) return hipErrorInvalidValue;
    switch (mode) {
    case BA_GLOBAL: return launch_long_affine_mode<BA_GLOBAL>(P, T, phases, st);
    case BA_LOCAL: return P.lane_max ? launch_long_affine_mode<BA_LOCAL>(P, T, phases, st) : hipErrorInvalidValue;
    case BA_FITTING: return launch_long_affine_mode<BA_FITTING>(P, T, phases, st);
    case BA_OVERLAP: return launch_long_affine_mode<BA_OVERLAP>(P, T, phases, st);
    }
    return hipErrorInvalidValue;
}

// rows per lane (T * K divides 32).  One pair is latency-bound: a lone wavefront spends about 65 + 11 T cycles on a
// step (measured, profiles/r02_notes.md: the serial chain DPP -> T dependent v_max3 and the waits around it, not the
// issue slots), a strip of 64 T rows starts 63 + 16 steps after the one above, so the DP takes about
// (m + 80 n / (64 T)) (65 + 11 T) cycles plus the HBM hops between workgroups: T = 2 and T = 4 come out the same at
// 100k x 100k, and the traceback likes the taller strips.  BIOALIGN_LONG_T=1|2 for experiments.
extern "C" int ba_long_rows_per_lane(int K)
{
    (void)K;
    const char *e = getenv("BIOALIGN_LONG_T");
    const int t = e ? atoi(e) : 0;
    return (t == 1 || t == 2) ? t : 4;
}
// strips per workgroup
extern "C" int ba_long_strips_per_group(void) { return BA_LONG_W; }

// phases: BA_LONG_PHASE_DP | BA_LONG_PHASE_WALK (the walk includes the expansion of its moves into bytes)
extern "C" hipError_t ba_launch_long(const BaLongParams &P, int mode, int T, int K, int phases, hipStream_t st)
{
    if (P.n == 0 || P.m == 0 || P.strips == 0 || (uint64_t)P.nsuper * BA_LONG_W < P.strips) return hipErrorInvalidValue;
    if (P.multi) {
        if (!P.order || P.wg_count == 0 || phases != BA_LONG_PHASE_DP) return hipErrorInvalidValue;
    } else if ((uint64_t)P.wg_begin + P.wg_count > P.nsuper || P.wg_begin >= P.nsuper) {
        return hipErrorInvalidValue;
    }
    switch (mode) {
    case BA_GLOBAL: return launch_long_mode<BA_GLOBAL>(P, T, K, phases, st);
    case BA_LOCAL:
        if (!P.lane_max || P.left_col || P.right_col || P.col0) return hipErrorInvalidValue;  // (unstriped only)
        return launch_long_mode<BA_LOCAL>(P, T, K, phases, st);
    case BA_FITTING: return launch_long_mode<BA_FITTING>(P, T, K, phases, st);
    case BA_OVERLAP: return launch_long_mode<BA_OVERLAP>(P, T, K, phases, st);
    }
    return hipErrorInvalidValue;
}
