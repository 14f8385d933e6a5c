// kernels_long.hip -- long mode: ONE pair aligned by many wavefronts (intra-sequence,
// anti-diagonal wavefront), linear gap, global / fitting / overlap.
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

#include "common.h"
#include "device_helpers.h"

#define BA_LONG_FINAL 0x7fffffffu

// lane l <- lane l-1; lane 0 keeps `first`
__device__ __forceinline__ int ba_wave_shr1(int first, int src)
{
    return __builtin_amdgcn_update_dpp(first, src, 0x138, 0xf, 0xf, false);
}
// lane l <- lane l+1; lane 63 keeps `last`
__device__ __forceinline__ int ba_wave_shl1(int last, int src)
{
    return __builtin_amdgcn_update_dpp(last, src, 0x130, 0xf, 0xf, false);
}

// Waits until every store this wave has issued has been acknowledged (s_waitcnt vmcnt(0);
// gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima),
// and keeps the compiler from moving memory operations across it.
__device__ __forceinline__ void ba_stores_done()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

template <int T, int K>
struct BaLongGeom {
    static constexpr int RS = 64 * T;          // rows per strip
    static constexpr int CPD = 32 / (T * K);   // steps per residue word
    static_assert(CPD >= 1 && CPD * T * K == 32, "T * K must divide 32");
};

// One step of one lane: T cells of one column.  ACTIVE_CHECK: the lane may be outside
// its column range (start-up and drain of the skew) and must then keep its state.
template <int T, int K, bool CHECK>
__device__ __forceinline__ void ba_long_step(int (&h)[T], const uint32_t (&prow)[T], uint32_t &acc, int top, int diag0,
                                             uint32_t bshift, bool active)
{
    int up = top, diag = diag0;
#pragma unroll
    for (int r = 0; r < T; ++r) {
        const int left = h[r];
        const int w = __builtin_amdgcn_sbfe((int)prow[r], bshift, 8);
        int hn = ba_max3(up, left, diag + w);
        if (CHECK) hn = active ? hn : left;
        acc = __builtin_amdgcn_alignbit((uint32_t)hn, acc, K);
        diag = left;
        h[r] = hn;
        up = hn;
    }
}

template <int MODE, int T, int K>
__global__ void __launch_bounds__(64) ba_long_dp_kernel(BaLongParams P)
{
    using G = BaLongGeom<T, K>;
    constexpr int RS = G::RS, CPD = G::CPD;
    __shared__ uint32_t code2s[64];  // byte -> code of the column side, packed
    const int lane = threadIdx.x;
    const int s = blockIdx.x;
    const int n = (int)P.n, m = (int)P.m, g = P.gap;
    const int strips = (int)P.strips;
    code2s[lane] = reinterpret_cast<const uint32_t *>(P.code2)[lane];
    const uint8_t *code2 = reinterpret_cast<const uint8_t *>(code2s);

    // bottom-aligned strips: strip s covers rows i0+1 .. i0+RS; rows <= 0 (first strip only)
    // carry the padding code and reproduce row 0 (see kernels.hip)
    const int i0 = n - (strips - s) * RS;
    const int b_top = (MODE == BA_FITTING) ? g : 0;   // Hs(0, j) = b_top * j
    const int b_left = (MODE == BA_OVERLAP) ? g : 0;  // Hs(i, 0) = b_left * i
    int h[T];
    uint32_t prow[T];
#pragma unroll
    for (int r = 0; r < T; ++r) {
        const int i = i0 + lane * T + r + 1;
        const int code = i >= 1 ? (int)P.code1[P.s1[i - 1]] : P.pad_code;
        prow[r] = P.rowprof[code];
        h[r] = b_left * max(i, 0);
    }
    int32_t *bnd_out = P.bnd + (size_t)s * m;
    const int32_t *bnd_in = P.bnd + (size_t)(s > 0 ? s - 1 : 0) * m;
    uint32_t *flag_out = P.flags + s;
    const uint32_t *flag_in = P.flags + (s > 0 ? s - 1 : 0);
    uint32_t *resid = P.resid + (size_t)s * P.words_per_strip * 64 + lane;

    // skew state: `bottom` = my last row after the previous step; the lane below me reads it
    int bottom = h[T - 1];
    // score above my first row in the column BEFORE my current one (the diagonal of my first row);
    // before the first step that is column 0 of the row above: my upper neighbour's initial bottom
    int topprev = ba_wave_shr1(b_left * max(i0, 0), bottom);
    int bsym = 0;    // my current column's symbol code * 8
    int sreg = 0;    // column symbols of the current block of 64 steps, lane 0 reads its own
    int bin = 0;     // boundary row of the strip above for the same block
    int bout = 0;    // my strip's bottom row being collected for a 64-column store
    uint32_t acc = 0;
    const int total_steps = m + 63;
    uint32_t seen = 0;
    bool failed = false;

    __syncthreads();
    for (int t0 = 0; t0 < total_steps; t0 += 64) {
        // inputs of this block of 64 steps: column (0-based) t0 + lane for lane 0's steps t0 .. t0+63
        {
            const int col = t0 + lane;
            sreg = col < m ? 8 * (int)code2[P.s2[col]] : 0;
            if (s > 0 && t0 < m) {
                const uint32_t need = (uint32_t)(t0 / 64 + 1);
                if (seen < need) {
                    int spin = 0;
                    for (; spin < (1 << 22); ++spin) {
                        seen = __hip_atomic_load(flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen >= need) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (spin == (1 << 22)) {  // give up waiting for good; the host reports the run as failed
                        failed = true;
                        seen = BA_LONG_FINAL;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                bin = col < m ? __hip_atomic_load(bnd_in + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
        }
        const int t_end = min(t0 + 64, total_steps);
        // all lanes inside their column range for the whole block?
        const bool steady = (t0 >= 63) && (t_end <= m);
        for (int t = t0; t < t_end; ++t) {
            // the row above my first row, in my current column: lane 0 takes the boundary
            const int top_in = (s == 0) ? b_top * (t + 1) : bin;
            const int top = ba_wave_shr1(top_in, bottom);
            bsym = ba_wave_shr1(sreg, bsym);
            if (steady) {
                ba_long_step<T, K, false>(h, prow, acc, top, topprev, (uint32_t)bsym, true);
            } else {
                const bool active = (t >= lane) && (t - lane < m);
                ba_long_step<T, K, true>(h, prow, acc, top, topprev, (uint32_t)bsym, active);
            }
            topprev = top;
            bottom = h[T - 1];
            bout = ba_wave_shl1(bottom, bout);
            sreg = ba_wave_shl1(sreg, sreg);
            bin = ba_wave_shl1(bin, bin);
            if ((t + 1) % CPD == 0) resid[(size_t)(t / CPD) * 64] = acc;
            if ((t & 63) == 62 && t >= 126) {
                // lanes hold the bottom row of columns (0-based) t-125 .. t-62: one 256-byte line
                const int q = (t - 126) / 64;
                __hip_atomic_store(bnd_out + q * 64 + lane, bout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ba_stores_done();  // the line is out before the word says so
                if (lane == 0) __hip_atomic_store(flag_out, (uint32_t)(q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // last, partial residue word: its pushes sit at the top, bring them down to their places
    if (total_steps % CPD != 0)
        resid[(size_t)(total_steps / CPD) * 64] = acc >> ((CPD - total_steps % CPD) * T * K);
    // the columns not yet handed over: after the last step (t = m + 62) lane l holds column (0-based) m - 64 + l
    {
        const int col = m - 64 + lane;
        const int done = (m / 64) * 64;  // full blocks stored above cover columns < done
        if (col >= done && col >= 0) __hip_atomic_store(bnd_out + col, bout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ba_stores_done();
        if (lane == 0) __hip_atomic_store(flag_out, BA_LONG_FINAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (failed && lane == 0) atomicAdd(P.status, 1u);
}

// ----------------------------------------------------------------------------
// traceback, long mode: one wavefront walks the path back from the end cell; all lanes
// run the same walk (uniform control flow, LDS broadcasts) over residue tiles staged in
// LDS: the tile of the current strip (64 residue words x 64 lanes, one contiguous 16 KB
// block of HBM) and, for the rows that touch the strip above, that strip's last lane.
// ----------------------------------------------------------------------------
template <int MODE, int T, int K>
__global__ void __launch_bounds__(64) ba_long_trace_kernel(BaLongParams P)
{
    using G = BaLongGeom<T, K>;
    constexpr int RS = G::RS, CPD = G::CPD;
    constexpr uint32_t MASK = (1u << K) - 1u;
    __shared__ uint32_t tile[64 * 64];
    __shared__ uint32_t above[64];
    __shared__ uint32_t seqbuf[2][64];  // 256 bytes of s1 / s2 around the walk
    __shared__ uint32_t code1s[64], code2s[64];
    const int lane = threadIdx.x;
    const int n = (int)P.n, m = (int)P.m, g = P.gap;
    const int strips = (int)P.strips;
    const int front = strips * RS - n;
    const size_t wps = P.words_per_strip;
    code1s[lane] = reinterpret_cast<const uint32_t *>(P.code1)[lane];
    code2s[lane] = reinterpret_cast<const uint32_t *>(P.code2)[lane];
    const uint8_t *code1 = reinterpret_cast<const uint8_t *>(code1s);
    const uint8_t *code2 = reinterpret_cast<const uint8_t *>(code2s);
    __syncthreads();

    const int b_top = (MODE == BA_FITTING) ? g : 0;
    const int b_left = (MODE == BA_OVERLAP) ? g : 0;
    constexpr bool top_free = (MODE == BA_FITTING);
    constexpr bool left_free = (MODE == BA_OVERLAP);
    const int32_t *bottom_row = P.bnd + (size_t)(strips - 1) * m;

    // end cell (align.go:123-124; align_pair.go:169-182): n >= 1 and m >= 1 here
    int i = n, j = m, cur;
    if (MODE == BA_GLOBAL) {
        cur = bottom_row[m - 1];
    } else {
        // leftmost strict maximum of the true scores Hs(n, j) - g*j of the bottom row, from column 0
        int bv = b_left * n, bj = 0, bc = b_left * n;  // column 0
        for (int c0 = 0; c0 < m; c0 += 64) {
            const int col = c0 + lane;  // 0-based array index = column col+1
            const int hv = col < m ? bottom_row[col] : 0;
            int v = col < m ? hv - g * (col + 1) : INT32_MIN;
            int jj = col + 1, cc = hv;
            // wave argmax, smaller column wins ties
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
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
    }
    if (lane == 0) *P.score = (int64_t)cur - (int64_t)g * (i + j);
    if (P.score_only) return;

    BaBackWriter w1, w2;
    const uint64_t cap = (uint64_t)n + (uint64_t)m;
    w1.init(P.t1, cap);
    w2.init(P.t2, cap);
    const uint32_t gs = P.gap_symbol;
    const int wbias = 2 * g;

    // staged sequence bytes: seqbuf[k] holds bytes [base_k, base_k + 256)
    int base1 = -1, base2 = -1;
    auto sym = [&](int which, const uint8_t *src, int len, int idx, int &base) {  // src[idx]
        if (base < 0 || idx < base || idx >= base + 256) {
            base = max(0, idx - 252) & ~3;  // idx lands in the last dword of the window
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int a = base + lane * 4 + b;
                v |= (a < len ? (uint32_t)src[a] : 0u) << (8 * b);
            }
            seqbuf[which][lane] = v;
        }
        return (uint32_t)reinterpret_cast<const uint8_t *>(seqbuf[which])[idx - base];
    };

    // residue tiles
    int tile_s = -1, tile_u = 0;   // strip and first word index of `tile`
    int ab_s = -1, ab_u = 0;       // strip and first word index of `above` (lane 63 of that strip)
    auto load_tile = [&](int s, int u) {
        tile_s = s;
        tile_u = max(0, min(u - 63 + 8, (int)wps - 64));  // the walk moves towards smaller words
        tile_u = max(tile_u, 0);
        const uint32_t *src = P.resid + ((size_t)s * wps + tile_u) * 64;
        const int words = min(64, (int)wps - tile_u);
#pragma unroll 4
        for (int q = 0; q < 64; ++q) tile[q * 64 + lane] = q < words ? src[(size_t)q * 64 + lane] : 0u;
    };
    auto load_above = [&](int s, int u) {
        ab_s = s;
        ab_u = max(0, min(u - 63 + 8, (int)wps - 64));
        const int uu = ab_u + lane;
        above[lane] = uu < (int)wps ? P.resid[((size_t)s * wps + uu) * 64 + 63] : 0u;
    };
    // residue of cell (ii, jj), 1 <= ii <= n, 1 <= jj <= m
    auto residue = [&](int ii, int jj) {
        const int tt = ii - 1 + front;
        const int s = tt / RS, l = (tt % RS) / T, r = tt % T;
        const int t = (jj - 1) + l;
        const int u = t / CPD;
        const uint32_t sh = (uint32_t)(((t % CPD) * T + r) * K);
        uint32_t word;
        if (s == tile_s || tile_s < 0 || l != 63) {
            if (s != tile_s || u < tile_u || u >= tile_u + 64) load_tile(s, u);
            word = tile[(u - tile_u) * 64 + l];
        } else {
            // a cell of the strip above the one the walk is in: always its last lane
            if (s != ab_s || u < ab_u || u >= ab_u + 64) load_above(s, u);
            word = above[u - ab_u];
        }
        return (word >> sh) & MASK;
    };
    auto neighbour = [&](int c, uint32_t rho) { return c - (int)(((uint32_t)c - rho) & MASK); };

    while (i > 0 && j > 0) {
        const int up = (i == 1) ? b_top * j : neighbour(cur, residue(i - 1, j));
        const int left = (j == 1) ? b_left * i : neighbour(cur, residue(i, j - 1));
        const int diag = (i == 1) ? b_top * (j - 1) : (j == 1) ? b_left * (i - 1) : neighbour(up, residue(i - 1, j - 1));
        const uint32_t x = sym(0, P.s1, n, i - 1, base1);
        const uint32_t y = sym(1, P.s2, m, j - 1, base2);
        const int w = P.table[(int)code1[x] * P.nb + (int)code2[y]] + wbias;
        const int d = diag + w;
        const int mx = max(up, left);
        if (d > mx) {
            w1.put(x);
            w2.put(y);
            cur = diag;
            --i;
            --j;
        } else if (left > up) {
            w1.put(gs);
            w2.put(y);
            cur = left;
            --j;
        } else {
            w1.put(x);
            w2.put(gs);
            cur = up;
            --i;
        }
    }
    // borders (align.go:92-100; align_pair.go:110-113, 158-159, 187-188)
    if (!left_free)
        while (i > 0) {
            w1.put(sym(0, P.s1, n, i - 1, base1));
            w2.put(gs);
            --i;
        }
    if (!top_free)
        while (j > 0) {
            w1.put(gs);
            w2.put(sym(1, P.s2, m, j - 1, base2));
            --j;
        }
    w1.finish();
    w2.finish();
    if (lane == 0) {
        const uint64_t len = (uint64_t)w1.length();
        *P.t_off = cap - len;
        *P.t_len = len;
    }
}

// ----------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------
template <int MODE, int T, int K>
static hipError_t launch_long(const BaLongParams &P, hipStream_t st)
{
    hipLaunchKernelGGL((ba_long_dp_kernel<MODE, T, K>), dim3(P.strips), dim3(64), 0, st, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ba_long_trace_kernel<MODE, T, K>), dim3(1), dim3(64), 0, st, P);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_long_mode(const BaLongParams &P, int T, int K, hipStream_t st)
{
    if (T == 8 && K == 2) return launch_long<MODE, 8, 2>(P, st);
    if (T == 8 && K == 4) return launch_long<MODE, 8, 4>(P, st);
    if (T == 4 && K == 8) return launch_long<MODE, 4, 8>(P, st);
    return hipErrorInvalidValue;
}

// rows per lane for a residue width: T * K divides 32
extern "C" int ba_long_rows_per_lane(int K) { return K == 8 ? 4 : 8; }

extern "C" hipError_t ba_launch_long(const BaLongParams &P, int mode, int T, int K, hipStream_t st)
{
    if (P.n == 0 || P.m == 0 || P.strips == 0) return hipErrorInvalidValue;
    switch (mode) {
    case BA_GLOBAL: return launch_long_mode<BA_GLOBAL>(P, T, K, st);
    case BA_FITTING: return launch_long_mode<BA_FITTING>(P, T, K, st);
    case BA_OVERLAP: return launch_long_mode<BA_OVERLAP>(P, T, K, st);
    }
    return hipErrorInvalidValue;
}
