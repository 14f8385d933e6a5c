// device_helpers.h -- device-side pieces shared by the linear-gap and the affine-gap kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

// ----------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------
__device__ __forceinline__ int ba_max3(int a, int b, int c) { return max(max(a, b), c); }
// the same, kept out of the compiler's hands: left to itself it rebalances the maxima of the local cell and its running
// maximum into 1.75 instructions per cell (v_max3 + v_max) where 1.5 v_max3 do
__device__ __forceinline__ int ba_max3_asm(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// signed byte k (0..3) of a dword; hipcc folds this into the SDWA operand of the add that uses it
__device__ __forceinline__ int ba_sbyte(uint32_t w, int k) { return (int)(int8_t)(w >> (8 * k)); }

// acc = (x << (32-K)) | (acc >> K): shifts the K low bits of x in at the top (one v_alignbit_b32)
template <int K>
__device__ __forceinline__ uint32_t ba_push_residue(uint32_t acc, int x)
{
    return __builtin_amdgcn_alignbit((uint32_t)x, acc, K);
}

// One record of the linear-gap walks (kernels.hip, kernels_bp.hip): 16 moves, 2 bits each (first move in the low
// bits), and the cell the first of them starts from.  ba_expand_kernel turns the records into the two gapped strings.
struct BaMoveRec {
    uint32_t ops, i, j, count;
};
// The records of one pair inside its bundle's block of rec_cap * 64 records, LANE-INTERLEAVED: record r of lane l lies
// at rec_off + (r + 1) * 64 + l; r = -1 is the header (trace length, where the trace ends, pair index).  Lanes that
// close a record in the same instruction -- a column-synchronous walk whose lanes took the same number of moves --
// store neighbouring 16-byte slots, and the expansion's threads of 16 neighbouring pairs read 256 contiguous bytes
// per record index (per-lane runs of records were 16-byte stores scattered over as many sectors).
struct BaRecs {
    BaMoveRec *base;
    __device__ __forceinline__ BaMoveRec &operator[](int r) const { return base[((ptrdiff_t)r + 1) * 64]; }
};
__device__ __forceinline__ BaRecs ba_recs(const void *rec, const BaBundle &B, uint32_t lane)
{
    return BaRecs{const_cast<BaMoveRec *>(reinterpret_cast<const BaMoveRec *>(rec)) + B.rec_off + lane};
}
enum { BA_MV_UP = 0, BA_MV_LEFT = 1, BA_MV_DIAG = 2 };

// ----------------------------------------------------------------------------
// strip profile: prof[b][half][lane][16 rows] = score(row r of the strip, column symbol b), int8,
// built from the transposed table wt[c][a] (dwords of 4 column symbols) in LDS.
// Rows of the strip that lie above the matrix (t < front) take the padding code.
// NIB: the table bytes are 0..15 and the profile keeps them as nibbles,
// prof[b][half][lane][16 rows] in 8 bytes (row r of a half = nibble r): half the LDS.
// ----------------------------------------------------------------------------
template <int R, bool NIB = false>
__device__ __forceinline__ void ba_build_strip_profile(const uint32_t *wt, int nb, int nb4, int na_pad,
                                                       const uint8_t *code1, int pad_code,
                                                       const uint32_t (&aw)[R / 4], int s, int front,
                                                       uint8_t *prof_lane)
{
    constexpr int HALVES = R / 16;
    for (int c = 0; c < nb4; ++c) {
        const uint32_t *wrow = wt + c * na_pad;
#pragma unroll
        for (int hf = 0; hf < HALVES; ++hf) {
            uint32_t o[4][4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const uint32_t wq = aw[hf * 4 + q4];
                const int tq = s * R + (hf * 4 + q4) * 4;
                const uint32_t t0 = wrow[tq + 0 >= front ? (uint32_t)code1[wq & 0xffu] : (uint32_t)pad_code];
                const uint32_t t1 = wrow[tq + 1 >= front ? (uint32_t)code1[(wq >> 8) & 0xffu] : (uint32_t)pad_code];
                const uint32_t t2 = wrow[tq + 2 >= front ? (uint32_t)code1[(wq >> 16) & 0xffu] : (uint32_t)pad_code];
                const uint32_t t3 = wrow[tq + 3 >= front ? (uint32_t)code1[wq >> 24] : (uint32_t)pad_code];
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
                if (b < nb) {
                    if (NIB) {
                        // bytes b0 b1 b2 b3 (each < 16) of a dword -> 16 bits b0 | b1 << 4 | b2 << 8 | b3 << 12
                        auto nib4 = [](uint32_t x) {
                            x = (x | (x >> 4)) & 0x00ff00ffu;
                            return (x | (x >> 8)) & 0xffffu;
                        };
                        *reinterpret_cast<uint2 *>(prof_lane + (size_t)(b * HALVES + hf) * (64 * 8)) =
                            make_uint2(nib4(o[bb][0]) | (nib4(o[bb][1]) << 16), nib4(o[bb][2]) | (nib4(o[bb][3]) << 16));
                    } else {
                        *reinterpret_cast<uint4 *>(prof_lane + (size_t)(b * HALVES + hf) * (64 * 16)) =
                            make_uint4(o[bb][0], o[bb][1], o[bb][2], o[bb][3]);
                    }
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------
// back-to-front trace writer
// ----------------------------------------------------------------------------
// Writes bytes back to front through a 16-byte shift register: a new byte (the
// lowest address so far) enters at byte 0 and the rest moves up, so byte k of
// the register always belongs to address a+k.  Every time the write position
// reaches a 16-byte boundary the register is exactly that block and goes out as
// one dwordx4 store.  The two blocks that can be shared with a neighbouring slot
// -- the one holding the slot's end and the one holding the trace's first byte
// -- are written byte by byte in finish().
struct BaBackWriter {
    uint8_t *abase;            // buffer base rounded down to 16 bytes
    int64_t end;               // offset (from abase) one past the slot
    int64_t a;                 // offset of the last byte written (== end when empty)
    uint32_t w0, w1, w2, w3;   // shift register: byte k <-> address a + k
    uint32_t t0, t1, t2, t3;   // the register as it was for the (partial) block that holds the slot's end
    __device__ __forceinline__ void init(uint8_t *buf, uint64_t end_index)
    {
        const int mis = (int)((uint64_t)buf & 15);
        abase = buf - mis;
        end = (int64_t)end_index + mis;
        a = end;
        w0 = w1 = w2 = w3 = 0;
        t0 = t1 = t2 = t3 = 0;
    }
    __device__ __forceinline__ void put(uint32_t byte)
    {
        --a;
        w3 = __builtin_amdgcn_alignbit(w3, w2, 24);
        w2 = __builtin_amdgcn_alignbit(w2, w1, 24);
        w1 = __builtin_amdgcn_alignbit(w1, w0, 24);
        w0 = (w0 << 8) | byte;
        if ((a & 15) == 0) {
            if (a + 16 <= end) {
                *reinterpret_cast<uint4 *>(abase + a) = make_uint4(w0, w1, w2, w3);
            } else {
                t0 = w0; t1 = w1; t2 = w2; t3 = w3;
            }
        }
    }
    __device__ __forceinline__ int64_t length() const { return end - a; }
    // abase[lo + k] = byte k of (r0..r3) for lo + k < hi
    static __device__ __forceinline__ void write_bytes(uint8_t *abase, int64_t lo, int64_t hi,
                                                      uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3)
    {
        for (int64_t q = lo; q < hi; ++q) {
            abase[q] = (uint8_t)r0;
            r0 = __builtin_amdgcn_alignbit(r1, r0, 8);
            r1 = __builtin_amdgcn_alignbit(r2, r1, 8);
            r2 = __builtin_amdgcn_alignbit(r3, r2, 8);
            r3 >>= 8;
        }
    }
    __device__ __forceinline__ void finish()
    {
        const int64_t top_blk = end & ~(int64_t)15;  // block that holds the slot's end (partial if end & 15)
        if (a & 15) {                                // partial block at the front of the trace
            const int64_t blk = a & ~(int64_t)15;
            write_bytes(abase, a, blk + 16 < end ? blk + 16 : end, w0, w1, w2, w3);
            if (blk == top_blk) return;  // the trace never left the end block
        }
        if ((end & 15) && a <= top_blk) write_bytes(abase, top_blk, end, t0, t1, t2, t3);
    }
};

