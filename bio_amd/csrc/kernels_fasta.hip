// kernels_fasta.hip -- FASTA text in HBM -> the batch format of the aligner (concatenated sequence bytes +
// uint64 offsets), without a host pass over the bytes.  SURVEY.md 8(f) rank 1: the step before the DP path.
//
// Semantics: ReadFASTA, reference fasta.go:55-84 -- split at '\n', bytes.TrimSpace on every line (ASCII white
// space here: \t \n \v \f \r, space and the UTF-8 encoded Unicode spaces of Go's unicode.IsSpace), blank lines skipped, '>' opens a record (also one with no data),
// other lines are appended to the last record, data before the first header is the error "No header".
//
// HBM-bound byte work, five small kernels:
//   fa_count    newlines per 16 KB chunk (SWAR byte compare on 16-byte loads)
//   fa_scan1    exclusive scan of a short array in one block (chunk counts; partial sums of the line scan)
//   fa_starts   the start offset of every line, in order
//   fa_classify one thread per line: trims both ends (a few bytes each), kind = blank / header / data
//   fa_partial / fa_emit   scan of (data bytes, headers) over the lines, then 16 lanes per line copy the
//               line's bytes to their place (dwords) and headers record their spans
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

#define FA_CHUNK 16384        // bytes per block of fa_count / fa_starts: 256 threads x 64 bytes
#define FA_LINES_PER_BLOCK 2048  // lines per block of the line scan: 256 threads x 8

__device__ __forceinline__ uint32_t fa_newline_bits(uint32_t v)
{
    // 0x80 in every byte of v that equals '\n' (exact per byte: no borrow between bytes)
    const uint32_t x = v ^ 0x0a0a0a0au;
    const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | x | 0x7f7f7f7fu);
}

__device__ __forceinline__ bool fa_is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
// bytes.TrimSpace (fasta.go:63) also trims UTF-8 encoded Unicode white space (unicode.IsSpace): U+0085, U+00A0, U+1680,
// U+2000..U+200A, U+2028, U+2029, U+202F, U+205F, U+3000; anything else above 0x7f ends the trimming
__device__ __forceinline__ bool fa_space3(uint8_t a, uint8_t b, uint8_t c)
{
    if (a == 0xE1) return b == 0x9A && c == 0x80;
    if (a == 0xE2) return (b == 0x80 && (c <= 0x8A ? c >= 0x80 : (c == 0xA8 || c == 0xA9 || c == 0xAF))) || (b == 0x81 && c == 0x9F);
    if (a == 0xE3) return b == 0x80 && c == 0x80;
    return false;
}
// bytes of the white-space rune that starts at text[b] / that ends at text[t - 1], the line being [b, t); 0: none
__device__ __forceinline__ int fa_space_at(const uint8_t *text, uint64_t b, uint64_t t)
{
    const uint8_t c = text[b];
    if (fa_is_space(c)) return 1;
    if (c < 0x80) return 0;
    if (c == 0xC2 && b + 1 < t && (text[b + 1] == 0x85 || text[b + 1] == 0xA0)) return 2;
    if (b + 2 < t && fa_space3(c, text[b + 1], text[b + 2])) return 3;
    return 0;
}
__device__ __forceinline__ int fa_space_before(const uint8_t *text, uint64_t b, uint64_t t)
{
    const uint8_t c = text[t - 1];
    if (fa_is_space(c)) return 1;
    if (c < 0x80) return 0;
    if (t - b >= 2 && text[t - 2] == 0xC2 && (c == 0x85 || c == 0xA0)) return 2;
    if (t - b >= 3 && fa_space3(text[t - 3], text[t - 2], c)) return 3;
    return 0;
}

// block-wide exclusive scan of one value per thread (256 threads); returns the prefix, *total = block sum
__device__ __forceinline__ uint32_t fa_block_scan(uint32_t v, uint32_t *lds4, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += lds4[w];
        tot += lds4[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// thread t of a block owns bytes [chunk + 64 t, chunk + 64 t + 64); bytes at and beyond `len` read as zero
// (only the last chunk takes the careful path: nothing past the end of the text is touched)
__device__ __forceinline__ void fa_load64(const uint8_t *text, uint64_t chunk, uint64_t len, uint4 (&q)[4])
{
    const uint64_t p0 = chunk + (uint64_t)threadIdx.x * 64;
    if (chunk + FA_CHUNK <= len) {  // uniform
        const uint4 *src = reinterpret_cast<const uint4 *>(text + p0);
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = src[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t p = p0 + 16 * k;
        if (p + 16 <= len) {
            q[k] = *reinterpret_cast<const uint4 *>(text + p);
        } else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int c = 0; c < 16; ++c)
                if (p + c < len) w[c >> 2] |= (uint32_t)text[p + c] << (8 * (c & 3));
            q[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

__global__ void __launch_bounds__(256) fa_count_kernel(BaFastaParams P)
{
    __shared__ uint32_t lds4[4];
    uint4 q[4];
    fa_load64(P.text, (uint64_t)blockIdx.x * FA_CHUNK, P.len, q);
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        c += __popc(fa_newline_bits(q[k].x)) + __popc(fa_newline_bits(q[k].y)) + __popc(fa_newline_bits(q[k].z)) +
             __popc(fa_newline_bits(q[k].w));
    uint32_t total;
    (void)fa_block_scan(c, lds4, &total);
    if (threadIdx.x == 0) P.chunk_count[blockIdx.x] = total;
}

// exclusive scan of (a, b) pairs of n entries in ONE block of 1024 threads; totals to tot[0], tot[1]
__global__ void __launch_bounds__(1024) fa_scan1_kernel(uint64_t *a, uint64_t *b, uint64_t n, uint64_t *tot)
{
    __shared__ uint64_t wa[16], wb[16];
    __shared__ uint64_t carry[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry[0] = carry[1] = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t va = i < n ? a[i] : 0, vb = (b && i < n) ? b[i] : 0;
        uint64_t ia = va, ib = vb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t oa = __shfl_up(ia, d), ob = __shfl_up(ib, d);
            if (lane >= d) {
                ia += oa;
                ib += ob;
            }
        }
        if (lane == 63) {
            wa[wave] = ia;
            wb[wave] = ib;
        }
        __syncthreads();
        uint64_t pa = carry[0], pb = carry[1], ta = 0, tb = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) {
                pa += wa[w];
                pb += wb[w];
            }
            ta += wa[w];
            tb += wb[w];
        }
        if (i < n) {
            a[i] = pa + ia - va;
            if (b) b[i] = pb + ib - vb;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry[0] += ta;
            carry[1] += tb;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tot[0] = carry[0];
        tot[1] = carry[1];
    }
}

// line k (k >= 1) starts right after the k-th newline; line 0 starts at 0 (written by the host side memset)
__global__ void __launch_bounds__(256) fa_starts_kernel(BaFastaParams P)
{
    __shared__ uint32_t lds4[4];
    const uint64_t chunk = (uint64_t)blockIdx.x * FA_CHUNK;
    uint4 q[4];
    fa_load64(P.text, chunk, P.len, q);
    const uint32_t w[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                            q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
    uint32_t bits[16], c = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        bits[k] = fa_newline_bits(w[k]);
        c += __popc(bits[k]);
    }
    uint32_t total;
    const uint32_t pre = fa_block_scan(c, lds4, &total);
    uint64_t idx = P.chunk_count[blockIdx.x] /* scanned: newlines before this chunk */ + pre + 1;
    const uint64_t pos0 = chunk + (uint64_t)threadIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t m = bits[k];
        while (m) {
            const int bit = __builtin_ctz(m);  // 7, 15, 23 or 31: byte bit / 8 of dword k
            m &= m - 1;
            P.line_start[idx++] = pos0 + (uint64_t)k * 4 + (uint64_t)(bit >> 3) + 1;
        }
    }
}

// four lines per thread (their loads in flight together: the kernel waits on memory, not on arithmetic): trimmed
// span and kind; ldata[l] = bytes the line adds to its record, lhdr[l] = 1 for a header
#define FA_CLASSIFY_LINES 4
__global__ void __launch_bounds__(256) fa_classify_kernel(BaFastaParams P)
{
    const uint64_t l0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t b[FA_CLASSIFY_LINES], t[FA_CLASSIFY_LINES];
    bool live[FA_CLASSIFY_LINES];
#pragma unroll
    for (int q = 0; q < FA_CLASSIFY_LINES; ++q) {
        const uint64_t l = l0 + q * stride;
        live[q] = l < P.n_lines;
        const uint64_t lc = live[q] ? l : 0;
        b[q] = P.line_start[lc];
        t[q] = (lc + 1 < P.n_lines) ? P.line_start[lc + 1] - 1 : P.len;  // the newline itself is not part of the line
    }
    uint8_t first[FA_CLASSIFY_LINES], last[FA_CLASSIFY_LINES];
#pragma unroll
    for (int q = 0; q < FA_CLASSIFY_LINES; ++q) {  // the common case needs these two bytes only
        first[q] = b[q] < t[q] ? P.text[b[q]] : (uint8_t)'x';
        last[q] = b[q] < t[q] ? P.text[t[q] - 1] : (uint8_t)'x';
    }
#pragma unroll
    for (int q = 0; q < FA_CLASSIFY_LINES; ++q) {
        if (!live[q]) continue;
        const uint64_t l = l0 + q * stride;
        uint64_t bb = b[q], tt = t[q];
        if (P.reader) {
            // bufio.Reader.ReadLine (fasta.go:117): "\n" or "\r\n" ends a line and is dropped, nothing else is;
            // the last line may lack its terminator, and then keeps a trailing "\r"
            const bool terminated = l + 1 < P.n_lines;
            const uint64_t raw = tt - bb + (terminated ? 1 : 0);
            if (terminated ? raw > P.line_limit : raw >= P.line_limit) atomicMin(P.too_long, (unsigned long long)l);
            if (terminated && tt > bb && last[q] == '\r') --tt;
        } else {
            if (fa_is_space(first[q]) || first[q] >= 0x80)
                for (int k; bb < tt && (k = fa_space_at(P.text, bb, tt)) != 0;) bb += k;
            if (fa_is_space(last[q]) || last[q] >= 0x80)
                for (int k; tt > bb && (k = fa_space_before(P.text, bb, tt)) != 0;) tt -= k;
        }
        const bool blank = (tt == bb);
        const bool header = !blank && (bb == b[q] ? first[q] : P.text[bb]) == '>';
        P.trim_begin[l] = bb;
        P.ldata[l] = (blank || header) ? 0 : tt - bb;
        P.lhdr[l] = header ? 1 : 0;
        P.trim_len[l] = (uint32_t)(tt - bb) | (header ? 0x80000000u : 0u);  // (a line is shorter than 2 GB)
    }
}

// block sums of (ldata, lhdr) over FA_LINES_PER_BLOCK lines
__global__ void __launch_bounds__(256) fa_partial_kernel(BaFastaParams P)
{
    __shared__ uint64_t sa[4], sb[4];
    const uint64_t base = (uint64_t)blockIdx.x * FA_LINES_PER_BLOCK;
    uint64_t a = 0, b = 0;
    for (int k = 0; k < FA_LINES_PER_BLOCK / 256; ++k) {
        const uint64_t l = base + (uint64_t)k * 256 + threadIdx.x;
        if (l < P.n_lines) {
            a += P.ldata[l];
            b += P.lhdr[l];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
    }
    if ((threadIdx.x & 63) == 0) {
        sa[threadIdx.x >> 6] = a;
        sb[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        P.part_data[blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3];
        P.part_hdr[blockIdx.x] = sb[0] + sb[1] + sb[2] + sb[3];
    }
}

// exclusive scan of (ldata, lhdr) inside each block of lines on top of the scanned block sums: after this
// ldata[l] = sequence bytes before line l, lhdr[l] = headers before line l
__global__ void __launch_bounds__(256) fa_line_scan_kernel(BaFastaParams P)
{
    __shared__ uint64_t wa[4], wb[4];
    __shared__ uint64_t carry[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * FA_LINES_PER_BLOCK;
    if (threadIdx.x == 0) {
        carry[0] = P.part_data[blockIdx.x];
        carry[1] = P.part_hdr[blockIdx.x];
    }
    __syncthreads();
    for (int k = 0; k < FA_LINES_PER_BLOCK / 256; ++k) {
        const uint64_t l = base + (uint64_t)k * 256 + threadIdx.x;
        const uint64_t va = l < P.n_lines ? P.ldata[l] : 0, vb = l < P.n_lines ? P.lhdr[l] : 0;
        uint64_t ia = va, ib = vb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t oa = __shfl_up(ia, d), ob = __shfl_up(ib, d);
            if (lane >= d) {
                ia += oa;
                ib += ob;
            }
        }
        if (lane == 63) {
            wa[wave] = ia;
            wb[wave] = ib;
        }
        __syncthreads();
        uint64_t pa = carry[0], pb = carry[1], ta = 0, tb = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) {
                pa += wa[w];
                pb += wb[w];
            }
            ta += wa[w];
            tb += wb[w];
        }
        if (l < P.n_lines) {
            P.ldata[l] = pa + ia - va;
            P.lhdr[l] = pb + ib - vb;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry[0] += ta;
            carry[1] += tb;
        }
        __syncthreads();
    }
}

// 16 lanes per line, four lines per wavefront: a data line goes to its place in the concatenated sequences
// four bytes per lane and round (dword loads and stores at whatever alignment the offsets have: global memory
// takes them), a header records where its record starts and the span of its own text
#define FA_EMIT_LINES 4  // lines a group of 16 lanes works on at once: their loads are all in flight together
__global__ void __launch_bounds__(256) fa_emit_kernel(BaFastaParams P)
{
    const int sub = threadIdx.x & 15;
    const uint64_t groups = (uint64_t)gridDim.x * 16;
    for (uint64_t l0 = ((uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4)) * FA_EMIT_LINES; l0 < P.n_lines;
         l0 += groups * FA_EMIT_LINES) {
        // the kernel is bound by the latency of its dependent loads (line record -> text -> store), not by
        // bytes: one line per group ran at 1.4 TB/s of traffic with every wavefront slot taken
        uint32_t tlk[FA_EMIT_LINES];
        uint64_t beg[FA_EMIT_LINES], out[FA_EMIT_LINES], rec[FA_EMIT_LINES];
#pragma unroll
        for (int q = 0; q < FA_EMIT_LINES; ++q) {
            const uint64_t l = min(l0 + q, P.n_lines - 1);
            tlk[q] = (l0 + q < P.n_lines) ? P.trim_len[l] : 0u;
            beg[q] = P.trim_begin[l];
            out[q] = P.ldata[l];
            rec[q] = P.lhdr[l];
        }
        const uint8_t *src[FA_EMIT_LINES];
        uint8_t *dst[FA_EMIT_LINES];
        uint32_t head[FA_EMIT_LINES], dwords[FA_EMIT_LINES], tl[FA_EMIT_LINES];
        uint32_t most = 0;
#pragma unroll
        for (int q = 0; q < FA_EMIT_LINES; ++q) {
            tl[q] = tlk[q] & 0x7fffffffu;
            if (tlk[q] >> 31) {  // header: where its record starts, and its own span of the text
                if (sub == 0) {
                    P.off[rec[q]] = out[q];
                    P.hdr_begin[rec[q]] = beg[q];
                    P.hdr_len[rec[q]] = tl[q];
                }
                tl[q] = 0;
            } else if (tl[q] != 0 && rec[q] == 0) {  // data before the first header (fasta.go:72-73)
                if (sub == 0 && !P.reader) atomicOr(P.error, 1u);  // (ReadSeq ignores it, fasta.go:137-138)
                tl[q] = 0;
            }
            src[q] = P.text + beg[q];
            dst[q] = P.seq + out[q];
            // bytes up to the first aligned dword of the destination, aligned dword stores (the loads take
            // whatever alignment is left), then the last few bytes
            head[q] = min((uint32_t)((0 - out[q]) & 3), tl[q]);
            dwords[q] = (tl[q] - head[q]) >> 2;
            most = max(most, dwords[q]);
        }
        for (uint32_t k = sub; k < most; k += 16) {
            uint32_t v[FA_EMIT_LINES];
#pragma unroll
            for (int q = 0; q < FA_EMIT_LINES; ++q)
                if (k < dwords[q]) __builtin_memcpy(&v[q], src[q] + head[q] + 4 * k, 4);
#pragma unroll
            for (int q = 0; q < FA_EMIT_LINES; ++q)
                if (k < dwords[q]) *reinterpret_cast<uint32_t *>(dst[q] + head[q] + 4 * k) = v[q];
        }
#pragma unroll
        for (int q = 0; q < FA_EMIT_LINES; ++q) {
            if ((uint32_t)sub < head[q]) dst[q][sub] = src[q][sub];
            const uint32_t done = head[q] + dwords[q] * 4;
            if (done + sub < tl[q]) dst[q][done + sub] = src[q][done + sub];
        }
    }
}

extern "C" hipError_t ba_launch_fasta_count(const BaFastaParams &P, hipStream_t st)
{
    hipLaunchKernelGGL(fa_count_kernel, dim3((unsigned)P.n_chunks), dim3(256), 0, st, P);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_fasta_scan1(uint64_t *a, uint64_t *b, uint64_t n, uint64_t *tot, hipStream_t st)
{
    hipLaunchKernelGGL(fa_scan1_kernel, dim3(1), dim3(1024), 0, st, a, b, n, tot);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_fasta_starts(const BaFastaParams &P, hipStream_t st)
{
    hipLaunchKernelGGL(fa_starts_kernel, dim3((unsigned)P.n_chunks), dim3(256), 0, st, P);
    return hipGetLastError();
}

// classify, block sums
extern "C" hipError_t ba_launch_fasta_lines_a(const BaFastaParams &P, hipStream_t st)
{
    hipLaunchKernelGGL(fa_classify_kernel, dim3((unsigned)((P.n_lines + 1023) / 1024)), dim3(256), 0, st, P);  // 4 lines per thread
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fa_partial_kernel, dim3((unsigned)P.n_line_blocks), dim3(256), 0, st, P);
    return hipGetLastError();
}

// line scan on top of the scanned block sums, emit
extern "C" hipError_t ba_launch_fasta_lines_b(const BaFastaParams &P, hipStream_t st)
{
    hipLaunchKernelGGL(fa_line_scan_kernel, dim3((unsigned)P.n_line_blocks), dim3(256), 0, st, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint64_t blocks = (P.n_lines + 16 * 4 - 1) / (16 * 4);  // FA_EMIT_LINES lines per group of 16 lanes
    hipLaunchKernelGGL(fa_emit_kernel, dim3((unsigned)(blocks < 65536 * 16 ? (blocks ? blocks : 1) : 65536 * 16)), dim3(256), 0, st, P);
    return hipGetLastError();
}
