// api.cpp -- host side of the C ABI declared in include/bioalign.h:
// contexts, planning (shape-sorting pairs into bundles, sizing the HBM
// workspace), and the staged run (pack -> DP -> traceback) on HIP streams.
//
// No CPU alignment code lives here: when HIP is unusable every entry point
// fails with BIOALIGN_ENODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <chrono>
#include <cmath>
#include <thread>
#include <vector>

#include "../../include/bioalign.h"
#include "common.h"

extern "C" size_t ba_dp_linear_lds_bytes(int nb, int nb4, int na_pad, int R, int waves);
extern "C" hipError_t ba_launch_pack(const BaPackParams &P, hipStream_t st);
extern "C" hipError_t ba_launch_dp_linear(const BaDpParams &P, int mode, int R, int K, int waves, hipStream_t st);
extern "C" hipError_t ba_launch_trace_linear(const BaTraceParams &P, int mode, int R, int K, int phases, hipStream_t st);
extern "C" hipError_t ba_launch_iota(uint32_t *out, uint64_t n, hipStream_t st);
extern "C" hipError_t ba_launch_len_partial(const uint64_t *len, uint64_t n, uint64_t *part, hipStream_t st);
extern "C" hipError_t ba_launch_len_apply(const uint64_t *len, uint64_t n, const uint64_t *part, uint64_t *t_off,
                                          uint64_t *dense_end, hipStream_t st);
extern "C" hipError_t ba_launch_fasta_scan1(uint64_t *a, uint64_t *b, uint64_t n, uint64_t *tot, hipStream_t st);
extern "C" hipError_t ba_launch_ops_sizes(const uint64_t *t_len, uint64_t n, uint64_t *sizes, hipStream_t st);
extern "C" hipError_t ba_launch_ops_emit(const BaTraceParams &P, hipStream_t st);
extern "C" size_t ba_dp_affine_lds_bytes(int nb, int nb4, int na_pad, int R, int waves, int nib);
extern "C" hipError_t ba_launch_dp_affine(const BaDpParams &P, int mode, int R, int waves, int pipe, hipStream_t st);
extern "C" int ba_bp_max_rows(void);
extern "C" int ba_bp_max_local_score(void);
extern "C" hipError_t ba_launch_dp_bp(const BaDpParams &P, int mode, int max_words, hipStream_t st);
extern "C" hipError_t ba_launch_trace_bp(const BaTraceParams &P, int mode, int max_words, hipStream_t st);
extern "C" hipError_t ba_launch_dp_wide(const BaDpParams &P, int mode, int affine, hipStream_t st);
extern "C" hipError_t ba_launch_trace_wide(const BaTraceParams &P, int mode, int affine, hipStream_t st);
extern "C" hipError_t ba_launch_trace_affine(const BaTraceParams &P, int mode, int R, hipStream_t st);

namespace {

thread_local std::string g_create_error;

std::string vfmt(const char *fmt, va_list ap)
{
    char buf[1024];
    vsnprintf(buf, sizeof buf, fmt, ap);
    return std::string(buf);
}

}  // namespace

// Device memory that bioalign_batch keeps between calls (its workspace and staging buffers are the same sizes
// call after call; hipMalloc / hipFree of GBs cost milliseconds each way).  At most `cap` bytes are kept while
// unused (BIOALIGN_POOL_BYTES, default 64 GiB); everything goes back at bioalign_destroy.  Plans that callers
// create themselves do not use it: their memory is theirs to free.
struct DevPool {
    struct Blk {
        void *p;
        size_t bytes;
        int dev;
        bool used;
    };
    std::mutex mu;
    std::vector<Blk> blks;
    size_t cap = 64ull << 30;
    // a block of the device of at least `bytes` (and not wastefully larger), or a new one; null: out of memory
    void *get(int dev, size_t bytes)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            Blk *best = nullptr;
            for (Blk &b : blks)
                if (!b.used && b.dev == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
                    (!best || b.bytes < best->bytes))
                    best = &b;
            if (best) {
                best->used = true;
                return best->p;
            }
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            trim(0);  // give back what lies unused and try once more
            if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        }
        std::lock_guard<std::mutex> g(mu);
        blks.push_back({p, bytes, dev, true});
        return p;
    }
    void put(void *p)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            for (Blk &b : blks)
                if (b.p == p) b.used = false;
        }
        trim(cap);
    }
    // frees unused blocks, largest first, until at most `keep` bytes lie unused
    void trim(size_t keep)
    {
        std::lock_guard<std::mutex> g(mu);
        for (;;) {
            size_t idle = 0;
            long big = -1;
            for (size_t k = 0; k < blks.size(); ++k)
                if (!blks[k].used) {
                    idle += blks[k].bytes;
                    if (big < 0 || blks[k].bytes > blks[(size_t)big].bytes) big = (long)k;
                }
            if (idle <= keep || big < 0) return;
            (void)hipFree(blks[(size_t)big].p);
            blks.erase(blks.begin() + big);
        }
    }
    ~DevPool()
    {
        for (Blk &b : blks) (void)hipFree(b.p);
    }
};

// Scratch of the allocation-free entry point (bioalign_run_noalloc, include/bioalign.h): the whole workspace of one
// small batch -- bundle table, pair order, score tables, packed sequence words, boundary rows, residue / decision
// words, move records, end cells -- lives in this file-scope device array instead of hipMalloc'ed memory.  Every
// GPU the library's code object is loaded on holds its own copy at its own address until the process ends.  The
// name is a host-side stand-in, NOT a device address: host code only ever uses the per-device pointer that
// hipGetSymbolAddress returns (bioalign_create resolves it once per device of the context, with that device
// current) -- as kernel arguments, like the hipMalloc'ed workspace of the other entry points.
static __device__ __align__(256) unsigned char scratch[BIOALIGN_NOALLOC_SCRATCH_BYTES];

struct bioalign_ctx {
    // handles that point back at the context (plans, stripes, parsed FASTA texts) hold a reference each, so that
    // bioalign_destroy before their own destroy calls is not a use-after-free: the context goes with the last of them
    std::atomic<int> refs{1};
    std::vector<int> devices;
    std::vector<unsigned char *> scratch_ptr;  // per entry of `devices`: that GPU's address of `scratch`
    std::shared_ptr<DevPool> pool;  // bioalign_batch's device memory (see DevPool)
    std::string err;
    int code = 0;  // code of the last failure
    int fail(int c, const char *fmt, ...)
    {
        va_list ap;
        va_start(ap, fmt);
        err = vfmt(fmt, ap);
        va_end(ap);
        code = c;
        return c;
    }
};

#define HIP_TRY(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return (ctx)->fail(BIOALIGN_EHIP, "%s failed: %s (%s:%d)", #call,                \
                               hipGetErrorString(e_), __FILE__, __LINE__);                  \
    } while (0)

struct Chunk {
    uint32_t bundle_begin, bundle_end;
};

struct bioalign_plan {
    bioalign_ctx *ctx = nullptr;
    bool ctx_ref = false;  // holds a reference on ctx (plans handed to the caller; the library's own plans live inside a call)
    int device = 0;
    uint64_t n_pairs = 0;
    int mode = 0;
    bool affine = false;
    bool score_only = false;
    bool end_cells = false;  // score_only: the end cell of every pair goes to t_off (row) / t_len (column)
    int gap = 0, gap_open = 0, gap_ext = 0;
    uint8_t gap_symbol = '-';
    int na = 0, nb = 0, nb4 = 0, na_pad = 0;
    int R = 32, waves = 4;
    int pipe = 0;  // affine: one block per bundle, strips pipelined over its waves (kernels_affine.hip)
    int K = 2;  // residue bits per cell in the traceback matrix
    int nib = 0, kappa = 0;  // affine, large alphabets: 4-bit strip profile (kernels_affine.hip)
    int fkey = 0;            // affine local: float-built best-cell keys (kernels_affine.hip)
    int max_n = 0, max_m = 0;  // longest s1 / s2 of the batch
    int bp_words = 0;   // > 0: unit scoring (+1 / -1 / gap 1), global: the bit-parallel kernels over this many words of rows (kernels_bp.hip)
    bool wide = false;  // scores or penalties the packed encodings of the fast kernels cannot hold: 32-bit kernels (kernels_wide.hip)
    uint64_t cells = 0;
    std::vector<BaBundle> bundles;
    std::vector<Chunk> chunks;
    // device memory
    BaBundle *d_bundles = nullptr;
    uint32_t *d_perm = nullptr;
    uint8_t *d_code1 = nullptr, *d_code2 = nullptr;
    int32_t *d_table = nullptr;
    uint32_t *d_wt = nullptr;
    uint32_t *d_s1w = nullptr, *d_s2w = nullptr;
    const bioalign_plan *packed_by = nullptr;  // bioalign_plan_share_pack: read that plan's packed words
    uint64_t lens_hash = 0;                    // of the pair lengths in batch order
    int32_t *d_hb = nullptr;
    uint32_t *d_dir = nullptr;
    void *d_rec = nullptr;  // move records of the linear-gap walk (16 bytes per 16 trace columns)
    // dense trace layout (bioalign_batch: what goes back over PCIe is the traces, not their slots)
    bool want_dense = false, dense = false;
    bool ops = false;  // BIOALIGN_TRACE_OPS: compact move records instead of the two strings
    uint64_t *d_sizes = nullptr;
    uint64_t *d_dense_end = nullptr, *d_len_part = nullptr;
    BaLocalBest *d_best = nullptr;
    uint64_t ws_bytes = 0;
    // timing hooks: a fixed pool of event pairs, one pair per DP launch since the last reset
    std::vector<hipEvent_t> ev0, ev1;
    size_t ev_used = 0;
    bool ran = false;
    // traceback of slice k runs on a second stream while the DP kernel of slice k+1 runs on the caller's
    int slices = 1;
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev_slice;
    hipEvent_t ev_side_done = nullptr;
    std::vector<void *> allocs;
    std::shared_ptr<DevPool> pool;  // set: the plan's memory comes from (and goes back to) bioalign_batch's pool
    // bioalign_run_noalloc: the plan's device memory is carved out of this GPU's `scratch` array (no hipMalloc, no
    // stream or event of the plan's own); uploads go on `arena_stream`, the caller's
    unsigned char *arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    hipStream_t arena_stream = nullptr;
};

namespace {

int dev_alloc(bioalign_plan *pl, void **p, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (pl->arena) {
        const size_t at = (pl->arena_used + 255) & ~(size_t)255;
        if (at + bytes > pl->arena_cap)
            return pl->ctx->fail(BIOALIGN_EUNSUPPORTED,
                                 "bioalign_run_noalloc: this batch needs more than the %zu bytes of scratch the entry point "
                                 "owns (BIOALIGN_NOALLOC_SCRATCH_BYTES); use bioalign_plan_create / bioalign_plan_run",
                                 pl->arena_cap);
        *p = pl->arena + at;
        pl->arena_used = at + bytes;
        pl->ws_bytes += bytes;
        return 0;
    }
    if (pl->pool) {
        *p = pl->pool->get(pl->ctx->devices[pl->device], bytes);
        if (!*p) return pl->ctx->fail(BIOALIGN_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
    } else {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess)
            return pl->ctx->fail(BIOALIGN_ENOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    pl->allocs.push_back(*p);
    pl->ws_bytes += bytes;
    return 0;
}

int bit_length(uint64_t v)
{
    int b = 0;
    while (v) {
        ++b;
        v >>= 1;
    }
    return b;
}

uint64_t env_u64(const char *name, uint64_t dflt)
{
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    return strtoull(s, nullptr, 10);
}

}  // namespace

extern "C" int bioalign_abi_version(void) { return BIOALIGN_ABI_VERSION; }

extern "C" bioalign_ctx *bioalign_create(const int *device_ids, int n_devices)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("no usable HIP device: ") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                         " (this library has no CPU fallback)";
        return nullptr;
    }
    bioalign_ctx *ctx = new bioalign_ctx;
    if (!device_ids || n_devices <= 0) {
        ctx->devices.push_back(0);
    } else {
        for (int i = 0; i < n_devices; ++i) {
            if (device_ids[i] < 0 || device_ids[i] >= count) {
                g_create_error = "device id out of range";
                delete ctx;
                return nullptr;
            }
            ctx->devices.push_back(device_ids[i]);
        }
    }
    // this GPU's copy of the allocation-free entry point's scratch array: resolved once per device, with that
    // device current (each GPU holds the array at its own address)
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    for (int d : ctx->devices) {
        void *sp = nullptr;
        e = hipSetDevice(d);
        if (e == hipSuccess) e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sp) {
            g_create_error = std::string("hipGetSymbolAddress(scratch) failed on device ") + std::to_string(d) + ": " +
                             hipGetErrorString(e);
            (void)hipSetDevice(prev_dev);
            delete ctx;
            return nullptr;
        }
        ctx->scratch_ptr.push_back(static_cast<unsigned char *>(sp));
    }
    (void)hipSetDevice(prev_dev);
    ctx->pool = std::make_shared<DevPool>();
    ctx->pool->cap = env_u64("BIOALIGN_POOL_BYTES", 64ull << 30);
    return ctx;
}

static void ctx_retain(bioalign_ctx *ctx) { ctx->refs.fetch_add(1, std::memory_order_relaxed); }
static void ctx_release(bioalign_ctx *ctx)
{
    if (ctx && ctx->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete ctx;
}

extern "C" void bioalign_destroy(bioalign_ctx *ctx) { ctx_release(ctx); }

extern "C" const char *bioalign_last_error(const bioalign_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int bioalign_device_count(const bioalign_ctx *ctx) { return ctx ? (int)ctx->devices.size() : 0; }

extern "C" void bioalign_plan_destroy(bioalign_plan *pl)
{
    if (!pl) return;
    (void)hipSetDevice(pl->ctx->devices[pl->device]);
    for (void *p : pl->allocs) {
        if (pl->pool)
            pl->pool->put(p);
        else
            (void)hipFree(p);
    }
    for (hipEvent_t e : pl->ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->ev1) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->ev_slice) (void)hipEventDestroy(e);
    if (pl->ev_side_done) (void)hipEventDestroy(pl->ev_side_done);
    if (pl->side) (void)hipStreamDestroy(pl->side);
    bioalign_ctx *held = pl->ctx_ref ? pl->ctx : nullptr;
    delete pl;
    ctx_release(held);
}

extern "C" uint64_t bioalign_plan_workspace_bytes(const bioalign_plan *pl) { return pl ? pl->ws_bytes : 0; }
extern "C" uint64_t bioalign_plan_cells(const bioalign_plan *pl) { return pl ? pl->cells : 0; }

// lens1 / lens2: the pairs' lengths when the caller has them already (bioalign_all_pairs: the plan then never looks at
// in->s1_off / s2_off); null: taken from the offsets
static int plan_build(bioalign_plan *pl, const bioalign_batch_t *in, const uint32_t *lens1 = nullptr,
                      const uint32_t *lens2 = nullptr)
{
    bioalign_ctx *ctx = pl->ctx;
    if (in->mode < BIOALIGN_GLOBAL || in->mode > BIOALIGN_OVERLAP)
        return ctx->fail(BIOALIGN_EINVAL, "unknown mode %d", in->mode);
    if ((!lens1 && (!in->s1_off || !in->s2_off)) || !in->score_table || !in->code1 || !in->code2)
        return ctx->fail(BIOALIGN_EINVAL, "null offsets / score table / code maps");
    if (in->na <= 0 || in->nb <= 0 || in->na > 255 || in->nb > 255)
        return ctx->fail(BIOALIGN_EINVAL, "alphabet sizes must be in 1..255 (got %d, %d)", in->na, in->nb);
    const uint64_t N = in->n_pairs;
    if (N > 0xfffffff0ull) return ctx->fail(BIOALIGN_EINVAL, "too many pairs");
    pl->n_pairs = N;
    pl->mode = in->mode;
    pl->affine = in->gap_open != in->gap_extend;
    pl->score_only = (in->flags & BIOALIGN_SCORE_ONLY) != 0;
    pl->end_cells = pl->score_only && (in->flags & BIOALIGN_END_CELLS) != 0;
    pl->want_dense = (in->flags & BIOALIGN_DENSE_TRACES) != 0;
    pl->ops = !pl->score_only && (in->flags & BIOALIGN_TRACE_OPS) != 0;
    pl->gap = pl->gap_open = in->gap_open;
    pl->gap_ext = in->gap_extend;
    pl->gap_symbol = in->gap_symbol;
    pl->na = in->na;
    pl->nb = in->nb;
    if (pl->gap_open < 0 || pl->gap_ext < 0)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "negative gap penalty (%d, %d) is outside the fast kernels",
                         pl->gap_open, pl->gap_ext);
    if (pl->affine && pl->gap_open < pl->gap_ext)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "gap_open %d < gap_extend %d is outside the fast kernels",
                         pl->gap_open, pl->gap_ext);

    const bool pb_timing = env_u64("BIOALIGN_TIMING", 0) > 1;
    const auto pb_t0 = std::chrono::steady_clock::now();
    auto pb_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pb_t0).count(); };
    double pb_a = 0, pb_b = 0, pb_c = 0, pb_d = 0;
    // shapes
    std::vector<uint32_t> len1_own, len2_own;
    if (!lens1) {
        len1_own.resize(N);
        len2_own.resize(N);
        for (uint64_t p = 0; p < N; ++p) {
            if (in->s1_off[p + 1] < in->s1_off[p] || in->s2_off[p + 1] < in->s2_off[p])
                return ctx->fail(BIOALIGN_EINVAL, "offsets must be non-decreasing (pair %llu)", (unsigned long long)p);
            const uint64_t a = in->s1_off[p + 1] - in->s1_off[p], b = in->s2_off[p + 1] - in->s2_off[p];
            len1_own[p] = (uint32_t)std::min<uint64_t>(a, 0xffffffffu);
            len2_own[p] = (uint32_t)std::min<uint64_t>(b, 0xffffffffu);
        }
        if (N && (in->s1_off[0] != 0 || in->s2_off[0] != 0))
            return ctx->fail(BIOALIGN_EINVAL, "offsets must start at 0");
    }
    const uint32_t *const len1 = lens1 ? lens1 : len1_own.data(), *const len2 = lens1 ? lens2 : len2_own.data();
    uint64_t max_n = 0, max_m = 0, min_nm_max = 0;
    for (uint64_t p = 0; p < N; ++p) {
        const uint64_t a = len1[p], b = len2[p];
        // (a sum of independent terms: no multiply chain from pair to pair)
        pl->lens_hash += ((a * 0x9e3779b97f4a7c15ull + b) ^ (p * 0x100000001b3ull)) * (2 * p + 1);
        max_n = std::max(max_n, a);
        max_m = std::max(max_m, b);
        min_nm_max = std::max(min_nm_max, std::min(a, b));
        pl->cells += a * b;
    }
    if (max_n > 0x7fffffu || max_m > 0x7fffffu)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "a pair is longer than 8M symbols; use bioalign_long");
    pl->max_n = (int)max_n;
    pl->max_m = (int)max_m;

    // int8 profile with the bias folded in (kernels.hip: biased score space)
    // (affine: every level is biased by ext*(i+j) and the strip keeps H' - (open - ext), kernels_affine.hip)
    int bias = pl->affine ? pl->gap_open + pl->gap_ext : (pl->mode == BIOALIGN_LOCAL) ? pl->gap : 2 * pl->gap;
    // What the fast kernels' encodings cannot hold -- profile bytes outside int8, a score span beyond 8 residue bits,
    // affine working scores beyond the float patterns' 2^22 -- runs on the WIDE kernels (kernels_wide.hip): 32-bit
    // scores, the caller's table, a decision byte per cell.  Slow, and the reference's result for every input whose
    // scores stay within 30 bits (align.go:76 takes any int).  BIOALIGN_WIDE=1 forces it (tests).
    bool wide = env_u64("BIOALIGN_WIDE", 0) != 0;
    int wmax = 0, wmin = 0;
    int64_t wabs = 0;
    for (int i = 0; i < pl->na * pl->nb; ++i) {
        const int64_t v = (int64_t)in->score_table[i] + bias;
        if (v < -127 || v > 127) wide = true;
        wmax = std::max(wmax, in->score_table[i]);
        wmin = std::min(wmin, in->score_table[i]);
        wabs = std::max<int64_t>(wabs, std::llabs((long long)in->score_table[i]));
    }

    // Affine gaps over a large alphabet (proteins): the strip profile of 16 rows x nb symbols is what limits the
    // wavefronts per CU.  Where the scores span at most 16 values the profile keeps w - min(w) in 4 bits and the
    // kernel adds kappa = min(w) + open + ext back with the diagonal.  The rows of the first strip that lie above
    // the matrix must score `open` (free top row; kernels_affine.hip): nibble open - kappa = -min(w) - ext.
    int nib_pad = 0;
    if (!wide && pl->affine && (pl->mode == BIOALIGN_LOCAL || pl->mode == BIOALIGN_FITTING) && pl->nb > 6 &&
        wmax - wmin <= 15 && -wmin - pl->gap_ext >= 0 && -wmin - pl->gap_ext <= 15 &&
        env_u64("BIOALIGN_NIBBLE_PROFILE", 1) != 0) {
        pl->nib = 1;
        pl->kappa = wmin + pl->gap_open + pl->gap_ext;
        nib_pad = -wmin - pl->gap_ext;
        bias = -wmin;
    }
    pl->nb4 = (pl->nb + 3) / 4;
    pl->na_pad = ((pl->na + 1 + 31) / 32) * 32;
    // residue bits: adjacent working-space scores differ by 0 .. max(w)+2g (kernels.hip)
    if (!pl->affine) {
        const int64_t span = std::max<int64_t>((int64_t)wmax + 2 * (int64_t)pl->gap, 0) + 1;
        pl->K = span <= 4 ? 2 : span <= 16 ? 4 : span <= 256 ? 8 : 0;
        if (!pl->K) {
            wide = true;  // (more than 8 residue bits)
            pl->K = 8;
        }
    }
    if (!wide && (max_n + max_m + 2) * 128 >= (1ull << 31)) wide = true;
    if (pl->affine) {
        // working-space magnitudes stay far below the kernels' "minus infinity" and floor offsets
        if ((max_n + max_m + 2) * (uint64_t)(128 + pl->gap_open + pl->gap_ext) >= (1ull << 22)) wide = true;
        // local: (true score << key_shift) must fit 31 bits (position key, kernels_affine.hip)
        const int key_shift = bit_length(max_m | 1) + 1;
        if (pl->mode == BIOALIGN_LOCAL && (min_nm_max + 1) * (uint64_t)std::max(wmax, 1) >= (1ull << (31 - key_shift))) wide = true;
        pl->fkey = pl->mode == BIOALIGN_LOCAL && (min_nm_max + 2) * (uint64_t)std::max(wmax, 1) < (1ull << (24 - std::min(key_shift, 24)));
        if (pl->gap_open > 127) wide = true;
        pl->K = 0;
    }
    if (wide) {
        // every score of the matrix (and every E / F value a finite step below one) stays within 30 bits
        if ((double)(max_n + max_m + 2) * (double)(wabs + pl->gap_open + pl->gap_ext + 1) >= (double)(1ull << 29))
            return ctx->fail(BIOALIGN_EUNSUPPORTED, "scores of up to %lld over %llu + %llu symbols do not fit the 32-bit scores of the GPU kernels",
                             (long long)wabs, (unsigned long long)max_n, (unsigned long long)max_m);
        if (pl->ops)
            return ctx->fail(BIOALIGN_EUNSUPPORTED, "BIOALIGN_TRACE_OPS: not with scores that need the wide kernels");
        // one lane walks a pair's whole matrix, a byte per cell and lane: fine for reads, not for one huge pair
        if (max_n * max_m > (1ull << 26))
            return ctx->fail(BIOALIGN_EUNSUPPORTED, "a pair of %llu x %llu with scores or penalties outside the fast batch kernels' "
                             "range: use bioalign_long for long pairs", (unsigned long long)max_n, (unsigned long long)max_m);
        pl->wide = true;
        pl->nib = 0;
        pl->fkey = 0;
        if (!pl->affine) pl->K = 8;
    }
    // Unit scoring -- every substitution score +1 or -1, gap 1: BASELINE's DNA configuration -- in any of the four modes, with
    // at most four column symbols and rows that fit eight words runs BIT-PARALLEL (kernels_bp.hip): same packed words,
    // same two bits per cell of traceback matrix, same records.  BIOALIGN_BITPARALLEL=0 keeps the general kernels.
    // (local mode keeps the scores bit-sliced in eight planes: no alignment can score above min(len(s1), len(s2)))
    if (!wide && !pl->affine && pl->gap == 1 && pl->nb <= 4 && wmax == 1 && max_n <= (uint64_t)ba_bp_max_rows() &&
        (pl->mode != BIOALIGN_LOCAL || min_nm_max <= (uint64_t)ba_bp_max_local_score()) &&
        env_u64("BIOALIGN_BITPARALLEL", 1) != 0) {
        bool unit = true;
        for (int i = 0; i < pl->na * pl->nb && unit; ++i) unit = in->score_table[i] == 1 || in->score_table[i] == -1;
        if (unit) pl->bp_words = std::max<int>(1, (int)((max_n + 31) / 32));
    }
    // strip height and block shape: the strip profile (nb * R bytes per lane) lives in LDS
    pl->R = (pl->nb <= 6 && pl->K <= 4) ? 32 : 16;
    if (pl->bp_words) pl->R = 32;  // (K = 2 and nb <= 4 already say so: the words of rows are the 32-row strips)
    if (pl->wide) pl->R = 32;  // (only the packing's row padding depends on it)
    if (const uint64_t rows = env_u64("BIOALIGN_STRIP_ROWS", 0)) {  // experiments: 16 where 32 would be chosen
        if (rows == 16) pl->R = 16;
    }
    pl->waves = 4;
    while (!pl->wide && pl->waves > 1 && ba_dp_linear_lds_bytes(pl->nb, pl->nb4, pl->na_pad, pl->R, pl->waves) > 40 * 1024)
        pl->waves >>= 1;
    if (!pl->wide && ba_dp_linear_lds_bytes(pl->nb, pl->nb4, pl->na_pad, pl->R, pl->waves) > 160 * 1024)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "alphabet of %d column symbols needs more LDS than a CU has", pl->nb);
    const int R = pl->R;
    const int chw = 4 * (R * pl->K / 32);  // residue dwords per lane per group of 4 columns
    const uint64_t cpw = 32 / R;           // affine: columns per decision word

    pb_a = pb_ms();
    // shape-sort the pairs; identical shapes become bundles of <= 64
    // (one shape for all: the order is the caller's, written on the device -- no host array, no copy)
    bool uniform = true;
    for (uint64_t p = 1; p < N && uniform; ++p) uniform = len1[p] == len1[0] && len2[p] == len2[0];
    std::vector<uint32_t> perm;
    if (!uniform) {
        perm.resize(N);
        std::iota(perm.begin(), perm.end(), 0u);
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
            return len1[a] != len1[b] ? len1[a] < len1[b] : len2[a] < len2[b];
        });
    }
    auto at = [&](uint64_t q) -> uint64_t { return uniform ? q : perm[q]; };
    uint64_t budget = env_u64("BIOALIGN_WORKSPACE_BYTES", 96ull << 30);
    if (pl->arena) budget = ~0ull >> 1;  // one chunk; dev_alloc refuses what does not fit the scratch array
    pl->bundles.clear();
    pl->chunks.clear();
    uint64_t o_s1w = 0, o_s2w = 0, o_hb = 0, o_dir = 0, o_rec = 0;
    uint64_t m_s1w = 0, m_s2w = 0, m_hb = 0, m_dir = 0, m_rec = 0;
    // linear gaps: the walk records its moves and a second kernel expands them (kernels.hip)
    const bool use_rec = !pl->affine && !pl->score_only && !pl->wide;  // (the wide traceback writes the strings itself)
    uint32_t chunk_begin = 0;
    for (uint64_t q = 0; q < N;) {
        const uint32_t n = len1[at(q)], m = len2[at(q)];
        uint64_t e = q + 1;
        if (uniform)
            e = std::min<uint64_t>(N, q + BA_WAVE);
        else
            while (e < N && e - q < BA_WAVE && len1[perm[e]] == n && len2[perm[e]] == m) ++e;
        const uint64_t strips = (n + R - 1) / R;
        const uint64_t sz_s1w = strips * R / 4 * 64, sz_s2w = (uint64_t)((m + 3) / 4) * 64;
        // boundary deltas: two buffers (strip parity) of ceil(m / F) words per lane, F >= 4
        // (affine: two buffers of (hq, F') per column and lane; four decision planes per column word)
        // (affine local: behind them, the best cell of every strip, 16 bytes per lane -- the strips of a bundle are
        // spread over the waves of a block)
        const uint64_t sz_hb = pl->wide     ? ((uint64_t)m + 1) * 64 * (pl->affine ? 2 : 1)
                               : pl->affine ? 2 * (uint64_t)m * 64 * 2 + (pl->mode == BIOALIGN_LOCAL ? strips * 64 * 4 : 0)
                                            : 2 * (uint64_t)((m + 3) / 4 + 1) * 64;
        const uint64_t sz_dir_fast = pl->affine ? strips * ((m + cpw - 1) / cpw) * 64 * 4
                                                : strips * (uint64_t)((m + 3) / 4) * 64 * chw;
        // (wide: a decision byte per cell and lane; the H row -- and the F row -- of m + 1 scores per lane)
        // (bit-parallel kernels, score-only plans: scores and end cells come out of the DP kernel itself)
        const uint64_t sz_dir = pl->wide ? (uint64_t)n * m * 16 + 64 : (pl->bp_words && pl->score_only) ? 0 : sz_dir_fast;
        const uint64_t rec_cap = ((uint64_t)n + m) / 16 + 2;  // header + ceil((n + m) / 16) records
        const uint64_t sz_rec = use_rec ? rec_cap * 64 : 0;  // records of 16 bytes
        const uint64_t after = (o_s1w + sz_s1w + o_s2w + sz_s2w + o_hb + sz_hb + o_dir + sz_dir) * 4 + (o_rec + sz_rec) * 16;
        if (after > budget && pl->bundles.size() > chunk_begin) {
            pl->chunks.push_back({chunk_begin, (uint32_t)pl->bundles.size()});
            chunk_begin = (uint32_t)pl->bundles.size();
            o_s1w = o_s2w = o_hb = o_dir = o_rec = 0;
        }
        BaBundle B;
        B.n = n;
        B.m = m;
        B.first = (uint32_t)q;
        B.count = (uint32_t)(e - q);
        B.s1w_off = o_s1w;
        B.s2w_off = o_s2w;
        B.hb_off = o_hb;
        B.dir_off = o_dir;
        B.rec_off = o_rec;
        B.rec_cap = (uint32_t)rec_cap;
        B.pad = 0;
        pl->bundles.push_back(B);
        o_s1w += sz_s1w;
        o_s2w += sz_s2w;
        o_hb += sz_hb;
        o_dir += sz_dir;
        o_rec += sz_rec;
        m_s1w = std::max(m_s1w, o_s1w);
        m_s2w = std::max(m_s2w, o_s2w);
        m_hb = std::max(m_hb, o_hb);
        m_dir = std::max(m_dir, o_dir);
        m_rec = std::max(m_rec, o_rec);
        q = e;
    }
    if (pl->bundles.size() > chunk_begin) pl->chunks.push_back({chunk_begin, (uint32_t)pl->bundles.size()});

    // transposed int8 table: dword [c][a] = bytes score(a, 4c..4c+3) + bias; the padding
    // row (code na) and unused columns hold -128
    std::vector<uint32_t> wt((size_t)pl->nb4 * pl->na_pad, pl->nib ? 0u : 0x80808080u);
    for (int c = 0; c < pl->nb4; ++c)
        for (int a = 0; a < pl->na; ++a) {
            uint32_t w = 0;
            for (int q = 0; q < 4; ++q) {
                const int b = 4 * c + q;
                const int v = b < pl->nb ? in->score_table[a * pl->nb + b] + bias : (pl->nib ? 0 : -128);
                w |= (uint32_t)(uint8_t)(int8_t)v << (8 * q);
            }
            wt[(size_t)c * pl->na_pad + a] = w;
        }
    if (pl->affine && (pl->mode == BIOALIGN_LOCAL || pl->mode == BIOALIGN_FITTING)) {
        // free top row: the rows of the first strip that lie above the matrix reproduce
        // H'(0, j) = ext*j through their diagonal when the padding code scores `open`
        // (kernels_affine.hip); other modes keep -128 there
        for (int c = 0; c < pl->nb4; ++c) {
            uint32_t w = 0;
            for (int q = 0; q < 4; ++q) {
                const int v = pl->nib ? (4 * c + q < pl->nb ? nib_pad : 0) : (4 * c + q < pl->nb ? pl->gap_open : -128);
                w |= (uint32_t)(uint8_t)(int8_t)v << (8 * q);
            }
            wt[(size_t)c * pl->na_pad + pl->na] = w;
        }
    }

    pb_b = pb_ms();
    // device memory
    HIP_TRY(ctx, hipSetDevice(ctx->devices[pl->device]));
    int rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_bundles, pl->bundles.size() * sizeof(BaBundle)))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_perm, N * 4))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_code1, 256))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_code2, 256))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_table, (size_t)pl->na * pl->nb * 4))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_wt, wt.size() * 4))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_s1w, m_s1w * 4))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_s2w, m_s2w * 4))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_hb, m_hb * 4))) return rc;
    if ((rc = dev_alloc(pl, (void **)&pl->d_dir, m_dir * 4))) return rc;
    if (use_rec && (rc = dev_alloc(pl, &pl->d_rec, m_rec * 16))) return rc;
    // dense layout: the walks of the whole batch come first, then one scan of the lengths, then the expansion --
    // which needs every pair's move records at once: one chunk, linear gaps
    if (pl->ops && (!use_rec || pl->chunks.size() != 1))
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "BIOALIGN_TRACE_OPS needs linear gaps and a batch that runs in one workspace chunk");
    pl->dense = (pl->want_dense || pl->ops) && use_rec && pl->chunks.size() == 1;
    if (pl->ops && (rc = dev_alloc(pl, (void **)&pl->d_sizes, N * 8))) return rc;
    if (pl->dense) {
        if ((rc = dev_alloc(pl, (void **)&pl->d_dense_end, N * 8))) return rc;
        if ((rc = dev_alloc(pl, (void **)&pl->d_len_part, ((N + 2047) / 2048 + 2) * 8))) return rc;
    }
    if ((rc = dev_alloc(pl, (void **)&pl->d_best, (N ? N : 1) * sizeof(BaLocalBest)))) return rc;
    if (pl->affine && !pl->wide) {
        // Run the strips of a bundle as a pipeline over the waves of a block (as many waves as strips, up to
        // 8).  Few bundles: one wavefront each cannot fill 1024 SIMDs.  Many bundles: the row a strip hands
        // to the next one is read back while it is still in the L2 instead of from HBM, and the smaller work
        // units leave a shorter tail (1M protein pairs, 16-row strips: 1399 -> 1561 GCUPS; with the 32-row
        // strips of small alphabets one wavefront per bundle stays ahead on large batches: 1752 vs 1404).
        // BIOALIGN_PIPE_BELOW=<bundles> moves the batch size below which it is used (0: never).
        const int strips_max = (int)((max_n + R - 1) / R);
        // (round 3: with the progress words in LDS proper and no drain per column group the pipeline is ahead at every
        // batch size -- 400k reads of config 4: 3.7 against 2.7 TCUPS -- except for local mode with 32-row strips, whose
        // pipeline kernel needs 200 VGPRs: 2.1 against 2.5 TCUPS on 500k DNA pairs)
        const uint64_t pipe_below = env_u64("BIOALIGN_PIPE_BELOW", (R == 16 || pl->mode != BIOALIGN_LOCAL) ? ~0ull : 6144);
        pl->pipe = strips_max >= 2 && pl->bundles.size() < pipe_below;
        if (pl->pipe) {
            const int waves_max = R == 16 ? 16 : (pl->mode == BIOALIGN_LOCAL ? 8 : 12);  // (1024 threads of 128 VGPRs, 768 of 168, local: 512 of 256: kernels_affine.hip)
            pl->waves = std::min(8, std::max(2, strips_max));
            // the waves of a block take strips w, w + W, ...: seven waves when that leaves fewer of them idle in the last
            // round than eight (19 strips: 7 + 7 + 5 instead of 8 + 8 + 3; config 3 1969 -> 2032 GCUPS)
            if (strips_max > 8 && (strips_max + 6) / 7 * 7 < (strips_max + 7) / 8 * 8) pl->waves = 7;
            if (const uint64_t w = env_u64("BIOALIGN_PIPE_WAVES", 0))
                pl->waves = (int)std::min<uint64_t>(std::max<uint64_t>(w, 2), (uint64_t)waves_max);
            while (pl->waves > 2 && ba_dp_affine_lds_bytes(pl->nb, pl->nb4, pl->na_pad, R, pl->waves, pl->nib) > 80 * 1024)
                --pl->waves;
            if ((max_m + 3) / 4 >= 65536) pl->pipe = 0;  // progress word: 16 bits of column groups
            // a large alphabet: two strip profiles do not fit a CU's LDS, one wavefront per bundle does
            if (ba_dp_affine_lds_bytes(pl->nb, pl->nb4, pl->na_pad, R, pl->waves, pl->nib) > 160 * 1024) pl->pipe = 0;
        }
        if (!pl->pipe) {
            pl->waves = 1;
            // bundles per block: whatever puts the most wavefronts on a CU (160 KB of LDS, at most 16 resident here)
            size_t best_waves_per_cu = 0;
            for (int w = 4; w >= 1; w >>= 1) {
                const size_t lds = ba_dp_affine_lds_bytes(pl->nb, pl->nb4, pl->na_pad, R, w, pl->nib);
                const size_t per_cu = std::min<size_t>((160 * 1024) / lds * w, 16);
                if (per_cu > best_waves_per_cu) {
                    best_waves_per_cu = per_cu;
                    pl->waves = w;
                }
            }
        }
        if (ba_dp_affine_lds_bytes(pl->nb, pl->nb4, pl->na_pad, R, pl->waves, pl->nib) > 160 * 1024)
            return ctx->fail(BIOALIGN_EUNSUPPORTED, "alphabet of %d column symbols needs more LDS than a CU has", pl->nb);
    }
    pb_c = pb_ms();
    {
        // on a stream of the plan's own that does not synchronise with the null stream: a plan that is built while
        // another range's kernels run (bioalign_batch, bioalign_all_pairs) does not wait for them
        hipStream_t cs = pl->arena_stream;  // (allocation-free entry point: the caller's stream, nothing is created)
        if (!pl->arena) HIP_TRY(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipError_t e = hipSuccess;
        auto up = [&](void *dst, const void *src, size_t bytes) {
            if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs);
        };
        up(pl->d_bundles, pl->bundles.data(), pl->bundles.size() * sizeof(BaBundle));
        if (uniform) {
            if (e == hipSuccess) e = ba_launch_iota(pl->d_perm, N, cs);
        } else {
            up(pl->d_perm, perm.data(), N * 4);
        }
        up(pl->d_code1, in->code1, 256);
        up(pl->d_code2, in->code2, 256);
        up(pl->d_table, in->score_table, (size_t)pl->na * pl->nb * 4);
        up(pl->d_wt, wt.data(), wt.size() * 4);
        if (e == hipSuccess) e = hipStreamSynchronize(cs);  // (the sources are locals; runs come on the caller's streams)
        if (!pl->arena) (void)hipStreamDestroy(cs);
        if (e != hipSuccess) return ctx->fail(BIOALIGN_EHIP, "plan upload failed: %s", hipGetErrorString(e));
    }
    // Large batches: cut every chunk into slices so that the (latency-bound) traceback of one slice
    // overlaps the (VALU-bound) DP of the next.  Slices have disjoint workspace by construction.
    {
        size_t biggest = 0;
        for (const Chunk &c : pl->chunks) biggest = std::max<size_t>(biggest, c.bundle_end - c.bundle_begin);
        // Off by default: measured on MI355X (profiles/r01_notes.md) the two kernels contend for the same CUs and the
        // step gets slower (10.5 -> 11.4 ms with 4 slices, 13.4 ms with 8); BIOALIGN_SLICES=n turns it on.
        const uint64_t want = env_u64("BIOALIGN_SLICES", 1);
        pl->slices = (!pl->score_only && biggest >= 4096 && want > 1) ? (int)std::min<uint64_t>(want, 16) : 1;
        if (pl->dense) pl->slices = 1;  // (the scan of the lengths sits between the walks and the expansion of the whole batch)
        if (pl->arena) pl->slices = 1;  // (no side stream, no events: nothing is created)
        if (pl->slices > 1) {
            HIP_TRY(ctx, hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking));
            pl->ev_slice.resize(pl->slices);
            for (auto &e : pl->ev_slice) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&pl->ev_side_done, hipEventDisableTiming));
        }
    }
    pb_d = pb_ms();
    const size_t pool = pl->arena ? 0 : 256;  // (allocation-free entry point: no timing events)
    pl->ev0.resize(pool);
    pl->ev1.resize(pool);
    for (size_t c = 0; c < pool; ++c) {
        HIP_TRY(ctx, hipEventCreate(&pl->ev0[c]));
        HIP_TRY(ctx, hipEventCreate(&pl->ev1[c]));
    }
    if (pb_timing)
        fprintf(stderr, "bioalign plan_build %llu pairs: lengths %.2f ms, sort + bundles %.2f, alloc %.2f, copies %.2f, events %.2f\n",
                (unsigned long long)N, pb_a, pb_b - pb_a, pb_c - pb_b, pb_d - pb_c, pb_ms() - pb_d);
    return 0;
}

extern "C" bioalign_plan *bioalign_plan_create(bioalign_ctx *ctx, int device_index, const bioalign_batch_t *in)
{
    if (!ctx) return nullptr;
    if (!in) {
        ctx->fail(BIOALIGN_EINVAL, "null batch");
        return nullptr;
    }
    if (device_index < 0 || device_index >= (int)ctx->devices.size()) {
        ctx->fail(BIOALIGN_EINVAL, "device index %d out of range", device_index);
        return nullptr;
    }
    bioalign_plan *pl = new bioalign_plan;
    pl->ctx = ctx;
    pl->ctx_ref = true;
    ctx_retain(ctx);
    pl->device = device_index;
    if (plan_build(pl, in) != 0) {
        bioalign_plan_destroy(pl);
        return nullptr;
    }
    return pl;
}

extern "C" int bioalign_plan_run(bioalign_plan *pl, const uint8_t *d_s1, const uint64_t *d_s1_off,
                                 const uint8_t *d_s2, const uint64_t *d_s2_off, int64_t *d_scores,
                                 uint8_t *d_t1, uint8_t *d_t2, uint64_t *d_t_off, uint64_t *d_t_len,
                                 void *stream)
{
    if (!pl) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = pl->ctx;
    if (pl->n_pairs == 0) return 0;
    if (!d_s1_off || !d_s2_off || !d_scores)
        return ctx->fail(BIOALIGN_EINVAL, "null device pointer");
    if (pl->end_cells && (!d_t_off || !d_t_len))
        return ctx->fail(BIOALIGN_EINVAL, "BIOALIGN_END_CELLS needs t_off and t_len");
    if (pl->score_only && !pl->end_cells) d_t_off = d_t_len = nullptr;  // (whatever the caller passed is not touched)
    if (!pl->score_only && (!d_t1 || (!d_t2 && !pl->ops) || !d_t_off || !d_t_len))
        return ctx->fail(BIOALIGN_EINVAL, "trace buffers are required unless BIOALIGN_SCORE_ONLY is set");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->devices[pl->device]));
    uint32_t *const s1w = pl->packed_by ? pl->packed_by->d_s1w : pl->d_s1w;
    uint32_t *const s2w = pl->packed_by ? pl->packed_by->d_s2w : pl->d_s2w;
    for (size_t c = 0; c < pl->chunks.size(); ++c) {
        const Chunk ch = pl->chunks[c];
        if (!pl->packed_by) {
            BaPackParams pk;
            pk.bundles = pl->d_bundles;
            pk.bundle_begin = ch.bundle_begin;
            pk.bundle_end = ch.bundle_end;
            pk.perm = pl->d_perm;
            pk.s1 = d_s1;
            pk.s2 = d_s2;
            pk.s1_off = d_s1_off;
            pk.s2_off = d_s2_off;
            pk.s1w = s1w;
            pk.s2w = s2w;
            pk.rows_per_strip = pl->R;
            HIP_TRY(ctx, ba_launch_pack(pk, st));
        }

        const uint32_t nbun = ch.bundle_end - ch.bundle_begin;
        const int slices = pl->slices > 1 && nbun >= 4096 ? pl->slices : 1;
        for (int k = 0; k < slices; ++k) {
            const uint32_t b0 = ch.bundle_begin + (uint32_t)((uint64_t)nbun * k / slices);
            const uint32_t b1 = ch.bundle_begin + (uint32_t)((uint64_t)nbun * (k + 1) / slices);
            BaDpParams dp;
            dp.bundles = pl->d_bundles;
            dp.bundle_begin = b0;
            dp.bundle_end = b1;
            dp.s1w = s1w;
            dp.s2w = s2w;
            dp.hb = pl->d_hb;
            dp.dir = pl->d_dir;
            dp.best = pl->d_best;
            dp.wt = pl->d_wt;
            dp.code1 = pl->d_code1;
            dp.code2 = pl->d_code2;
            dp.pad_code = pl->na;
            dp.nb = pl->nb;
            dp.nb4 = pl->nb4;
            dp.na_pad = pl->na_pad;
            dp.gap = pl->gap;
            dp.gap_open = pl->gap_open;
            dp.gap_ext = pl->gap_ext;
            dp.nib = pl->nib;
            dp.kappa = pl->kappa;
            dp.fkey = pl->fkey;
            dp.pipe_bundles = (int)env_u64("BIOALIGN_PIPE_BUNDLES", 1);
            dp.no_dir = (pl->bp_words && pl->score_only) ? 1 : 0;
            const bool timed = pl->ev_used < pl->ev0.size();
            if (timed) HIP_TRY(ctx, hipEventRecord(pl->ev0[pl->ev_used], st));
            dp.table = pl->d_table;
            dp.wide_rows = pl->R;
            if (pl->bp_words)
                HIP_TRY(ctx, ba_launch_dp_bp(dp, pl->mode, pl->bp_words, st));
            else if (pl->wide)
                HIP_TRY(ctx, ba_launch_dp_wide(dp, pl->mode, pl->affine ? 1 : 0, st));
            else if (pl->affine)
                HIP_TRY(ctx, ba_launch_dp_affine(dp, pl->mode, pl->R, pl->waves, pl->pipe, st));
            else
                HIP_TRY(ctx, ba_launch_dp_linear(dp, pl->mode, pl->R, pl->K, pl->waves, st));
            if (timed) {
                HIP_TRY(ctx, hipEventRecord(pl->ev1[pl->ev_used], st));
                ++pl->ev_used;
            }

            hipStream_t ts = st;
            if (slices > 1) {
                ts = pl->side;
                HIP_TRY(ctx, hipEventRecord(pl->ev_slice[k], st));
                HIP_TRY(ctx, hipStreamWaitEvent(ts, pl->ev_slice[k], 0));
            }
            BaTraceParams tr;
            tr.bundles = pl->d_bundles;
            tr.bundle_begin = b0;
            tr.bundle_end = b1;
            tr.perm = pl->d_perm;
            tr.s1 = d_s1;
            tr.s2 = d_s2;
            tr.s1_off = d_s1_off;
            tr.s2_off = d_s2_off;
            tr.code1 = pl->d_code1;
            tr.code2 = pl->d_code2;
            tr.table = pl->d_table;
            tr.na = pl->na;
            tr.nb = pl->nb;
            tr.s1w = s1w;
            tr.s2w = s2w;
            tr.hb = pl->d_hb;
            tr.dir = pl->d_dir;
            tr.best = pl->d_best;
            tr.gap = pl->gap;
            tr.gap_open = pl->gap_open;
            tr.gap_ext = pl->gap_ext;
            tr.rows_per_strip = pl->R;
            tr.residue_bits = pl->K;
            tr.score_only = pl->score_only ? 1 : 0;
            tr.gap_symbol = pl->gap_symbol;
            tr.scores = d_scores;
            tr.t1 = d_t1;
            tr.t2 = d_t2;
            tr.t_off = d_t_off;
            tr.t_len = d_t_len;
            tr.rec = pl->d_rec;
            tr.dense_end = nullptr;
            tr.max_n = pl->max_n;
            tr.max_m = pl->max_m;
            if (pl->wide) {
                HIP_TRY(ctx, ba_launch_trace_wide(tr, pl->mode, pl->affine ? 1 : 0, ts));
            } else if (pl->affine) {
                HIP_TRY(ctx, ba_launch_trace_affine(tr, pl->mode, pl->R, ts));
            } else if (!pl->dense) {
                if (pl->bp_words) {
                    HIP_TRY(ctx, ba_launch_trace_bp(tr, pl->mode, pl->bp_words, ts));
                    HIP_TRY(ctx, ba_launch_trace_linear(tr, pl->mode, pl->R, pl->K, 2, ts));
                } else {
                    HIP_TRY(ctx, ba_launch_trace_linear(tr, pl->mode, pl->R, pl->K, 3, ts));
                }
            } else {
                // walk -> prefix sums of the lengths in pair order -> expansion to the dense places
                if (pl->bp_words)
                    HIP_TRY(ctx, ba_launch_trace_bp(tr, pl->mode, pl->bp_words, ts));
                else
                    HIP_TRY(ctx, ba_launch_trace_linear(tr, pl->mode, pl->R, pl->K, 1, ts));
                const uint64_t nparts = (pl->n_pairs + 2047) / 2048;
                // what is laid back to back: the traces themselves, or (BIOALIGN_TRACE_OPS) the compact records
                const uint64_t *lens = d_t_len;
                if (pl->ops) {
                    HIP_TRY(ctx, ba_launch_ops_sizes(d_t_len, pl->n_pairs, pl->d_sizes, ts));
                    lens = pl->d_sizes;
                }
                HIP_TRY(ctx, ba_launch_len_partial(lens, pl->n_pairs, pl->d_len_part, ts));
                HIP_TRY(ctx, ba_launch_fasta_scan1(pl->d_len_part, nullptr, nparts, pl->d_len_part + nparts, ts));
                HIP_TRY(ctx, ba_launch_len_apply(lens, pl->n_pairs, pl->d_len_part, d_t_off, pl->d_dense_end, ts));
                tr.dense_end = pl->d_dense_end;
                if (pl->ops)
                    HIP_TRY(ctx, ba_launch_ops_emit(tr, ts));
                else
                    HIP_TRY(ctx, ba_launch_trace_linear(tr, pl->mode, pl->R, pl->K, 2, ts));
            }
        }
        if (slices > 1) {
            // the caller's stream owns the results (and the workspace, for the next chunk or run) again
            HIP_TRY(ctx, hipEventRecord(pl->ev_side_done, pl->side));
            HIP_TRY(ctx, hipStreamWaitEvent(st, pl->ev_side_done, 0));
        }
    }
    pl->ran = true;
    return 0;
}

extern "C" int bioalign_plan_share_pack(bioalign_plan *pl, const bioalign_plan *src)
{
    if (!pl) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = pl->ctx;
    if (!src || src == pl) {
        pl->packed_by = nullptr;
        return 0;
    }
    if (src->ctx != ctx || src->device != pl->device)
        return ctx->fail(BIOALIGN_EINVAL, "share_pack: the plans belong to different contexts or devices");
    if (src->packed_by) return ctx->fail(BIOALIGN_EINVAL, "share_pack: the other plan does not pack for itself");
    if (src->n_pairs != pl->n_pairs || src->lens_hash != pl->lens_hash || src->R != pl->R ||
        src->bundles.size() != pl->bundles.size())
        return ctx->fail(BIOALIGN_EINVAL, "share_pack: the plans were not made for the same batch");
    // one pass over the batch each: a chunked plan reuses its packed buffers from chunk to chunk
    if (src->chunks.size() > 1 || pl->chunks.size() > 1)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "share_pack: a plan runs in several chunks (BIOALIGN_WORKSPACE_BYTES)");
    for (size_t b = 0; b < pl->bundles.size(); ++b) {
        const BaBundle &x = pl->bundles[b], &y = src->bundles[b];
        if (x.n != y.n || x.m != y.m || x.first != y.first || x.count != y.count || x.s1w_off != y.s1w_off ||
            x.s2w_off != y.s2w_off)
            return ctx->fail(BIOALIGN_EINVAL, "share_pack: the plans lay the batch out differently");
    }
    pl->packed_by = src;
    return 0;
}

extern "C" int bioalign_plan_dp_ms(bioalign_plan *pl, float *ms, int *n_launches)
{
    if (!pl || !ms) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = pl->ctx;
    *ms = 0.f;
    if (n_launches) *n_launches = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->devices[pl->device]));
    for (size_t c = 0; c < pl->ev_used; ++c) {
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, pl->ev0[c], pl->ev1[c]));
        *ms += t;
    }
    if (n_launches) *n_launches = (int)pl->ev_used;
    pl->ev_used = 0;
    return 0;
}

extern "C" const char *bioalign_plan_kernel(const bioalign_plan *pl)
{
    if (!pl) return "";
    return pl->bp_words ? "bitparallel" : pl->wide ? "wide" : pl->affine ? "affine" : "linear";
}

// The allocation-free variant of bioalign_plan_create + bioalign_plan_run (include/bioalign.h): the same planner and
// the same kernels, with the plan's device memory carved out of this GPU's `scratch` array.  Calls no hipMalloc /
// hipFree, creates no stream and no event; uploads its small tables on the caller's stream and returns when the
// work queued on that stream is done (its host-side tables are locals).
extern "C" int bioalign_run_noalloc(bioalign_ctx *ctx, int device_index, const bioalign_batch_t *in, const uint8_t *d_s1,
                                    const uint64_t *d_s1_off, const uint8_t *d_s2, const uint64_t *d_s2_off,
                                    int64_t *d_scores, uint8_t *d_t1, uint8_t *d_t2, uint64_t *d_t_off,
                                    uint64_t *d_t_len, void *stream)
{
    if (!ctx) return BIOALIGN_EINVAL;
    if (!in) return ctx->fail(BIOALIGN_EINVAL, "null batch");
    if (device_index < 0 || device_index >= (int)ctx->devices.size())
        return ctx->fail(BIOALIGN_EINVAL, "device index %d out of range", device_index);
    if (in->n_pairs == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->devices[device_index]));
    // One call at a time per GPU, whatever the context, thread or stream: every call carves its workspace out of that
    // GPU's one `scratch` array (the array belongs to the code object, not to a context).  The call blocks until its
    // work is done, so holding the lock for its duration costs nothing a correct caller could have had.
    static std::mutex noalloc_mu[64];
    std::lock_guard<std::mutex> noalloc_guard(noalloc_mu[(unsigned)ctx->devices[device_index] % 64u]);
    bioalign_plan pl;  // (host object only; nothing of it outlives the call)
    pl.ctx = ctx;
    pl.device = device_index;
    pl.arena = ctx->scratch_ptr[device_index];  // the pointer hipGetSymbolAddress gave for THIS device, never the array's name
    pl.arena_cap = BIOALIGN_NOALLOC_SCRATCH_BYTES;
    pl.arena_stream = (hipStream_t)stream;
    int rc = plan_build(&pl, in);
    if (rc == 0 && pl.chunks.size() > 1) rc = ctx->fail(BIOALIGN_EUNSUPPORTED, "bioalign_run_noalloc: the batch does not run in one chunk");
    if (rc == 0)
        rc = bioalign_plan_run(&pl, d_s1, d_s1_off, d_s2, d_s2_off, d_scores, d_t1, d_t2, d_t_off, d_t_len, stream);
    // the kernels read the plan's tables from the scratch array: the next call on this GPU overwrites them
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (rc == 0 && e != hipSuccess) rc = ctx->fail(BIOALIGN_EHIP, "bioalign_run_noalloc: %s", hipGetErrorString(e));
    return rc;
}

// One device's share of bioalign_batch: pairs [lo, hi) of the caller's batch.
static int run_share(bioalign_ctx *ctx, int dev_index, const bioalign_batch_t *in, bioalign_result_t *out,
                     uint64_t lo, uint64_t hi, std::string *err)
{
    // a private context object so that concurrent shares do not race on the error string
    bioalign_ctx local;
    local.devices = ctx->devices;
    auto fail = [&](int rc) {
        *err = local.err;
        return rc;
    };
    const uint64_t N = hi - lo;
    if (N == 0) return 0;
    // rebase the offsets of the share to 0
    std::vector<uint64_t> o1(N + 1), o2(N + 1);
    const uint64_t b1 = in->s1_off[lo], b2 = in->s2_off[lo];
    for (uint64_t p = 0; p <= N; ++p) {
        o1[p] = in->s1_off[lo + p] - b1;
        o2[p] = in->s2_off[lo + p] - b2;
    }
    // BIOALIGN_TIMING=1: where the wall time of a share goes, one line on stderr
    const bool timing = env_u64("BIOALIGN_TIMING", 0) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count();
    };
    const auto t_start = now();
    double t_plan = 0, t_alloc = 0, t_h2d = 0, t_kern = 0, t_d2h = 0;
    bioalign_batch_t sub = *in;
    sub.s1 = in->s1 ? in->s1 + b1 : nullptr;
    sub.s2 = in->s2 ? in->s2 + b2 : nullptr;
    sub.s1_off = o1.data();
    sub.s2_off = o2.data();
    sub.n_pairs = N;
    sub.flags |= BIOALIGN_DENSE_TRACES;  // where the plan can: only the traces themselves go back over PCIe
    // (a plan of bioalign_batch's own: its memory comes from the context's pool)
    const std::shared_ptr<DevPool> pool = ctx->pool;
    const int hip_dev = ctx->devices[dev_index];
    if (hipSetDevice(hip_dev) != hipSuccess) return fail(local.fail(BIOALIGN_EHIP, "hipSetDevice failed"));
    bioalign_plan *pl = new bioalign_plan;
    pl->ctx = &local;
    pl->device = dev_index;
    pl->pool = pool;
    if (plan_build(pl, &sub) != 0) {
        bioalign_plan_destroy(pl);
        return fail(local.code ? local.code : BIOALIGN_EINVAL);
    }
    t_plan = ms_since(t_start);
    auto t_mark = now();
    const bool traces = !pl->score_only;
    const bool end_cells = pl->end_cells;
    const uint64_t bytes1 = o1[N], bytes2 = o2[N], cap = bytes1 + bytes2;
    const bool ops = pl->ops;
    // BIOALIGN_TRACE_OPS: the share's records go to the caller's t1 from byte ops_base(lo) on (see bioalign_ops_capacity)
    auto ops_base = [&](uint64_t p) { return 16 * p + 4 * ((in->s1_off[p] + in->s2_off[p]) / 16); };
    const uint64_t t1_bytes = ops ? ops_base(hi) - ops_base(lo) : cap;
    uint8_t *d_s1 = nullptr, *d_s2 = nullptr, *d_t1 = nullptr, *d_t2 = nullptr;
    uint64_t *d_o1 = nullptr, *d_o2 = nullptr, *d_toff = nullptr, *d_tlen = nullptr;
    int64_t *d_sc = nullptr;
    int rc = 0;
    hipStream_t st = nullptr;
    std::vector<uint64_t> toff, tlen;
#define SHARE_TRY(call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            rc = local.fail(BIOALIGN_EHIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
            goto done;                                                                       \
        }                                                                                    \
    } while (0)
    SHARE_TRY(hipSetDevice(ctx->devices[dev_index]));
    SHARE_TRY(hipStreamCreate(&st));
#define SHARE_GET(ptr, bytes)                                                                \
    do {                                                                                     \
        *(void **)&(ptr) = pool->get(hip_dev, (bytes));                                      \
        if (!(ptr)) {                                                                        \
            rc = local.fail(BIOALIGN_ENOMEM, "hipMalloc of %zu bytes failed", (size_t)(bytes)); \
            goto done;                                                                       \
        }                                                                                    \
    } while (0)
    SHARE_GET(d_s1, bytes1 + 16);
    SHARE_GET(d_s2, bytes2 + 16);
    SHARE_GET(d_o1, (N + 1) * 8);
    SHARE_GET(d_o2, (N + 1) * 8);
    SHARE_GET(d_sc, N * 8);
    if (traces) {
        SHARE_GET(d_t1, t1_bytes + 16);
        if (!ops) SHARE_GET(d_t2, cap + 16);
        SHARE_GET(d_toff, N * 8);
        SHARE_GET(d_tlen, N * 8);
    } else if (end_cells) {
        SHARE_GET(d_toff, N * 8);
        SHARE_GET(d_tlen, N * 8);
    }
#undef SHARE_GET
    t_alloc = ms_since(t_mark);
    t_mark = now();
    if (bytes1) SHARE_TRY(hipMemcpyAsync(d_s1, sub.s1, bytes1, hipMemcpyHostToDevice, st));
    if (bytes2) SHARE_TRY(hipMemcpyAsync(d_s2, sub.s2, bytes2, hipMemcpyHostToDevice, st));
    SHARE_TRY(hipMemcpyAsync(d_o1, o1.data(), (N + 1) * 8, hipMemcpyHostToDevice, st));
    SHARE_TRY(hipMemcpyAsync(d_o2, o2.data(), (N + 1) * 8, hipMemcpyHostToDevice, st));
    if (timing) {
        SHARE_TRY(hipStreamSynchronize(st));
        t_h2d = ms_since(t_mark);
        t_mark = now();
    }
    rc = bioalign_plan_run(pl, d_s1, d_o1, d_s2, d_o2, d_sc, d_t1, d_t2, d_toff, d_tlen, st);
    if (rc) goto done;
    if (timing) {
        SHARE_TRY(hipStreamSynchronize(st));
        t_kern = ms_since(t_mark);
        t_mark = now();
    }
    SHARE_TRY(hipMemcpyAsync(out->scores + lo, d_sc, N * 8, hipMemcpyDeviceToHost, st));
    if (end_cells) {
        SHARE_TRY(hipMemcpyAsync(out->t_off + lo, d_toff, N * 8, hipMemcpyDeviceToHost, st));
        SHARE_TRY(hipMemcpyAsync(out->t_len + lo, d_tlen, N * 8, hipMemcpyDeviceToHost, st));
    }
    if (traces) {
        toff.resize(N);
        tlen.resize(N);
        SHARE_TRY(hipMemcpyAsync(toff.data(), d_toff, N * 8, hipMemcpyDeviceToHost, st));
        SHARE_TRY(hipMemcpyAsync(tlen.data(), d_tlen, N * 8, hipMemcpyDeviceToHost, st));
        // the share's part of the caller's trace buffers starts at byte b1+b2; dense layout: the traces fill
        // its first sum-of-lengths bytes
        uint64_t used = cap;
        if (pl->dense) {
            SHARE_TRY(hipStreamSynchronize(st));
            used = toff[N - 1] + (ops ? 8 + 4 * ((tlen[N - 1] + 15) / 16) : tlen[N - 1]);
        }
        if (ops) {
            if (used > t1_bytes) {
                rc = local.fail(BIOALIGN_EHIP, "compact records exceed their capacity");
                goto done;
            }
            SHARE_TRY(hipMemcpyAsync(out->t1 + ops_base(lo), d_t1, used, hipMemcpyDeviceToHost, st));
        } else if (used) {
            SHARE_TRY(hipMemcpyAsync(out->t1 + b1 + b2, d_t1, used, hipMemcpyDeviceToHost, st));
            SHARE_TRY(hipMemcpyAsync(out->t2 + b1 + b2, d_t2, used, hipMemcpyDeviceToHost, st));
        }
    }
    SHARE_TRY(hipStreamSynchronize(st));
    if (traces)
        for (uint64_t p = 0; p < N; ++p) {
            out->t_off[lo + p] = toff[p] + (ops ? ops_base(lo) : b1 + b2);
            out->t_len[lo + p] = tlen[p];
        }
    t_d2h = ms_since(t_mark);
    t_mark = now();
done:
    if (st) (void)hipStreamSynchronize(st);  // nothing of this share still runs on memory that goes back to the pool
    pool->put(d_s1);
    pool->put(d_s2);
    pool->put(d_o1);
    pool->put(d_o2);
    pool->put(d_sc);
    pool->put(d_t1);
    pool->put(d_t2);
    pool->put(d_toff);
    pool->put(d_tlen);
    if (st) (void)hipStreamDestroy(st);
    // bioalign_plan_destroy touches pl->ctx (== &local), still alive here
    bioalign_plan_destroy(pl);
    if (timing)
        fprintf(stderr, "bioalign share dev %d, %llu pairs: plan %.2f ms, alloc %.2f, h2d %.2f, kernels %.2f, d2h %.2f, free %.2f, total %.2f\n",
                dev_index, (unsigned long long)N, t_plan, t_alloc, t_h2d, t_kern, t_d2h, ms_since(t_mark), ms_since(t_start));
    if (rc) *err = local.err;
    return rc;
#undef SHARE_TRY
}

extern "C" int bioalign_batch(bioalign_ctx *ctx, const bioalign_batch_t *in, bioalign_result_t *out)
{
    if (!ctx) return BIOALIGN_EINVAL;
    if (!in || !out) return ctx->fail(BIOALIGN_EINVAL, "null argument");
    if (!in->s1_off || !in->s2_off) return ctx->fail(BIOALIGN_EINVAL, "null offsets");
    const uint64_t N = in->n_pairs;
    if (N == 0) return 0;
    if (!out->scores) return ctx->fail(BIOALIGN_EINVAL, "null scores buffer");
    const bool traces = !(in->flags & BIOALIGN_SCORE_ONLY);
    if (traces && (!out->t1 || (!out->t2 && !(in->flags & BIOALIGN_TRACE_OPS)) || !out->t_off || !out->t_len))
        return ctx->fail(BIOALIGN_EINVAL, "trace buffers are required unless BIOALIGN_SCORE_ONLY is set");
    if (!traces && (in->flags & BIOALIGN_END_CELLS) && (!out->t_off || !out->t_len))
        return ctx->fail(BIOALIGN_EINVAL, "BIOALIGN_END_CELLS needs t_off and t_len");
    if ((in->s1_off[N] && !in->s1) || (in->s2_off[N] && !in->s2))
        return ctx->fail(BIOALIGN_EINVAL, "null sequence bytes");
    // split the pairs into contiguous ranges balanced by cell count, one per device -- and, for
    // large batches, two per device: each range is driven by a host thread of its own (copy in, kernels, copy
    // out), so the copies of one range overlap the kernels and the opposite copies of the other
    // (1M pairs x 150 bp on one GPU, host to host: 41.7 -> 31.8 ms; three and four ranges per device are slower
    // again).  BIOALIGN_HOST_SHARES sets the ranges per device.
    const int ndev = (int)ctx->devices.size();
    const int per_dev = (int)std::min<uint64_t>(std::max<uint64_t>(env_u64("BIOALIGN_HOST_SHARES", N >= 65536 ? 2 : 1), 1), 8);
    const int D = ndev * per_dev;
    std::vector<uint64_t> cut(D + 1, N);
    cut[0] = 0;
    if (D > 1) {
        std::vector<double> pre(N + 1, 0.0);
        for (uint64_t p = 0; p < N; ++p)
            pre[p + 1] = pre[p] + (double)(in->s1_off[p + 1] - in->s1_off[p] + 1) *
                                      (double)(in->s2_off[p + 1] - in->s2_off[p] + 1);
        for (int d = 1; d < D; ++d) {
            const double target = pre[N] * d / D;
            cut[d] = (uint64_t)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin());
            if (cut[d] < cut[d - 1]) cut[d] = cut[d - 1];
            if (cut[d] > N) cut[d] = N;
        }
    }
    std::vector<int> rcs(D, 0);
    std::vector<std::string> errs(D);
    if (D == 1) {
        rcs[0] = run_share(ctx, 0, in, out, 0, N, &errs[0]);
    } else {
        std::vector<std::thread> th;
        for (int d = 0; d < D; ++d)
            th.emplace_back([&, d] { rcs[d] = run_share(ctx, d / per_dev, in, out, cut[d], cut[d + 1], &errs[d]); });
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < D; ++d)
        if (rcs[d]) {
            ctx->err = errs[d];
            ctx->code = rcs[d];
            return rcs[d];
        }
    return 0;
}

// ----------------------------------------------------------------------------
// long mode: one pair at a time on the intra-sequence wavefront kernels (kernels_long.hip)
// ----------------------------------------------------------------------------
extern "C" int ba_long_rows_per_lane(int K);
extern "C" int ba_long_strips_per_group(void);
extern "C" hipError_t ba_launch_long(const BaLongParams &P, int mode, int T, int K, int phases, hipStream_t st);
extern "C" int ba_long_affine_rows_per_lane(void);
extern "C" hipError_t ba_launch_long_affine(const BaLongParams &P, int mode, int T, int phases, hipStream_t st);

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
    int alloc(bioalign_ctx *ctx, size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess)
            return ctx->fail(BIOALIGN_ENOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return 0;
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

int long_geometry(bioalign_ctx *ctx, const bioalign_batch_t *in, uint64_t n, uint64_t m, int *K_out, int *T_out,
                  std::vector<uint32_t> *rowprof);
int long_geometry_affine(bioalign_ctx *ctx, const bioalign_batch_t *in, uint64_t n, uint64_t m, int *K_out, int *T_out,
                         std::vector<uint32_t> *rowprof);

// one pair, n >= 1 and m >= 1, results into slot p of `out`
int long_pair(bioalign_ctx *ctx, const bioalign_batch_t *in, bioalign_result_t *out, uint64_t p, float *dp_ms)
{
    const uint64_t a1 = in->s1_off[p], a2 = in->s2_off[p];
    const uint64_t n = in->s1_off[p + 1] - a1, m = in->s2_off[p + 1] - a2;
    const int g = in->gap_open;
    const bool affine = in->gap_open != in->gap_extend;  // three-level Gotoh (align_affine.go) on the long-mode mapping
    const bool traces = !(in->flags & BIOALIGN_SCORE_ONLY);
    int K = 0, T = 0;
    std::vector<uint32_t> rowprof;
    if (int rc0 = affine ? long_geometry_affine(ctx, in, n, m, &K, &T, &rowprof) : long_geometry(ctx, in, n, m, &K, &T, &rowprof))
        return rc0;
    const int cpd = 32 / (T * K);
    const uint64_t rs = 64ull * T;
    const uint64_t strips = (n + rs - 1) / rs;
    const uint64_t nsuper = (strips + ba_long_strips_per_group() - 1) / ba_long_strips_per_group();
    const uint64_t wps = (m + 63 + cpd - 1) / cpd;

    HIP_TRY(ctx, hipSetDevice(ctx->devices[0]));
    DevBuf d_s1, d_s2, d_prof, d_c1, d_c2, d_tab, d_bnd, d_res, d_misc, d_t1, d_t2, d_ops, d_lmax;
    int rc;
    // Residue words: strips * wps * 256 bytes (K bits per cell).  Score only: none are stored.  A matrix whose residues
    // exceed the budget (BIOALIGN_LONG_RESIDUE_BYTES; default 70 % of the device's free memory) is traced back in
    // LINEAR SPACE (north_star's Hirschberg item): the DP runs once without storing residues -- every
    // workgroup's bottom row stays in HBM as a checkpoint -- and then row blocks of workgroups are recomputed bottom-up
    // with residues, each from the checkpoint above it and only as far right as the walk has come, and walked at
    // once.  At most twice the cells, the traces the same as with the whole matrix (the walk is the same walk).
    // All four modes.  Local mode with a linear gap (align.go:216-226): the first pass leaves the maximum and the first group
    // of rows that holds it; the first block recomputed is the one that ends with that group's workgroup, over all columns,
    // the end cell is found in its residues, and the walk carries its exact score from block to block until it reads 0.
    // Affine gaps (align_affine.go:12-167): the blocks hold decision words, the walk is handed from block to block as
    // (level, cell) and, in local mode, its score; the local end cell is known after the first pass (per-row maxima).
    const uint64_t resid_all = strips * wps * 256;
    uint64_t budget = env_u64("BIOALIGN_LONG_RESIDUE_BYTES", 0);
    if (budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        // 70 % of what is free once the pair's other buffers are taken: the checkpoint rows (8 bytes per column and
        // workgroup), the sequences, the two traces and their move codes, the local-mode maxima
        const uint64_t other = nsuper * ((m + 15) / 16 * 16) * 8 + (n + m) * 4 + strips * 64 * 4 * (affine ? 2 * (uint64_t)T : 1) + (1u << 20);
        budget = (uint64_t)(0.7 * (double)(free_b > other ? free_b - other : 0));
    }
    const bool checkpointed = traces && resid_all > budget;
    const uint64_t W_ = (uint64_t)ba_long_strips_per_group();
    // workgroups per recomputed block: as many as the budget holds (at least one)
    const uint64_t wgs_per_block = checkpointed ? std::max<uint64_t>(1, budget / (W_ * wps * 256)) : nsuper;
    // (score only: nothing reads residues or decision words, none are stored -- local mode takes its maximum from the
    // lanes' running maxima, the other modes read the bottom row)
    const bool no_resid = !traces;
    const uint64_t resid_bytes = no_resid ? 0 : checkpointed ? std::min(nsuper, wgs_per_block) * W_ * wps * 256 : resid_all;
    // local: per lane its largest score (affine: per row its largest true score and the step it was reached at)
    if (in->mode == BIOALIGN_LOCAL && (rc = d_lmax.alloc(ctx, strips * 64 * 4 * (affine ? 2 * (size_t)T : 1)))) return rc;
    if ((rc = d_s1.alloc(ctx, n + 16)) || (rc = d_s2.alloc(ctx, m + 16)) || (rc = d_prof.alloc(ctx, rowprof.size() * 4)) ||
        (rc = d_c1.alloc(ctx, 256)) || (rc = d_c2.alloc(ctx, 256)) || (rc = d_tab.alloc(ctx, (size_t)in->na * in->nb * 4)) ||
        (rc = d_bnd.alloc(ctx, nsuper * ((m + 15) / 16 * 16) * 8 + 256)) ||
        (rc = d_res.alloc(ctx, resid_bytes)) || (rc = d_misc.alloc(ctx, 64)))
        return rc;
    const uint64_t cap = n + m;
    if (traces && ((rc = d_t1.alloc(ctx, cap + 32)) || (rc = d_t2.alloc(ctx, cap + 32)) ||
                   (rc = d_ops.alloc(ctx, (cap / 16 + 2) * 4))))
        return rc;
    hipStream_t st = nullptr;
    HIP_TRY(ctx, hipMemcpyAsync(d_s1.p, in->s1 + a1, n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_s2.p, in->s2 + a2, m, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_prof.p, rowprof.data(), rowprof.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_c1.p, in->code1, 256, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_c2.p, in->code2, 256, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_tab.p, in->score_table, (size_t)in->na * in->nb * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(d_bnd.p, 0, nsuper * ((m + 15) / 16 * 16) * 8 + 256, st));  // tag 0: not handed over yet
    HIP_TRY(ctx, hipMemsetAsync(d_misc.p, 0, 64, st));

    BaLongParams P{};
    P.nsuper = (uint32_t)nsuper;
    P.bnd_stride = (uint32_t)((m + 15) / 16 * 16);
    P.epoch = 1;
    P.wg_begin = P.wg_count = 0;
    P.s1 = d_s1.as<uint8_t>();
    P.s2 = d_s2.as<uint8_t>();
    P.n = (uint32_t)n;
    P.m = (uint32_t)m;
    P.strips = (uint32_t)strips;
    P.words_per_strip = (uint32_t)wps;
    P.rowprof = d_prof.as<uint32_t>();
    P.code1 = d_c1.as<uint8_t>();
    P.code2 = d_c2.as<uint8_t>();
    P.pad_code = in->na;
    P.gap = g;
    P.gap_ext = in->gap_extend;
    P.bnd = d_bnd.as<int32_t>();
    P.resid = d_res.as<uint32_t>();
    // misc: [0] status, [8] score, [16] t_off, [24] t_len, [32..43] walk
    P.status = d_misc.as<uint32_t>();
    P.lane_max = d_lmax.as<int32_t>();
    P.table = d_tab.as<int32_t>();
    P.nb = in->nb;
    P.gap_symbol = in->gap_symbol;
    P.score_only = traces ? 0 : 1;
    P.score = reinterpret_cast<int64_t *>(d_misc.as<uint8_t>() + 8);
    P.t1 = d_t1.as<uint8_t>();
    P.t2 = d_t2.as<uint8_t>();
    P.t_off = reinterpret_cast<uint64_t *>(d_misc.as<uint8_t>() + 16);
    P.t_len = reinterpret_cast<uint64_t *>(d_misc.as<uint8_t>() + 24);
    P.walk = reinterpret_cast<uint32_t *>(d_misc.as<uint8_t>() + 32);
    P.ops = d_ops.as<uint32_t>();
    P.col0 = 0;  // the whole matrix on this device
    P.left_col = nullptr;
    P.right_col = nullptr;
    P.start_i = 0;
    P.start_j = -1;
    P.scan_only = 0;
    P.no_resid = no_resid ? 1 : 0;

    // diagnostics: when every strip started and ended (printed below)
    DevBuf d_stamps;
    const bool want_stamps = env_u64("BIOALIGN_LONG_STAMPS", 0) != 0;
    if (want_stamps) {
        if ((rc = d_stamps.alloc(ctx, strips * 16))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(d_stamps.p, 0, strips * 16, st));
        P.stamps = d_stamps.as<unsigned long long>();
    }

    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    hipError_t le = hipSuccess;
    uint64_t misc[8] = {0};
    if (!checkpointed) {
        le = affine ? ba_launch_long_affine(P, in->mode, T, BA_LONG_PHASE_DP | BA_LONG_PHASE_WALK, st)
                    : ba_launch_long(P, in->mode, T, K, BA_LONG_PHASE_DP | BA_LONG_PHASE_WALK, st);
    } else {
        // pass 1: scores only, then the end cell (the walk kernel's scan of the last workgroup's bottom row)
        BaLongParams P1 = P;
        P1.no_resid = 1;
        P1.scan_only = 1;
        auto launch = [&](const BaLongParams &Q, int phases) {
            return affine ? ba_launch_long_affine(Q, in->mode, T, phases, st) : ba_launch_long(Q, in->mode, T, K, phases, st);
        };
        le = launch(P1, BA_LONG_PHASE_DP | BA_LONG_PHASE_WALK);
        if (le == hipSuccess) le = hipMemcpyAsync(misc, d_misc.p, 64, hipMemcpyDeviceToHost, st);
        if (le == hipSuccess) le = hipStreamSynchronize(st);
        const uint32_t *mw = reinterpret_cast<const uint32_t *>(misc);
        uint64_t ci = mw[8 + 0], cj = mw[8 + 1], done = 0;  // the cell the walk is in; moves expanded so far
        const int64_t score_seen = (int64_t)misc[1];
        const uint64_t padw = nsuper * W_ - strips;
        const bool local = in->mode == BIOALIGN_LOCAL;
        // local: the end cell is not known yet, only the group of rows it lies in (walk[6] = strip * 64 + lane); the
        // blocks start with that strip's workgroup, and the first one is recomputed over all columns to find the cell
        uint64_t hi0 = nsuper;
        bool find_cell = false;
        if (local && le == hipSuccess && mw[0] == 0) {
            if (score_seen <= 0) {
                ci = cj = 0;  // an empty alignment (align.go:222-226 returns at once)
            } else if (affine) {
                hi0 = ((ci - 1 + (strips * rs - n)) / rs + padw) / W_ + 1;  // the workgroup of the end cell's row
            } else {
                hi0 = ((uint64_t)(mw[8 + 6] >> 6) + padw) / W_ + 1;
                find_cell = true;
                ci = n;  // (placeholders: somewhere in the block; the kernels take the cell from the search)
                cj = m;
            }
        }
        int64_t cur_score = score_seen;  // local: the score of the cell the walk is in
        uint32_t level = 0;              // affine: the level the walk is on (0: the middle one)
        // pass 2: row blocks bottom-up
        for (uint64_t hi = hi0; le == hipSuccess && mw[0] == 0 && ci > 0 && cj > 0 && hi > 0 && (!local || cur_score > 0 || level != 0);) {
            const uint64_t lo = hi > wgs_per_block ? hi - wgs_per_block : 0;
            const int64_t s_first = std::max<int64_t>((int64_t)(lo * W_) - (int64_t)padw, 0);
            const int64_t stop_row = lo == 0 ? 0 : (int64_t)n - (int64_t)(strips - (uint64_t)s_first) * (int64_t)rs;
            if ((int64_t)ci <= stop_row) {  // (the walk has left this block's rows already: cannot happen, blocks are contiguous)
                hi = lo;
                continue;
            }
            BaLongParams Pb = P;
            Pb.m = (uint32_t)cj;  // nothing right of the walk's column is needed again
            Pb.words_per_strip = (uint32_t)((cj + 63 + cpd - 1) / cpd);
            Pb.wg_begin = (uint32_t)lo;
            Pb.wg_count = (uint32_t)(hi - lo);
            Pb.first_strip = (int)s_first;
            Pb.stop_row = (int)stop_row;
            Pb.start_i = find_cell ? 0 : (int)ci;
            Pb.start_j = find_cell ? -1 : (int)cj;
            Pb.trace_done = (uint32_t)done;
            Pb.trace_cap = cap;
            find_cell = false;
            le = launch(Pb, BA_LONG_PHASE_DP | BA_LONG_PHASE_WALK);
            if (le == hipSuccess) le = hipMemcpyAsync(misc, d_misc.p, 64, hipMemcpyDeviceToHost, st);
            if (le == hipSuccess) le = hipStreamSynchronize(st);
            done += mw[8 + 2];
            ci = mw[8 + 4];
            cj = mw[8 + 5];
            if (local) cur_score = (int64_t)(int32_t)mw[8 + 3];
            if (affine) level = mw[8 + 6];
            hi = lo;
        }
        if (le == hipSuccess && mw[0] == 0 && (ci > 0 || cj > 0) && !local) {
            // the walk has reached (or started on) the matrix' left or top border: what remains are the border's own moves
            // (align.go:92-100; align_pair.go:110-113, 158-159, 187-188), which the walk emits without reading a residue
            BaLongParams Pb = P;
            Pb.start_i = (int)ci;
            Pb.start_j = (int)cj;
            Pb.trace_done = (uint32_t)done;
            Pb.trace_cap = cap;
            le = launch(Pb, BA_LONG_PHASE_WALK);
            if (le == hipSuccess) le = hipMemcpyAsync(misc, d_misc.p, 64, hipMemcpyDeviceToHost, st);
            if (le == hipSuccess) le = hipStreamSynchronize(st);
            done += mw[8 + 2];
        }
        misc[1] = (uint64_t)score_seen;  // (the blocks' walks start from a given cell and leave the score alone)
        if (le == hipSuccess && done == 0) {  // a walk of no moves (cannot happen for n, m >= 1 in these modes)
            misc[2] = cap;
            misc[3] = 0;
        }
    }
    if (le == hipSuccess) le = hipEventRecord(e1, st);
    if (le == hipSuccess && !checkpointed) le = hipMemcpyAsync(misc, d_misc.p, 64, hipMemcpyDeviceToHost, st);
    if (le == hipSuccess) le = hipStreamSynchronize(st);
    float ms = 0.f;
    if (le == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (le != hipSuccess) return ctx->fail(BIOALIGN_EHIP, "long-mode kernels failed: %s", hipGetErrorString(le));
    if (dp_ms) *dp_ms += ms;
    if (env_u64("BIOALIGN_TIMING", 0) > 1) {
        const uint32_t *mw = reinterpret_cast<const uint32_t *>(misc);
        fprintf(stderr, "bioalign_long: %.3f ms, walk: %u moves, %u tile requests, %u rounds\n", ms, mw[8 + 2], mw[8 + 6], mw[8 + 7]);
    }
    if (want_stamps) {
        std::vector<unsigned long long> sv(strips * 2);
        HIP_TRY(ctx, hipMemcpy(sv.data(), d_stamps.p, strips * 16, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (uint64_t k = 0; k < strips; ++k)
            if (sv[2 * k]) t0 = std::min(t0, sv[2 * k]);
        for (uint64_t k = 0; k < strips; ++k)
            fprintf(stderr, "bioalign_long strip %llu: start %.2f us, end %.2f us\n", (unsigned long long)k,
                    (double)(sv[2 * k] - t0) * 0.01, (double)(sv[2 * k + 1] - t0) * 0.01);
    }
    if ((uint32_t)misc[0] != 0)
        return ctx->fail(BIOALIGN_EHIP, "long mode: a strip gave up waiting for the strip above it (%u waits ran out)",
                         (unsigned)(uint32_t)misc[0]);
    out->scores[p] = (int64_t)misc[1];
    if (traces) {
        const uint64_t slot = a1 + a2, off = misc[2], len = misc[3];
        if (off + len != cap) return ctx->fail(BIOALIGN_EHIP, "long mode: inconsistent trace extent");
        HIP_TRY(ctx, hipMemcpy(out->t1 + slot + off, d_t1.as<uint8_t>() + off, len, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(out->t2 + slot + off, d_t2.as<uint8_t>() + off, len, hipMemcpyDeviceToHost));
        out->t_off[p] = slot + off;
        out->t_len[p] = len;
    }
    return 0;
}

// ----------------------------------------------------------------------------
// Long mode over column stripes: stripe g holds columns [c_g, c_g+1) of the matrix and all its
// rows, on device g (mod the context's devices).  A stripe's column 0 is its left neighbour's last column:
// n + 1 working-space scores, copied device to device (over xGMI between GPUs) once the neighbour's DP is
// done.  Every stripe keeps its own residue words; the walk starts in the stripe that holds the end cell and
// is handed on at each stripe's column 0 as one cell index; each stripe expands its own piece of the trace.
// One 100k x 100k pair is latency-bound on ONE GPU already, so stripes buy capacity -- a
// matrix whose residues do not fit one GPU's HBM -- not time.
// ----------------------------------------------------------------------------
}  // namespace

// One column stripe of one pair on one device (also the C ABI's bioalign_stripe).
struct bioalign_stripe {
    bioalign_ctx *ctx = nullptr;
    bool ctx_ref = false;  // holds a reference on ctx (stripes handed to the caller)
    int dev = 0;          // index into ctx->devices
    int mode = 0, T = 8, K = 2;
    uint64_t n = 0, c0 = 0, m = 0, m_total = 0, wps = 0, strips = 0, nsuper = 0;
    bool traces = true, dp_done = false;
    // row blocks: block k = workgroups [blk[k], blk[k+1]) = rows [blk_row[k], blk_row[k+1]) of column 0 .. (row 0 in block 0)
    std::vector<uint32_t> blk;
    std::vector<uint64_t> blk_row;
    uint32_t blocks_done = 0;
    DevBuf s1, s2, prof, c1, c2, bnd, res, misc, t1, t2, ops, left, right, rgran;
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    BaLongParams P{};
    ~bioalign_stripe()
    {
        if (ctx) (void)hipSetDevice(ctx->devices[dev]);
        if (done) (void)hipEventDestroy(done);
        if (st) (void)hipStreamDestroy(st);
    }
};

namespace {

// geometry and checks shared by the long-mode paths
int long_geometry(bioalign_ctx *ctx, const bioalign_batch_t *in, uint64_t n, uint64_t m, int *K_out, int *T_out,
                  std::vector<uint32_t> *rowprof)
{
    const int g = in->gap_open;
    const int bias = in->mode == BIOALIGN_LOCAL ? g : 2 * g;  // what the profile adds to w (kernels_long.hip)
    if (in->nb > 100)  // (four strip profiles of 256 bytes per symbol next to 47 KB of rings in a CU's 160 KB)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "long mode keeps a strip profile of 256 bytes per column symbol and wavefront "
                                                "in LDS: at most 100 distinct column symbols (got %d)", in->nb);
    int wmax = 0;
    for (int i = 0; i < in->na * in->nb; ++i) {
        const int64_t v = (int64_t)in->score_table[i] + bias;
        if (v < -127 || v > 127)
            return ctx->fail(BIOALIGN_EUNSUPPORTED, "score %d with gap %d does not fit the int8 row profile",
                             in->score_table[i], g);
        wmax = std::max(wmax, in->score_table[i]);
    }
    const int64_t span = std::max<int64_t>((int64_t)wmax + 2 * (int64_t)g, 0) + 1;
    const int K = span <= 4 ? 2 : span <= 16 ? 4 : span <= 256 ? 8 : 0;
    if (!K) return ctx->fail(BIOALIGN_EUNSUPPORTED, "score span %lld does not fit 8 residue bits", (long long)span);
    if ((n + m + 2) * 128 >= (1ull << 31))
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "pair too long for 32-bit working scores");
    *K_out = K;
    *T_out = ba_long_rows_per_lane(K);
    // signed bytes [na + 1][nb], packed into dwords for the upload; the padding row scores -128 against everything
    const size_t bytes = (size_t)(in->na + 1) * in->nb;
    rowprof->assign((bytes + 3) / 4, 0x80808080u);
    int8_t *tab = reinterpret_cast<int8_t *>(rowprof->data());
    for (int a = 0; a < in->na; ++a)
        for (int b = 0; b < in->nb; ++b) tab[(size_t)a * in->nb + b] = (int8_t)(in->score_table[a * in->nb + b] + bias);
    return 0;
}

// Affine gaps in long mode: the cell of kernels_affine.hip (levels biased by ext*(i+j), profile byte = w + open + ext,
// four decision bits per cell).  Rows of the first strip that lie above the matrix: -128 against everything, or `open`
// in the modes with a free top row (they reproduce H'(0, j) = ext*j through their diagonal, as in the batch planner).
int long_geometry_affine(bioalign_ctx *ctx, const bioalign_batch_t *in, uint64_t n, uint64_t m, int *K_out, int *T_out,
                         std::vector<uint32_t> *rowprof)
{
    const int go = in->gap_open, ge = in->gap_extend;
    if (ge < 0 || go < ge)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "gap penalties (%d, %d): the affine kernels need open >= extend >= 0", go, ge);
    if (go > 127) return ctx->fail(BIOALIGN_EUNSUPPORTED, "gap_open %d does not fit the int8 profile", go);
    if (in->nb > 64)
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "long mode keeps a strip profile of 256 bytes per column symbol and wavefront "
                                                "in LDS: at most 64 distinct column symbols (got %d)", in->nb);
    int wmax = 0;
    for (int i = 0; i < in->na * in->nb; ++i) {
        const int64_t v = (int64_t)in->score_table[i] + go + ge;
        if (v < -127 || v > 127)
            return ctx->fail(BIOALIGN_EUNSUPPORTED, "score %d with gaps (%d, %d) does not fit the int8 row profile",
                             in->score_table[i], go, ge);
        wmax = std::max(wmax, in->score_table[i]);
    }
    // working-space magnitudes stay far from the levels' minus infinity (-2^28)
    if ((n + m + 2) * (uint64_t)(128 + go + ge) >= (1ull << 27))
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "pair too long for the affine long-mode kernels' 32-bit working scores");
    if (in->mode == BIOALIGN_LOCAL && (std::min(n, m) + 1) * (uint64_t)std::max(wmax, 1) >= (1ull << 31))
        return ctx->fail(BIOALIGN_EUNSUPPORTED, "local scores do not fit 32 bits");
    *K_out = 4;
    *T_out = ba_long_affine_rows_per_lane();
    const bool top_free = in->mode == BIOALIGN_LOCAL || in->mode == BIOALIGN_FITTING;
    const size_t bytes = (size_t)(in->na + 1) * in->nb;
    rowprof->assign((bytes + 3) / 4, 0x80808080u);
    int8_t *tab = reinterpret_cast<int8_t *>(rowprof->data());
    for (int a = 0; a < in->na; ++a)
        for (int b = 0; b < in->nb; ++b) tab[(size_t)a * in->nb + b] = (int8_t)(in->score_table[a * in->nb + b] + go + ge);
    for (int b = 0; b < in->nb; ++b) tab[(size_t)in->na * in->nb + b] = (int8_t)(top_free ? go : -128);
    return 0;
}

// stripe of pair p of `in`: columns [c_begin, c_end) of the pair's matrix, all rows
int stripe_init(bioalign_stripe &x, bioalign_ctx *ctx, int dev, const bioalign_batch_t *in, uint64_t p, uint64_t c_begin,
                uint64_t c_end)
{
    const uint64_t a1 = in->s1_off[p], a2 = in->s2_off[p];
    const uint64_t n = in->s1_off[p + 1] - a1, m = in->s2_off[p + 1] - a2;
    if (n == 0 || c_begin >= c_end || c_end > m) return ctx->fail(BIOALIGN_EINVAL, "stripe: empty pair or column range");
    std::vector<uint32_t> rowprof;
    int rc = long_geometry(ctx, in, n, m, &x.K, &x.T, &rowprof);
    if (rc) return rc;
    x.ctx = ctx;
    x.dev = dev;
    x.mode = in->mode;
    x.n = n;
    x.m_total = m;
    x.c0 = c_begin;
    x.m = c_end - c_begin;
    x.traces = !(in->flags & BIOALIGN_SCORE_ONLY);
    const int cpd = 32 / (x.T * x.K);
    const uint64_t rs = 64ull * x.T;
    x.strips = (n + rs - 1) / rs;
    x.nsuper = (x.strips + ba_long_strips_per_group() - 1) / ba_long_strips_per_group();
    x.wps = (x.m + 63 + cpd - 1) / cpd;
    {
        // Row blocks: the DP of a stripe runs block by block, each block one launch over some of the
        // workgroups, so that a stripe can start on a block as soon as its left neighbour has handed that block's
        // rows of its last column over, while the neighbour goes on with the next block.  Same blocks in every
        // stripe of a pair (they depend on n alone).  BIOALIGN_STRIPE_BLOCKS sets their number (default 8).
        const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(env_u64("BIOALIGN_STRIPE_BLOCKS", 8), x.nsuper));
        const uint64_t per = (x.nsuper + want - 1) / want;
        const uint64_t W = (uint64_t)ba_long_strips_per_group(), pad = x.nsuper * W - x.strips;
        x.blk.clear();
        x.blk_row.clear();
        for (uint64_t b0 = 0; b0 < x.nsuper; b0 += per) {
            x.blk.push_back((uint32_t)b0);
            // first row of the first strip of workgroup b0 (strip b0 * W - pad, rows above the matrix belong to block 0)
            const int64_t s_first = (int64_t)(b0 * W) - (int64_t)pad;
            const int64_t row = s_first <= 0 ? 0 : (int64_t)n - (int64_t)(x.strips - (uint64_t)s_first) * (int64_t)rs + 1;
            x.blk_row.push_back((uint64_t)std::max<int64_t>(row, 0));
        }
        x.blk.push_back((uint32_t)x.nsuper);
        x.blk_row.push_back(n + 1);
        x.blk_row[0] = 0;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->devices[dev]));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&x.done, hipEventDisableTiming));
    const uint64_t cap = n + x.m;
    if ((rc = x.s1.alloc(ctx, n + 16)) || (rc = x.s2.alloc(ctx, x.m + 16)) || (rc = x.prof.alloc(ctx, rowprof.size() * 4)) ||
        (rc = x.c1.alloc(ctx, 256)) || (rc = x.c2.alloc(ctx, 256)) || (rc = x.bnd.alloc(ctx, x.nsuper * ((x.m + 15) / 16 * 16) * 8 + 256)) ||
        (rc = x.res.alloc(ctx, x.strips * x.wps * 256)) ||
        (rc = x.misc.alloc(ctx, 64)) || (rc = x.left.alloc(ctx, (n + 1) * 4)) || (rc = x.right.alloc(ctx, (n + 1) * 4)))
        return rc;
    if (x.traces && ((rc = x.t1.alloc(ctx, cap + 32)) || (rc = x.t2.alloc(ctx, cap + 32)) ||
                     (rc = x.ops.alloc(ctx, (cap / 16 + 2) * 4))))
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(x.s1.p, in->s1 + a1, n, hipMemcpyHostToDevice, x.st));
    HIP_TRY(ctx, hipMemcpyAsync(x.s2.p, in->s2 + a2 + x.c0, x.m, hipMemcpyHostToDevice, x.st));
    HIP_TRY(ctx, hipMemcpyAsync(x.prof.p, rowprof.data(), rowprof.size() * 4, hipMemcpyHostToDevice, x.st));
    HIP_TRY(ctx, hipMemcpyAsync(x.c1.p, in->code1, 256, hipMemcpyHostToDevice, x.st));
    HIP_TRY(ctx, hipMemcpyAsync(x.c2.p, in->code2, 256, hipMemcpyHostToDevice, x.st));
    HIP_TRY(ctx, hipMemsetAsync(x.bnd.p, 0, x.nsuper * ((x.m + 15) / 16 * 16) * 8 + 256, x.st));  // tag 0: not handed over yet
    HIP_TRY(ctx, hipMemsetAsync(x.misc.p, 0, 64, x.st));
    HIP_TRY(ctx, hipStreamSynchronize(x.st));  // rowprof is a local
    BaLongParams &P = x.P;
    P.s1 = x.s1.as<uint8_t>();
    P.s2 = x.s2.as<uint8_t>();
    P.n = (uint32_t)n;
    P.m = (uint32_t)x.m;
    P.strips = (uint32_t)x.strips;
    P.nsuper = (uint32_t)x.nsuper;
    P.bnd_stride = (uint32_t)((x.m + 15) / 16 * 16);
    P.epoch = 1;
    P.wg_begin = P.wg_count = 0;
    P.words_per_strip = (uint32_t)x.wps;
    P.rowprof = x.prof.as<uint32_t>();
    P.code1 = x.c1.as<uint8_t>();
    P.code2 = x.c2.as<uint8_t>();
    P.pad_code = in->na;
    P.gap = in->gap_open;
    P.bnd = x.bnd.as<int32_t>();
    P.resid = x.res.as<uint32_t>();
    P.status = x.misc.as<uint32_t>();
    P.table = nullptr;
    P.nb = in->nb;
    P.gap_symbol = in->gap_symbol;
    P.score_only = 0;
    P.score = reinterpret_cast<int64_t *>(x.misc.as<uint8_t>() + 8);
    P.t1 = x.t1.as<uint8_t>();
    P.t2 = x.t2.as<uint8_t>();
    P.t_off = reinterpret_cast<uint64_t *>(x.misc.as<uint8_t>() + 16);
    P.t_len = reinterpret_cast<uint64_t *>(x.misc.as<uint8_t>() + 24);
    P.walk = reinterpret_cast<uint32_t *>(x.misc.as<uint8_t>() + 32);
    P.ops = x.ops.as<uint32_t>();
    P.col0 = (uint32_t)x.c0;
    P.left_col = x.c0 > 0 ? x.left.as<int32_t>() : nullptr;
    P.right_col = x.right.as<int32_t>();
    P.start_i = 0;
    P.start_j = -1;
    P.scan_only = 0;
    return 0;
}

// queues row block k of the stripe's DP on its stream (blocks in order); the rows of x.left that the block reads --
// [blk_row[k], blk_row[k+1]) and the row above them -- must be in place on that stream by then
int stripe_dp_block(bioalign_stripe &x, uint32_t k)
{
    bioalign_ctx *ctx = x.ctx;
    if (k != x.blocks_done || k + 1 >= x.blk.size()) return ctx->fail(BIOALIGN_EINVAL, "stripe: row blocks run once each, in order");
    HIP_TRY(ctx, hipSetDevice(ctx->devices[x.dev]));
    BaLongParams Pk = x.P;
    Pk.wg_begin = x.blk[k];
    Pk.wg_count = x.blk[k + 1] - x.blk[k];
    hipError_t le = ba_launch_long(Pk, x.mode, x.T, x.K, BA_LONG_PHASE_DP, x.st);
    if (le != hipSuccess) return ctx->fail(BIOALIGN_EHIP, "long-mode DP kernel failed: %s", hipGetErrorString(le));
    x.blocks_done = k + 1;
    if (x.blocks_done + 1 == x.blk.size()) {
        HIP_TRY(ctx, hipEventRecord(x.done, x.st));
        x.dp_done = true;
    }
    return 0;
}

// queues the whole DP of the stripe on its stream; a stripe that is not the first one reads its left column (x.left)
int stripe_dp(bioalign_stripe &x)
{
    for (uint32_t k = x.blocks_done; k + 1 < x.blk.size(); ++k)
        if (int rc = stripe_dp_block(x, k)) return rc;
    return 0;
}

// the end cell this stripe offers (align.go:123-124; align_pair.go:169-182): its true score, its column inside
// the stripe, and the score to report if it is chosen.  Waits for the stripe's DP.
int stripe_end_cell(bioalign_stripe &x, int64_t *true_score, uint64_t *j_local, int64_t *score)
{
    bioalign_ctx *ctx = x.ctx;
    if (!x.dp_done) return ctx->fail(BIOALIGN_EINVAL, "stripe: end cell asked for before the DP");
    HIP_TRY(ctx, hipSetDevice(ctx->devices[x.dev]));
    BaLongParams Pq = x.P;
    Pq.scan_only = 1;
    uint32_t misc[16];
    hipError_t le = ba_launch_long(Pq, x.mode, x.T, x.K, BA_LONG_PHASE_WALK, x.st);
    if (le == hipSuccess) le = hipMemcpyAsync(misc, x.misc.p, 64, hipMemcpyDeviceToHost, x.st);
    if (le == hipSuccess) le = hipStreamSynchronize(x.st);
    if (le != hipSuccess) return ctx->fail(BIOALIGN_EHIP, "long-mode kernels failed: %s", hipGetErrorString(le));
    if (misc[0] != 0)
        return ctx->fail(BIOALIGN_EHIP, "long mode: a strip gave up waiting for the strip above it (%u waits ran out)", misc[0]);
    *true_score = (int32_t)misc[8 + 3];
    *j_local = misc[8 + 1];
    *score = (int64_t)((uint64_t)misc[2] | ((uint64_t)misc[3] << 32));
    return 0;
}

// walks from cell (i, j_local) to the stripe's border and expands the moves: the piece's bytes go to t1 / t2
// (host, room for n + width each), its length to *len, the cell the walk stopped in to (*exit_i, *exit_j)
int stripe_walk(bioalign_stripe &x, uint64_t i, uint64_t j_local, uint8_t *t1, uint8_t *t2, uint64_t *len_out,
                uint64_t *exit_i, uint64_t *exit_j)
{
    bioalign_ctx *ctx = x.ctx;
    if (!x.dp_done || !x.traces) return ctx->fail(BIOALIGN_EINVAL, "stripe: walk without DP or without trace buffers");
    if (i > x.n || j_local > x.m) return ctx->fail(BIOALIGN_EINVAL, "stripe: start cell outside the stripe");
    HIP_TRY(ctx, hipSetDevice(ctx->devices[x.dev]));
    BaLongParams Pq = x.P;
    Pq.start_i = (int)i;
    Pq.start_j = (int)j_local;
    uint32_t misc[16];
    hipError_t le = ba_launch_long(Pq, x.mode, x.T, x.K, BA_LONG_PHASE_WALK, x.st);
    if (le == hipSuccess) le = hipMemcpyAsync(misc, x.misc.p, 64, hipMemcpyDeviceToHost, x.st);
    if (le == hipSuccess) le = hipStreamSynchronize(x.st);
    if (le != hipSuccess) return ctx->fail(BIOALIGN_EHIP, "long-mode kernels failed: %s", hipGetErrorString(le));
    const uint64_t off = (uint64_t)misc[4] | ((uint64_t)misc[5] << 32), len = (uint64_t)misc[6] | ((uint64_t)misc[7] << 32);
    if (off + len != x.n + x.m) return ctx->fail(BIOALIGN_EHIP, "long mode: inconsistent trace extent in a stripe");
    if (len) {
        HIP_TRY(ctx, hipMemcpy(t1, x.t1.as<uint8_t>() + off, len, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(t2, x.t2.as<uint8_t>() + off, len, hipMemcpyDeviceToHost));
    }
    *len_out = len;
    *exit_i = misc[8 + 4];
    *exit_j = misc[8 + 5];
    return 0;
}

// after a stripe's walk stopped in (i, j): does the walk go on in the stripe to the left?
bool walk_goes_on(int mode, uint64_t i) { return !(i == 0 && mode == BIOALIGN_FITTING); }  // free top row: it ends there

int long_pair_striped(bioalign_ctx *ctx, const bioalign_batch_t *in, bioalign_result_t *out, uint64_t p, int G,
                      float *kernel_ms)
{
    const uint64_t a1 = in->s1_off[p], a2 = in->s2_off[p];
    const uint64_t n = in->s1_off[p + 1] - a1, m = in->s2_off[p + 1] - a2;
    const bool traces = !(in->flags & BIOALIGN_SCORE_ONLY);
    G = (int)std::min<uint64_t>((uint64_t)G, m);
    const int D = (int)ctx->devices.size();
    std::vector<bioalign_stripe> S(G);
    int rc;
    for (int q = 0; q < G; ++q)
        if ((rc = stripe_init(S[q], ctx, q % D, in, p, m * (uint64_t)q / G, m * (uint64_t)(q + 1) / G))) return rc;

    const auto t_begin = std::chrono::steady_clock::now();
    // All stripes on one device: ONE launch holds the workgroups of every stripe, and a strip starts as soon as the strip
    // at its height in the stripe to its left has finished (tagged granules in HBM, as between workgroups): the DP is
    // the same wavefront as the unstriped one.  BIOALIGN_STRIPE_MERGED=0 keeps the host-driven row blocks below.
    bool merged = D == 1 && G > 1 && env_u64("BIOALIGN_STRIPE_MERGED", 1) != 0;
    uint64_t total_wgs = 0;
    for (int q = 0; q < G; ++q) total_wgs += S[q].nsuper;
    if (total_wgs > 0x7fffffffull) merged = false;
    DevBuf d_multi, d_order;
    if (merged) {
        HIP_TRY(ctx, hipSetDevice(ctx->devices[S[0].dev]));
        for (int q = 0; q + 1 < G; ++q) {
            if ((rc = S[q].rgran.alloc(ctx, (n + 1) * 8))) return rc;
            HIP_TRY(ctx, hipMemsetAsync(S[q].rgran.p, 0, (n + 1) * 8, S[0].st));  // tag 0: not handed over yet
        }
        std::vector<BaLongParams> multi(G);
        for (int q = 0; q < G; ++q) {
            BaLongParams Pq = S[q].P;
            Pq.right_col = nullptr;
            Pq.left_gran = q > 0 ? S[q - 1].rgran.as<unsigned long long>() : nullptr;
            Pq.right_gran = q + 1 < G ? S[q].rgran.as<unsigned long long>() : nullptr;
            multi[q] = Pq;
        }
        // workgroups in the order they can start: (columns left of the stripe) + (strips above) * (steps a strip lags)
        struct Wg {
            uint64_t key;
            uint32_t q, s;
        };
        std::vector<Wg> wg;
        wg.reserve(total_wgs);
        const uint64_t lag = 80ull * (uint64_t)ba_long_strips_per_group();
        for (int q = 0; q < G; ++q)
            for (uint64_t s = 0; s < S[q].nsuper; ++s) wg.push_back({S[q].c0 + s * lag, (uint32_t)q, (uint32_t)s});
        std::sort(wg.begin(), wg.end(), [](const Wg &a, const Wg &b) {
            return a.key != b.key ? a.key < b.key : a.q != b.q ? a.q < b.q : a.s < b.s;
        });
        std::vector<uint2> order(wg.size());
        for (size_t k = 0; k < wg.size(); ++k) order[k] = make_uint2(wg[k].q, wg[k].s);
        if ((rc = d_multi.alloc(ctx, multi.size() * sizeof(BaLongParams))) || (rc = d_order.alloc(ctx, order.size() * sizeof(uint2)))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(d_multi.p, multi.data(), multi.size() * sizeof(BaLongParams), hipMemcpyHostToDevice, S[0].st));
        HIP_TRY(ctx, hipMemcpyAsync(d_order.p, order.data(), order.size() * sizeof(uint2), hipMemcpyHostToDevice, S[0].st));
        HIP_TRY(ctx, hipStreamSynchronize(S[0].st));  // (the sources are locals)
        BaLongParams P0 = S[0].P;
        P0.multi = d_multi.as<BaLongParams>();
        P0.order = d_order.as<uint2>();
        P0.wg_begin = 0;
        P0.wg_count = (uint32_t)total_wgs;
        hipError_t le = ba_launch_long(P0, S[0].mode, S[0].T, S[0].K, BA_LONG_PHASE_DP, S[0].st);
        if (le != hipSuccess) return ctx->fail(BIOALIGN_EHIP, "long-mode DP kernel failed: %s", hipGetErrorString(le));
        HIP_TRY(ctx, hipEventRecord(S[0].done, S[0].st));
        for (int q = 0; q < G; ++q) {
            S[q].blocks_done = (uint32_t)S[q].blk.size() - 1;
            S[q].dp_done = true;
            if (q > 0) {
                HIP_TRY(ctx, hipStreamWaitEvent(S[q].st, S[0].done, 0));
                HIP_TRY(ctx, hipEventRecord(S[q].done, S[q].st));
            }
        }
    }
    // DP as a pipeline over row blocks: stripe q runs block r once stripe q-1 has finished block r and
    // those rows of its last column have come over (a device-to-device copy, xGMI between GPUs; stream-ordered behind
    // an event, no host round trip), while stripe q-1 goes on with block r+1.  Every stripe has its own stream.
    const uint32_t R = (uint32_t)S[0].blk.size() - 1;
    std::vector<hipEvent_t> ev((size_t)G * R, nullptr);
    struct EvGuard {
        std::vector<hipEvent_t> &v;
        ~EvGuard()
        {
            for (hipEvent_t e : v)
                if (e) (void)hipEventDestroy(e);
        }
    } ev_guard{ev};
    // Queued by anti-diagonals of the (stripe, block) grid: the blocks of one diagonal depend on the diagonal before it
    // alone, so a hardware queue that several of these streams share never holds a block in front of one it waits for.
    for (uint32_t d = 0; !merged && d + 1 < R + (uint32_t)G; ++d)
        for (int q = 0; q < G; ++q) {
            if ((uint32_t)q > d || d - (uint32_t)q >= R) continue;
            const uint32_t r = d - (uint32_t)q;
            bioalign_stripe &x = S[q];
            HIP_TRY(ctx, hipSetDevice(ctx->devices[x.dev]));
            if (q > 0) {
                bioalign_stripe &l = S[q - 1];
                const uint64_t lo = x.blk_row[r], cnt = x.blk_row[r + 1] - lo;
                HIP_TRY(ctx, hipStreamWaitEvent(x.st, ev[(size_t)(q - 1) * R + r], 0));
                int32_t *dst = x.left.as<int32_t>() + lo;
                const int32_t *src = l.right.as<int32_t>() + lo;
                if (ctx->devices[l.dev] == ctx->devices[x.dev])
                    HIP_TRY(ctx, hipMemcpyAsync(dst, src, cnt * 4, hipMemcpyDeviceToDevice, x.st));
                else
                    HIP_TRY(ctx, hipMemcpyPeerAsync(dst, ctx->devices[x.dev], src, ctx->devices[l.dev], cnt * 4, x.st));
            }
            if ((rc = stripe_dp_block(x, r))) return rc;
            HIP_TRY(ctx, hipEventCreateWithFlags(&ev[(size_t)q * R + r], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ev[(size_t)q * R + r], x.st));
        }
    // end cell: global = the last cell of the last stripe; fitting / overlap = leftmost strict maximum of the
    // bottom row over all stripes (align_pair.go:169-182): every stripe reports its own
    int end_stripe = G - 1;
    int64_t score = 0, best = INT64_MIN;
    uint64_t j = 0;
    for (int q = (in->mode == BIOALIGN_GLOBAL ? G - 1 : 0); q < G; ++q) {
        int64_t v, sc;
        uint64_t jl;
        if ((rc = stripe_end_cell(S[q], &v, &jl, &sc))) return rc;
        if (v > best) {
            best = v;
            end_stripe = q;
            j = jl;
            score = sc;
        }
    }
    out->scores[p] = score;
    if (traces) {
        const uint64_t slot = a1 + a2;
        uint64_t pos_end = slot + n + m, total = 0, i = n;
        std::vector<uint8_t> piece1, piece2;
        for (int q = end_stripe; q >= 0; --q) {
            bioalign_stripe &x = S[q];
            piece1.resize(n + x.m + 1);
            piece2.resize(n + x.m + 1);
            uint64_t len = 0, ei = 0, ej = 0;
            if ((rc = stripe_walk(x, i, j, piece1.data(), piece2.data(), &len, &ei, &ej))) return rc;
            if (len > pos_end - slot) return ctx->fail(BIOALIGN_EHIP, "long mode: the pieces of the trace exceed its slot");
            memcpy(out->t1 + pos_end - len, piece1.data(), len);
            memcpy(out->t2 + pos_end - len, piece2.data(), len);
            pos_end -= len;
            total += len;
            if (q == 0 || !walk_goes_on(in->mode, ei)) break;
            if (ej != 0) return ctx->fail(BIOALIGN_EHIP, "long mode: the walk left stripe %d inside it", q);
            i = ei;
            j = S[q - 1].m;  // the same cell, as the left neighbour numbers it
        }
        out->t_off[p] = pos_end;
        out->t_len[p] = total;
    }
    for (int q = 0; q < G; ++q) {
        HIP_TRY(ctx, hipSetDevice(ctx->devices[S[q].dev]));
        HIP_TRY(ctx, hipStreamSynchronize(S[q].st));
    }
    if (kernel_ms)
        *kernel_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

}  // namespace

extern "C" int bioalign_long(bioalign_ctx *ctx, const bioalign_batch_t *in, bioalign_result_t *out, float *kernel_ms)
{
    if (!ctx) return BIOALIGN_EINVAL;
    if (!in || !out) return ctx->fail(BIOALIGN_EINVAL, "null argument");
    if (!in->s1_off || !in->s2_off || !in->score_table || !in->code1 || !in->code2)
        return ctx->fail(BIOALIGN_EINVAL, "null offsets / score table / code maps");
    if (in->mode < BIOALIGN_GLOBAL || in->mode > BIOALIGN_OVERLAP) return ctx->fail(BIOALIGN_EINVAL, "unknown mode %d", in->mode);
    if (in->na <= 0 || in->nb <= 0 || in->na > 255 || in->nb > 255)
        return ctx->fail(BIOALIGN_EINVAL, "alphabet sizes must be in 1..255 (got %d, %d)", in->na, in->nb);
    const uint64_t N = in->n_pairs;
    if (N == 0) return 0;
    if (!out->scores) return ctx->fail(BIOALIGN_EINVAL, "null scores buffer");
    const bool traces = !(in->flags & BIOALIGN_SCORE_ONLY);
    if (traces && (!out->t1 || !out->t2 || !out->t_off || !out->t_len))
        return ctx->fail(BIOALIGN_EINVAL, "trace buffers are required unless BIOALIGN_SCORE_ONLY is set");
    if (kernel_ms) *kernel_ms = 0.f;
    if (in->gap_open < 0) return ctx->fail(BIOALIGN_EUNSUPPORTED, "negative gap penalty %d", in->gap_open);
    for (uint64_t p = 0; p < N; ++p) {
        const uint64_t n = in->s1_off[p + 1] - in->s1_off[p], m = in->s2_off[p + 1] - in->s2_off[p];
        int rc;
        if (n == 0 || m == 0) {
            // nothing to parallelise inside the pair: the batch kernels handle the borders
            std::string err;
            rc = run_share(ctx, 0, in, out, p, p + 1, &err);
            if (rc) {
                ctx->err = err;
                ctx->code = rc;
            }
        } else {
            // column stripes over the context's devices when the residue words (K bits per cell, K <= 8) may not
            // fit one device, or when BIOALIGN_LONG_STRIPES asks for them
            // (local mode runs unstriped: its end cell is a maximum over the whole matrix)
            // (affine gaps too: the stripes hand over one value per row, the affine kernels need two)
            int stripes = (in->mode == BIOALIGN_LOCAL || in->gap_open != in->gap_extend) ? 1 : (int)env_u64("BIOALIGN_LONG_STRIPES", 0);
            if (stripes == 0 && ctx->devices.size() > 1) {
                size_t free_b = 0, total_b = 0;
                HIP_TRY(ctx, hipSetDevice(ctx->devices[0]));
                HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
                if ((double)n * (double)m > 0.8 * (double)free_b) stripes = (int)ctx->devices.size();
            }
            rc = stripes > 1 ? long_pair_striped(ctx, in, out, p, stripes, kernel_ms) : long_pair(ctx, in, out, p, kernel_ms);
        }
        if (rc) return rc;
    }
    return 0;
}

// ----------------------------------------------------------------------------
// C ABI of one column stripe (include/bioalign.h): what a process that owns ONE GPU of a group needs to take
// part in a long-mode alignment whose matrix is spread over the group (bio_amd/long_stripes.py drives it over
// torch.distributed, boundary columns as RCCL send / recv).
// ----------------------------------------------------------------------------
extern "C" bioalign_stripe *bioalign_stripe_create(bioalign_ctx *ctx, int device_index, const bioalign_batch_t *in,
                                                   uint64_t col_begin, uint64_t col_end)
{
    if (!ctx) return nullptr;
    if (!in || in->n_pairs != 1 || !in->s1 || !in->s2 || !in->s1_off || !in->s2_off || !in->score_table || !in->code1 ||
        !in->code2) {
        ctx->fail(BIOALIGN_EINVAL, "stripe: a batch of exactly one pair with sequences, offsets, score table and code maps");
        return nullptr;
    }
    if (device_index < 0 || device_index >= (int)ctx->devices.size()) {
        ctx->fail(BIOALIGN_EINVAL, "device index %d out of range", device_index);
        return nullptr;
    }
    if (in->mode != BIOALIGN_GLOBAL && in->mode != BIOALIGN_FITTING && in->mode != BIOALIGN_OVERLAP) {
        ctx->fail(BIOALIGN_EUNSUPPORTED, "column stripes cover global, fitting and overlap (local mode runs unstriped)");
        return nullptr;
    }
    if (in->gap_open != in->gap_extend || in->gap_open < 0) {
        ctx->fail(BIOALIGN_EUNSUPPORTED, "long mode covers linear gaps with a non-negative penalty");
        return nullptr;
    }
    if (in->na <= 0 || in->nb <= 0 || in->na > 255 || in->nb > 255) {
        ctx->fail(BIOALIGN_EINVAL, "alphabet sizes must be in 1..255 (got %d, %d)", in->na, in->nb);
        return nullptr;
    }
    bioalign_stripe *x = new bioalign_stripe;
    if (stripe_init(*x, ctx, device_index, in, 0, col_begin, col_end) != 0) {
        delete x;
        return nullptr;
    }
    x->ctx_ref = true;
    ctx_retain(ctx);
    return x;
}

extern "C" void bioalign_stripe_destroy(bioalign_stripe *x)
{
    if (!x) return;
    bioalign_ctx *held = x->ctx_ref ? x->ctx : nullptr;
    delete x;
    ctx_release(held);
}

extern "C" uint64_t bioalign_stripe_rows(const bioalign_stripe *x) { return x ? x->n : 0; }

extern "C" int bioalign_stripe_dp(bioalign_stripe *x, const int32_t *d_left_col, int32_t *d_right_col)
{
    if (!x) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = x->ctx;
    if ((x->c0 > 0) != (d_left_col != nullptr))
        return ctx->fail(BIOALIGN_EINVAL, "stripe: the left column is required for every stripe but the first, and only there");
    HIP_TRY(ctx, hipSetDevice(ctx->devices[x->dev]));
    if (d_left_col)
        HIP_TRY(ctx, hipMemcpyAsync(x->left.p, d_left_col, (x->n + 1) * 4, hipMemcpyDeviceToDevice, x->st));
    int rc = stripe_dp(*x);
    if (rc) return rc;
    if (d_right_col)
        HIP_TRY(ctx, hipMemcpyAsync(d_right_col, x->right.p, (x->n + 1) * 4, hipMemcpyDeviceToDevice, x->st));
    HIP_TRY(ctx, hipStreamSynchronize(x->st));
    return 0;
}

extern "C" uint32_t bioalign_stripe_blocks(const bioalign_stripe *x) { return x ? (uint32_t)x->blk.size() - 1 : 0; }

extern "C" int bioalign_stripe_block_rows(const bioalign_stripe *x, uint32_t block, uint64_t *row_begin, uint64_t *row_end)
{
    if (!x || !row_begin || !row_end || block + 1 >= x->blk.size()) return BIOALIGN_EINVAL;
    *row_begin = x->blk_row[block];
    *row_end = x->blk_row[block + 1];
    return 0;
}

extern "C" int bioalign_stripe_dp_block(bioalign_stripe *x, uint32_t block, const int32_t *d_left_rows, int32_t *d_right_rows)
{
    if (!x) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = x->ctx;
    if (block + 1 >= x->blk.size()) return ctx->fail(BIOALIGN_EINVAL, "stripe: no such row block");
    if ((x->c0 > 0) != (d_left_rows != nullptr))
        return ctx->fail(BIOALIGN_EINVAL, "stripe: the left column is required for every stripe but the first, and only there");
    const uint64_t lo = x->blk_row[block], cnt = x->blk_row[block + 1] - lo;
    HIP_TRY(ctx, hipSetDevice(ctx->devices[x->dev]));
    if (d_left_rows)
        HIP_TRY(ctx, hipMemcpyAsync(x->left.as<int32_t>() + lo, d_left_rows, cnt * 4, hipMemcpyDeviceToDevice, x->st));
    int rc = stripe_dp_block(*x, block);
    if (rc) return rc;
    if (d_right_rows)
        HIP_TRY(ctx, hipMemcpyAsync(d_right_rows, x->right.as<int32_t>() + lo, cnt * 4, hipMemcpyDeviceToDevice, x->st));
    HIP_TRY(ctx, hipStreamSynchronize(x->st));
    return 0;
}

extern "C" int bioalign_stripe_end_cell(bioalign_stripe *x, int64_t *true_score, uint64_t *j_local, int64_t *score)
{
    if (!x) return BIOALIGN_EINVAL;
    if (!true_score || !j_local || !score) return x->ctx->fail(BIOALIGN_EINVAL, "null argument");
    return stripe_end_cell(*x, true_score, j_local, score);
}

extern "C" int bioalign_stripe_walk(bioalign_stripe *x, uint64_t i, uint64_t j_local, uint8_t *t1, uint8_t *t2,
                                    uint64_t *len, uint64_t *exit_i, uint64_t *exit_j)
{
    if (!x) return BIOALIGN_EINVAL;
    if (!t1 || !t2 || !len || !exit_i || !exit_j) return x->ctx->fail(BIOALIGN_EINVAL, "null argument");
    return stripe_walk(*x, i, j_local, t1, t2, len, exit_i, exit_j);
}

// ----------------------------------------------------------------------------
// FASTA text -> the batch format in HBM (ReadFASTA, fasta.go:55-84; kernels_fasta.hip)
// ----------------------------------------------------------------------------
extern "C" hipError_t ba_launch_fasta_count(const BaFastaParams &P, hipStream_t st);
extern "C" hipError_t ba_launch_fasta_scan1(uint64_t *a, uint64_t *b, uint64_t n, uint64_t *tot, hipStream_t st);
extern "C" hipError_t ba_launch_fasta_starts(const BaFastaParams &P, hipStream_t st);
extern "C" hipError_t ba_launch_fasta_lines_a(const BaFastaParams &P, hipStream_t st);
extern "C" hipError_t ba_launch_fasta_lines_b(const BaFastaParams &P, hipStream_t st);

// device memory from the context's pool (DevPool), given back when the object goes
struct PoolBuf {
    std::shared_ptr<DevPool> pool;
    void *p = nullptr;
    ~PoolBuf()
    {
        if (pool) pool->put(p);
    }
    int alloc(bioalign_ctx *ctx, int hip_dev, size_t bytes)
    {
        pool = ctx->pool;
        p = pool->get(hip_dev, bytes ? bytes : 16);
        if (!p) return ctx->fail(BIOALIGN_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
        return 0;
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

struct bioalign_fasta {
    bioalign_ctx *ctx = nullptr;
    int dev = 0;
    uint64_t n_records = 0, total = 0, n_lines = 0;
    PoolBuf seq, off;  // device: the records' data back to back, [n_records + 1] offsets
    std::vector<uint64_t> h_off, h_hdr_begin, h_hdr_len;
    float kernel_ms = 0.f;
    int64_t too_long_at = -1;  // reader rules: the ReadSeq call (0-based) that meets a line too long for the buffer
};

namespace {

#define FA_TRY(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->fail(BIOALIGN_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BIOALIGN_EHIP;                                                                      \
        }                                                                                              \
    } while (0)

// line_limit == 0: ReadFASTA's rules; else FASTAReader.ReadSeq's with a bufio buffer of that many bytes
int fasta_parse(bioalign_fasta &F, const uint8_t *text, uint64_t len, bool on_device, uint64_t line_limit)
{
    bioalign_ctx *ctx = F.ctx;
    const uint64_t CHUNK = 16384, LPB = 2048;
    FA_TRY(hipSetDevice(ctx->devices[F.dev]));
    const uint64_t n_chunks = std::max<uint64_t>((len + CHUNK - 1) / CHUNK, 1);
    if (n_chunks > 0x7fffffffull) return ctx->fail(BIOALIGN_EUNSUPPORTED, "FASTA text of %llu bytes is too long for one call", (unsigned long long)len);
    const int hip_dev = ctx->devices[F.dev];
    PoolBuf d_text, d_cnt;  // (the context's pool: a pipeline parses file after file)
    int rc;
    if ((rc = d_cnt.alloc(ctx, hip_dev, n_chunks * 8 + 32))) return rc;
    uint64_t *const d_tot = d_cnt.as<uint64_t>() + n_chunks;
    unsigned long long *const d_long = reinterpret_cast<unsigned long long *>(d_tot + 2);
    hipStream_t st = nullptr;
    const uint8_t *dtext = text;  // text already in HBM: parsed where it lies
    if (!on_device) {
        if ((rc = d_text.alloc(ctx, hip_dev, len + 16))) return rc;
        if (len) FA_TRY(hipMemcpyAsync(d_text.p, text, len, hipMemcpyHostToDevice, st));
        dtext = d_text.as<uint8_t>();
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FA_TRY(hipEventCreate(&e0));
    FA_TRY(hipEventCreate(&e1));
    struct Ev {
        hipEvent_t &a, &b;
        ~Ev()
        {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
        }
    } ev{e0, e1};
    FA_TRY(hipEventRecord(e0, st));

    BaFastaParams P{};
    P.text = dtext;
    P.len = len;
    P.n_chunks = n_chunks;
    P.chunk_count = d_cnt.as<uint64_t>();
    P.reader = line_limit != 0;
    P.line_limit = line_limit;
    P.too_long = d_long;
    FA_TRY(hipMemsetAsync(d_long, 0xff, 8, st));
    FA_TRY(ba_launch_fasta_count(P, st));
    FA_TRY(ba_launch_fasta_scan1(P.chunk_count, nullptr, n_chunks, d_tot, st));
    uint64_t tot[2] = {0, 0};
    FA_TRY(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, st));
    FA_TRY(hipStreamSynchronize(st));
    const uint64_t n_lines = tot[0] + 1;  // bytes.Split: one piece more than separators (fasta.go:62)
    const uint64_t n_lb = (n_lines + LPB - 1) / LPB;
    F.n_lines = n_lines;

    // the per-line arrays in one allocation
    auto up = [](uint64_t v) { return (v + 255) & ~255ull; };
    const uint64_t o_ls = 0, o_tb = o_ls + up(n_lines * 8), o_ld = o_tb + up(n_lines * 8), o_lh = o_ld + up(n_lines * 8),
                   o_pd = o_lh + up(n_lines * 8), o_ph = o_pd + up(n_lb * 8), o_tl = o_ph + up(n_lb * 8),
                   ws_bytes = o_tl + up(n_lines * 4);
    PoolBuf d_ws;
    if ((rc = d_ws.alloc(ctx, hip_dev, ws_bytes))) return rc;
    uint8_t *const ws = d_ws.as<uint8_t>();
    P.line_start = reinterpret_cast<uint64_t *>(ws + o_ls);
    P.n_lines = n_lines;
    P.n_line_blocks = n_lb;
    P.trim_begin = reinterpret_cast<uint64_t *>(ws + o_tb);
    P.trim_len = reinterpret_cast<uint32_t *>(ws + o_tl);
    P.ldata = reinterpret_cast<uint64_t *>(ws + o_ld);
    P.lhdr = reinterpret_cast<uint64_t *>(ws + o_lh);
    P.part_data = reinterpret_cast<uint64_t *>(ws + o_pd);
    P.part_hdr = reinterpret_cast<uint64_t *>(ws + o_ph);
    FA_TRY(hipMemsetAsync(P.line_start, 0, 8, st));  // line 0 starts at 0
    FA_TRY(ba_launch_fasta_starts(P, st));
    FA_TRY(ba_launch_fasta_lines_a(P, st));
    FA_TRY(ba_launch_fasta_scan1(P.part_data, P.part_hdr, n_lb, d_tot, st));
    FA_TRY(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, st));
    FA_TRY(hipStreamSynchronize(st));
    F.total = tot[0];
    F.n_records = tot[1];

    PoolBuf d_tmp;  // header spans + the error word
    if ((rc = F.seq.alloc(ctx, hip_dev, F.total + 16)) || (rc = F.off.alloc(ctx, hip_dev, (F.n_records + 1) * 8)) ||
        (rc = d_tmp.alloc(ctx, hip_dev, F.n_records * 16 + 16)))
        return rc;
    uint64_t *const d_hb = d_tmp.as<uint64_t>(), *const d_hl = d_hb + F.n_records;
    uint32_t *const d_err = reinterpret_cast<uint32_t *>(d_hl + F.n_records);
    P.seq = F.seq.as<uint8_t>();
    P.off = F.off.as<uint64_t>();
    P.hdr_begin = d_hb;
    P.hdr_len = d_hl;
    P.error = d_err;
    FA_TRY(hipMemsetAsync(d_err, 0, 4, st));
    FA_TRY(ba_launch_fasta_lines_b(P, st));
    FA_TRY(hipMemcpyAsync(F.off.as<uint64_t>() + F.n_records, &F.total, 8, hipMemcpyHostToDevice, st));
    FA_TRY(hipEventRecord(e1, st));
    uint32_t err = 0;
    F.h_off.assign(F.n_records + 1, 0);
    F.h_hdr_begin.assign(F.n_records, 0);
    F.h_hdr_len.assign(F.n_records, 0);
    FA_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
    FA_TRY(hipMemcpyAsync(F.h_off.data(), F.off.p, (F.n_records + 1) * 8, hipMemcpyDeviceToHost, st));
    if (F.n_records) {
        FA_TRY(hipMemcpyAsync(F.h_hdr_begin.data(), d_hb, F.n_records * 8, hipMemcpyDeviceToHost, st));
        FA_TRY(hipMemcpyAsync(F.h_hdr_len.data(), d_hl, F.n_records * 8, hipMemcpyDeviceToHost, st));
    }
    unsigned long long too_long = ~0ull;
    if (P.reader) FA_TRY(hipMemcpyAsync(&too_long, d_long, 8, hipMemcpyDeviceToHost, st));
    FA_TRY(hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&F.kernel_ms, e0, e1);
    if (err) return ctx->fail(BIOALIGN_EINVAL, "No header");  // the reference's error text (fasta.go:73)
    if (too_long != ~0ull) {
        // the call that is reading record h - 1 meets the line (h headers before it; also when the line is itself
        // a header: it is read by the call that looks for the end of the record before); no header yet: the first
        uint64_t h = 0;
        FA_TRY(hipMemcpy(&h, P.lhdr + too_long, 8, hipMemcpyDeviceToHost));
        F.too_long_at = h ? (int64_t)h - 1 : 0;
    }
    return 0;
}

}  // namespace

extern "C" bioalign_fasta *bioalign_fasta_parse(bioalign_ctx *ctx, int device_index, const uint8_t *text, uint64_t len,
                                                int text_on_device)
{
    if (!ctx) return nullptr;
    if (device_index < 0 || device_index >= (int)ctx->devices.size()) {
        ctx->fail(BIOALIGN_EINVAL, "device index %d out of range", device_index);
        return nullptr;
    }
    if (!text && len) {
        ctx->fail(BIOALIGN_EINVAL, "null text");
        return nullptr;
    }
    bioalign_fasta *F = new bioalign_fasta;
    F->ctx = ctx;
    F->dev = device_index;
    if (fasta_parse(*F, text, len, text_on_device != 0, 0) != 0) {
        delete F;
        return nullptr;
    }
    ctx_retain(ctx);
    return F;
}

extern "C" bioalign_fasta *bioalign_fasta_parse_reader(bioalign_ctx *ctx, int device_index, const uint8_t *text,
                                                       uint64_t len, int text_on_device, uint64_t line_limit)
{
    if (!ctx) return nullptr;
    if (device_index < 0 || device_index >= (int)ctx->devices.size()) {
        ctx->fail(BIOALIGN_EINVAL, "device index %d out of range", device_index);
        return nullptr;
    }
    if (!text && len) {
        ctx->fail(BIOALIGN_EINVAL, "null text");
        return nullptr;
    }
    bioalign_fasta *F = new bioalign_fasta;
    F->ctx = ctx;
    F->dev = device_index;
    if (fasta_parse(*F, text, len, text_on_device != 0, line_limit ? line_limit : 4096) != 0) {
        delete F;
        return nullptr;
    }
    ctx_retain(ctx);
    return F;
}
extern "C" int64_t bioalign_fasta_too_long_at(const bioalign_fasta *F) { return F ? F->too_long_at : -1; }

extern "C" void bioalign_fasta_destroy(bioalign_fasta *F)
{
    if (!F) return;
    (void)hipSetDevice(F->ctx->devices[F->dev]);
    bioalign_ctx *held = F->ctx;
    delete F;
    ctx_release(held);
}
extern "C" uint64_t bioalign_fasta_count(const bioalign_fasta *F) { return F ? F->n_records : 0; }
extern "C" uint64_t bioalign_fasta_bytes(const bioalign_fasta *F) { return F ? F->total : 0; }
extern "C" float bioalign_fasta_kernel_ms(const bioalign_fasta *F) { return F ? F->kernel_ms : 0.f; }
extern "C" const uint8_t *bioalign_fasta_device_seq(const bioalign_fasta *F) { return F ? F->seq.as<uint8_t>() : nullptr; }
extern "C" const uint64_t *bioalign_fasta_device_off(const bioalign_fasta *F) { return F ? F->off.as<uint64_t>() : nullptr; }
extern "C" const uint64_t *bioalign_fasta_off(const bioalign_fasta *F) { return F ? F->h_off.data() : nullptr; }
extern "C" const uint64_t *bioalign_fasta_header_begin(const bioalign_fasta *F) { return F ? F->h_hdr_begin.data() : nullptr; }
extern "C" const uint64_t *bioalign_fasta_header_len(const bioalign_fasta *F) { return F ? F->h_hdr_len.data() : nullptr; }
extern "C" int bioalign_fasta_copy_seq(const bioalign_fasta *F, uint8_t *dst)
{
    if (!F) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = F->ctx;
    if (!dst && F->total) return ctx->fail(BIOALIGN_EINVAL, "null destination");
    HIP_TRY(ctx, hipSetDevice(ctx->devices[F->dev]));
    if (F->total) HIP_TRY(ctx, hipMemcpy(dst, F->seq.p, F->total, hipMemcpyDeviceToHost));
    return 0;
}

// ----------------------------------------------------------------------------
// Compact traces (BIOALIGN_TRACE_OPS): capacity, and the host-side expansion of one record into the reference's
// two gapped strings (align.go:126-143; align_pair.go:197-216).  Format conversion of O(trace) bytes, no DP.
// ----------------------------------------------------------------------------
extern "C" uint64_t bioalign_ops_capacity(uint64_t n_pairs, uint64_t total_s1, uint64_t total_s2)
{
    return 16 * n_pairs + 4 * ((total_s1 + total_s2) / 16) + 16;
}

extern "C" int bioalign_expand_ops(const uint8_t *s1, uint64_t n, const uint8_t *s2, uint64_t m, const uint8_t *record,
                                   uint64_t moves, uint8_t gap_symbol, uint8_t *t1_out, uint8_t *t2_out)
{
    if (!record || (moves && (!t1_out || !t2_out))) return BIOALIGN_EINVAL;
    uint32_t ij[2];
    memcpy(ij, record, 8);
    uint64_t i = ij[0], j = ij[1];
    if (i > n || j > m) return BIOALIGN_EINVAL;
    for (uint64_t k = 0; k < moves; ++k) {
        uint32_t w;
        memcpy(&w, record + 8 + 4 * (k / 16), 4);
        const uint32_t mv = (w >> (2 * (k % 16))) & 3u;
        const uint64_t pos = moves - 1 - k;  // the first move is the last column
        if (mv == 0) {
            if (i == 0) return BIOALIGN_EINVAL;
            t1_out[pos] = s1[--i];
            t2_out[pos] = gap_symbol;
        } else if (mv == 1) {
            if (j == 0) return BIOALIGN_EINVAL;
            t1_out[pos] = gap_symbol;
            t2_out[pos] = s2[--j];
        } else if (mv == 2) {
            if (i == 0 || j == 0) return BIOALIGN_EINVAL;
            t1_out[pos] = s1[--i];
            t2_out[pos] = s2[--j];
        } else {
            return BIOALIGN_EINVAL;
        }
    }
    return 0;
}

// ----------------------------------------------------------------------------
// All pairs of one list: the batched client under DNA8List.DistanceMatrix
// (dna_list.go:951-969) and StrList.DistanceMatrix (str.go:450-469) when the distance function is an alignment
// score.  The reference calls f(l[i], l[j]) for every j < i; here those n(n-1)/2 calls are one score-only batch.
// The list lies in HBM once: a pair's two sequences are addressed by their offsets into it (the kernels read a
// pair's bytes from s + s_off[p] and take its lengths from the plan), so nothing is replicated per pair.
// ----------------------------------------------------------------------------
namespace {

// pair p of the lower triangle, row by row: p = i(i-1)/2 + j, j < i
void pair_of(uint64_t p, uint64_t *i, uint64_t *j)
{
    uint64_t r = (uint64_t)((1.0 + std::sqrt(1.0 + 8.0 * (double)p)) / 2.0);
    while (r > 1 && r * (r - 1) / 2 > p) --r;
    while ((r + 1) * r / 2 <= p) ++r;
    *i = r;
    *j = p - r * (r - 1) / 2;
}

extern "C" hipError_t ba_launch_pair_offsets(const uint64_t *off, uint64_t c0, uint64_t n, uint64_t n_seqs, int ordered,
                                             uint64_t *q1, uint64_t *q2, hipStream_t st);

// what bioalign_all_pairs / bioalign_all_ordered_pairs enumerate, and where the results go
struct PairSpace {
    bool ordered = false;          // every (i, j) with i != j instead of j < i
    int64_t *scores = nullptr;     // [0] is pair `base` of the numbering
    uint64_t *end_i = nullptr, *end_j = nullptr;  // end cells (BIOALIGN_END_CELLS), or null
    uint64_t base = 0;
};

// Pairs [lo, hi) on one device, in chunks of at most BIOALIGN_ALL_PAIRS_CHUNK pairs (one plan each).
// d_list / d_off: the list's bytes and offsets if they already lie on this device, else null (h_list and off are
// copied in).  The plan sees lengths only (virtual offsets); where each pair's sequences lie in the list is
// computed on the device (ba_pair_offsets_kernel).
int all_pairs_share(bioalign_ctx *ctx, int dev_index, const bioalign_batch_t *in, const uint64_t *off,
                    const uint8_t *h_list, const uint8_t *d_list, const uint64_t *d_off, const PairSpace &S, uint64_t lo,
                    uint64_t hi, std::string *err)
{
    bioalign_ctx local;  // (as run_share: concurrent shares do not race on the error string)
    local.devices = ctx->devices;
    local.pool = ctx->pool;
    if (lo >= hi) return 0;
    const uint64_t n_seqs = in->n_pairs, total = off[n_seqs];
    const int hip_dev = ctx->devices[dev_index];
    const uint64_t CH = std::max<uint64_t>(env_u64("BIOALIGN_ALL_PAIRS_CHUNK", 4ull << 20), 1);
    hipStream_t st = nullptr;
    bioalign_plan *pl = nullptr;
    int rc = 0;
    PoolBuf list, loff, q1, q2, sc, e1, e2;
    std::vector<uint32_t> v1, v2;
    const bool ends = S.end_i && S.end_j;
    const bool timing = env_u64("BIOALIGN_TIMING", 0) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count();
    };
    const auto t_start = now();
    auto t_mark = t_start;
    double t_fill = 0, t_plan = 0, t_run = 0;
#define AP_TRY(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            rc = local.fail(BIOALIGN_EHIP, "%s failed: %s", #call, hipGetErrorString(e_));  \
            goto done;                                                                      \
        }                                                                                   \
    } while (0)
    AP_TRY(hipSetDevice(hip_dev));
    AP_TRY(hipStreamCreate(&st));
    if (!d_list) {
        if ((rc = list.alloc(&local, hip_dev, total + 16)) != 0 || (rc = loff.alloc(&local, hip_dev, (n_seqs + 1) * 8)) != 0)
            goto done;
        if (total) AP_TRY(hipMemcpyAsync(list.p, h_list, total, hipMemcpyHostToDevice, st));
        AP_TRY(hipMemcpyAsync(loff.p, off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
        d_list = list.as<uint8_t>();
        d_off = loff.as<uint64_t>();
    }
    {
        const uint64_t cap = std::min(CH, hi - lo);
        if ((rc = q1.alloc(&local, hip_dev, (cap + 1) * 8)) != 0 || (rc = q2.alloc(&local, hip_dev, (cap + 1) * 8)) != 0 ||
            (rc = sc.alloc(&local, hip_dev, cap * 8)) != 0)
            goto done;
        if (ends && ((rc = e1.alloc(&local, hip_dev, cap * 8)) != 0 || (rc = e2.alloc(&local, hip_dev, cap * 8)) != 0)) goto done;
        v1.resize(cap);
        v2.resize(cap);
    }
    for (uint64_t c0 = lo; c0 < hi; c0 += CH) {
        const uint64_t N = std::min(CH, hi - c0);
        AP_TRY(ba_launch_pair_offsets(d_off, c0, N, n_seqs, S.ordered ? 1 : 0, q1.as<uint64_t>(), q2.as<uint64_t>(), st));
        uint64_t i, j;
        if (S.ordered) {
            i = c0 / (n_seqs - 1);
            uint64_t r = c0 % (n_seqs - 1);
            for (uint64_t k = 0; k < N; ++k) {
                j = r < i ? r : r + 1;
                v1[k] = (uint32_t)std::min<uint64_t>(off[i + 1] - off[i], 0xffffffffu);
                v2[k] = (uint32_t)std::min<uint64_t>(off[j + 1] - off[j], 0xffffffffu);
                if (++r == n_seqs - 1) {
                    r = 0;
                    ++i;
                }
            }
        } else {
            pair_of(c0, &i, &j);
        }
        for (uint64_t k = 0; !S.ordered && k < N;) {  // row i: its own length against the lengths of sequences j .. i - 1
            const uint32_t li = (uint32_t)std::min<uint64_t>(off[i + 1] - off[i], 0xffffffffu);
            const uint64_t stop = std::min<uint64_t>(N, k + (i - j));
            for (; k < stop; ++k, ++j) {
                v1[k] = li;
                v2[k] = (uint32_t)std::min<uint64_t>(off[j + 1] - off[j], 0xffffffffu);
            }
            if (j == i) {
                ++i;
                j = 0;
            }
        }
        t_fill += ms_since(t_mark);
        t_mark = now();
        bioalign_batch_t sub = *in;
        sub.s1 = sub.s2 = nullptr;
        sub.s1_off = sub.s2_off = nullptr;
        sub.n_pairs = N;
        sub.flags = BIOALIGN_SCORE_ONLY | (ends ? BIOALIGN_END_CELLS : 0u);
        pl = new bioalign_plan;
        pl->ctx = &local;
        pl->device = dev_index;
        pl->pool = ctx->pool;
        if (plan_build(pl, &sub, v1.data(), v2.data()) != 0) {
            rc = local.code ? local.code : BIOALIGN_EINVAL;
            goto done;
        }
        t_plan += ms_since(t_mark);
        t_mark = now();
        rc = bioalign_plan_run(pl, d_list, q1.as<uint64_t>(), d_list, q2.as<uint64_t>(), sc.as<int64_t>(), nullptr,
                               nullptr, e1.as<uint64_t>(), e2.as<uint64_t>(), st);
        if (rc) goto done;
        AP_TRY(hipMemcpyAsync(S.scores + (c0 - S.base), sc.p, N * 8, hipMemcpyDeviceToHost, st));
        if (ends) {
            AP_TRY(hipMemcpyAsync(S.end_i + (c0 - S.base), e1.p, N * 8, hipMemcpyDeviceToHost, st));
            AP_TRY(hipMemcpyAsync(S.end_j + (c0 - S.base), e2.p, N * 8, hipMemcpyDeviceToHost, st));
        }
        AP_TRY(hipStreamSynchronize(st));
        bioalign_plan_destroy(pl);
        pl = nullptr;
        t_run += ms_since(t_mark);
        t_mark = now();
    }
    if (timing)
        fprintf(stderr, "bioalign all-pairs dev %d, %llu pairs: lengths %.2f ms, plan %.2f, kernels + copy out %.2f, total %.2f\n",
                dev_index, (unsigned long long)(hi - lo), t_fill, t_plan, t_run, ms_since(t_start));
done:
    if (st) (void)hipStreamSynchronize(st);
    if (pl) bioalign_plan_destroy(pl);  // touches pl->ctx (== &local), still alive here
    if (st) (void)hipStreamDestroy(st);
    if (rc) *err = local.err;
    return rc;
#undef AP_TRY
}

// pairs [first, first + count) of the numbering; S.scores[0] is pair `first`
int all_pairs(bioalign_ctx *ctx, const bioalign_batch_t *in, const uint64_t *off, const uint8_t *h_list,
              const uint8_t *d_list, const uint64_t *d_off, int d_list_dev, uint64_t first, uint64_t count, PairSpace S)
{
    const uint64_t n = in->n_pairs;
    int64_t *const scores = S.scores;
    S.base = first;
    if (n > (1ull << 31)) return ctx->fail(BIOALIGN_EINVAL, "too many sequences");
    const uint64_t P = n < 2 ? 0 : (S.ordered ? n * (n - 1) : n * (n - 1) / 2);
    if (count == ~0ull) count = P - std::min(first, P);
    if (first > P || count > P - first) return ctx->fail(BIOALIGN_EINVAL, "pair range outside the triangle of %llu pairs", (unsigned long long)P);
    if (count == 0) return 0;
    if (!scores) return ctx->fail(BIOALIGN_EINVAL, "null scores buffer");
    for (uint64_t k = 0; k < n; ++k)
        if (off[k + 1] < off[k]) return ctx->fail(BIOALIGN_EINVAL, "offsets must be non-decreasing (sequence %llu)", (unsigned long long)k);
    // Ranges of pairs balanced by cell count: one per device (as bioalign_batch; a list that already lies on one
    // device stays there), and for large lists several per device, each driven by a host thread of its own, so
    // that the host work of one range (lengths, plan) overlaps the kernels of another.  BIOALIGN_HOST_SHARES
    // sets the ranges per device.
    const int ndev = d_list ? 1 : (int)ctx->devices.size();
    const int per_dev = (int)std::min<uint64_t>(std::max<uint64_t>(env_u64("BIOALIGN_HOST_SHARES", count >= 262144 ? 3 : 1), 1), 8);
    const int D = ndev * per_dev;
    std::vector<uint64_t> cut(D + 1, first + count);
    cut[0] = first;
    if (D > 1 && (count != P || S.ordered)) {
        for (int d = 1; d < D; ++d) cut[d] = first + count * d / D;  // a caller's range, ordered pairs: by pair count
    } else if (D > 1) {
        // row i costs (len_i + 1) * sum over j < i of (len_j + 1)
        std::vector<double> cum(n + 1, 0.0), rowpre(n + 1, 0.0);
        for (uint64_t k = 0; k < n; ++k) cum[k + 1] = cum[k] + (double)(off[k + 1] - off[k] + 1);
        for (uint64_t k = 0; k < n; ++k) rowpre[k + 1] = rowpre[k] + (double)(off[k + 1] - off[k] + 1) * cum[k];
        for (int d = 1; d < D; ++d) {
            const double target = rowpre[n] * d / D;
            uint64_t i = (uint64_t)(std::upper_bound(rowpre.begin(), rowpre.end(), target) - rowpre.begin());
            i = std::min<uint64_t>(std::max<uint64_t>(i, 1), n) - 1;  // rowpre[i] <= target
            uint64_t j = 0;
            if (i >= 1) {
                const double rem = (target - rowpre[i]) / (double)(off[i + 1] - off[i] + 1);
                j = (uint64_t)(std::lower_bound(cum.begin(), cum.begin() + i + 1, rem) - cum.begin());
                j = std::min(j, i);
            }
            cut[d] = std::min(std::max(i * (i > 0 ? i - 1 : 0) / 2 + j, cut[d - 1]), P);
        }
    }
    std::vector<int> rcs(D, 0);
    std::vector<std::string> errs(D);
    auto share = [&](int d) {
        rcs[d] = all_pairs_share(ctx, d_list ? d_list_dev : d / per_dev, in, off, h_list, d_list, d_off, S, cut[d],
                                 cut[d + 1], &errs[d]);
    };
    if (D == 1) {
        share(0);
    } else {
        std::vector<std::thread> th;
        for (int d = 0; d < D; ++d) th.emplace_back(share, d);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < D; ++d)
        if (rcs[d]) {
            ctx->err = errs[d];
            ctx->code = rcs[d];
            return rcs[d];
        }
    return 0;
}

}  // namespace

extern "C" int bioalign_all_pairs_range(bioalign_ctx *ctx, const bioalign_batch_t *in, uint64_t first_pair,
                                        uint64_t n_range, int64_t *scores)
{
    if (!ctx) return BIOALIGN_EINVAL;
    if (!in || !in->s1_off) return ctx->fail(BIOALIGN_EINVAL, "null argument");
    if (in->s1_off[in->n_pairs] && !in->s1) return ctx->fail(BIOALIGN_EINVAL, "null sequence bytes");
    PairSpace S;
    S.scores = scores;
    return all_pairs(ctx, in, in->s1_off, in->s1, nullptr, nullptr, 0, first_pair, n_range, S);
}

extern "C" int bioalign_all_ordered_pairs(bioalign_ctx *ctx, const bioalign_batch_t *in, int64_t *scores, uint64_t *end_i,
                                          uint64_t *end_j)
{
    if (!ctx) return BIOALIGN_EINVAL;
    if (!in || !in->s1_off) return ctx->fail(BIOALIGN_EINVAL, "null argument");
    if (in->s1_off[in->n_pairs] && !in->s1) return ctx->fail(BIOALIGN_EINVAL, "null sequence bytes");
    if ((end_i == nullptr) != (end_j == nullptr)) return ctx->fail(BIOALIGN_EINVAL, "end_i and end_j come together");
    PairSpace S;
    S.ordered = true;
    S.scores = scores;
    S.end_i = end_i;
    S.end_j = end_j;
    return all_pairs(ctx, in, in->s1_off, in->s1, nullptr, nullptr, 0, 0, ~0ull, S);
}

extern "C" int bioalign_all_pairs(bioalign_ctx *ctx, const bioalign_batch_t *in, int64_t *scores)
{
    return bioalign_all_pairs_range(ctx, in, 0, ~0ull, scores);
}

extern "C" int bioalign_fasta_all_pairs(const bioalign_fasta *F, const bioalign_batch_t *in, int64_t *scores)
{
    if (!F) return BIOALIGN_EINVAL;
    bioalign_ctx *ctx = F->ctx;
    if (!in) return ctx->fail(BIOALIGN_EINVAL, "null argument");
    bioalign_batch_t sub = *in;
    sub.n_pairs = F->n_records;
    PairSpace S;
    S.scores = scores;
    return all_pairs(ctx, &sub, F->h_off.data(), nullptr, F->seq.as<uint8_t>(), F->off.as<uint64_t>(), F->dev, 0, ~0ull, S);
}
