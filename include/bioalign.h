/*
 * bioalign.h -- C ABI of the MI355X (gfx950) batched pairwise-alignment engine.
 *
 * This is the drop-in boundary for soniakeys/bio's DP alignment path.  The
 * reference has no FFI layer: its boundary is a handful of
 * exported Go functions, so each entry point below names the Go function(s) a
 * cgo wrapper would route through it.  All citations are file:line relative to
 * the reference tree.  INTEGRATION.md shows the cgo binding.
 *
 * Conventions
 *  - plain C types only; the library never keeps a caller pointer after a call
 *    returns (cgo rule) except inside an explicit bioalign_plan, which only
 *    keeps copies;
 *  - return 0 on success, a negative BIOALIGN_E* code on failure, message via
 *    bioalign_last_error();
 *  - a context is used from one host thread at a time; separate contexts may
 *    be used concurrently;
 *  - the library fails loudly (BIOALIGN_ENODEVICE) when no GPU is usable:
 *    there is no CPU fallback.
 */
#ifndef BIOALIGN_H
#define BIOALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIOALIGN_ABI_VERSION 1

/* Alignment modes = the `mode` strings of AlignPair (align_pair.go:5-16). */
enum {
    BIOALIGN_GLOBAL = 0,  /* AlignGlobal (align.go:76), AlignPair "global", AlignGlobalLS (align_linear.go:9), AlignGlobalAffine (align_affine.go:12) */
    BIOALIGN_LOCAL = 1,   /* AlignLocal (align.go:152), AlignPair "local" */
    BIOALIGN_FITTING = 2, /* AlignPair "fitting" (align_pair.go:157-166) */
    BIOALIGN_OVERLAP = 3  /* AlignPair "overlap" (align_pair.go:186-195) */
};

/* flags */
enum {
    BIOALIGN_SCORE_ONLY = 1u << 0,  /* skip the direction matrix and the traceback */
    BIOALIGN_DENSE_TRACES = 1u << 1,/* plans: lay the traces back to back in pair order (t_off[p] = sum of the
                                       lengths before p) instead of into the pairs' slots; honoured for linear gaps
                                       when the batch runs in one workspace chunk, ignored otherwise -- t_off / t_len
                                       are right either way.  bioalign_batch lays its results out like this
                                       whenever it can: what crosses PCIe is then the traces, not their slots. */
    BIOALIGN_TRACE_OPS = 1u << 2,   /* instead of the two gapped strings, t1 receives one compact record per pair (t2
                                       is not touched and may be NULL): uint32 i, uint32 j -- the cell the alignment
                                       ends in ((0, 0) when it has no moves) -- then the moves of the traceback, 2 bits each (0: s1[i-1] against a
                                       gap, i -= 1; 1: a gap against s2[j-1], j -= 1; 2: s1[i-1] against s2[j-1], both
                                       -= 1), first move = LAST column of the alignment, 16 moves per little-endian
                                       uint32.  t_off[p] = byte offset of the pair's record in t1, t_len[p] = number
                                       of moves.  t1 needs bioalign_ops_capacity() bytes.  bioalign_expand_ops() turns
                                       a record into the reference's two strings.  Linear gaps, batches that run in
                                       one workspace chunk; anything else answers BIOALIGN_EUNSUPPORTED. */
    BIOALIGN_END_CELLS = 1u << 3    /* with BIOALIGN_SCORE_ONLY: no traceback, but t_off[p] = row i and t_len[p] =
                                       column j of the cell the alignment ends in (global: (n, m); fitting /
                                       overlap: (n, leftmost best column of the bottom row), align_pair.go:169-182;
                                       local: the first best cell in row-major order, (0, 0) for score 0) -- e.g. the
                                       length of the overlap for an assembler.  t1 / t2 are not touched.  Batch
                                       kernels only (bioalign_long ignores it). */
};

/* error codes */
enum {
    BIOALIGN_OK = 0,
    BIOALIGN_EINVAL = -1,      /* bad argument (mode, sizes, null pointer) */
    BIOALIGN_ENODEVICE = -2,   /* no usable HIP device / HIP runtime failure at start-up */
    BIOALIGN_ENOMEM = -3,      /* host or device allocation failed */
    BIOALIGN_EHIP = -4,        /* a HIP call failed; see bioalign_last_error */
    BIOALIGN_EUNSUPPORTED = -5 /* parameters outside what the kernels cover (message says which) */
};

typedef struct bioalign_ctx bioalign_ctx;
typedef struct bioalign_plan bioalign_plan;

/*
 * A batch of independent pairs.  Pair p is s1[s1_off[p], s1_off[p+1]) against
 * s2[s2_off[p], s2_off[p+1]); offsets start at 0 and have n_pairs+1 entries.
 *
 * Scoring replaces the Aligner interface (align.go:5-8) with a table the
 * wrapper materialises once per batch: code1/code2 map every
 * byte of the s1 / s2 side to a dense code (< na / < nb) and
 * score_table[code1[x]*nb + code2[y]] == Aligner.Score(x, y).  Bytes that do not
 * occur in the batch may map to anything.
 *
 * gap_open == gap_extend selects the linear-gap recurrences of align.go /
 * align_pair.go (indelPenalty = that value); otherwise the three-level Gotoh
 * recurrences of align_affine.go:44-103 in integers, where a gap of length L
 * costs gap_open + (L-1)*gap_extend.
 */
typedef struct {
    const uint8_t *s1;
    const uint64_t *s1_off;
    const uint8_t *s2;
    const uint64_t *s2_off;
    uint64_t n_pairs;
    int32_t mode;
    const int32_t *score_table; /* na * nb, row-major */
    int32_t na, nb;
    const uint8_t *code1; /* [256] */
    const uint8_t *code2; /* [256] */
    int32_t gap_open;
    int32_t gap_extend;
    uint8_t gap_symbol; /* the package variable GapSymbol (align.go:3) */
    uint32_t flags;
} bioalign_batch_t;

/*
 * Results, in caller-allocated buffers.  scores has n_pairs entries (Go int).
 * t1 and t2 each have capacity s1_off[n_pairs] + s2_off[n_pairs] bytes (the sum
 * of the longest possible traces).  The traces of pair p are the t_len[p] bytes
 * starting at t_off[p] of t1 and of t2; WHERE in the buffers they lie is the
 * library's choice -- inside the pair's own slot
 * [s1_off[p]+s2_off[p], s1_off[p+1]+s2_off[p+1]), or back to back in pair order
 * (BIOALIGN_DENSE_TRACES) -- so read them through t_off / t_len only.
 * t_len[p] == 0 is the reference's nil trace.
 * With BIOALIGN_SCORE_ONLY, t1/t2/t_off/t_len may be NULL.
 */
typedef struct {
    int64_t *scores;
    uint8_t *t1;
    uint8_t *t2;
    uint64_t *t_off;
    uint64_t *t_len;
} bioalign_result_t;

/* Opens the engine on the given HIP devices (NULL / 0 = device 0 only). */
bioalign_ctx *bioalign_create(const int *device_ids, int n_devices);
/* Closes the engine.  Plans, fasta objects and stripes made from a context hold a reference on it: the context's
 * memory goes with the last of them, so the order of the destroy calls does not matter (no other call may use the
 * context handle after bioalign_destroy). */
void bioalign_destroy(bioalign_ctx *ctx);
/* Message of the last failure on this context (or of a failed create when ctx is NULL). */
const char *bioalign_last_error(const bioalign_ctx *ctx);
int bioalign_abi_version(void);
int bioalign_device_count(const bioalign_ctx *ctx);

/*
 * End to end on host buffers: plan, upload, run on the context's devices
 * (pairs are split across them by cell count, no collective), download.
 * Replaces the per-pair bodies of AlignGlobal / AlignLocal (align.go:76-249),
 * AlignPair (align_pair.go:3-246), AlignGlobalLS (align_linear.go:9-45; here it
 * returns what AlignGlobal returns, as its disabled test asks, align_test.go:55-66)
 * and AlignGlobalAffine (align_affine.go:12-167; score as written, canonical
 * traceback instead of the broken one).
 */
int bioalign_batch(bioalign_ctx *ctx, const bioalign_batch_t *in, bioalign_result_t *out);

/*
 * The same work split into stages, for callers that keep data resident in HBM
 * (pipelines, the benchmark).  A plan depends on the pair lengths, the mode and
 * the scoring, not on the sequence bytes: it groups pairs into wavefront-sized
 * bundles of equal shape, sizes and allocates the device workspace on device
 * `device_index` of the context, and uploads the score tables.
 * `in->s1` / `in->s2` may be NULL here.
 */
bioalign_plan *bioalign_plan_create(bioalign_ctx *ctx, int device_index, const bioalign_batch_t *in);
void bioalign_plan_destroy(bioalign_plan *plan);
/* Bytes of device memory the plan holds (workspace + tables). */
uint64_t bioalign_plan_workspace_bytes(const bioalign_plan *plan);
/* Interior cells (sum of len1*len2) the plan covers. */
uint64_t bioalign_plan_cells(const bioalign_plan *plan);

/*
 * Runs the planned batch.  Every pointer is DEVICE memory on the plan's device:
 * d_s1/d_s2 the concatenated bytes, d_s1_off/d_s2_off the uint64 offsets, and
 * the outputs laid out as in bioalign_result_t.  Work is queued on `stream`
 * (a hipStream_t, NULL = the default stream) and the call returns without
 * waiting for it.  Packing, DP, and traceback kernels all run inside this call.
 */
int bioalign_plan_run(bioalign_plan *plan, const uint8_t *d_s1, const uint64_t *d_s1_off,
                      const uint8_t *d_s2, const uint64_t *d_s2_off,
                      int64_t *d_scores, uint8_t *d_t1, uint8_t *d_t2,
                      uint64_t *d_t_off, uint64_t *d_t_len, void *stream);

/*
 * Several modes over the SAME batch (the batched form of calling AlignGlobal and AlignLocal,
 * align.go:76 / 152, or several AlignPair modes, align_pair.go:3, on the same pairs): a plan
 * may read the packed copy of the sequences that another plan of the same batch has made
 * instead of packing them again.  `plan` and `packed_by` must come from the same context and
 * device and from batches with identical lengths (checked); from then on every
 * bioalign_plan_run(plan, ...) must be queued on the same stream AFTER a run of `packed_by`
 * on the same sequence bytes, and before `packed_by` runs on other bytes or is destroyed.
 * packed_by == NULL makes the plan pack for itself again.
 */
int bioalign_plan_share_pack(bioalign_plan *plan, const bioalign_plan *packed_by);

/*
 * Long mode: the same call shape as bioalign_batch, but every pair is aligned on its own by
 * the intra-sequence kernels -- rows spread over the lanes of many wavefronts, anti-diagonal
 * wavefront, strips pipelined through HBM -- which is what a single long pair needs
 * (AlignGlobalLS, align_linear.go:9, on inputs the per-lane batch kernels would crawl through;
 * AlignPair in its four modes, align_pair.go:3, and AlignLocal, align.go:152, likewise; with
 * gap_open != gap_extend Gotoh's three levels, align_affine.go:12, in four modes).  At most 100
 * distinct column symbols (64 with affine gaps); anything else answers BIOALIGN_EUNSUPPORTED.  A pair whose traceback
 * matrix exceeds BIOALIGN_LONG_RESIDUE_BYTES (default: 70 % of the device's free memory) is traced
 * back in linear space -- one DP pass without the matrix, then row blocks recomputed from kept
 * rows and walked bottom-up: same traces, at most twice the cells (every mode, linear and affine
 * gaps).  Runs on
 * the first device of the context -- unless the traceback matrix of a pair (up to one byte per
 * cell) would not fit that device and the context holds several: the matrix is then cut into
 * column stripes, one per device in turn, each stripe's last column handed to the next in row
 * blocks by device-to-device copies (BIOALIGN_LONG_STRIPES=<count> forces a number of stripes;
 * stripes that share a device run in one launch; local mode and affine gaps always run unstriped).
 * kernel_ms (may be NULL) receives the DP + traceback time in milliseconds.
 */
int bioalign_long(bioalign_ctx *ctx, const bioalign_batch_t *in, bioalign_result_t *out, float *kernel_ms);

/*
 * One column stripe of a long-mode alignment, for a process that owns one GPU of a group
 * (no counterpart in the reference; the stripes of bioalign_long, one at a time).  The pair's
 * matrix is cut into column ranges [col_begin, col_end), one per process, left to right:
 *   create      in = a batch of ONE pair (the whole s1 and the whole s2), linear gaps,
 *               global / fitting / overlap; the stripe keeps what it needs on the device;
 *   dp          d_left_col: device memory, rows()+1 int32 = the last column of the stripe on
 *               the left in the kernels' working space (NULL for the stripe with col_begin 0);
 *               d_right_col: device memory, rows()+1 int32, receives this stripe's last column
 *               (may be NULL).  Returns when the stripe's DP is done;
 *   end_cell    the end cell the stripe offers (align.go:123-124; align_pair.go:169-182): its
 *               true score, its column inside the stripe (0 = the matrix' column 0, first stripe
 *               only) and the score to report if it is the one chosen.  Global: ask the last
 *               stripe; fitting / overlap: take the largest true score, leftmost stripe on ties;
 *   walk        from cell (i, j_local) to the stripe's border: the piece of both gapped strings
 *               (host memory, room for rows() + width each; a piece belongs in front of the
 *               pieces of the stripes to its right), and the cell the walk stopped in.  The walk
 *               goes on in the stripe on the left from (exit_i, that stripe's width) unless this
 *               was the first stripe, or exit_i == 0 in fitting mode (free top row).
 */
typedef struct bioalign_stripe bioalign_stripe;
bioalign_stripe *bioalign_stripe_create(bioalign_ctx *ctx, int device_index, const bioalign_batch_t *in,
                                        uint64_t col_begin, uint64_t col_end);
void bioalign_stripe_destroy(bioalign_stripe *stripe);
uint64_t bioalign_stripe_rows(const bioalign_stripe *stripe);
int bioalign_stripe_dp(bioalign_stripe *stripe, const int32_t *d_left_col, int32_t *d_right_col);
/* The stripe's DP in row blocks, so that the stripes of a pair overlap: block k covers rows
 * [row_begin, row_end) of column 0 .. -- the same rows in every stripe of the pair -- and is one launch; the stripe on
 * the right may run block k as soon as it has these rows of this stripe's last column, while this stripe goes on with
 * block k+1.  dp_block: d_left_rows / d_right_rows are device arrays of row_end - row_begin int32 (the block's rows
 * of the neighbour's last column in, of this stripe's last column out; NULL as for bioalign_stripe_dp).  Blocks run
 * once each, in order; bioalign_stripe_dp is all of them. */
uint32_t bioalign_stripe_blocks(const bioalign_stripe *stripe);
int bioalign_stripe_block_rows(const bioalign_stripe *stripe, uint32_t block, uint64_t *row_begin, uint64_t *row_end);
int bioalign_stripe_dp_block(bioalign_stripe *stripe, uint32_t block, const int32_t *d_left_rows, int32_t *d_right_rows);
int bioalign_stripe_end_cell(bioalign_stripe *stripe, int64_t *true_score, uint64_t *j_local, int64_t *score);
int bioalign_stripe_walk(bioalign_stripe *stripe, uint64_t i, uint64_t j_local, uint8_t *t1, uint8_t *t2,
                         uint64_t *len, uint64_t *exit_i, uint64_t *exit_j);

/*
 * Compact traces (BIOALIGN_TRACE_OPS; no counterpart in the reference, whose callers get the two gapped strings of
 * align.go:126-144): bytes t1 must hold for a batch of n_pairs pairs with total_s1 + total_s2 sequence bytes, and
 * the host-side expansion of one pair's record into those strings (t1_out / t2_out: `moves` bytes each).
 */
uint64_t bioalign_ops_capacity(uint64_t n_pairs, uint64_t total_s1, uint64_t total_s2);
int bioalign_expand_ops(const uint8_t *s1, uint64_t n, const uint8_t *s2, uint64_t m, const uint8_t *record,
                        uint64_t moves, uint8_t gap_symbol, uint8_t *t1_out, uint8_t *t2_out);

/*
 * FASTA text -> the batch format, on the GPU (the step before the DP path; ReadFASTA, fasta.go:55-84).
 * The text (host memory, or device memory with text_on_device != 0) is split at '\n', every line trimmed of
 * ASCII white space at both ends, blank lines skipped, a line starting with '>' opens a record, other lines
 * are appended to the last record; data before the first header fails with the reference's "No header".
 * What comes back stays in HBM in the shape bioalign_plan_run takes: device_seq = the records' data back to
 * back, device_off = count + 1 uint64 offsets.  off (host copy of the offsets, what bioalign_plan_create
 * needs) and header_begin / header_len (every header line, trimmed, as a span of the caller's text: FASTASeq's
 * Header, fasta.go:17-20) are host arrays owned by the object.  (Go's TrimSpace also trims UTF-8 encoded
 * Unicode spaces; bytes >= 0x80 are never trimmed here.)
 */
typedef struct bioalign_fasta bioalign_fasta;
bioalign_fasta *bioalign_fasta_parse(bioalign_ctx *ctx, int device_index, const uint8_t *text, uint64_t len,
                                     int text_on_device);
/*
 * The same with the rules of the reference's streaming reader, FASTAReader.ReadSeq (fasta.go:93-142), which differ
 * from ReadFASTA's: a line ends at "\n" or "\r\n" and is otherwise taken as it is (no trimming: a header is a line
 * whose FIRST byte is '>'), data before the first header is ignored instead of being an error, and a line that does
 * not fit bufio.Reader's buffer (line_limit bytes, terminator included; 0 = bufio's default 4096) is the error
 * "FASTA line too long".  The whole stream is parsed in one call; record k is what the (k+1)-th ReadSeq call
 * returns.  bioalign_fasta_too_long_at: -1, or the 0-based ReadSeq call that returns that error (records before it
 * are good, those from it on are not meaningful).
 */
bioalign_fasta *bioalign_fasta_parse_reader(bioalign_ctx *ctx, int device_index, const uint8_t *text, uint64_t len,
                                            int text_on_device, uint64_t line_limit);
int64_t bioalign_fasta_too_long_at(const bioalign_fasta *fasta);
void bioalign_fasta_destroy(bioalign_fasta *fasta);
uint64_t bioalign_fasta_count(const bioalign_fasta *fasta);           /* records */
uint64_t bioalign_fasta_bytes(const bioalign_fasta *fasta);           /* sequence bytes of all records */
float bioalign_fasta_kernel_ms(const bioalign_fasta *fasta);          /* the parse on the device, milliseconds */
const uint8_t *bioalign_fasta_device_seq(const bioalign_fasta *fasta);
const uint64_t *bioalign_fasta_device_off(const bioalign_fasta *fasta);
const uint64_t *bioalign_fasta_off(const bioalign_fasta *fasta);
const uint64_t *bioalign_fasta_header_begin(const bioalign_fasta *fasta);
const uint64_t *bioalign_fasta_header_len(const bioalign_fasta *fasta);
int bioalign_fasta_copy_seq(const bioalign_fasta *fasta, uint8_t *dst); /* all sequence bytes to host memory */

/*
 * All pairs of one list.  The batched client under
 * DNA8List.DistanceMatrix (dna_list.go:951-969) and StrList.DistanceMatrix
 * (str.go:450-469) for distance functions that are a function of an alignment
 * score: the reference calls f(l[i], l[j]) for every j < i, one alignment per
 * call; here those n(n-1)/2 alignments are one score-only batch.
 *
 * `in` describes the LIST, not pairs: n_pairs = number of sequences n, s1 / s1_off
 * = their bytes back to back and n+1 offsets (host memory); s2 / s2_off are
 * ignored.  Sequence i is the row sequence (code1, table row) and sequence j the
 * column sequence (code2) of pair (i, j), as in f(l[i], l[j]).  mode, score table,
 * gap penalties as in bioalign_batch; flags are ignored (scores only).
 *
 * scores[i*(i-1)/2 + j] (j < i; n(n-1)/2 values, host memory) = the score of that
 * alignment.  The list is copied to each device once; pairs address it by offset,
 * nothing is replicated per pair.  Pairs are split over the context's devices by
 * cell count, no collective.  n < 2: nothing is written.
 *
 * bioalign_fasta_all_pairs: the same over the records of a parsed FASTA text,
 * which already lie in HBM (on the parse's device; in->n_pairs, s1, s1_off are
 * ignored too).
 */
int bioalign_all_pairs(bioalign_ctx *ctx, const bioalign_batch_t *in, int64_t *scores);
/* Pairs [first_pair, first_pair + n_range) of the same numbering only; scores[0] is pair first_pair.  For programs
 * that run one process per GPU and give every rank a range of the triangle (bio_amd/distance.py). */
int bioalign_all_pairs_range(bioalign_ctx *ctx, const bioalign_batch_t *in, uint64_t first_pair, uint64_t n_range,
                             int64_t *scores);
int bioalign_fasta_all_pairs(const bioalign_fasta *fasta, const bioalign_batch_t *in, int64_t *scores);
/* Every ORDERED pair (i, j), i != j, of the same list -- n(n-1) alignments, pair number i*(n-1) + (j < i ? j : j-1) --
 * for modes that are not symmetric in their two sequences: the all-pairs step of a greedy assembler
 * (align_long.go:35-45) asks for the "overlap" alignment of every fragment's suffix with every other fragment's
 * prefix.  scores as above; end_i / end_j (both or neither; n(n-1) entries each) receive the end cell of every
 * alignment as BIOALIGN_END_CELLS describes it: for "overlap", end_j is the length of the overlap in sequence j. */
int bioalign_all_ordered_pairs(bioalign_ctx *ctx, const bioalign_batch_t *in, int64_t *scores, uint64_t *end_i,
                               uint64_t *end_j);

/*
 * One entry point that allocates nothing.  It has no counterpart in the reference; it is for callers that cannot
 * call hipMalloc: tight serving loops, harnesses that forbid allocation.  (Not for a stream under graph capture: the
 * call uploads its tables from pageable host memory and synchronises `stream` before it returns.)  It is
 * bioalign_plan_create + bioalign_plan_run + bioalign_plan_destroy in one call -- same arguments (`in` as for
 * bioalign_plan_create: host offsets give the lengths, in->s1 / s2 may be NULL; the d_ pointers as for
 * bioalign_plan_run: device memory on device `device_index` of the context), same planner, same kernels, same
 * results -- except that the plan's whole device workspace (bundle table, packed sequence words, boundary rows,
 * residue / decision words, move records, end cells) lies in a fixed array of BIOALIGN_NOALLOC_SCRATCH_BYTES
 * that the library's code object holds on every GPU it is loaded on until the process ends.  The call makes no
 * hipMalloc / hipFree, creates no stream or event, uploads its few KB of tables on `stream` and returns once the
 * work queued on `stream` is done.  A batch whose workspace does not fit the array (about 10 000 pairs of 150 bp
 * with linear gaps, 2 bits per cell; half that many cells with affine gaps) answers BIOALIGN_EUNSUPPORTED and
 * nothing is run; BIOALIGN_DENSE_TRACES / BIOALIGN_TRACE_OPS / BIOALIGN_SCORE_ONLY / BIOALIGN_END_CELLS work as in
 * plans.  Every call on a GPU shares that GPU's array, which keeps the last call's contents (the kernels write
 * every word before they read it, nothing has to start at zero).  Calls that would overlap on one GPU -- from two
 * threads, contexts or streams of this process -- are serialised by a per-GPU lock inside the call (it blocks until
 * its work is done anyway); calls on different GPUs run side by side.  Two PROCESSES each hold their own copy of
 * the array.
 */
#define BIOALIGN_NOALLOC_SCRATCH_BYTES (64u << 20)
int bioalign_run_noalloc(bioalign_ctx *ctx, int device_index, const bioalign_batch_t *in, const uint8_t *d_s1,
                         const uint64_t *d_s1_off, const uint8_t *d_s2, const uint64_t *d_s2_off, int64_t *d_scores,
                         uint8_t *d_t1, uint8_t *d_t2, uint64_t *d_t_off, uint64_t *d_t_len, void *stream);

/*
 * Timing hooks for the benchmark (no counterpart in the reference): the plan
 * records a pair of HIP events around every DP kernel launch, on the launch
 * stream, into a fixed pool of 256 pairs.  Once the stream has been
 * synchronised this returns the summed duration in milliseconds of the DP
 * launches recorded since the previous call, and their number, and empties the
 * pool.  Launches beyond the pool size are not timed.
 */
int bioalign_plan_dp_ms(bioalign_plan *plan, float *ms, int *n_launches);

/*
 * Which family of DP kernels the planner chose for this plan (no counterpart
 * in the reference; for benchmarks and tests): "linear", "bitparallel" (unit
 * scoring +1 / -1 / gap 1: align.go:76-249, align_pair.go:3), "affine"
 * or "wide".  A static string.
 */
const char *bioalign_plan_kernel(const bioalign_plan *plan);

#ifdef __cplusplus
}
#endif
#endif
