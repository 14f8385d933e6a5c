// common.h -- structures shared by the host planner and the gfx950 kernels.
//
// Vocabulary (follows the reference's domain):
//   pair    : one (s1, s2) alignment problem; s1 labels the rows (index i),
//             s2 the columns (index j), as in align_affine.go:8-9.
//   bundle  : up to 64 pairs of identical shape (len(s1), len(s2)); one
//             wavefront aligns one bundle, one pair per lane.
//   strip   : R consecutive rows of the DP matrix held in VGPRs while the
//             wavefront sweeps the columns.
//   boundary row : the last row of the previous strip (m+1 scores per lane),
//             kept in HBM/L2, lane-interleaved so every access is one
//             coalesced 256-byte line per wavefront.
//   direction words : the bit-packed traceback matrix, one word per
//             (strip, column, lane) and bit plane.
#pragma once
#include <stdint.h>

#define BA_WAVE 64

enum { BA_GLOBAL = 0, BA_LOCAL = 1, BA_FITTING = 2, BA_OVERLAP = 3 };

struct BaBundle {
    uint32_t n, m;     // len(s1), len(s2) of every pair in the bundle
    uint32_t first;    // first position in the shape-sorted pair order
    uint32_t count;    // pairs in the bundle (1..64)
    uint64_t s1w_off;  // dword offset of the bundle's lane-interleaved s1 words (raw bytes, front padded)
    uint64_t s2w_off;  // dword offset of the bundle's lane-interleaved s2 words (raw bytes)
    uint64_t hb_off;   // dword offset of the bundle's boundary-delta words (strip hand-over row)
    uint64_t dir_off;  // dword offset of the bundle's direction words
    uint64_t rec_off;  // record offset of the bundle's move records (lane l: rec_off + l * rec_cap)
    uint32_t rec_cap;  // records per pair = header + ceil((n + m) / 16)
    uint32_t pad;
};

// end cell of a local alignment, found by the DP kernel
struct BaLocalBest {
    int32_t score;
    int32_t i, j;
    int32_t pad;
};

struct BaPackParams {
    const BaBundle *bundles;
    uint32_t bundle_begin, bundle_end;
    const uint32_t *perm;      // shape-sorted order -> caller's pair index
    const uint8_t *s1, *s2;    // caller's concatenated bytes (device)
    const uint64_t *s1_off, *s2_off;
    uint32_t *s1w, *s2w;
    int rows_per_strip;
};

struct BaDpParams {
    const BaBundle *bundles;
    uint32_t bundle_begin, bundle_end;
    const uint32_t *s1w, *s2w;
    int32_t *hb;
    uint32_t *dir;
    BaLocalBest *best;     // indexed by sorted position
    const uint32_t *wt;    // transposed int8 score table, dwords [nb4][na_pad]
    const uint8_t *code1, *code2;  // [256] byte -> dense code (256-byte aligned device arrays)
    int pad_code;          // s1-side code of the rows above the matrix
    int nb, nb4, na_pad;
    int gap;               // linear gap penalty (>= 0 on this path)
    int gap_open, gap_ext; // affine
    int nib;               // affine, large alphabets: the strip profile holds 4-bit scores w - kappa0 (kernels_affine.hip)
    int kappa;             // ... and this is added back with the diagonal: min(w) + open + ext
    int fkey;              // affine local: the best-cell keys fit 24 bits and are built as floats (kernels_affine.hip)
    const int32_t *table;  // wide kernels: na * nb scores as the caller gave them (kernels_wide.hip)
    int wide_rows;         // wide kernels: rows per strip of the packed s1 words (their front padding)
    int pipe_bundles;      // affine strip pipeline: bundles a block runs back to back (kernels_affine.hip)
    int no_dir;            // bit-parallel kernels, score-only plans: nothing is stored for a traceback (`dir` has no room)
};

struct BaTraceParams {
    const BaBundle *bundles;
    uint32_t bundle_begin, bundle_end;
    const uint32_t *perm;
    const uint8_t *s1, *s2;
    const uint64_t *s1_off, *s2_off;
    const uint8_t *code1, *code2;
    const int32_t *table;  // na * nb scores, as the caller gave them
    int na, nb;
    const uint32_t *s1w, *s2w;  // lane-interleaved raw bytes (pack kernel)
    const int32_t *hb;
    const uint32_t *dir;
    const BaLocalBest *best;
    int gap, gap_open, gap_ext;
    int rows_per_strip;
    int residue_bits;
    int score_only;
    uint8_t gap_symbol;
    int64_t *scores;
    uint8_t *t1, *t2;
    uint64_t *t_off, *t_len;
    void *rec;             // move records (linear-gap walk + expansion); null only with score_only
    int max_n, max_m;           // longest s1 / s2 of the plan (0, 0: unknown)
    const uint64_t *dense_end;  // [n_pairs] by pair: where the pair's trace ends when the traces are laid back to back
                                // in pair order instead of into the pairs' slots (null: slots)
};

// ----------------------------------------------------------------------------
// long mode: ONE pair, rows spread over the lanes of a wavefront (kernels_long.hip)
//   strip   : 64*T consecutive rows, T per lane; one wavefront sweeps it over all columns,
//             lane l working on column t - l at step t (anti-diagonal skew)
//   step    : one column per lane; residue words are stored by step, not by column
// ----------------------------------------------------------------------------
enum { BA_LONG_PHASE_DP = 1, BA_LONG_PHASE_WALK = 2 };

struct BaLongParams {
    const uint8_t *s1, *s2;        // the pair's raw bytes (device)
    uint32_t n, m;
    uint32_t strips;               // ceil(n / (64*T)), bottom aligned
    uint32_t nsuper;               // workgroups = ceil(strips / strips per workgroup); the last strip is the last wave of the last one
    uint32_t wg_begin, wg_count;   // this launch runs workgroups [wg_begin, wg_begin + wg_count) (0, 0: all): a row block of the matrix;
                                   // the workgroups above must have run (their bottom rows lie in bnd)
    uint32_t words_per_strip;      // residue dwords per lane per strip = ceil((m + 63) / CPD)
    const uint32_t *rowprof;       // signed bytes [na + 1][nb]: w(a, b) + 2g; row na = padding (-128 against everything)
    const uint8_t *code1, *code2;  // [256] byte -> dense code
    int pad_code;                  // = na
    int gap;                       // linear: the gap penalty; affine: gap open
    int gap_ext;                   // affine: gap extend (the affine kernels are chosen by the host, not by this field)
    int32_t *bnd;                  // [nsuper][bnd_stride] granules {working-space score, tag}, 8 bytes each: the bottom row of every
                                   // workgroup's last strip; zeroed before the launch (tag 0 = not there yet)
    uint32_t bnd_stride;           // granules per row: m rounded up to 16 (a sub-block of 16 columns is one whole 128-byte line)
    uint32_t epoch;                // tag of this launch's granules: 1 .. 0x7ffffffe (top bit: poisoned)
    uint32_t *resid;               // [strips][words_per_strip][64 lanes]
    uint32_t *status;              // != 0: a wait ran out (should never happen)
    int32_t *lane_max;             // local mode: [strips][64] the largest score each lane saw in its T rows
                                   // (affine: [strips][64][T][2] = largest true score of the row, step it was first seen at)
    // traceback
    const int32_t *table;          // na * nb scores as the caller gave them
    int nb;
    uint8_t gap_symbol;
    int score_only;
    int64_t *score;
    uint8_t *t1, *t2;              // capacity n + m each, written right aligned
    uint64_t *t_off, *t_len;
    uint32_t *ops;                 // 2-bit move codes of the walk, 16 per dword, last column first
    uint32_t *walk;                // [0], [1]: cell the walk starts from (i, j); [2]: number of moves;
                                   // [3]: true score of the end cell found (int32); [4], [5]: cell the walk stopped in
    // column stripes (one matrix over several devices): this launch covers columns col0+1 .. col0+m of the matrix;
    // n, m, s2, bnd, resid, ops, t1, t2 are the stripe's own.  All zero / null: the whole matrix.
    uint32_t col0;                 // columns of the matrix left of this stripe
    const int32_t *left_col;       // [n + 1] working-space scores Hs(i, col0), i = 0..n (null: the matrix' own column 0)
    int32_t *right_col;            // [n + 1] out: Hs(i, col0 + m) (null: not wanted)
    int start_i, start_j;          // walk: start from this cell instead of looking for the end cell (start_j < 0: look for it)
    int scan_only;                 // walk: find the end cell of this stripe's bottom row, report it, stop
    // linear-space traceback (checkpointed): the DP runs once without storing residues (no_resid; every workgroup's
    // bottom row stays in bnd), then row blocks of workgroups are recomputed bottom-up WITH residues (wg_begin /
    // wg_count, resid = the block's own buffer) and walked one after the other
    int no_resid;                  // DP: do not store residue words
    int first_strip;               // walk: strips below this index are not in resid (resid[0] is strip first_strip); the
                                   // row just above the block is read from bnd row wg_begin - 1
    int stop_row;                  // walk: stop on reaching this row (the last row above the block; 0: the matrix' top)
    uint32_t trace_done;           // expansion: moves already expanded by the blocks below (they lie behind this block's)
    uint64_t trace_cap;            // expansion: capacity of t1 / t2 (0: n + m; a recomputed block runs with a smaller m)
    // several stripes of one pair in ONE launch (all on this device): workgroup b of the grid is workgroup order[b].y of
    // stripe order[b].x and takes its parameters from multi[order[b].x].  A stripe's last column goes to its right
    // neighbour as tagged granules {score, tag} (right_gran, [n + 1], zeroed before the launch); the neighbour's strips
    // wait for the granules of their own rows (left_gran) and copy the scores to their plain left_col for the walk.
    unsigned long long *stamps;    // diagnostics (BIOALIGN_LONG_STAMPS=1): [strips][2] real-time counter at a strip's first step
                                   // and behind its last one (100 MHz); null: none
    const BaLongParams *multi;
    const uint2 *order;
    const unsigned long long *left_gran;
    unsigned long long *right_gran;
};

// FASTA text -> concatenated sequences + offsets (kernels_fasta.hip)
struct BaFastaParams {
    const uint8_t *text;      // device, len bytes
    uint64_t len;
    uint64_t n_chunks;
    uint64_t *chunk_count;    // [n_chunks] newlines per chunk, then (scanned) newlines before the chunk
    uint64_t *line_start;     // [n_lines]
    uint64_t n_lines, n_line_blocks;
    uint64_t *trim_begin;     // [n_lines] first byte of the trimmed line
    uint32_t *trim_len;       // [n_lines] its length (0: blank)
    uint64_t *ldata, *lhdr;   // [n_lines] sequence bytes / header flag of the line, then (scanned) the totals before it
    uint64_t *part_data, *part_hdr;  // [n_line_blocks] block sums, then scanned
    uint8_t *seq;             // out: the records' data back to back
    uint64_t *off;            // out: [n_records + 1]
    uint64_t *hdr_begin, *hdr_len;   // out: [n_records] header lines as spans of the text
    uint32_t *error;          // out: != 0: data before the first header
    // FASTAReader.ReadSeq's rules instead of ReadFASTA's (fasta.go:109-142): lines end at "\n" or "\r\n" and are
    // otherwise taken as they are (no trimming), data before the first header is ignored, and a line that does not
    // fit bufio's buffer of `line_limit` bytes (terminator included) is the error "FASTA line too long"
    int reader;
    uint64_t line_limit;
    unsigned long long *too_long;  // out: the first such line (min over lines), ~0 if none
};
