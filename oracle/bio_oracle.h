/*
 * bio_oracle.h -- CPU restatement of soniakeys/bio's pairwise alignment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bio_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity status: pinned by the reference's own golden vectors G1..G10
 * (reference align_test.go:9-114, align_fitting_test.go:18-30,
 * align_overlap_test.go:21-33; see tests/golden/align_golden.json) for the
 * linear-gap functions.  Everything affine except the global score recurrence
 * is "parity unpinned": the reference has no test for it and its traceback is
 * broken.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef BIO_ORACLE_H
#define BIO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return codes.  BO_PANIC mirrors a Go run-time panic in the reference. */
enum { BO_OK = 0, BO_PANIC = -1, BO_BAD_MODE = -2, BO_NOMEM = -3 };

enum { BO_GLOBAL = 0, BO_LOCAL = 1, BO_FITTING = 2, BO_OVERLAP = 3 };

/*
 * Aligner: the one-method Go interface `Score(x, y byte) int` (align.go:5-8),
 * expressed as a function pointer + context so that every DP cell pays one
 * indirect call, like the reference's interface dispatch.
 * `panicked` is raised by a scorer where the Go code would panic (index out of
 * range in SubstMatrix.Score, subst_matrix.go:15-17).
 */
typedef struct bo_aligner bo_aligner;
struct bo_aligner {
    int64_t (*score)(const bo_aligner *self, uint8_t x, uint8_t y);
    /* SubstMatrix fields (subst_matrix.go:8-12) */
    const uint8_t *alphabet;
    int n_alpha;
    const int32_t *matrix; /* n_alpha * n_alpha, row-major */
    int32_t *index;        /* length index_len = max(alphabet)+1, zero filled */
    int index_len;
    /* match / mismatch scorer (align_fitting_test.go:9-16, align_overlap_test.go:9-19) */
    int64_t match, mismatch;
    /* dense 256x256 table scorer (used to model arbitrary user Aligners) */
    const int32_t *table256;
    int panicked;
};

/* NewSubstMatrix (subst_matrix.go:20-32).  Caller frees with bo_aligner_free. */
bo_aligner *bo_new_subst_matrix(const uint8_t *alphabet, int n_alpha, const int32_t *matrix);
bo_aligner *bo_new_match_aligner(int64_t match, int64_t mismatch);
bo_aligner *bo_new_table_aligner(const int32_t *table256);
bo_aligner *bo_blosum62(void); /* subst_matrix.go:63-86 */
bo_aligner *bo_pam250(void);   /* subst_matrix.go:88-131 */
void bo_aligner_free(bo_aligner *a);
int64_t bo_score(bo_aligner *a, uint8_t x, uint8_t y);

/* package variable GapSymbol (align.go:3) */
void bo_set_gap_symbol(uint8_t g);
uint8_t bo_get_gap_symbol(void);

/* LinearGap / ConstantGap (align.go:18-69). BO_PANIC on unequal lengths. */
int bo_linear_gap(const uint8_t *s, size_t ns, const uint8_t *t, size_t nt,
                  bo_aligner *a, int64_t gap, int64_t *score);
int bo_constant_gap(const uint8_t *s, size_t ns, const uint8_t *t, size_t nt,
                    bo_aligner *a, int64_t gap, int64_t *score);

/*
 * Alignment results: t1/t2 are caller buffers of capacity n+m each; *tlen1 and
 * *tlen2 receive the trace lengths (they differ only through the NUL-byte quirk
 * of AlignPair's trace(), align_pair.go:202-207).  A zero length corresponds to
 * the reference's nil slice.
 */
int bo_align_global(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                    bo_aligner *a, int64_t indel,
                    int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen); /* align.go:76-145 */
int bo_align_local(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                   bo_aligner *a, int64_t indel,
                   int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen); /* align.go:152-249 */
int bo_align_pair(int mode, const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                  bo_aligner *a, int64_t indel,
                  int64_t *score, uint8_t *t1, size_t *tlen1, uint8_t *t2, size_t *tlen2); /* align_pair.go:3-246 */

/* lastColumnScores (align_linear.go:122-145): score-only two-column pass.
 * sc0/sc1 are caller buffers of n+1 entries: last and next-to-last columns. */
int bo_last_column_scores(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                          bo_aligner *a, int64_t indel_signed, int64_t *sc0, int64_t *sc1);

/* AlignGlobalLS exactly as written (align_linear.go:9-119), WIP in the
 * reference: returns BO_PANIC where the Go code indexes out of range. */
int bo_align_global_ls_as_written(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                                  bo_aligner *a, int64_t indel,
                                  int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen);

/* AlignGlobalAffine score recurrence in float64 exactly as written
 * (align_affine.go:36-104,139-140), without its debug output and without its
 * (broken) traceback. */
int bo_affine_global_score_f64(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                               bo_aligner *a, double gap_open, double gap_extend, double *score);

/*
 * Integer Gotoh in the four modes with a canonical traceback.  PARITY UNPINNED
 * beyond the global score: this is the build's own definition:
 * the three-level recurrences and tie priorities of align_affine.go:44-103
 * (extend beats open; close-g2 -> diagonal -> close-g1, strict >), with the
 * border and end-cell rules of align_pair.go:109-195 applied to the middle level.
 */
int bo_align_affine(int mode, const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                    bo_aligner *a, int64_t gap_open, int64_t gap_extend,
                    int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen);

/* Rescore an affine trace: a run of L gap columns on one side costs
 * open + (L-1)*extend (align_affine.go:51-57). */
int bo_affine_gap(const uint8_t *s, size_t ns, const uint8_t *t, size_t nt,
                  bo_aligner *a, int64_t gap_open, int64_t gap_extend, int64_t *score);

/*
 * Batch driver used by tests and by bench.py's cpu_baseline: aligns pairs
 * [0,n_pairs) given concatenated bytes + offsets, `threads` worker threads, one
 * pair at a time per thread.  kind: 0 AlignGlobal, 1 AlignLocal, 2 AlignPair,
 * 3 integer affine.  Traces land right where the C ABI of the product puts them:
 * pair p owns t1[s1_off[p]+s2_off[p] ...) of capacity n_p+m_p, left aligned.
 * Returns cells processed, or a negative code.
 */
int64_t bo_batch(int kind, int mode, const uint8_t *s1, const uint64_t *s1_off,
                 const uint8_t *s2, const uint64_t *s2_off, uint64_t n_pairs,
                 bo_aligner *a, int64_t gap_open, int64_t gap_extend, int threads,
                 int64_t *scores, uint8_t *t1, uint8_t *t2, uint64_t *tlen);

/*
 * ReadFASTA (fasta.go:55-84): the text is split at '\n', every line trimmed of white space at both ends
 * (bytes.TrimSpace; restated for ASCII white space: \t \n \v \f \r and space -- Go would also trim UTF-8
 * encoded Unicode spaces, which FASTA files do not hold), blank lines skipped, a line that starts with '>'
 * opens a new record (even if no data follows), other lines are appended to the last record.
 * Data before the first header is the error "No header": returns -1.
 * Outputs: seq = the records' data back to back (capacity: len), off[0..n] (capacity: lines + 2),
 * hdr_begin / hdr_len = every header line, trimmed, as a span of the text.  Returns the number of records.
 */
int64_t bo_read_fasta(const uint8_t *text, uint64_t len, uint8_t *seq, uint64_t *off, uint64_t *hdr_begin,
                      uint64_t *hdr_len);

#ifdef __cplusplus
}
#endif
#endif
