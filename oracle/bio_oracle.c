/*
 * bio_oracle.c -- CPU restatement of soniakeys/bio's pairwise alignment path.
 *
 * TEST INFRASTRUCTURE ONLY (see bio_oracle.h).  Deliberately keeps the
 * reference's data structures and costs: full (n+1)(m+1) score and backtrack
 * matrices allocated per call, 64-bit scores (Go int), and one indirect
 * Aligner call per interior cell.  All citations are relative to /root/reference.
 */
#include "bio_oracle.h"

#include <malloc.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* GapSymbol (align.go:3)                                              */
/* ------------------------------------------------------------------ */
static uint8_t g_gap_symbol = '-';
void bo_set_gap_symbol(uint8_t g) { g_gap_symbol = g; }
uint8_t bo_get_gap_symbol(void) { return g_gap_symbol; }

/* ------------------------------------------------------------------ */
/* Aligners                                                            */
/* ------------------------------------------------------------------ */

/* SubstMatrix.Score (subst_matrix.go:15-17): Matrix[Index[x]][Index[y]].
 * Index has max(alphabet)+1 zero-filled entries (subst_matrix.go:27-30), so a
 * byte <= max that is not in the alphabet aliases to symbol 0 and a byte > max
 * is an index-out-of-range panic. */
static int64_t subst_score(const bo_aligner *self, uint8_t x, uint8_t y)
{
    if ((int)x >= self->index_len || (int)y >= self->index_len) {
        ((bo_aligner *)self)->panicked = 1;
        return 0;
    }
    return self->matrix[self->index[x] * self->n_alpha + self->index[y]];
}

static int64_t match_score(const bo_aligner *self, uint8_t x, uint8_t y)
{
    return x == y ? self->match : self->mismatch;
}

static int64_t table_score(const bo_aligner *self, uint8_t x, uint8_t y)
{
    return self->table256[(int)x * 256 + (int)y];
}

bo_aligner *bo_new_subst_matrix(const uint8_t *alphabet, int n_alpha, const int32_t *matrix)
{
    bo_aligner *a = (bo_aligner *)calloc(1, sizeof *a);
    if (!a) return NULL;
    int max = 0;
    for (int i = 0; i < n_alpha; i++)
        if (alphabet[i] > max) max = alphabet[i];
    uint8_t *al = (uint8_t *)malloc(n_alpha > 0 ? n_alpha : 1);
    int32_t *mx = (int32_t *)malloc(sizeof(int32_t) * (n_alpha > 0 ? n_alpha * n_alpha : 1));
    a->index = (int32_t *)calloc(max + 1, sizeof(int32_t));
    if (!al || !mx || !a->index) { free(al); free(mx); free(a->index); free(a); return NULL; }
    memcpy(al, alphabet, n_alpha);
    memcpy(mx, matrix, sizeof(int32_t) * n_alpha * n_alpha);
    for (int i = 0; i < n_alpha; i++) a->index[alphabet[i]] = i;
    a->alphabet = al;
    a->matrix = mx;
    a->n_alpha = n_alpha;
    a->index_len = max + 1;
    a->score = subst_score;
    return a;
}

bo_aligner *bo_new_match_aligner(int64_t match, int64_t mismatch)
{
    bo_aligner *a = (bo_aligner *)calloc(1, sizeof *a);
    if (!a) return NULL;
    a->match = match;
    a->mismatch = mismatch;
    a->score = match_score;
    return a;
}

bo_aligner *bo_new_table_aligner(const int32_t *table256)
{
    bo_aligner *a = (bo_aligner *)calloc(1, sizeof *a);
    if (!a) return NULL;
    int32_t *t = (int32_t *)malloc(sizeof(int32_t) * 65536);
    if (!t) { free(a); return NULL; }
    memcpy(t, table256, sizeof(int32_t) * 65536);
    a->table256 = t;
    a->score = table_score;
    return a;
}

void bo_aligner_free(bo_aligner *a)
{
    if (!a) return;
    free((void *)a->alphabet);
    free((void *)a->matrix);
    free(a->index);
    free((void *)a->table256);
    free(a);
}

int64_t bo_score(bo_aligner *a, uint8_t x, uint8_t y) { return a->score(a, x, y); }

/* BLOSUM62 in the reference's alphabet order ARNDCQEGHILKMFPSTWYV (subst_matrix.go:64). */
static const char BLOSUM62_ALPHA[] = "ARNDCQEGHILKMFPSTWYV";
static const int32_t BLOSUM62_M[400] = {
     4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,
    -1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,
    -2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3,
    -2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3,
     0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,
    -1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2,
    -1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2,
     0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,
    -2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3,
    -1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,
    -1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,
    -1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2,
    -1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,
    -2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,
    -1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,
     1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2,
     0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,
    -3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,
    -2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,
     0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,
};

/* PAM250 in the reference's alphabet order ACDEFGHIKLMNPQRSTVWY (subst_matrix.go:90). */
static const char PAM250_ALPHA[] = "ACDEFGHIKLMNPQRSTVWY";
static const int32_t PAM250_M[400] = {
     2,-2, 0, 0,-3, 1,-1,-1,-1,-2,-1, 0, 1, 0,-2, 1, 1, 0,-6,-3,
    -2,12,-5,-5,-4,-3,-3,-2,-5,-6,-5,-4,-3,-5,-4, 0,-2,-2,-8, 0,
     0,-5, 4, 3,-6, 1, 1,-2, 0,-4,-3, 2,-1, 2,-1, 0, 0,-2,-7,-4,
     0,-5, 3, 4,-5, 0, 1,-2, 0,-3,-2, 1,-1, 2,-1, 0, 0,-2,-7,-4,
    -3,-4,-6,-5, 9,-5,-2, 1,-5, 2, 0,-3,-5,-5,-4,-3,-3,-1, 0, 7,
     1,-3, 1, 0,-5, 5,-2,-3,-2,-4,-3, 0, 0,-1,-3, 1, 0,-1,-7,-5,
    -1,-3, 1, 1,-2,-2, 6,-2, 0,-2,-2, 2, 0, 3, 2,-1,-1,-2,-3, 0,
    -1,-2,-2,-2, 1,-3,-2, 5,-2, 2, 2,-2,-2,-2,-2,-1, 0, 4,-5,-1,
    -1,-5, 0, 0,-5,-2, 0,-2, 5,-3, 0, 1,-1, 1, 3, 0, 0,-2,-3,-4,
    -2,-6,-4,-3, 2,-4,-2, 2,-3, 6, 4,-3,-3,-2,-3,-3,-2, 2,-2,-1,
    -1,-5,-3,-2, 0,-3,-2, 2, 0, 4, 6,-2,-2,-1, 0,-2,-1, 2,-4,-2,
     0,-4, 2, 1,-3, 0, 2,-2, 1,-3,-2, 2, 0, 1, 0, 1, 0,-2,-4,-2,
     1,-3,-1,-1,-5, 0, 0,-2,-1,-3,-2, 0, 6, 0, 0, 1, 0,-1,-6,-5,
     0,-5, 2, 2,-5,-1, 3,-2, 1,-2,-1, 1, 0, 4, 1,-1,-1,-2,-5,-4,
    -2,-4,-1,-1,-4,-3, 2,-2, 3,-3, 0, 0, 0, 1, 6, 0,-1,-2, 2,-4,
     1, 0, 0, 0,-3, 1,-1,-1, 0,-3,-2, 1, 1,-1, 0, 2, 1,-1,-2,-3,
     1,-2, 0, 0,-3, 0,-1, 0, 0,-2,-1, 0, 0,-1,-1, 1, 3, 0,-5,-3,
     0,-2,-2,-2,-1,-1,-2, 4,-2, 2, 2,-2,-1,-2,-2,-1, 0, 4,-6,-2,
    -6,-8,-7,-7, 0,-7,-3,-5,-3,-2,-4,-4,-6,-5, 2,-2,-5,-6,17, 0,
    -3, 0,-4,-4, 7,-5, 0,-1,-4,-1,-2,-2,-5,-4,-4,-3,-3,-2, 0,10,
};

bo_aligner *bo_blosum62(void)
{
    return bo_new_subst_matrix((const uint8_t *)BLOSUM62_ALPHA, 20, BLOSUM62_M);
}
bo_aligner *bo_pam250(void)
{
    return bo_new_subst_matrix((const uint8_t *)PAM250_ALPHA, 20, PAM250_M);
}

/* ------------------------------------------------------------------ */
/* LinearGap / ConstantGap (align.go:18-69)                            */
/* ------------------------------------------------------------------ */
int bo_linear_gap(const uint8_t *s, size_t ns, const uint8_t *t, size_t nt,
                  bo_aligner *a, int64_t gap, int64_t *score)
{
    if (ns != nt) return BO_PANIC; /* align.go:19-21 */
    int64_t sc = 0;
    for (size_t i = 0; i < ns; i++) {
        /* `switch ta := t[i]; GapSymbol { case sa, ta: ...` (align.go:23-28) */
        if (s[i] == g_gap_symbol || t[i] == g_gap_symbol)
            sc -= gap;
        else
            sc += a->score(a, s[i], t[i]);
    }
    *score = sc;
    return a->panicked ? BO_PANIC : BO_OK;
}

int bo_constant_gap(const uint8_t *s, size_t ns, const uint8_t *t, size_t nt,
                    bo_aligner *a, int64_t gap, int64_t *score)
{
    if (ns != nt) return BO_PANIC; /* align.go:42-44 */
    enum { NO_GAP, S_GAP, T_GAP } open = NO_GAP;
    int64_t sc = 0;
    for (size_t i = 0; i < ns; i++) {
        if (s[i] == g_gap_symbol) { /* case sa is tested first (align.go:53) */
            if (open != S_GAP) { sc -= gap; open = S_GAP; }
        } else if (t[i] == g_gap_symbol) {
            if (open != T_GAP) { sc -= gap; open = T_GAP; }
        } else {
            sc += a->score(a, s[i], t[i]);
            open = NO_GAP;
        }
    }
    *score = sc;
    return a->panicked ? BO_PANIC : BO_OK;
}

/* reverse the first `len` bytes in place */
static void reverse_bytes(uint8_t *p, size_t len)
{
    for (size_t i = 0; i < len / 2; i++) {
        uint8_t tmp = p[i];
        p[i] = p[len - 1 - i];
        p[len - 1 - i] = tmp;
    }
}

/* ------------------------------------------------------------------ */
/* AlignGlobal (align.go:76-145)                                       */
/* ------------------------------------------------------------------ */
int bo_align_global(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                    bo_aligner *a, int64_t indel,
                    int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen)
{
    size_t stride = m + 1;
    size_t total = (n + 1) * stride;
    int64_t *s = (int64_t *)calloc(total, sizeof(int64_t)); /* score matrix, align.go:79 */
    int64_t *b = (int64_t *)calloc(total, sizeof(int64_t)); /* backtrack deltas, align.go:81 */
    if (!s || !b) { free(s); free(b); return BO_NOMEM; }
    const int64_t g1 = 1, g2 = (int64_t)stride, mm = (int64_t)stride + 1; /* align.go:85-87 */
    size_t x;
    for (x = 1; x < stride; x++) { /* top row: gap in s1 all across, align.go:92-95 */
        s[x] = s[x - 1] - indel;
        b[x] = g1;
    }
    for (size_t i = 0; i < n; i++) {
        s[x] = s[x - stride] - indel; /* first column: gap in s2, align.go:99-100 */
        b[x] = g2;
        x++;
        for (size_t j = 0; j < m; j++) {
            int64_t smax = s[x - stride] - indel; /* gap in s2, align.go:105-106 */
            int64_t bmax = g2;
            int64_t s0 = s[x - 1] - indel; /* gap in s1, align.go:108-111 */
            if (s0 > smax) { smax = s0; bmax = g1; }
            s0 = s[x - stride - 1] + a->score(a, s1[i], s2[j]); /* align.go:113-116 */
            if (s0 > smax) { smax = s0; bmax = mm; }
            s[x] = smax;
            b[x] = bmax;
            x++;
        }
    }
    x = total - 1;
    *score = s[x]; /* align.go:123-124 */
    size_t k = 0;
    int rc = BO_OK;
    for (; x > 0; x -= (size_t)b[x]) { /* align.go:126-138 */
        if (b[x] == g1) {
            /* With an empty s2 the stride is 1, so g2 == g1 and this case
             * catches the first-column steps: s2[x%stride-1] is s2[-1], a Go
             * index-out-of-range panic (align.go:128-130). */
            if (x % stride == 0) { rc = BO_PANIC; break; }
            t1[k] = g_gap_symbol;
            t2[k] = s2[x % stride - 1];
        } else if (b[x] == g2) {
            t1[k] = s1[x / stride - 1];
            t2[k] = g_gap_symbol;
        } else {
            t1[k] = s1[x / stride - 1];
            t2[k] = s2[x % stride - 1];
        }
        k++;
    }
    reverse_bytes(t1, k);
    reverse_bytes(t2, k);
    *tlen = k;
    free(s);
    free(b);
    if (rc) return rc;
    return a->panicked ? BO_PANIC : BO_OK;
}

/* ------------------------------------------------------------------ */
/* AlignLocal (align.go:152-249)                                       */
/* ------------------------------------------------------------------ */
int bo_align_local(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                   bo_aligner *a, int64_t indel,
                   int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen)
{
    size_t stride = m + 1;
    size_t total = (n + 1) * stride;
    int64_t *s = (int64_t *)calloc(total, sizeof(int64_t));
    uint8_t *b = (uint8_t *)calloc(total, 1); /* align.go:157 */
    if (!s || !b) { free(s); free(b); return BO_NOMEM; }
    enum { SP = 'p', G1 = '1', G2 = '2', MM = 'm' }; /* align.go:159-164 */
    size_t x;
    for (x = 1; x < stride; x++) b[x] = SP; /* align.go:175-177 */
    for (size_t i = 0; i < n; i++) {
        b[x] = SP; /* align.go:181 */
        x++;
        for (size_t j = 0; j < m; j++) {
            int64_t smax = 0; /* free taxi ride, align.go:186-187 */
            uint8_t bmax = SP;
            int64_t s0 = s[x - stride] - indel; /* gap in s2, align.go:189-192 */
            if (s0 > smax) { smax = s0; bmax = G2; }
            s0 = s[x - 1] - indel; /* gap in s1, align.go:194-197 */
            if (s0 > smax) { smax = s0; bmax = G1; }
            s0 = s[x - stride - 1] + a->score(a, s1[i], s2[j]); /* align.go:199-202 */
            if (s0 > smax) { smax = s0; bmax = MM; }
            s[x] = smax;
            b[x] = bmax;
            x++;
        }
    }
    /* first strict maximum of the whole matrix in row-major order, starting
     * from score 0 at node 0 (align.go:214-221) */
    int64_t best = 0;
    x = 0;
    for (size_t xx = 0; xx < total; xx++)
        if (s[xx] > best) { best = s[xx]; x = xx; }
    *score = best;
    size_t k = 0;
    while (x > 0) { /* align.go:225-242 */
        uint8_t bx = b[x];
        if (bx == G1) {
            t1[k] = g_gap_symbol;
            t2[k] = s2[x % stride - 1];
            x -= 1;
        } else if (bx == G2) {
            t1[k] = s1[x / stride - 1];
            t2[k] = g_gap_symbol;
            x -= stride;
        } else if (bx == MM) {
            t1[k] = s1[x / stride - 1];
            t2[k] = s2[x % stride - 1];
            x -= stride + 1;
        } else {
            break;
        }
        k++;
    }
    reverse_bytes(t1, k);
    reverse_bytes(t2, k);
    *tlen = k;
    free(s);
    free(b);
    return a->panicked ? BO_PANIC : BO_OK;
}

/* ------------------------------------------------------------------ */
/* AlignPair (align_pair.go:3-246): the rule engine                    */
/* ------------------------------------------------------------------ */
typedef struct pair_aligner pair_aligner;
typedef int64_t (*rule_score_fn)(pair_aligner *pa, size_t x);
/* back: returns previous node and the two symbols to append (0 = nothing),
 * align_pair.go:23,102-107 */
typedef size_t (*rule_back_fn)(pair_aligner *pa, size_t x, uint8_t *b1, uint8_t *b2);
typedef struct { rule_score_fn score; rule_back_fn back; } pa_rule;

struct pair_aligner {
    const uint8_t *s1, *s2;
    size_t n, m;
    bo_aligner *a;
    int64_t indel;
    size_t stride, mmx; /* align_pair.go:33-34 */
    const pa_rule *top, *left;
    const pa_rule *interior[4];
    int n_interior;
    int64_t *s;          /* score matrix, align_pair.go:42 */
    rule_back_fn *b;     /* backtrack matrix of function values, align_pair.go:43 */
    size_t total;
    size_t x_last;
    int64_t score;
};

/* skip prefix, "free taxi ride" (align_pair.go:62-64) */
static int64_t sp_score(pair_aligner *pa, size_t x) { (void)pa; (void)x; return 0; }
static size_t sp_back(pair_aligner *pa, size_t x, uint8_t *b1, uint8_t *b2)
{
    (void)pa; (void)x;
    *b1 = 0; *b2 = 0;
    return 0;
}
/* gap in s2 (align_pair.go:67-72) */
static int64_t g2_score(pair_aligner *pa, size_t x) { return pa->s[x - pa->stride] - pa->indel; }
static size_t g2_back(pair_aligner *pa, size_t x, uint8_t *b1, uint8_t *b2)
{
    *b1 = pa->s1[x / pa->stride - 1];
    *b2 = g_gap_symbol;
    return x - pa->stride;
}
/* gap in s1 (align_pair.go:78-83) */
static int64_t g1_score(pair_aligner *pa, size_t x) { return pa->s[x - 1] - pa->indel; }
static size_t g1_back(pair_aligner *pa, size_t x, uint8_t *b1, uint8_t *b2)
{
    *b1 = g_gap_symbol;
    *b2 = pa->s2[x % pa->stride - 1];
    return x - 1;
}
/* match / mismatch (align_pair.go:89-96): recomputes i, j by divide and modulo */
static int64_t mm_score(pair_aligner *pa, size_t x)
{
    size_t i = x / pa->stride - 1;
    size_t j = x % pa->stride - 1;
    return pa->s[x - pa->mmx] + pa->a->score(pa->a, pa->s1[i], pa->s2[j]);
}
static size_t mm_back(pair_aligner *pa, size_t x, uint8_t *b1, uint8_t *b2)
{
    *b1 = pa->s1[x / pa->stride - 1];
    *b2 = pa->s2[x % pa->stride - 1];
    return x - pa->mmx;
}

static const pa_rule RULE_SP = { sp_score, sp_back };
static const pa_rule RULE_G2 = { g2_score, g2_back };
static const pa_rule RULE_G1 = { g1_score, g1_back };
static const pa_rule RULE_MM = { mm_score, mm_back };

/* final rule of global (align_pair.go:123-126) */
static void final_last_node(pair_aligner *pa)
{
    pa->x_last = pa->total - 1;
    pa->score = pa->s[pa->x_last];
}
/* final rule of local (align_pair.go:138-152) */
static void final_whole_matrix(pair_aligner *pa)
{
    int64_t smax = 0;
    size_t xmax = 0;
    for (size_t x = 0; x < pa->total; x++)
        if (pa->s[x] > smax) { smax = pa->s[x]; xmax = x; }
    pa->x_last = xmax;
    pa->score = smax;
}
/* skipS2Suffix, shared by fitting and overlap (align_pair.go:169-182) */
static void final_bottom_row(pair_aligner *pa)
{
    size_t xmax = pa->total - pa->stride;
    int64_t smax = pa->s[xmax];
    for (size_t x = xmax + 1; x < pa->total; x++)
        if (pa->s[x] > smax) { smax = pa->s[x]; xmax = x; }
    pa->x_last = xmax;
    pa->score = smax;
}

int bo_align_pair(int mode, const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                  bo_aligner *a, int64_t indel,
                  int64_t *score, uint8_t *t1, size_t *tlen1, uint8_t *t2, size_t *tlen2)
{
    pair_aligner pa;
    memset(&pa, 0, sizeof pa);
    void (*final)(pair_aligner *) = NULL;
    switch (mode) {
    case BO_GLOBAL: /* setGlobal, align_pair.go:109-127 */
        pa.top = &RULE_G1; pa.left = &RULE_G2;
        pa.interior[0] = &RULE_G2; pa.interior[1] = &RULE_G1; pa.interior[2] = &RULE_MM;
        pa.n_interior = 3;
        final = final_last_node;
        break;
    case BO_LOCAL: /* setLocal, align_pair.go:129-153 */
        pa.top = &RULE_SP; pa.left = &RULE_SP;
        pa.interior[0] = &RULE_SP; pa.interior[1] = &RULE_G2;
        pa.interior[2] = &RULE_G1; pa.interior[3] = &RULE_MM;
        pa.n_interior = 4;
        final = final_whole_matrix;
        break;
    case BO_FITTING: /* setFitting, align_pair.go:157-166 */
        pa.top = &RULE_SP; pa.left = &RULE_G2;
        pa.interior[0] = &RULE_G2; pa.interior[1] = &RULE_G1; pa.interior[2] = &RULE_MM;
        pa.n_interior = 3;
        final = final_bottom_row;
        break;
    case BO_OVERLAP: /* setOverlap, align_pair.go:186-195 */
        pa.top = &RULE_G1; pa.left = &RULE_SP;
        pa.interior[0] = &RULE_G2; pa.interior[1] = &RULE_G1; pa.interior[2] = &RULE_MM;
        pa.n_interior = 3;
        final = final_bottom_row;
        break;
    default: /* align_pair.go:14-15 */
        *score = -1;
        *tlen1 = *tlen2 = 0;
        return BO_BAD_MODE;
    }
    /* newPairAligner (align_pair.go:52-59) */
    pa.s1 = s1; pa.s2 = s2; pa.n = n; pa.m = m; pa.a = a; pa.indel = indel;
    pa.stride = m + 1;
    pa.mmx = pa.stride + 1;
    pa.total = (n + 1) * (m + 1);
    pa.s = (int64_t *)calloc(pa.total, sizeof(int64_t));
    pa.b = (rule_back_fn *)calloc(pa.total, sizeof(rule_back_fn));
    if (!pa.s || !pa.b) { free(pa.s); free(pa.b); return BO_NOMEM; }

    /* align() (align_pair.go:218-246) */
    size_t x = 1;
    for (size_t j = 0; j < m; j++) {
        pa.s[x] = pa.top->score(&pa, x);
        pa.b[x] = pa.top->back;
        x++;
    }
    for (size_t i = 0; i < n; i++) {
        pa.s[x] = pa.left->score(&pa, x);
        pa.b[x] = pa.left->back;
        x++;
        for (size_t j = 0; j < m; j++) {
            int64_t smax = pa.interior[0]->score(&pa, x);
            rule_back_fn bmax = pa.interior[0]->back;
            for (int r = 1; r < pa.n_interior; r++) {
                int64_t s0 = pa.interior[r]->score(&pa, x);
                if (s0 > smax) { smax = s0; bmax = pa.interior[r]->back; }
            }
            pa.s[x] = smax;
            pa.b[x] = bmax;
            x++;
        }
    }
    final(&pa);
    *score = pa.score;

    /* trace() (align_pair.go:197-216).  A symbol value 0 means "append
     * nothing" (align_pair.go:202-207), so a NUL byte inside a sequence is
     * silently dropped from its trace -- kept as is. */
    size_t k1 = 0, k2 = 0;
    x = pa.x_last;
    while (x > 0) {
        uint8_t b1, b2;
        size_t px = pa.b[x](&pa, x, &b1, &b2);
        if (b1 != 0) t1[k1++] = b1;
        if (b2 != 0) t2[k2++] = b2;
        x = px;
    }
    int rc = BO_OK;
    /* the reversal loop runs over len(t1)/2 and indexes both traces with
     * len(t1)-1 (align_pair.go:210-214) */
    if (k1 > 0) {
        size_t last = k1 - 1;
        for (size_t i = 0; i < k1 / 2; i++) {
            uint8_t tmp = t1[i]; t1[i] = t1[last - i]; t1[last - i] = tmp;
            if (last - i >= k2) { rc = BO_PANIC; break; }
            tmp = t2[i]; t2[i] = t2[last - i]; t2[last - i] = tmp;
        }
    }
    *tlen1 = k1;
    *tlen2 = k2;
    free(pa.s);
    free(pa.b);
    if (a->panicked) rc = BO_PANIC;
    return rc;
}

/* ------------------------------------------------------------------ */
/* lastColumnScores (align_linear.go:122-145)                          */
/* ------------------------------------------------------------------ */
int bo_last_column_scores(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                          bo_aligner *a, int64_t indel, int64_t *out0, int64_t *out1)
{
    int64_t *sc0 = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    int64_t *sc1 = (int64_t *)calloc(n + 1, sizeof(int64_t));
    if (!sc0 || !sc1) { free(sc0); free(sc1); return BO_NOMEM; }
    for (size_t i = 0; i <= n; i++) sc0[i] = (int64_t)i * indel; /* align_linear.go:125-127 */
    for (size_t j = 0; j < m; j++) {
        sc1[0] = sc0[0] + indel; /* align_linear.go:131 */
        for (size_t i = 0; i < n; i++) {
            int64_t max = sc1[i] + indel; /* "down", align_linear.go:133 */
            int64_t mmv = sc0[i] + a->score(a, s1[i], s2[j]); /* diagonal, 134-136 */
            if (mmv > max) max = mmv;
            int64_t g = sc0[i + 1] + indel; /* "across", 137-139 */
            if (g > max) max = g;
            sc1[i + 1] = max;
        }
        int64_t *tmp = sc0; sc0 = sc1; sc1 = tmp; /* align_linear.go:142 */
    }
    memcpy(out0, sc0, sizeof(int64_t) * (n + 1));
    if (out1) memcpy(out1, sc1, sizeof(int64_t) * (n + 1));
    free(sc0);
    free(sc1);
    return a->panicked ? BO_PANIC : BO_OK;
}

/* ------------------------------------------------------------------ */
/* AlignGlobalLS as written (align_linear.go:9-119)                    */
/* ------------------------------------------------------------------ */
typedef struct { size_t r, c; } m_pair;

typedef struct {
    bo_aligner *a;
    int64_t indel_penalty;
    int64_t score;
    uint8_t *a1, *a2;
    size_t k;
    int rc;
} ls_state;

/* middleNodes + middleEdge (align_linear.go:51-119).  Returns BO_PANIC where
 * the Go code would index rev[] at -1 (align_linear.go:77) or s1 out of range. */
static int ls_middle_edge(ls_state *st, const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                          m_pair *p1, m_pair *p2)
{
    int64_t indel = -st->indel_penalty; /* align_linear.go:62 */
    int rc = BO_OK;
    uint8_t *r1 = (uint8_t *)malloc(n ? n : 1), *r2 = (uint8_t *)malloc(m ? m : 1);
    int64_t *fwd = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    int64_t *rev = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    int64_t *rev1 = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    m_pair *mn = (m_pair *)malloc(sizeof(m_pair) * (n + 1));
    if (!r1 || !r2 || !fwd || !rev || !rev1 || !mn) { rc = BO_NOMEM; goto done; }
    for (size_t i = 0; i < n; i++) r1[i] = s1[n - 1 - i]; /* Seq.Reverse, seq.go:57-64 */
    for (size_t j = 0; j < m; j++) r2[j] = s2[m - 1 - j];
    size_t mc = m / 2; /* align_linear.go:100 */
    rc = bo_last_column_scores(s1, n, s2, mc, st->a, indel, fwd, NULL);
    if (rc) goto done;
    size_t rcols = m - mc;
    rc = bo_last_column_scores(r1, n, r2, rcols, st->a, indel, rev, rev1);
    if (rc) goto done;
    size_t last = n; /* len(rev)-1 */
    size_t nmn = 0;
    mn[nmn++] = (m_pair){ 0, mc }; /* align_linear.go:107-117 */
    int64_t max = fwd[0] + rev[last];
    for (size_t i = 1; i <= n; i++) {
        int64_t s = fwd[i] + rev[last - i];
        if (s > max) { max = s; nmn = 0; mn[nmn++] = (m_pair){ i, mc }; }
        else if (s == max) mn[nmn++] = (m_pair){ i, mc };
    }
    /* middleEdge's scan: `set` is never assigned, so the first branch always
     * fires and the last tied node wins (align_linear.go:66-93) */
    int64_t mx = 0;
    for (size_t q = 0; q < nmn; q++) {
        m_pair p = mn[q];
        int64_t g1 = rev1[last - p.r] + indel;
        mx = g1; *p1 = p; *p2 = (m_pair){ p.r, p.c + 1 };
        if (p.r + 1 > last) { rc = BO_PANIC; goto done; } /* rev[-1], align_linear.go:77 */
        int64_t g2 = rev[last - (p.r + 1)] + indel;
        if (g2 > mx) { mx = g2; *p1 = p; *p2 = (m_pair){ p.r + 1, p.c }; }
        int64_t mmv = rev1[last - (p.r + 1)] + st->a->score(st->a, s1[p.r], s2[p.c]);
        if (mmv > mx) { mx = mmv; *p1 = p; *p2 = (m_pair){ p.r + 1, p.c + 1 }; }
    }
done:
    free(r1); free(r2); free(fwd); free(rev); free(rev1); free(mn);
    return rc;
}

static void ls_recurse(ls_state *st, const uint8_t *s1, size_t n, const uint8_t *s2, size_t m)
{
    if (st->rc) return;
    int64_t indel = -st->indel_penalty;
    if (n == 0) { /* align_linear.go:13-18 */
        st->score += indel * (int64_t)m;
        for (size_t j = 0; j < m; j++) { st->a1[st->k] = g_gap_symbol; st->a2[st->k] = s2[j]; st->k++; }
        return;
    }
    if (m == 0) { /* align_linear.go:19-24 */
        st->score += indel * (int64_t)n;
        for (size_t i = 0; i < n; i++) { st->a1[st->k] = s1[i]; st->a2[st->k] = g_gap_symbol; st->k++; }
        return;
    }
    m_pair p1 = { 0, 0 }, p2 = { 0, 0 };
    st->rc = ls_middle_edge(st, s1, n, s2, m, &p1, &p2);
    if (st->rc) return;
    ls_recurse(st, s1, p1.r, s2, p1.c);
    if (st->rc) return;
    if (p1.r == p2.r) { /* align_linear.go:28-31 */
        st->score += indel;
        st->a1[st->k] = g_gap_symbol; st->a2[st->k] = s2[p1.c]; st->k++;
    } else if (p1.c == p2.c) { /* 32-35 */
        st->score += indel;
        st->a1[st->k] = s1[p1.r]; st->a2[st->k] = g_gap_symbol; st->k++;
    } else { /* 36-39 */
        st->score += st->a->score(st->a, s1[p1.r], s2[p1.c]);
        st->a1[st->k] = s1[p1.r]; st->a2[st->k] = s2[p1.c]; st->k++;
    }
    ls_recurse(st, s1 + p2.r, n - p2.r, s2 + p2.c, m - p2.c);
}

int bo_align_global_ls_as_written(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                                  bo_aligner *a, int64_t indel,
                                  int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen)
{
    ls_state st = { a, indel, 0, t1, t2, 0, BO_OK };
    ls_recurse(&st, s1, n, s2, m);
    *score = st.score;
    *tlen = st.k;
    if (a->panicked) return BO_PANIC;
    return st.rc;
}

/* ------------------------------------------------------------------ */
/* AlignGlobalAffine score recurrence, float64 (align_affine.go:36-104) */
/* ------------------------------------------------------------------ */
int bo_affine_global_score_f64(const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                               bo_aligner *a, double gap_open, double gap_extend, double *score)
{
    size_t stride = m + 1, total = (n + 1) * stride;
    double *sg1 = (double *)calloc(total, sizeof(double));
    double *smm = (double *)calloc(total, sizeof(double));
    double *sg2 = (double *)calloc(total, sizeof(double));
    if (!sg1 || !smm || !sg2) { free(sg1); free(smm); free(sg2); return BO_NOMEM; }
    sg1[0] = -INFINITY; /* align_affine.go:36-37 */
    sg2[0] = -INFINITY;
    for (size_t x = 1; x < total; x++) {
        double smax;
        if (x % stride == 0) {
            sg1[x] = -INFINITY; /* align_affine.go:45-46 */
        } else {
            smax = sg1[x - 1] - gap_extend; /* extend first, 51 */
            double s0 = smm[x - 1] - gap_open; /* open only if strictly better, 54 */
            if (s0 > smax) smax = s0;
            sg1[x] = smax;
        }
        if (x < stride) {
            sg2[x] = -INFINITY; /* 63-64 */
        } else {
            smax = sg2[x - stride] - gap_extend; /* 69 */
            double s0 = smm[x - stride] - gap_open; /* 72 */
            if (s0 > smax) smax = s0;
            sg2[x] = smax;
        }
        smax = sg2[x]; /* close gap in s2 first, 82 */
        size_t i = x / stride, j = x % stride;
        if (i > 0 && j > 0) {
            double s0 = smm[x - stride - 1] + (double)a->score(a, s1[i - 1], s2[j - 1]); /* 90 */
            if (s0 > smax) smax = s0;
        }
        if (sg1[x] > smax) smax = sg1[x]; /* close gap in s1, 97 */
        smm[x] = smax;
    }
    *score = smm[total - 1]; /* 139-140 */
    free(sg1); free(smm); free(sg2);
    return a->panicked ? BO_PANIC : BO_OK;
}

/* ------------------------------------------------------------------ */
/* Integer Gotoh, four modes, canonical traceback (PARITY UNPINNED)    */
/* ------------------------------------------------------------------ */
#define BO_NEG (INT64_MIN / 4)
/* -inf - p stays -inf, as in the reference's float64 arithmetic */
static inline int64_t neg_sub(int64_t v, int64_t p) { return v == BO_NEG ? BO_NEG : v - p; }

enum { LV_SP = 0, LV_G2 = 1, LV_G1 = 2, LV_MM = 3 };

int bo_align_affine(int mode, const uint8_t *s1, size_t n, const uint8_t *s2, size_t m,
                    bo_aligner *a, int64_t gap_open, int64_t gap_extend,
                    int64_t *score, uint8_t *t1, uint8_t *t2, size_t *tlen)
{
    if (mode < BO_GLOBAL || mode > BO_OVERLAP) { *score = -1; *tlen = 0; return BO_BAD_MODE; }
    size_t stride = m + 1, total = (n + 1) * stride;
    int64_t *sg1 = (int64_t *)calloc(total, sizeof(int64_t));
    int64_t *smm = (int64_t *)calloc(total, sizeof(int64_t));
    int64_t *sg2 = (int64_t *)calloc(total, sizeof(int64_t));
    /* per node: bmm (2 bits: LV_*), bg1 (1 = opened from middle level), bg2 (same) */
    uint8_t *bmm = (uint8_t *)calloc(total, 1);
    uint8_t *bg1 = (uint8_t *)calloc(total, 1);
    uint8_t *bg2 = (uint8_t *)calloc(total, 1);
    if (!sg1 || !smm || !sg2 || !bmm || !bg1 || !bg2) {
        free(sg1); free(smm); free(sg2); free(bmm); free(bg1); free(bg2);
        return BO_NOMEM;
    }
    /* which borders of the middle level are "skip prefix" (score 0, stop):
     * align_pair.go:130-131 (local), 158 (fitting top), 188 (overlap left) */
    int top_free = (mode == BO_LOCAL || mode == BO_FITTING);
    int left_free = (mode == BO_LOCAL || mode == BO_OVERLAP);
    sg1[0] = BO_NEG; /* align_affine.go:36-37 */
    sg2[0] = BO_NEG;
    bmm[0] = LV_SP;
    for (size_t x = 1; x < total; x++) {
        size_t i = x / stride, j = x % stride;
        if (j == 0) {
            sg1[x] = BO_NEG; /* align_affine.go:45-46 */
        } else {
            int64_t smax = neg_sub(sg1[x - 1], gap_extend); /* extend wins ties, 51-57 */
            uint8_t opened = 0;
            int64_t s0 = smm[x - 1] - gap_open;
            if (s0 > smax) { smax = s0; opened = 1; }
            sg1[x] = smax;
            bg1[x] = opened;
        }
        if (i == 0) {
            sg2[x] = BO_NEG; /* 63-64 */
        } else {
            int64_t smax = neg_sub(sg2[x - stride], gap_extend); /* 67-75 */
            uint8_t opened = 0;
            int64_t s0 = smm[x - stride] - gap_open;
            if (s0 > smax) { smax = s0; opened = 1; }
            sg2[x] = smax;
            bg2[x] = opened;
        }
        if ((i == 0 && top_free) || (j == 0 && left_free)) {
            smm[x] = 0; /* spRule border */
            bmm[x] = LV_SP;
            continue;
        }
        int64_t smax;
        uint8_t bmax;
        if (mode == BO_LOCAL) { /* sp first, then the three below, align_pair.go:132-137 */
            smax = 0; bmax = LV_SP;
            if (sg2[x] > smax) { smax = sg2[x]; bmax = LV_G2; }
        } else {
            smax = sg2[x]; bmax = LV_G2; /* close gap in s2, align_affine.go:82-83 */
        }
        if (i > 0 && j > 0) { /* diagonal, 87-94 */
            int64_t s0 = smm[x - stride - 1] + a->score(a, s1[i - 1], s2[j - 1]);
            if (s0 > smax) { smax = s0; bmax = LV_MM; }
        }
        if (sg1[x] > smax) { smax = sg1[x]; bmax = LV_G1; } /* close gap in s1, 97-100 */
        smm[x] = smax;
        bmm[x] = bmax;
    }
    /* end node: align_pair.go:123-126, 138-152, 169-182 on the middle level */
    size_t xl;
    int64_t best;
    if (mode == BO_GLOBAL) {
        xl = total - 1; best = smm[xl];
    } else if (mode == BO_LOCAL) {
        xl = 0; best = 0;
        for (size_t x = 0; x < total; x++)
            if (smm[x] > best) { best = smm[x]; xl = x; }
    } else {
        xl = total - stride; best = smm[xl];
        for (size_t x = xl + 1; x < total; x++)
            if (smm[x] > best) { best = smm[x]; xl = x; }
    }
    *score = best;
    /* canonical traceback over (level, node) */
    size_t k = 0, x = xl;
    int level = LV_MM;
    while (x > 0) {
        size_t i = x / stride, j = x % stride;
        if (level == LV_MM) {
            uint8_t bx = bmm[x];
            if (bx == LV_SP) break;
            if (bx == LV_MM) {
                t1[k] = s1[i - 1]; t2[k] = s2[j - 1]; k++;
                x -= stride + 1;
            } else {
                level = bx; /* close a gap: switch level at the same node */
            }
        } else if (level == LV_G1) {
            t1[k] = g_gap_symbol; t2[k] = s2[j - 1]; k++;
            level = bg1[x] ? LV_MM : LV_G1;
            x -= 1;
        } else {
            t1[k] = s1[i - 1]; t2[k] = g_gap_symbol; k++;
            level = bg2[x] ? LV_MM : LV_G2;
            x -= stride;
        }
    }
    reverse_bytes(t1, k);
    reverse_bytes(t2, k);
    *tlen = k;
    free(sg1); free(smm); free(sg2); free(bmm); free(bg1); free(bg2);
    return a->panicked ? BO_PANIC : BO_OK;
}

int bo_affine_gap(const uint8_t *s, size_t ns, const uint8_t *t, size_t nt,
                  bo_aligner *a, int64_t gap_open, int64_t gap_extend, int64_t *score)
{
    if (ns != nt) return BO_PANIC;
    enum { NO_GAP, S_GAP, T_GAP } open = NO_GAP;
    int64_t sc = 0;
    for (size_t i = 0; i < ns; i++) {
        if (s[i] == g_gap_symbol) {
            sc -= (open == S_GAP) ? gap_extend : gap_open;
            open = S_GAP;
        } else if (t[i] == g_gap_symbol) {
            sc -= (open == T_GAP) ? gap_extend : gap_open;
            open = T_GAP;
        } else {
            sc += a->score(a, s[i], t[i]);
            open = NO_GAP;
        }
    }
    *score = sc;
    return a->panicked ? BO_PANIC : BO_OK;
}

/* ------------------------------------------------------------------ */
/* Batch driver (tests + cpu_baseline)                                 */
/* ------------------------------------------------------------------ */
typedef struct {
    int kind, mode;
    const uint8_t *s1, *s2;
    const uint64_t *s1_off, *s2_off;
    uint64_t n_pairs;
    bo_aligner a; /* private copy: `panicked` is per thread */
    int64_t gap_open, gap_extend;
    int64_t *scores;
    uint8_t *t1, *t2;
    uint64_t *tlen;
    uint64_t *next; /* shared work counter */
    pthread_mutex_t *mu;
    int64_t cells;
    int rc;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *jb = (batch_job *)arg;
    const uint64_t CHUNK = 16;
    for (;;) {
        pthread_mutex_lock(jb->mu);
        uint64_t lo = *jb->next;
        *jb->next = lo + CHUNK;
        pthread_mutex_unlock(jb->mu);
        if (lo >= jb->n_pairs) break;
        uint64_t hi = lo + CHUNK < jb->n_pairs ? lo + CHUNK : jb->n_pairs;
        for (uint64_t p = lo; p < hi; p++) {
            const uint8_t *a1 = jb->s1 + jb->s1_off[p];
            const uint8_t *a2 = jb->s2 + jb->s2_off[p];
            size_t n = (size_t)(jb->s1_off[p + 1] - jb->s1_off[p]);
            size_t m = (size_t)(jb->s2_off[p + 1] - jb->s2_off[p]);
            uint64_t slot = jb->s1_off[p] + jb->s2_off[p];
            uint8_t *o1 = jb->t1 + slot, *o2 = jb->t2 + slot;
            size_t l1 = 0, l2 = 0;
            int64_t sc = 0;
            int rc;
            switch (jb->kind) {
            case 0: rc = bo_align_global(a1, n, a2, m, &jb->a, jb->gap_open, &sc, o1, o2, &l1); break;
            case 1: rc = bo_align_local(a1, n, a2, m, &jb->a, jb->gap_open, &sc, o1, o2, &l1); break;
            case 2: rc = bo_align_pair(jb->mode, a1, n, a2, m, &jb->a, jb->gap_open, &sc, o1, &l1, o2, &l2); break;
            default: rc = bo_align_affine(jb->mode, a1, n, a2, m, &jb->a, jb->gap_open, jb->gap_extend, &sc, o1, o2, &l1); break;
            }
            if (rc && !jb->rc) jb->rc = rc;
            jb->scores[p] = sc;
            jb->tlen[p] = l1;
            jb->cells += (int64_t)n * (int64_t)m;
        }
    }
    return NULL;
}

int64_t bo_batch(int kind, int mode, const uint8_t *s1, const uint64_t *s1_off,
                 const uint8_t *s2, const uint64_t *s2_off, uint64_t n_pairs,
                 bo_aligner *a, int64_t gap_open, int64_t gap_extend, int threads,
                 int64_t *scores, uint8_t *t1, uint8_t *t2, uint64_t *tlen)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    /* keep the per-call matrices on the per-thread heaps: with glibc's default
     * 128 KiB mmap threshold every call would mmap/munmap and the threads would
     * serialise in the kernel (Go's allocator does not behave like that) */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    pthread_t th[256];
    batch_job *jobs = (batch_job *)calloc(threads, sizeof(batch_job));
    if (!jobs) return BO_NOMEM;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    uint64_t next = 0;
    for (int t = 0; t < threads; t++) {
        batch_job *jb = &jobs[t];
        jb->kind = kind; jb->mode = mode;
        jb->s1 = s1; jb->s2 = s2; jb->s1_off = s1_off; jb->s2_off = s2_off;
        jb->n_pairs = n_pairs;
        jb->a = *a;
        jb->gap_open = gap_open; jb->gap_extend = gap_extend;
        jb->scores = scores; jb->t1 = t1; jb->t2 = t2; jb->tlen = tlen;
        jb->next = &next; jb->mu = &mu;
    }
    for (int t = 1; t < threads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    batch_worker(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    int64_t cells = 0;
    int rc = 0;
    for (int t = 0; t < threads; t++) {
        cells += jobs[t].cells;
        if (jobs[t].rc && !rc) rc = jobs[t].rc;
        if (jobs[t].a.panicked) { a->panicked = 1; if (!rc) rc = BO_PANIC; }
    }
    free(jobs);
    return rc ? rc : cells;
}

/* ---------------------------------------------------------------------------
 * ReadFASTA, fasta.go:55-84
 * ------------------------------------------------------------------------- */
static int bo_is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
/* bytes.TrimSpace (fasta.go:63; Go standard library, bytes.TrimSpace -> unicode.IsSpace) trims UTF-8 encoded Unicode
 * white space as well: U+0085, U+00A0 and category Z -- U+1680, U+2000..U+200A, U+2028, U+2029, U+202F, U+205F,
 * U+3000.  A byte sequence that is not one of these encodings (or is cut short) decodes to something that is not
 * white space and ends the trimming. */
static int bo_space3(uint8_t a, uint8_t b, uint8_t c)
{
    if (a == 0xE1) return b == 0x9A && c == 0x80;
    if (a == 0xE2) return (b == 0x80 && (c <= 0x8A ? c >= 0x80 : (c == 0xA8 || c == 0xA9 || c == 0xAF))) || (b == 0x81 && c == 0x9F);
    if (a == 0xE3) return b == 0x80 && c == 0x80;
    return 0;
}
/* bytes of the white-space rune that starts at text[b] (line is [b, t)); 0: none */
static int bo_space_at(const uint8_t *text, uint64_t b, uint64_t t)
{
    uint8_t c = text[b];
    if (bo_is_space(c)) return 1;
    if (c < 0x80) return 0;
    if (c == 0xC2 && b + 1 < t && (text[b + 1] == 0x85 || text[b + 1] == 0xA0)) return 2;
    if (b + 2 < t && bo_space3(c, text[b + 1], text[b + 2])) return 3;
    return 0;
}
/* bytes of the white-space rune that ends at text[t - 1] */
static int bo_space_before(const uint8_t *text, uint64_t b, uint64_t t)
{
    uint8_t c = text[t - 1];
    if (bo_is_space(c)) return 1;
    if (c < 0x80) return 0;
    if (t - b >= 2 && text[t - 2] == 0xC2 && (c == 0x85 || c == 0xA0)) return 2;
    if (t - b >= 3 && bo_space3(text[t - 3], text[t - 2], c)) return 3;
    return 0;
}

int64_t bo_read_fasta(const uint8_t *text, uint64_t len, uint8_t *seq, uint64_t *off, uint64_t *hdr_begin,
                      uint64_t *hdr_len)
{
    int64_t nrec = 0;
    uint64_t out = 0, pos = 0;
    int found_header = 0;
    while (pos <= len) {                       /* bytes.Split yields a last, possibly empty, piece (fasta.go:62) */
        uint64_t e = pos;
        while (e < len && text[e] != '\n') ++e;
        uint64_t b = pos, t = e;               /* bytes.TrimSpace (fasta.go:63) */
        for (int k; b < t && (k = bo_space_at(text, b, t)) != 0;) b += k;
        for (int k; t > b && (k = bo_space_before(text, b, t)) != 0;) t -= k;
        if (t > b) {                           /* blank lines are skipped (fasta.go:65-66) */
            if (text[b] == '>') {              /* new record (fasta.go:67-71) */
                found_header = 1;
                hdr_begin[nrec] = b;
                hdr_len[nrec] = t - b;
                off[nrec] = out;
                ++nrec;
            } else {
                if (!found_header) return -1;  /* "No header" (fasta.go:72-73) */
                memcpy(seq + out, text + b, t - b);  /* fasta.go:75 */
                out += t - b;
            }
        }
        pos = e + 1;
    }
    off[nrec] = out;
    return nrec;
}
