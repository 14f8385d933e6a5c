// bio.hpp -- C++ host mirror of soniakeys/bio's alignment API over the C ABI of bioalign.h.
//
// The reference is compiled code (Go); this image has no Go toolchain, so the host side a Go
// maintainer would write with cgo is mirrored here in C++: same names, same
// argument meaning, same error behaviour (a Go panic is a C++ exception, a nil slice an empty
// string).  Everything is computed by the HIP library; nothing here aligns on the CPU.
//
//   reference                                   here
//   type Seq []byte              (seq.go:18)    bio::Seq = std::string (bytes)
//   var GapSymbol byte           (align.go:3)   bio::GapSymbol
//   type Aligner interface       (align.go:5-8) bio::Aligner
//   SubstMatrix, NewSubstMatrix  (subst_matrix.go:8-32)
//   Blosum62, Pam250             (subst_matrix.go:34-61)
//   LinearGap, ConstantGap       (align.go:18-69)
//   AlignGlobal, AlignLocal      (align.go:76, 152)
//   AlignPair                    (align_pair.go:3)
//   AlignGlobalLS                (align_linear.go:9)
//   AlignGlobalAffine            (align_affine.go:12)
//   + AlignPairBatch, AlignAffineBatch, AlignPairLong (additive), AlignLongOverlap (align_long.go:17 on
//     overlap alignments)
#pragma once
#include <array>
#include <functional>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "bioalign.h"

namespace bio {

using Seq = std::string;
inline uint8_t GapSymbol = '-';  // align.go:3, read at call time

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("bioalign error " + std::to_string(c) + ": " + m), code(c) {}
};

struct Aligner {  // align.go:5-8
    virtual ~Aligner() = default;
    virtual int Score(uint8_t x, uint8_t y) const = 0;
};

struct SubstMatrix : Aligner {  // subst_matrix.go:8-17
    std::string Alphabet;
    std::vector<std::vector<int>> Matrix;
    std::vector<int> Index;  // zero filled up to the largest alphabet byte: smaller foreign bytes alias to symbol 0
    int Score(uint8_t x, uint8_t y) const override { return Matrix.at(Index.at(x)).at(Index.at(y)); }  // .at: the Go panic
};

inline SubstMatrix NewSubstMatrix(const std::string &alphabet, std::vector<std::vector<int>> matrix)
{  // subst_matrix.go:20-32
    SubstMatrix m;
    m.Alphabet = alphabet;
    m.Matrix = std::move(matrix);
    int mx = 0;
    for (unsigned char b : alphabet) mx = b > mx ? b : mx;
    m.Index.assign(alphabet.empty() ? 1 : mx + 1, 0);
    for (size_t i = 0; i < alphabet.size(); ++i) m.Index[(unsigned char)alphabet[i]] = (int)i;
    return m;
}

namespace detail {
#include "bio_matrices.inc"
inline SubstMatrix builtin(const char *alphabet, const int (&t)[20][20])
{
    std::vector<std::vector<int>> m(20, std::vector<int>(20));
    for (int i = 0; i < 20; ++i)
        for (int j = 0; j < 20; ++j) m[i][j] = t[i][j];
    return NewSubstMatrix(alphabet, std::move(m));
}
}  // namespace detail

inline const SubstMatrix &Blosum62()
{
    static const SubstMatrix m = detail::builtin(detail::kBlosum62Alphabet, detail::kBlosum62);
    return m;
}
inline const SubstMatrix &Pam250()
{
    static const SubstMatrix m = detail::builtin(detail::kPam250Alphabet, detail::kPam250);
    return m;
}

// match / mismatch scorer, as the reference's tests define locally (align_fitting_test.go:9-16)
struct MatchAligner : Aligner {
    int match, mismatch;
    MatchAligner(int ma, int mi) : match(ma), mismatch(mi) {}
    int Score(uint8_t x, uint8_t y) const override { return x == y ? match : mismatch; }
};

inline int LinearGap(const Seq &s, const Seq &t, const Aligner &a, int gapPenalty)
{  // align.go:18-31
    if (s.size() != t.size()) throw std::runtime_error("Sequences have different lengths");
    int score = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const uint8_t sa = (uint8_t)s[i], ta = (uint8_t)t[i];
        if (sa == GapSymbol || ta == GapSymbol) score -= gapPenalty;
        else score += a.Score(sa, ta);
    }
    return score;
}

inline int ConstantGap(const Seq &s, const Seq &t, const Aligner &a, int gapPenalty)
{  // align.go:41-69
    if (s.size() != t.size()) throw std::runtime_error("Sequences have different lengths");
    int score = 0, open = 0;  // 0 none, 1 gap in s, 2 gap in t
    for (size_t i = 0; i < s.size(); ++i) {
        const uint8_t sa = (uint8_t)s[i], ta = (uint8_t)t[i];
        if (sa == GapSymbol) {
            if (open != 1) score -= gapPenalty, open = 1;
        } else if (ta == GapSymbol) {
            if (open != 2) score -= gapPenalty, open = 2;
        } else {
            score += a.Score(sa, ta);
            open = 0;
        }
    }
    return score;
}

struct Alignment {  // (score, t1, t2); empty strings are the reference's nil slices
    long long score = 0;
    Seq t1, t2;
    bool operator==(const Alignment &o) const { return score == o.score && t1 == o.t1 && t2 == o.t2; }
};

namespace detail {
inline bioalign_ctx *engine()
{
    static bioalign_ctx *ctx = bioalign_create(nullptr, 0);
    if (!ctx) throw Error(BIOALIGN_ENODEVICE, bioalign_last_error(nullptr));  // no CPU fallback
    return ctx;
}
inline int mode_of(const std::string &mode)
{
    if (mode == "global") return BIOALIGN_GLOBAL;
    if (mode == "local") return BIOALIGN_LOCAL;
    if (mode == "fitting") return BIOALIGN_FITTING;
    if (mode == "overlap") return BIOALIGN_OVERLAP;
    return -1;
}
// Score is called once per (s1 byte, s2 byte) pair that occurs; an exception propagates
inline std::vector<Alignment> run(const std::string &mode, const std::vector<Seq> &s1s, const std::vector<Seq> &s2s,
                                  const Aligner &a, int gap_open, int gap_extend, bool longmode)
{
    if (s1s.size() != s2s.size()) throw std::invalid_argument("s1s and s2s must have the same length");
    const size_t n = s1s.size();
    std::vector<Alignment> res(n);
    const int m = mode_of(mode);
    if (m < 0) {  // align_pair.go:14-15
        for (auto &r : res) r.score = -1;
        return res;
    }
    if (n == 0) return res;
    std::vector<uint64_t> o1(n + 1, 0), o2(n + 1, 0);
    std::string c1, c2;
    std::array<bool, 256> seen1{}, seen2{};
    for (size_t p = 0; p < n; ++p) {
        c1 += s1s[p];
        c2 += s2s[p];
        o1[p + 1] = c1.size();
        o2[p + 1] = c2.size();
        for (unsigned char b : s1s[p]) seen1[b] = true;
        for (unsigned char b : s2s[p]) seen2[b] = true;
    }
    std::array<uint8_t, 256> code1{}, code2{};
    std::vector<uint8_t> b1, b2;
    for (int b = 0; b < 256; ++b) {
        if (seen1[b]) code1[b] = (uint8_t)b1.size(), b1.push_back((uint8_t)b);
        if (seen2[b]) code2[b] = (uint8_t)b2.size(), b2.push_back((uint8_t)b);
    }
    const int na = b1.empty() ? 1 : (int)b1.size(), nb = b2.empty() ? 1 : (int)b2.size();
    std::vector<int32_t> table((size_t)na * nb, 0);
    for (size_t i = 0; i < b1.size(); ++i)
        for (size_t j = 0; j < b2.size(); ++j) table[i * nb + j] = a.Score(b1[i], b2[j]);
    const size_t cap = c1.size() + c2.size();
    std::vector<int64_t> sc(n);
    std::vector<uint8_t> t1(cap + 1), t2(cap + 1);
    std::vector<uint64_t> toff(n), tlen(n);
    c1.push_back(0);
    c2.push_back(0);
    bioalign_batch_t in{};
    in.s1 = (const uint8_t *)c1.data();
    in.s1_off = o1.data();
    in.s2 = (const uint8_t *)c2.data();
    in.s2_off = o2.data();
    in.n_pairs = n;
    in.mode = m;
    in.score_table = table.data();
    in.na = na;
    in.nb = nb;
    in.code1 = code1.data();
    in.code2 = code2.data();
    in.gap_open = gap_open;
    in.gap_extend = gap_extend;
    in.gap_symbol = GapSymbol;
    bioalign_result_t out{sc.data(), t1.data(), t2.data(), toff.data(), tlen.data()};
    bioalign_ctx *ctx = engine();
    const int rc = longmode ? bioalign_long(ctx, &in, &out, nullptr) : bioalign_batch(ctx, &in, &out);
    if (rc != 0) throw Error(rc, bioalign_last_error(ctx));
    for (size_t p = 0; p < n; ++p) {
        res[p].score = sc[p];
        res[p].t1.assign((const char *)t1.data() + toff[p], tlen[p]);
        res[p].t2.assign((const char *)t2.data() + toff[p], tlen[p]);
    }
    return res;
}
// pairs of 2^24 cells or more go to the intra-sequence kernels (bioalign_long) when their column alphabet allows it
inline bool long_pair(const Seq &s1, const Seq &s2, int max_symbols)
{
    if ((unsigned long long)s1.size() * s2.size() < (1ull << 24)) return false;
    std::array<bool, 256> seen{};
    int distinct = 0;
    for (unsigned char b : s2) distinct += !seen[b], seen[b] = true;
    return distinct <= max_symbols;
}
// one pair; `try_long`: on the long kernels if they take the parameters, else on the batch kernels
inline Alignment run1(const std::string &mode, const Seq &s1, const Seq &s2, const Aligner &a, int go, int ge, bool try_long)
{
    if (try_long) {
        try {
            return run(mode, {s1}, {s2}, a, go, ge, true)[0];
        } catch (const Error &e) {
            if (e.code != BIOALIGN_EUNSUPPORTED) throw;
        }
    }
    return run(mode, {s1}, {s2}, a, go, ge, false)[0];
}
// AlignPair's trace() (align_pair.go:197-216): a symbol 0 is not appended to its trace, and the reversal loop runs over
// len(t1)/2 indexing both traces with len(t1)-1 -- reproduced on the aligned traces the kernels return, only when one
// of them holds a 0 byte (a NUL in a sequence, or a GapSymbol of 0)
inline void trace_quirk(Alignment &r)
{
    if (r.t1.find('\0') == std::string::npos && r.t2.find('\0') == std::string::npos) return;
    std::string r1, r2;
    for (auto it = r.t1.rbegin(); it != r.t1.rend(); ++it) if (*it) r1 += *it;
    for (auto it = r.t2.rbegin(); it != r.t2.rend(); ++it) if (*it) r2 += *it;
    const size_t last = r1.size() - 1;
    for (size_t i = 0; i < r1.size() / 2; ++i) {
        std::swap(r1[i], r1[last - i]);
        if (last - i >= r2.size()) throw std::out_of_range("index out of range (align_pair.go:213)");  // the reference panics
        std::swap(r2[i], r2[last - i]);
    }
    r.t1 = r1;
    r.t2 = r2;
}
}  // namespace detail

// additive batch entry points
inline std::vector<Alignment> AlignPairBatch(const std::string &mode, const std::vector<Seq> &s1s, const std::vector<Seq> &s2s,
                                             const Aligner &a, int indelPenalty)
{
    return detail::run(mode, s1s, s2s, a, indelPenalty, indelPenalty, false);
}
inline std::vector<Alignment> AlignAffineBatch(const std::string &mode, const std::vector<Seq> &s1s, const std::vector<Seq> &s2s,
                                               const Aligner &a, int gapOpenPenalty, int gapExtendPenalty)
{
    return detail::run(mode, s1s, s2s, a, gapOpenPenalty, gapExtendPenalty, false);
}

inline Alignment AlignPair(const std::string &mode, const Seq &s1, const Seq &s2, const Aligner &a, int indelPenalty)
{  // align_pair.go:3-21 (long pairs: the intra-sequence kernels, same result)
    Alignment r = detail::run1(mode, s1, s2, a, indelPenalty, indelPenalty, detail::long_pair(s1, s2, 100));
    detail::trace_quirk(r);
    return r;
}
inline Alignment AlignPairLong(const std::string &mode, const Seq &s1, const Seq &s2, const Aligner &a, int indelPenalty)
{  // AlignPair for one long pair: the intra-sequence wavefront kernels (bioalign_long)
    Alignment r = detail::run(mode, {s1}, {s2}, a, indelPenalty, indelPenalty, true)[0];
    detail::trace_quirk(r);
    return r;
}
inline Alignment AlignGlobal(const Seq &s1, const Seq &s2, const Aligner &a, int indelPenalty)
{  // align.go:76-145; with an empty s2 the reference indexes s2[-1] (align.go:128-130) and panics
    if (s2.empty() && !s1.empty()) throw std::out_of_range("index out of range [-1]");
    return detail::run1("global", s1, s2, a, indelPenalty, indelPenalty, detail::long_pair(s1, s2, 100));
}
inline Alignment AlignLocal(const Seq &s1, const Seq &s2, const Aligner &a, int indelPenalty)
{  // align.go:152-249
    return detail::run1("local", s1, s2, a, indelPenalty, indelPenalty, detail::long_pair(s1, s2, 100));
}
inline Alignment AlignGlobalLS(const Seq &s1, const Seq &s2, const Aligner &a, int indelPenalty)
{  // align_linear.go:9-45: AlignGlobal's result (align_test.go:55-66); long inputs take the long-mode kernels
    return detail::run1("global", s1, s2, a, indelPenalty, indelPenalty, detail::long_pair(s1, s2, 100));
}
inline std::tuple<double, Seq, Seq> AlignGlobalAffine(const Seq &s1, const Seq &s2, const Aligner &a, double gapOpenPenalty,
                                                      double gapExtendPenalty)
{  // align_affine.go:12-167: float64 penalties and score, integer arithmetic underneath.  Fractional penalties are not
   // truncated: multiples of 2^-e (e <= 6) run in units of 2^-e -- scores and penalties times 2^e, exact integer Gotoh,
   // score divided back -- which is what the reference's float64 sums compute exactly; any other fraction is refused
   // (there the reference's result depends on float64 rounding along the path).
    int scale = 0;
    for (int e = 0; e <= 6 && !scale; ++e) {
        const double f = (double)(1 << e);
        if (gapOpenPenalty * f == (double)(long long)(gapOpenPenalty * f) && gapExtendPenalty * f == (double)(long long)(gapExtendPenalty * f))
            scale = 1 << e;
    }
    if (!scale) throw std::invalid_argument("AlignGlobalAffine on the GPU path needs penalties that are multiples of 1/64");
    struct Scaled : Aligner {
        const Aligner &a;
        int k;
        Scaled(const Aligner &a_, int k_) : a(a_), k(k_) {}
        int Score(uint8_t x, uint8_t y) const override { return k * a.Score(x, y); }
    } scaled(a, scale);
    const bool lng = gapOpenPenalty != gapExtendPenalty && detail::long_pair(s1, s2, 64);
    Alignment r = detail::run1("global", s1, s2, scaled, (int)(gapOpenPenalty * scale), (int)(gapExtendPenalty * scale), lng);
    return {(double)r.score / scale, r.t1, r.t2};
}


// ---- fasta.go -------------------------------------------------------------------------------------------------
struct FASTASeq {  // fasta.go:17-20
    std::string Header;  // complete header line, including '>'
    Seq seq;             // sequence, all lines concatenated (the reference embeds Seq)
    std::string ID() const
    {  // fasta.go:23-33
        if (Header.empty()) return "";
        const std::string id = Header.substr(1);
        const size_t sp = id.find(' ');
        return sp == std::string::npos ? id : id.substr(0, sp);
    }
    std::string Desc() const
    {  // fasta.go:36-45
        if (Header.empty()) return "";
        const std::string id = Header.substr(1);
        const size_t sp = id.find(' ');
        return sp == std::string::npos ? "" : id.substr(sp + 1);
    }
};

// ReadFASTA (fasta.go:55-84) over the GPU parser.  The reference returns the error "No header" for data before the
// first header line: here a bio::Error whose what() is that text.
inline std::vector<FASTASeq> ReadFASTA(const std::string &text)
{
    bioalign_ctx *ctx = detail::engine();
    bioalign_fasta *fa = bioalign_fasta_parse(ctx, 0, reinterpret_cast<const uint8_t *>(text.data()), text.size(), 0);
    if (!fa) throw Error(BIOALIGN_EINVAL, bioalign_last_error(ctx));
    const uint64_t n = bioalign_fasta_count(fa);
    std::string raw(bioalign_fasta_bytes(fa), '\0');
    const int rc = bioalign_fasta_copy_seq(fa, reinterpret_cast<uint8_t *>(&raw[0]));
    std::vector<FASTASeq> out(n);
    if (rc == 0) {
        const uint64_t *off = bioalign_fasta_off(fa), *hb = bioalign_fasta_header_begin(fa), *hl = bioalign_fasta_header_len(fa);
        for (uint64_t k = 0; k < n; ++k) {
            out[k].Header = text.substr(hb[k], hl[k]);
            out[k].seq = raw.substr(off[k], off[k + 1] - off[k]);
        }
    }
    const std::string msg = rc ? bioalign_last_error(ctx) : "";
    bioalign_fasta_destroy(fa);
    if (rc) throw Error(rc, msg);
    return out;
}

// FASTAReader (fasta.go:93-142), the reference's second reader: one record per ReadSeq call, and rules of its own
// (bufio.ReadLine: "\n" or "\r\n" ends a line, nothing is trimmed, data before the first header is ignored, a line
// beyond bufio's 4096-byte buffer is an error).  The stream is parsed on the GPU when the reader is made
// (bioalign_fasta_parse_reader).  ReadSeq returns the reference's error as text: "" for nil, "EOF" for io.EOF,
// "FASTA line too long"; after that error the reader stays in error.
class FASTAReader {
public:
    explicit FASTAReader(const std::string &text, uint64_t lineLimit = 0)
    {
        bioalign_ctx *ctx = detail::engine();
        bioalign_fasta *fa = bioalign_fasta_parse_reader(ctx, 0, reinterpret_cast<const uint8_t *>(text.data()), text.size(), 0, lineLimit);
        if (!fa) throw Error(BIOALIGN_EINVAL, bioalign_last_error(ctx));
        const uint64_t n = bioalign_fasta_count(fa);
        std::string raw(bioalign_fasta_bytes(fa), '\0');
        const int rc = bioalign_fasta_copy_seq(fa, reinterpret_cast<uint8_t *>(&raw[0]));
        if (rc == 0) {
            const uint64_t *off = bioalign_fasta_off(fa), *hb = bioalign_fasta_header_begin(fa), *hl = bioalign_fasta_header_len(fa);
            recs_.resize(n);
            for (uint64_t k = 0; k < n; ++k) {
                recs_[k].Header = text.substr(hb[k], hl[k]);
                recs_[k].seq = raw.substr(off[k], off[k + 1] - off[k]);
            }
            tooLongAt_ = bioalign_fasta_too_long_at(fa);
        }
        const std::string msg = rc ? bioalign_last_error(ctx) : "";
        bioalign_fasta_destroy(fa);
        if (rc) throw Error(rc, msg);
    }
    std::string ReadSeq(FASTASeq &out)
    {  // fasta.go:109-142
        out = FASTASeq{};
        if (tooLongAt_ >= 0 && (int64_t)next_ >= tooLongAt_) return "FASTA line too long";
        if (next_ >= recs_.size()) return "EOF";
        out = recs_[next_++];
        return "";
    }

private:
    std::vector<FASTASeq> recs_;
    size_t next_ = 0;
    int64_t tooLongAt_ = -1;
};
inline FASTAReader NewFASTAReader(const std::string &text) { return FASTAReader(text); }  // fasta.go:100-102

// ---- DistanceMatrix (dna_list.go:951-969, str.go:450-469) ----------------------------------------------------
// The reference takes any func(a, b) float64 and calls it for every j < i.  AlignDist is the distance function
// that is a function of an alignment score: its n(n-1)/2 alignments are ONE score-only batch on the GPU
// (bioalign_all_pairs); the list lies in HBM once.  toDist maps a score to the distance (default: -score).
struct AlignDist {
    std::string mode;
    const Aligner *a;
    int gapOpen, gapExtend;
    std::function<double(long long)> toDist;
    AlignDist(std::string mode_, const Aligner &a_, int gapOpen_, int gapExtend_ = -1,
              std::function<double(long long)> toDist_ = [](long long s) { return (double)-s; })
        : mode(std::move(mode_)), a(&a_), gapOpen(gapOpen_), gapExtend(gapExtend_ < 0 ? gapOpen_ : gapExtend_),
          toDist(std::move(toDist_))
    {
    }
};

namespace detail {
// a list of sequences as bioalign_all_pairs / bioalign_all_ordered_pairs take it: the bytes back to back, n + 1 offsets,
// one code map over the symbols that occur, the aligner's scores over those symbols
struct PackedList {
    std::string bytes;
    std::vector<uint64_t> off;
    std::array<uint8_t, 256> code{};
    std::vector<int32_t> table;
    bioalign_batch_t in{};
    PackedList(const std::vector<Seq> &l, const Aligner &a, int mode, int gapOpen, int gapExtend) : off(l.size() + 1, 0)
    {
        std::array<bool, 256> seen{};
        for (size_t k = 0; k < l.size(); ++k) {
            bytes += l[k];
            off[k + 1] = bytes.size();
            for (unsigned char b : l[k]) seen[b] = true;
        }
        std::vector<uint8_t> syms;
        for (int b = 0; b < 256; ++b)
            if (seen[b]) code[b] = (uint8_t)syms.size(), syms.push_back((uint8_t)b);
        const int na = syms.empty() ? 1 : (int)syms.size();
        table.assign((size_t)na * na, 0);
        for (size_t i = 0; i < syms.size(); ++i)
            for (size_t j = 0; j < syms.size(); ++j) table[i * na + j] = a.Score(syms[i], syms[j]);
        bytes.push_back(0);
        in.s1 = (const uint8_t *)bytes.data();
        in.s1_off = off.data();
        in.n_pairs = l.size();
        in.mode = mode;
        in.score_table = table.data();
        in.na = in.nb = na;
        in.code1 = in.code2 = code.data();
        in.gap_open = gapOpen;
        in.gap_extend = gapExtend;
        in.gap_symbol = GapSymbol;
        in.flags = BIOALIGN_SCORE_ONLY;
    }
    PackedList(const PackedList &) = delete;
};
}  // namespace detail

inline std::vector<std::vector<double>> DistanceMatrix(const std::vector<Seq> &l, const AlignDist &f)
{
    if (l.empty()) throw std::out_of_range("index out of range [0] with length 0");  // d[0] of an empty slice
    const size_t n = l.size();
    std::vector<std::vector<double>> d(n, std::vector<double>(n, 0.0));
    const int m = detail::mode_of(f.mode);
    if (m < 0) throw std::invalid_argument("unknown mode " + f.mode);
    if (n < 2) return d;
    detail::PackedList pl(l, *f.a, m, f.gapOpen, f.gapExtend);
    std::vector<int64_t> sc(n * (n - 1) / 2);
    bioalign_ctx *ctx = detail::engine();
    const int rc = bioalign_all_pairs(ctx, &pl.in, sc.data());
    if (rc != 0) throw Error(rc, bioalign_last_error(ctx));
    for (size_t i = 1; i < n; ++i)
        for (size_t j = 0; j < i; ++j) d[i][j] = d[j][i] = f.toDist(sc[i * (i - 1) / 2 + j]);
    return d;
}
// any other distance function: the reference's loop as it stands (host code calling the caller's f)
template <class F>
inline std::vector<std::vector<double>> DistanceMatrix(const std::vector<Seq> &l, F f)
{
    if (l.empty()) throw std::out_of_range("index out of range [0] with length 0");
    const size_t n = l.size();
    std::vector<std::vector<double>> d(n, std::vector<double>(n, 0.0));
    for (size_t i = 1; i < n; ++i)
        for (size_t j = 0; j < i; ++j) d[i][j] = d[j][i] = f(l[i], l[j]);
    return d;
}

// ---- AlignLong on overlap alignments (align_long.go:17-90) -----------------------------
// The reference's greedy assembler finds overlaps by exact substring search (strings.Index of s's last half in t,
// align_long.go:35-45): one read error breaks the assembly.  AlignLongOverlap keeps its merge (steps 2-4,
// align_long.go:46-89: largest overlap first, the first one wins ties, overlaps from the merged fragment deleted by
// swapping with the last, overlaps from the absorbed one re-pointed -- the order of that list decides the output, so
// it is kept as the reference has it) and replaces step 1 by ONE call, bioalign_all_ordered_pairs: every ordered pair
// of fragments aligned in "overlap" mode (suffix of s against prefix of t, align_pair.go:184-195), scores and end
// cells only.  An overlap counts when it covers at least half a fragment of t (align_long.go:10-13) and scores at
// least minScorePerColumn per covered column.  All fragments have the same length.  Returns the assembled string, or
// "" if the fragments do not assemble into one.  Behaviour differs from AlignLong on purpose (overlaps may hold
// mismatches and gaps; repeats: bio_amd/assembly.py); parity unpinned, the reference has no test of AlignLong.
inline Seq AlignLongOverlap(std::vector<Seq> ss, const Aligner &a = MatchAligner(1, -1), int indelPenalty = 2,
                            double minScorePerColumn = 0.8)
{
    const size_t n = ss.size();
    if (n == 0) return "";
    if (n == 1) return ss[0];
    const size_t half = ss[0].size() / 2;
    struct Overlap {
        size_t si, ti;
        long long x;  // x + half = length of t's prefix inside the overlap
    };
    std::vector<Overlap> overlaps;
    {
        detail::PackedList pl(ss, a, BIOALIGN_OVERLAP, indelPenalty, indelPenalty);
        std::vector<int64_t> sc(n * (n - 1));
        std::vector<uint64_t> ei(n * (n - 1)), ej(n * (n - 1));
        bioalign_ctx *ctx = detail::engine();
        const int rc = bioalign_all_ordered_pairs(ctx, &pl.in, sc.data(), ei.data(), ej.data());
        if (rc != 0) throw Error(rc, bioalign_last_error(ctx));
        for (size_t si = 0; si < n; ++si)
            for (size_t ti = 0; ti < n; ++ti) {
                if (ss[ti] == ss[si]) continue;  // (align_long.go:38-40; also si == ti)
                const size_t q = si * (n - 1) + (ti < si ? ti : ti - 1);
                const uint64_t j = ej[q];
                if (j >= half && j > 0 && (double)sc[q] >= minScorePerColumn * (double)j)
                    overlaps.push_back({si, ti, (long long)j - (long long)half});
            }
    }
    size_t left = 0;
    for (size_t fragments = n; fragments > 1; --fragments) {
        long long xMax = -1;
        size_t oMax = 0;
        for (size_t i = 0; i < overlaps.size(); ++i) {
            const Seq &s = ss[overlaps[i].si], &t = ss[overlaps[i].ti];
            if (s.empty() || t.empty() || t == s) continue;
            if (overlaps[i].x > xMax) xMax = overlaps[i].x, oMax = i;
        }
        if (xMax == -1) return "";
        left = overlaps[oMax].si;
        const size_t right = overlaps[oMax].ti;
        ss[left] += ss[right].substr((size_t)xMax + half);
        ss[right].clear();
        for (size_t i = 0; i < overlaps.size();) {
            if (overlaps[i].si == left) {
                overlaps[i] = overlaps.back();
                overlaps.pop_back();
                continue;
            }
            if (overlaps[i].si == right) overlaps[i].si = left;
            ++i;
        }
    }
    return ss[left];
}

}  // namespace bio
