// C++ host-mirror tests, written like the reference's own Example tests (align_test.go,
// align_fitting_test.go, align_overlap_test.go): every golden vector through bio.hpp, plus the
// reference's error behaviour.   bio_test --cpu : what needs no GPU;   bio_test --gpu : everything.
#include <cstdio>
#include <cstring>
#include <memory>

#include "../../include/bio.hpp"
#include "golden_vectors.inc"

static int failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            ++failures;                                    \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                      \
            std::printf("\n");                             \
        }                                                  \
    } while (0)

static std::unique_ptr<bio::Aligner> aligner_of(const Golden &g)
{
    if (!std::strcmp(g.aligner, "Blosum62")) return std::make_unique<bio::SubstMatrix>(bio::Blosum62());
    if (!std::strcmp(g.aligner, "Pam250")) return std::make_unique<bio::SubstMatrix>(bio::Pam250());
    return std::make_unique<bio::MatchAligner>(g.match, g.mismatch);
}

static void cpu_tests()
{
    for (const Golden &g : kGolden) {
        auto a = aligner_of(g);
        if (!std::strcmp(g.fn, "LinearGap")) CHECK(bio::LinearGap(g.s1, g.s2, *a, g.gap) == g.score, "%s", g.id);
        if (!std::strcmp(g.fn, "ConstantGap")) CHECK(bio::ConstantGap(g.s1, g.s2, *a, g.gap) == g.score, "%s", g.id);
    }
    const bio::SubstMatrix &b = bio::Blosum62();
    CHECK(b.Score('W', 'W') == 11 && b.Score('A', 'R') == -1, "Blosum62 entries");
    CHECK(b.Score('B', 'W') == b.Score('A', 'W'), "foreign byte below max aliases to symbol 0 (subst_matrix.go:27-30)");
    bool threw = false;
    try { b.Score('a', 'A'); } catch (const std::out_of_range &) { threw = true; }
    CHECK(threw, "byte above max panics");
    CHECK(bio::Pam250().Score('W', 'W') == 17, "Pam250 entry");
    threw = false;
    try { bio::LinearGap("AC", "A", b, 1); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw, "LinearGap panics on unequal lengths (align.go:19-21)");
    const uint8_t old = bio::GapSymbol;
    bio::GapSymbol = '.';
    CHECK(bio::LinearGap("A.", "AC", bio::MatchAligner(1, -1), 3) == 1 - 3, "GapSymbol is read at call time");
    bio::GapSymbol = old;
    // unknown mode: (-1, nil, nil), never reaches the device (align_pair.go:14-15)
    bio::Alignment r = bio::AlignPair("semi-global", "AC", "AC", b, 1);
    CHECK(r.score == -1 && r.t1.empty() && r.t2.empty(), "unknown mode");
    threw = false;
    try { bio::AlignGlobal("AC", "", b, 1); } catch (const std::out_of_range &) { threw = true; }
    CHECK(threw, "AlignGlobal with an empty s2 panics (align.go:128-130)");
}

static void gpu_tests()
{
    for (const Golden &g : kGolden) {
        auto a = aligner_of(g);
        bio::Alignment want{g.score, g.t1, g.t2}, got;
        if (!std::strcmp(g.fn, "AlignGlobal")) got = bio::AlignGlobal(g.s1, g.s2, *a, g.gap);
        else if (!std::strcmp(g.fn, "AlignLocal")) got = bio::AlignLocal(g.s1, g.s2, *a, g.gap);
        else if (!std::strcmp(g.fn, "AlignGlobalLS")) got = bio::AlignGlobalLS(g.s1, g.s2, *a, g.gap);
        else if (!std::strcmp(g.fn, "AlignPair")) got = bio::AlignPair(g.mode, g.s1, g.s2, *a, g.gap);
        else continue;
        CHECK(got == want, "%s: got (%lld, %s, %s)", g.id, got.score, got.t1.c_str(), got.t2.c_str());
    }
    const bio::SubstMatrix &b = bio::Blosum62();
    CHECK((bio::AlignGlobal("", "", b, 5) == bio::Alignment{0, "", ""}), "empty inputs give nil traces");
    CHECK((bio::AlignPair("global", "AC", "", b, 5) == bio::Alignment{-10, "AC", "--"}), "AlignPair global with empty s2");
    CHECK((bio::AlignLocal("W", "A", b, 5) == bio::Alignment{0, "", ""}), "nothing positive: 0, nil, nil");
    bool threw = false;
    try { bio::AlignGlobal("a", "A", b, 5); } catch (const std::out_of_range &) { threw = true; }
    CHECK(threw, "lower case is beyond Index: panic while the score table is materialised");
    // batch = the single-pair results
    std::vector<bio::Seq> s1s = {"PLEASANTLY", "MEANLY", "", "PAWHEAE"}, s2s = {"MEANLY", "PENALTY", "AC", "HEAGAWGHEE"};
    auto batch = bio::AlignPairBatch("local", s1s, s2s, b, 5);
    for (size_t p = 0; p < s1s.size(); ++p) CHECK(batch[p] == bio::AlignLocal(s1s[p], s2s[p], b, 5), "batch pair %zu", p);
    // affine: the trace spells the inputs and rescoring it gives the score (open 5, extend 1)
    auto [sc, t1, t2] = bio::AlignGlobalAffine("ACDDE", "ACE", b, 5, 1);
    std::string u1, u2;
    for (char ch : t1) if (ch != '-') u1 += ch;
    for (char ch : t2) if (ch != '-') u2 += ch;
    CHECK(u1 == "ACDDE" && u2 == "ACE" && t1.size() == t2.size(), "affine trace spells its inputs");
    double resc = 0;
    int open = 0;
    for (size_t k = 0; k < t1.size(); ++k) {
        if (t1[k] == '-') resc -= (open == 1 ? 1 : 5), open = 1;
        else if (t2[k] == '-') resc -= (open == 2 ? 1 : 5), open = 2;
        else resc += b.Score((uint8_t)t1[k], (uint8_t)t2[k]), open = 0;
    }
    CHECK(resc == sc, "affine rescoring: %g vs %g", resc, sc);
    {   // fractional penalties run in units of the fraction, nothing is truncated (open 2.5, extend 0.5)
        auto [fs, f1, f2] = bio::AlignGlobalAffine("ACDDE", "ACE", b, 2.5, 0.5);
        double r2 = 0;
        int op2 = 0;
        for (size_t k = 0; k < f1.size(); ++k) {
            if (f1[k] == '-') r2 -= (op2 == 1 ? 0.5 : 2.5), op2 = 1;
            else if (f2[k] == '-') r2 -= (op2 == 2 ? 0.5 : 2.5), op2 = 2;
            else r2 += b.Score((uint8_t)f1[k], (uint8_t)f2[k]), op2 = 0;
        }
        CHECK(r2 == fs && f1.size() == f2.size(), "fractional affine rescoring: %g vs %g", r2, fs);
        bool refused = false;
        try { bio::AlignGlobalAffine("ACDDE", "ACE", b, 0.1, 1); } catch (const std::invalid_argument &) { refused = true; }
        CHECK(refused, "a penalty that is not a multiple of 1/64 is refused");
    }
    threw = false;
    try { bio::AlignAffineBatch("global", {"ACGT"}, {"ACG"}, bio::MatchAligner(1, -1), 1, 3); } catch (const bio::Error &e) { threw = e.code == BIOALIGN_EUNSUPPORTED; }
    CHECK(threw, "open < extend fails loudly");
    // long mode = the batch result
    bio::Seq x, y;
    uint64_t st = 12345;
    for (int k = 0; k < 900; ++k) { st = st * 6364136223846793005ull + 1442695040888963407ull; x += "ACGT"[(st >> 33) & 3]; }
    y = x;
    for (int k = 7; k < 900; k += 37) y[k] = y[k] == 'A' ? 'C' : 'A';
    y.erase(300, 5);
    bio::MatchAligner dna(1, -1);
    CHECK(bio::AlignPairLong("global", x, y, dna, 1) == bio::AlignPair("global", x, y, dna, 1), "long mode = batch mode");

    // ReadFASTA: the reference's example (fasta_test.go:10-44) and its documented behaviour (fasta.go:47-54)
    {
        const auto recs = bio::ReadFASTA(">db|123|abc example sequence\nAGACCA\nTACCA");
        CHECK(recs.size() == 1 && recs[0].Header == ">db|123|abc example sequence" && recs[0].seq == "AGACCATACCA", "ReadFASTA example");
        CHECK(recs[0].ID() == "db|123|abc" && recs[0].Desc() == "example sequence", "FASTASeq ID / Desc");
        const auto r2 = bio::ReadFASTA("\n  >a x \r\nAC GT \r\n\r\n>b\n>c\n  TT\n");
        CHECK(r2.size() == 3 && r2[0].Header == ">a x" && r2[0].seq == "AC GT" && r2[1].seq.empty() && r2[2].seq == "TT", "ReadFASTA trimming, crlf, empty record");
        CHECK(bio::ReadFASTA("").empty(), "ReadFASTA of nothing");
        bool no_header = false;
        try { bio::ReadFASTA("ACGT\n>a\n"); } catch (const bio::Error &e) { no_header = std::string(e.what()).find("No header") != std::string::npos; }
        CHECK(no_header, "ReadFASTA: data before the first header is the error \"No header\" (fasta.go:72-73)");
    }

    // FASTAReader.ReadSeq: the reference's example (fasta_test.go:46-64) and the rules that differ from ReadFASTA's
    {
        bio::FASTAReader r = bio::NewFASTAReader(">db|123|abc example sequence\nAGACCA\nTACCA");
        bio::FASTASeq s;
        CHECK(r.ReadSeq(s).empty() && s.Header == ">db|123|abc example sequence" && s.seq == "AGACCATACCA", "ReadSeq example");
        CHECK(r.ReadSeq(s) == "EOF", "ReadSeq: EOF after the last record");
        bio::FASTAReader r2("junk\n>a x \r\n AC GT \r\n\r\n>b\n");
        CHECK(r2.ReadSeq(s).empty() && s.Header == ">a x " && s.seq == " AC GT ", "ReadSeq: nothing trimmed, crlf, leading data ignored");
        CHECK(r2.ReadSeq(s).empty() && s.Header == ">b" && s.seq.empty() && r2.ReadSeq(s) == "EOF", "ReadSeq: empty last record");
        bio::FASTAReader r3(">a\nAC\n>b\n" + std::string(4096, 'A') + "\n>c\nC\n");
        CHECK(r3.ReadSeq(s).empty() && s.seq == "AC" && r3.ReadSeq(s) == "FASTA line too long", "ReadSeq: line beyond bufio's buffer (fasta.go:124-126)");
    }

    // DistanceMatrix: the reference's example (dna_list_test.go:320-337 = str_test.go:218-235), Hamming distances.
    // Equal lengths, match 0 / mismatch -1 and 2 * gap >= length: -(global score) is the Hamming distance.
    {
        const std::vector<bio::Seq> l = {"TTTCCATTTA", "GATTCATTTC", "TTTCCATTTT", "GTTCCATTTA"};
        const std::vector<std::vector<double>> want = {{0, 4, 1, 1}, {4, 0, 4, 3}, {1, 4, 0, 2}, {1, 3, 2, 0}};
        bio::MatchAligner ham(0, -1);
        CHECK(bio::DistanceMatrix(l, bio::AlignDist("global", ham, 5)) == want, "DistanceMatrix example through the engine");
        auto hamming = [](const bio::Seq &a, const bio::Seq &b) {
            double d = 0;
            for (size_t k = 0; k < a.size() && k < b.size(); ++k) d += a[k] != b[k];
            return d;
        };
        CHECK(bio::DistanceMatrix(l, hamming) == want, "DistanceMatrix example, caller's function");
        bool threw = false;
        try { bio::DistanceMatrix(std::vector<bio::Seq>{}, bio::AlignDist("global", ham, 5)); } catch (const std::out_of_range &) { threw = true; }
        CHECK(threw, "DistanceMatrix of an empty list panics (dna_list.go:957)");
        CHECK(bio::DistanceMatrix({"ACGT"}, bio::AlignDist("local", ham, 5)) == std::vector<std::vector<double>>{{0}}, "one sequence");
    }

    // AlignLongOverlap (align_long.go:17-90 with overlap alignments for step 1): reads of 60 every 20 bases of a
    // pseudo-random genome, out of order; error-free reads give the genome back (what AlignLong returns for them), and
    // so do reads with a substitution in every third one, which AlignLong's exact substring search cannot assemble
    {
        std::string genome;
        uint32_t lcg = 12345;
        for (int k = 0; k < 400; ++k) {
            lcg = lcg * 1664525u + 1013904223u;
            genome.push_back("ACGT"[(lcg >> 24) & 3]);
        }
        std::vector<bio::Seq> reads;
        for (size_t s0 = 0; s0 + 60 <= genome.size(); s0 += 20) reads.push_back(genome.substr(s0, 60));
        std::vector<bio::Seq> shuffled;
        for (size_t k = 0; k < reads.size(); ++k) shuffled.push_back(reads[(k * 7) % reads.size()]);  // 18 reads, 7 is coprime
        CHECK(reads.size() == 18, "18 reads");
        CHECK(bio::AlignLongOverlap(shuffled) == genome, "AlignLongOverlap: error-free reads assemble to the genome");
        std::vector<bio::Seq> noisy = shuffled;
        int errors = 0;
        for (size_t k = 0; k < noisy.size(); k += 3) {
            char &c = noisy[k][30 + (int)(k % 5) - 2];
            c = c == 'A' ? 'C' : c == 'C' ? 'G' : c == 'G' ? 'T' : 'A';
            ++errors;
        }
        const bio::Seq got = bio::AlignLongOverlap(noisy);
        int diff = 0;
        for (size_t k = 0; k < got.size() && k < genome.size(); ++k) diff += got[k] != genome[k];
        CHECK(got.size() == genome.size() && diff <= errors, "AlignLongOverlap: reads with substitutions still assemble");
        CHECK(bio::AlignLongOverlap({}) == "" && bio::AlignLongOverlap({"ACGT"}) == "ACGT", "AlignLongOverlap: no reads, one read (align_long.go:18-23)");
        CHECK(bio::AlignLongOverlap({"AAAACCCC", "GGGGTTTT"}) == "", "AlignLongOverlap: nothing overlaps (align_long.go:62-64)");
    }
}

int main(int argc, char **argv)
{
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "--gpu");
    cpu_tests();
    if (gpu) {
        try {
            gpu_tests();
        } catch (const std::exception &e) {
            std::printf("FAIL: exception: %s\n", e.what());
            ++failures;
        }
    } else {
        // without a GPU the engine must refuse loudly
        bool refused = false;
        try { bio::AlignPair("global", "ACGT", "ACG", bio::MatchAligner(1, -1), 1); } catch (const bio::Error &e) { refused = e.code == BIOALIGN_ENODEVICE; }
        if (argc > 2 && !std::strcmp(argv[2], "--expect-no-device")) CHECK(refused, "no GPU: BIOALIGN_ENODEVICE");
    }
    std::printf(failures ? "%d failure(s)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
