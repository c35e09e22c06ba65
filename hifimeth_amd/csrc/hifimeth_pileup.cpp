// hifimeth_pileup.cpp -- `hifimeth-hip pileup [OPTIONS] reference mod-bam output-prefix`
// Same command line, stderr messages and output files as `hifimeth pileup` (src/app/hifimeth/pileup.cpp:22-112,
// 461-606): <prefix>.CpG.cov.bed, <prefix>.CHG.cov.bed, <prefix>.CHH.cov.bed with rows
//   chrom <tab> soff <tab> soff+1 <tab> 100*pcov/(pcov+ncov) <tab> pcov <tab> ncov
// The host parses the BAM and the MM/ML lists (parallel over the reads of a batch); alignment projection,
// histograms and per-locus counting run on the GPU through the hm_pileup_* C ABI.  No temporary file is written:
// the projected calls stay in HBM until the thresholds are known.
#include <strings.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/hifimeth_hip.h"
#include "hm_bam.h"

using namespace hmbam;

// CIGAR of a record as uint32 ops.  A CIGAR of more than 65535 operations does not fit the 16-bit n_cigar_op field: the
// record then carries the placeholder <l_seq>S<ref_len>N and the real operations in a CG:B,I tag (SAMv1 section 4.2.2);
// htslib's sam_read1 -- what the reference's pileup reads with -- swaps them back in, and so does this.
static void real_cigar(const BamRecord& r, std::vector<uint32_t>& cig) {
    cig.resize((size_t)r.n_cigar());
    if (!cig.empty()) memcpy(cig.data(), r.cigar_bytes(), 4 * cig.size());
    if (cig.size() != 2 || (cig[0] & 15) != 4 || (int32_t)(cig[0] >> 4) != r.l_qseq() || (cig[1] & 15) != 3) return;
    const uint8_t* p = r.data.data() + r.aux_offset();
    const uint8_t* end = r.data.data() + r.data.size();
    AuxField f;
    while (p < end && next_aux(p, end, f))
        if (f.tag[0] == 'C' && f.tag[1] == 'G' && f.type == 'B' && (f.subtype == 'I' || f.subtype == 'i')) {
            cig.resize(f.count);
            memcpy(cig.data(), f.payload, 4 * (size_t)f.count);
            return;
        }
}


namespace {
void report_thresholds(const uint64_t* bins, uint8_t thr[3]);
}  // namespace

namespace {

struct PileupOptions {
    int min_mapq = 0;     // kMinMapQ
    double min_pi = 0.0;  // kMinPi
    int threads = 8;      // kNumThreads
    int device = 0;
    int read_batch = 512;
    std::string ref, bam, prefix;
};

void pileup_usage(const char* exe) {
    fprintf(stderr,
            "USAGE:\n  %s pileup [OPTIONS] reference mod-bam output-prefix\n\n"
            "DESCRIPTION:\n  Compute aggregate cytosine methylation states on the genomic reference\n\n"
            "OPTIONAL ARGUMENTS:\n"
            "  -q <mapQ>\n    Minimum mapping quality score\n    Default: 0\n"
            "  -f <Alignment identity>\n    Default: 0\n"
            "  -t <CPU threads>\n    Number of CPU threads\n    Default: 8\n"
            "  -d <int>\n    GPU ordinal\n    Default: 0\n"
            "  -b <int>\n    BAM records per GPU batch\n    Default: 512\n",
            exe);
}

// s_bam_is_mapped_and_sorted (pileup.cpp:438-459)
bool mapped_and_sorted(const BamHeader& h) {
    bool sorted = false;
    size_t p = 0;
    while (p < h.text.size()) {
        size_t e = h.text.find('\n', p);
        if (e == std::string::npos) e = h.text.size();
        const std::string line = h.text.substr(p, e - p);
        if (line.rfind("@HD", 0) == 0) {
            size_t q = 0;
            while ((q = line.find('\t', q)) != std::string::npos) {
                ++q;
                if (line.compare(q, 3, "SO:") == 0) {
                    size_t t = line.find('\t', q);
                    sorted = line.substr(q + 3, t == std::string::npos ? t : t - q - 3) == "coordinate";
                }
            }
        }
        p = e + 1;
    }
    const bool mapped = !h.refs.empty();
    if (!mapped || !sorted) {
        fprintf(stderr, "ERROR: Methylation frequency could not be computed due to the following errors:\n");
        if (!mapped) fprintf(stderr, "BAM is not mapped\n");
        if (!sorted) fprintf(stderr, "BAM is not sorted\n");
        return false;
    }
    return true;
}

}  // namespace

// corr [-c min_cov] bed1 bed2 : Pearson correlation of the methylation frequencies of the loci two *.cov.bed files
// share (src/app/hifimeth/pileup_correlation.cpp:101-210): rows with pcov + ncov < min_cov (default 5) are dropped,
// loci are keyed by (chromosome id in order of first appearance over both files, start); host only, no GPU.
int cmd_corr(int argc, char** argv) {
    int min_cov = 5;
    int i = 2;
    for (; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.size() < 2 || a[0] != '-') break;
        if (a == "-c" && i + 1 < argc) min_cov = atoi(argv[++i]);
        else { fprintf(stderr, "ERROR: unrecognised option %s", a.c_str()); return 1; }
    }
    if (argc - i != 2) {
        fprintf(stderr, "USAGE:\n  %s corr [-c <min coverage, default 5>] bed1 bed2\n", argv[0]);
        return 1;
    }
    fprintf(stderr, "\n\n====================> Parameters:\nmin-cov: %d\nbed1: %s\nbed2: %s\n\n\n", min_cov, argv[i], argv[i + 1]);
    std::vector<std::string> chr_names;
    auto chr_id = [&](const std::string& nm) {
        for (size_t k = 0; k < chr_names.size(); ++k)
            if (chr_names[k] == nm) return (uint64_t)k;
        chr_names.push_back(nm);
        return (uint64_t)chr_names.size() - 1;
    };
    auto load = [&](const char* path, std::vector<std::pair<uint64_t, double>>& out) {
        gzFile f = gzopen(path, "rb");
        if (!f) { fprintf(stderr, "ERROR: cannot open %s\n", path); return false; }
        static char line[1 << 16];
        std::string last;
        uint64_t sid = 0;
        while (gzgets(f, line, sizeof line)) {
            char* col[6];
            int nc = 0;
            char* p = line;
            col[nc++] = p;
            for (; *p && nc < 6; ++p)
                if (*p == '\t') { *p = 0; col[nc++] = p + 1; }
            if (nc < 6) continue;
            const int pcov = atoi(col[4]), ncov = atoi(col[5]);
            if (pcov + ncov < min_cov) continue;
            if (last != col[0]) { last = col[0]; sid = chr_id(last); }
            out.emplace_back((sid << 32) | (uint64_t)(uint32_t)atoi(col[1]), 1.0 * pcov / (pcov + ncov));
        }
        gzclose(f);
        return true;
    };
    std::vector<std::pair<uint64_t, double>> m1, m2;
    if (!load(argv[i], m1) || !load(argv[i + 1], m2)) return 1;
    auto by_key = [](const std::pair<uint64_t, double>& x, const std::pair<uint64_t, double>& y) { return x.first < y.first; };
    std::sort(m1.begin(), m1.end(), by_key);
    std::sort(m2.begin(), m2.end(), by_key);
    std::vector<double> x, y;
    for (size_t a = 0, b = 0; a < m1.size() && b < m2.size();) {
        if (m1[a].first < m2[b].first) ++a;
        else if (m1[a].first > m2[b].first) ++b;
        else { x.push_back(m1[a++].second); y.push_back(m2[b++].second); }
    }
    const size_t n = x.size();
    if (n < 5) { fprintf(stderr, "Intersect genomic loci is less than 5. Skip computation\n"); return 0; }
    double mx = 0, my = 0;
    for (size_t k = 0; k < n; ++k) { mx += x[k]; my += y[k]; }
    mx /= (double)n;
    my /= (double)n;
    double cov = 0, vx = 0, vy = 0;
    for (size_t k = 0; k < n; ++k) {
        const double dx = x[k] - mx, dy = y[k] - my;
        cov += dx * dy;
        vx += dx * dx;
        vy += dy * dy;
    }
    const double corr = (vx == 0 || vy == 0) ? 0.0 : cov / std::sqrt(vx * vy);
    fprintf(stdout, "Intersect loci: %zu\n", n);
    fprintf(stderr, "correlation: %g\n", corr);
    return 0;
}

// cov2bed REF.fa CONTEXT bismark.cov out.bed : 1-based Bismark coverage rows -> 0-based BED rows with a motif column
// (src/app/hifimeth/cov_to_bed.cpp).  Each input row "chr pos pos freq pcov ncov" is looked up on the reference and
// follows one rule of the table below: rows on a 'C' start a locus, rows on a 'G' either fold into the locus of the
// palindromic partner on the forward strand (CpG: one base left; CAG/CTG: two bases left) or stay where they are
// (CCG's partner CGG, every CHH motif, which is always named by its forward-strand spelling).  A chromosome's loci are
// written when the input moves on to another chromosome, as the reference does.  Neighbours outside the chromosome
// never match (the reference reads into the adjacent sequence there).  Host only, no GPU.
int cmd_cov2bed(int argc, char** argv) {
    if (argc != 6) {
        fprintf(stderr, "USAGE:\n%s %s reference context bismark-call bed\n", argv[0], argv[1]);
        return 1;
    }
    const std::string context = argv[3];
    int ctx = -1;
    if (context.size() == 3) {
        if (strcasecmp(context.c_str(), "CpG") == 0) ctx = 0;
        else if (strcasecmp(context.c_str(), "CHG") == 0) ctx = 1;
        else if (strcasecmp(context.c_str(), "CHH") == 0) ctx = 2;
    }
    if (ctx < 0) {
        fprintf(stderr, "Illegal 5mc context: %s\nPlausible contexts: CpG, CHG, CHH\n", context.c_str());
        return 1;
    }
    Fasta fa;
    std::string err;
    if (!load_fasta(argv[2], fa, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return EXIT_FAILURE; }
    std::vector<int64_t> start(fa.names.size() + 1, 0);
    for (size_t s = 0; s < fa.names.size(); ++s) start[s + 1] = start[s] + fa.length[s];

    // one rule: the row's own base, the 3-mer window it is matched against (offset of the window's first base relative
    // to the row), where the counts go (relative to the row), whether they replace or add, and the motif written
    struct Rule { char base; int win; const char* kmer; int dst; bool add; const char* motif; };
    static const Rule cpg[] = {{'C', 0, "CG", 0, false, "CG"}, {'G', -1, "CG", -1, true, "CG"}};
    static const Rule chg[] = {{'C', 0, "CCG", 0, false, "CCG"}, {'G', -2, "CGG", 0, false, "CCG"},
                               {'C', 0, "CAG", 0, false, "CAG"}, {'G', -2, "CAG", -2, true, "CAG"},
                               {'C', 0, "CTG", 0, false, "CTG"}, {'G', -2, "CTG", -2, true, "CTG"}};
    static const Rule chh[] = {{'C', 0, "CAA", 0, false, "CAA"}, {'C', 0, "CCA", 0, false, "CCA"}, {'C', 0, "CTA", 0, false, "CTA"},
                               {'C', 0, "CAC", 0, false, "CAC"}, {'C', 0, "CCC", 0, false, "CCC"}, {'C', 0, "CTC", 0, false, "CTC"},
                               {'C', 0, "CAT", 0, false, "CAT"}, {'C', 0, "CCT", 0, false, "CCT"}, {'C', 0, "CTT", 0, false, "CTT"},
                               {'G', -2, "TTG", 0, false, "CAA"}, {'G', -2, "TGG", 0, false, "CCA"}, {'G', -2, "TAG", 0, false, "CTA"},
                               {'G', -2, "GTG", 0, false, "CAC"}, {'G', -2, "GGG", 0, false, "CCC"}, {'G', -2, "GAG", 0, false, "CTC"},
                               {'G', -2, "ATG", 0, false, "CAT"}, {'G', -2, "AGG", 0, false, "CCT"}, {'G', -2, "AAG", 0, false, "CTT"}};
    const Rule* rules = ctx == 0 ? cpg : ctx == 1 ? chg : chh;
    const int n_rules = ctx == 0 ? 2 : ctx == 1 ? 6 : 18;

    struct Locus { int pcov, ncov; const char* motif; };
    std::vector<Locus> loci;
    FILE* out = fopen(argv[5], "w");
    if (!out) { fprintf(stderr, "ERROR: cannot open %s for writing\n", argv[5]); return EXIT_FAILURE; }
    gzFile in = gzopen(argv[4], "rb");
    if (!in) { fprintf(stderr, "ERROR: cannot open %s\n", argv[4]); fclose(out); return EXIT_FAILURE; }
    int cur = -1;
    bool bad = false;
    auto dump = [&]() {
        if (cur < 0) return;
        std::string text;
        char row[256];
        for (size_t i = 0; i < loci.size(); ++i) {
            const Locus& l = loci[i];
            if (!l.motif) continue;
            const int cov = l.pcov + l.ncov;
            if (cov <= 0) { fprintf(stderr, "ERROR: locus %s:%zu has no coverage\n", fa.names[cur].c_str(), i); bad = true; return; }
            const int len = snprintf(row, sizeof row, "\t%zu\t%zu\t%g\t%d\t%d\t%s\n", i, i + 1, 100.0 * l.pcov / cov, l.pcov, l.ncov, l.motif);
            text += fa.names[cur];
            text.append(row, (size_t)len);
        }
        fwrite(text.data(), 1, text.size(), out);
    };
    size_t fs = 0, rs = 0;
    static char line[1 << 16];
    std::string last_name;
    while (!bad && gzgets(in, line, sizeof line)) {
        size_t ll = strlen(line);
        while (ll && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
        char* col[6];
        int nc = 0;
        char* p = line;
        col[nc++] = p;
        for (; *p && nc < 6; ++p)
            if (*p == '\t') { *p = 0; col[nc++] = p + 1; }
        if (nc < 6) { fprintf(stderr, "ERROR: corrupted bismark record %s\n", line); bad = true; break; }
        if (cur < 0 || last_name != col[0]) {
            const int sid = fa.find(col[0]);
            if (sid < 0) { fprintf(stderr, "ERROR: sequence %s is not in %s\n", col[0], argv[2]); bad = true; break; }
            dump();
            if (bad) break;
            cur = sid;
            last_name = col[0];
            loci.assign((size_t)fa.length[sid], Locus{0, 0, nullptr});
        }
        const int64_t pos1 = atoll(col[1]);
        if (atoll(col[2]) != pos1) { fprintf(stderr, "ERROR: start and end differ: %s:%s-%s\n", col[0], col[1], col[2]); bad = true; break; }
        const int pcov = atoi(col[4]), ncov = atoi(col[5]);
        const int64_t len = fa.length[cur], soff = pos1 - 1;
        if (soff < 0 || soff >= len) { fprintf(stderr, "ERROR: position %s is outside %s\n", col[1], col[0]); bad = true; break; }
        const char* chr = fa.bases.data() + start[cur];
        for (int r = 0; r < n_rules; ++r) {
            const Rule& R = rules[r];
            if (chr[soff] != R.base) continue;
            const int k = (int)strlen(R.kmer);
            const int64_t w = soff + R.win;
            if (w < 0 || w + k > len || strncmp(chr + w, R.kmer, (size_t)k) != 0) continue;
            Locus& l = loci[(size_t)(soff + R.dst)];
            if (R.add) {
                l.pcov += pcov;
                l.ncov += ncov;
                if (!l.motif) l.motif = R.motif;
            } else {
                l = Locus{pcov, ncov, R.motif};
            }
            ++(R.base == 'C' ? fs : rs);
        }
    }
    gzclose(in);
    if (!bad) dump();
    fclose(out);
    if (bad) return EXIT_FAILURE;
    fprintf(stderr, "forward-strand-sites: %zu, reverse-strand-sites: %zu\n", fs, rs);
    return 0;
}

// sample [-s seed] REF.fa in.bam COVERAGE out.bam : random subset of the usable reads of an (unaligned) HiFi BAM adding
// up to COVERAGE x the reference size (src/app/hifimeth/subsample_bam.cpp).  A read is usable when it has >= 5000
// bases and all four kinetics arrays (:18-28); usable reads are shuffled, taken until the base target is reached (the
// read that crosses it included, :97-103) and written in input order (:105-117).  The reference seeds its shuffle from
// std::random_device, so which reads come out is not reproducible there either; `-s` (ours) fixes the seed for tests.
namespace {
std::string human_size(uint64_t bytes) {  // bytes_to_datasize (src/corelib/hbn_aux.cpp:447-490): 1024-based, <= 3 digits
    static const char* unit[] = {"B", "KB", "MB", "GB", "TB", "PB", "EB"};
    if (bytes == 0) return "0B";
    double v = (double)bytes;
    int u = 0;
    while (v >= 1024.0 && u < 6) { v /= 1024.0; ++u; }
    char buf[64];
    if (u == 0 || v >= 100) snprintf(buf, sizeof buf, "%llu", (unsigned long long)std::llround(v));
    else snprintf(buf, sizeof buf, v >= 10 ? "%.1f" : "%.2f", v);
    std::string r = buf;
    if (r.find('.') != std::string::npos) {
        r.erase(r.find_last_not_of('0') + 1);
        if (r.back() == '.') r.pop_back();
    }
    return r + unit[u];
}
}  // namespace

int cmd_sample(int argc, char** argv) {
    int a = 2;
    bool seeded = false;
    uint64_t seed = 0;
    if (argc >= 4 && std::string(argv[2]) == "-s") { seeded = true; seed = strtoull(argv[3], nullptr, 10); a = 4; }
    if (argc - a != 4) {
        fprintf(stderr, "USAGE:\n  %s %s [-s seed] reference input-bam coverage output-bam\n", argv[0], argv[1]);
        return 1;
    }
    const char* ref_path = argv[a];
    const char* in_path = argv[a + 1];
    const int cov = atoi(argv[a + 2]);
    const char* out_path = argv[a + 3];
    Fasta fa;
    std::string err;
    if (!load_fasta(ref_path, fa, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return EXIT_FAILURE; }
    const uint64_t dbsize = fa.bases.size(), target = dbsize * (uint64_t)std::max(cov, 0);

    struct Info { uint32_t id; int32_t length; bool valid, selected; };
    std::vector<Info> list;
    uint64_t total = 0;
    {
        BgzfReader in(in_path, 8);
        BamHeader h;
        if (!in.ok() || !read_header(in, h, err)) { fprintf(stderr, "ERROR: %s: %s\n", in_path, err.empty() ? "cannot open" : err.c_str()); return EXIT_FAILURE; }
        BamRecord r;
        while (read_record(in, r, err)) {
            Info f{(uint32_t)list.size(), r.l_qseq(), false, false};
            if (f.length >= 5000) {
                const KineticsView kv = kinetics_of(r);
                f.valid = kv.arr[0] && kv.arr[1] && kv.arr[2] && kv.arr[3];
            }
            if (f.valid) total += (uint64_t)f.length;
            list.push_back(f);
        }
        if (!err.empty()) { fprintf(stderr, "ERROR: %s: %s\n", in_path, err.c_str()); return EXIT_FAILURE; }
    }
    fprintf(stderr, "DB size: %s\ncoverage: %d, target size: %s\nBAM size: %s\n", human_size(dbsize).c_str(), cov,
            human_size(target).c_str(), human_size(total).c_str());
    std::mt19937 gen(seeded ? (uint32_t)seed : std::random_device{}());
    std::shuffle(list.begin(), list.end(), gen);
    uint64_t picked = 0;
    for (Info& f : list) {
        if (!f.valid) continue;
        picked += (uint64_t)f.length;
        f.selected = true;
        if (picked >= target) break;
    }
    std::sort(list.begin(), list.end(), [](const Info& x, const Info& y) { return x.id < y.id; });

    BgzfReader in(in_path, 8);
    BamHeader h;
    if (!in.ok() || !read_header(in, h, err)) { fprintf(stderr, "ERROR: %s: %s\n", in_path, err.c_str()); return EXIT_FAILURE; }
    BgzfWriter out(out_path, 8, 6);
    if (!out.ok()) { fprintf(stderr, "ERROR: cannot open %s for writing\n", out_path); return EXIT_FAILURE; }
    write_header(out, h);
    BamRecord r;
    size_t id = 0;
    int reads = 0;
    uint64_t bases = 0;
    while (read_record(in, r, err)) {
        if (id >= list.size()) { fprintf(stderr, "ERROR: %s changed between the two passes\n", in_path); return EXIT_FAILURE; }
        const Info& f = list[id++];
        if (!f.valid || !f.selected) continue;
        write_record(out, r);
        ++reads;
        bases += (uint64_t)f.length;
    }
    if (!err.empty() || !out.close()) { fprintf(stderr, "ERROR: %s\n", err.empty() ? "write failed" : err.c_str()); return EXIT_FAILURE; }
    fprintf(stderr, "Target: %s\nExtracted reads: %d (%s)\n", human_size(target).c_str(), reads, human_size(bases).c_str());
    return 0;
}

// eval [-s seed] [-d DUMP.json] [-g device] REF.fa bismark.bed mod.bam PREFIX : read-level benchmark samples
// (src/app/hifimeth/eval.cpp).  Truth labels come from a 0-based Bismark BED (rows with >= 10 reads: all unmethylated -> 0,
// all methylated -> 1, :103-112); every 5mC call of a mapped read that projects onto a labelled locus -- the same
// CpG / CHG / CHH walks as `pileup` (:503-560), here on the GPU through the pileup engine -- becomes one (label,
// probability) sample.  Per context: CHH negatives are thinned to one in ten (:556), small sample sets are replicated
// (:350-440), and five files PREFIX.<ctx>.<i> receive 100 000 positives and 100 000 negatives each, drawn without
// replacement, as "label<TAB>prediction<TAB>probability" (:580-611).  The reference draws with random_device seeds, so
// its files are not reproducible; the counts behind them are: `-d` (ours) writes the thresholds and the
// counts[context][label][scaled_prob] table before thinning, `-s` (ours) fixes the seed of every draw.
int cmd_eval(int argc, char** argv) {
    uint64_t seed = std::random_device{}();
    std::string dump_path;
    int device = 0;
    int a = 2;
    for (; a + 1 < argc && argv[a][0] == '-' && argv[a][1]; a += 2) {
        const std::string k = argv[a];
        if (k == "-s") seed = strtoull(argv[a + 1], nullptr, 10);
        else if (k == "-d") dump_path = argv[a + 1];
        else if (k == "-g") device = atoi(argv[a + 1]);
        else { fprintf(stderr, "ERROR: unrecognised option %s\n", argv[a]); return 1; }
    }
    if (argc - a != 4) {
        fprintf(stderr, "USAGE:\n%s %s [-s seed] [-d counts.json] [-g device] reference bismark mod-bam output-prefix\n", argv[0], argv[1]);
        return 1;
    }
    const char* ref_path = argv[a];
    const char* bed_path = argv[a + 1];
    const char* bam_path = argv[a + 2];
    const std::string prefix = argv[a + 3];
    constexpr uint64_t kTarget = 100000;
    static const char* cn[3] = {"CpG", "CHG", "CHH"};

    Fasta fa;
    std::string err;
    if (!load_fasta(ref_path, fa, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return EXIT_FAILURE; }
    if (fa.names.empty()) { fprintf(stderr, "ERROR: no sequence in %s\n", ref_path); return EXIT_FAILURE; }
    std::vector<int64_t> start(fa.names.size() + 1, 0);
    for (size_t s = 0; s < fa.names.size(); ++s) start[s + 1] = start[s] + fa.length[s];

    // truth labels (s_fill_chr_base_label_with_bismark, eval.cpp:42-114)
    std::vector<int8_t> labels(fa.bases.size(), (int8_t)-1);
    {
        gzFile in = gzopen(bed_path, "rb");
        if (!in) { fprintf(stderr, "ERROR: cannot open %s\n", bed_path); return EXIT_FAILURE; }
        static char line[1 << 16];
        std::string last;
        int sid = -1;
        size_t np = 0, nn = 0;
        while (gzgets(in, line, sizeof line)) {
            size_t ll = strlen(line);
            while (ll && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
            if (!ll) continue;
            char* col[6];
            int nc = 0;
            char* p = line;
            col[nc++] = p;
            for (; *p && nc < 6; ++p)
                if (*p == '\t') { *p = 0; col[nc++] = p + 1; }
            if (nc < 6) { fprintf(stderr, "ERROR: corrupted bismark record %s\n", line); gzclose(in); return EXIT_FAILURE; }
            if (sid < 0 || last != col[0]) {
                last = col[0];
                sid = fa.find(last);
                if (sid < 0) { fprintf(stderr, "ERROR: sequence %s is not in %s\n", col[0], ref_path); gzclose(in); return EXIT_FAILURE; }
            }
            const int64_t soff = atoll(col[1]), send = atoll(col[2]);
            if (send - soff != 1 || soff < 0 || soff >= fa.length[(size_t)sid]) {
                fprintf(stderr, "ERROR: bad interval %s:%s-%s\n", col[0], col[1], col[2]);
                gzclose(in);
                return EXIT_FAILURE;
            }
            const int pcov = atoi(col[4]), ncov = atoi(col[5]);
            if (pcov + ncov < 10) continue;
            if (pcov == 0) { labels[(size_t)(start[(size_t)sid] + soff)] = 0; ++nn; }
            else if (ncov == 0) { labels[(size_t)(start[(size_t)sid] + soff)] = 1; ++np; }
        }
        gzclose(in);
        fprintf(stderr, "Load %zu methylated sites and %zu unmethylated sites from %s\n", np, nn, bed_path);
    }

    BgzfReader in(bam_path, 8);
    BamHeader hdr;
    if (!in.ok() || !read_header(in, hdr, err)) { fprintf(stderr, "ERROR: %s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    std::vector<int> tid2sid(hdr.refs.size(), -2);
    hm_pileup_t* pe = nullptr;
    if (hm_pileup_create(&pe, device) != HM_OK) { fprintf(stderr, "ERROR: %s\n", hm_pileup_last_error(nullptr)); return EXIT_FAILURE; }
    auto die = [&](const std::string& what) {
        fprintf(stderr, "ERROR: %s: %s\n", what.c_str(), hm_pileup_last_error(pe));
        hm_pileup_destroy(pe);
        return EXIT_FAILURE;
    };
    if (hm_pileup_set_reference(pe, (int32_t)fa.names.size(), fa.length.data(), fa.bases.data()) != HM_OK) return die("reference");

    // one pass: the threshold histograms count every primary record with calls, mapped or not (s_prob_bin_thread,
    // eval.cpp:153-211) -- the engine counts the records it is given, the unmapped ones are counted here --; the samples
    // come from the mapped ones, without mapQ / identity filters (:484-489)
    static uint64_t extra[768];
    std::fill(extra, extra + 768, 0);
    constexpr int kBatch = 512;
    std::vector<BamRecord> recs((size_t)kBatch);
    std::vector<std::vector<BaseMod>> mods((size_t)kBatch);
    std::vector<std::string> perr((size_t)kBatch);
    std::vector<uint32_t> cig;
    uint64_t order = 0;
    bool more = true;
    while (more) {
        int n = 0;
        while (n < kBatch && (more = read_record(in, recs[(size_t)n], err))) ++n;
        if (!err.empty()) { fprintf(stderr, "ERROR: Could not read BAM record: %s\n", err.c_str()); hm_pileup_destroy(pe); return EXIT_FAILURE; }
        parallel_run(n, 8, [&](int k) {
            mods[(size_t)k].clear();
            perr[(size_t)k].clear();
            if (!parse_mods(recs[(size_t)k], mods[(size_t)k], perr[(size_t)k])) mods[(size_t)k].clear();
        });
        for (int k = 0; k < n; ++k, ++order) {
            const BamRecord& r = recs[(size_t)k];
            if (!perr[(size_t)k].empty()) {
                fprintf(stderr, "ERROR at parsing read %s\n%s\n", reinterpret_cast<const char*>(r.data.data() + 32), perr[(size_t)k].c_str());
                hm_pileup_destroy(pe);
                return EXIT_FAILURE;
            }
            if (mods[(size_t)k].empty()) continue;
            if (r.flag() & 4) {
                if (!(r.flag() & 0x900))
                    for (const BaseMod& m : mods[(size_t)k]) {
                        const int c = mod_context(r, m.qoff);
                        if (c >= 0) ++extra[c * 256 + m.prob];
                    }
                continue;
            }
            const int tid = r.ref_id();
            if (tid < 0 || tid >= (int)hdr.refs.size()) { fprintf(stderr, "ERROR: mapped record without a reference id\n"); hm_pileup_destroy(pe); return EXIT_FAILURE; }
            if (tid2sid[(size_t)tid] == -2) tid2sid[(size_t)tid] = fa.find(hdr.refs[(size_t)tid].first);
            if (tid2sid[(size_t)tid] < 0) {
                fprintf(stderr, "ERROR: Sequence name %s does not exist\n", hdr.refs[(size_t)tid].first.c_str());
                hm_pileup_destroy(pe);
                return EXIT_FAILURE;
            }
            real_cigar(r, cig);
            if (hm_pileup_submit_read(pe, (uint32_t)order, r.flag(), tid2sid[(size_t)tid], r.pos(), r.mapq(), r.l_qseq(), r.seq4(),
                                      (int32_t)cig.size(), cig.data(), (int64_t)mods[(size_t)k].size(), mods[(size_t)k].data()) < 0)
                return die(std::string("read ") + reinterpret_cast<const char*>(r.data.data() + 32));
        }
        if (hm_pileup_run(pe) != HM_OK) return die("projection");
    }
    static uint64_t bins[768];
    if (hm_pileup_histograms(pe, bins) != HM_OK) return die("histograms");
    for (int i = 0; i < 768; ++i) bins[i] += extra[i];
    uint8_t thr[3];
    report_thresholds(bins, thr);

    static uint64_t cnt[1536];  // [ctx][label][prob]
    if (hm_pileup_label_histograms(pe, labels.data(), (int64_t)labels.size(), cnt) != HM_OK) return die("labels");
    hm_pileup_destroy(pe);
    if (!dump_path.empty()) {
        FILE* f = fopen(dump_path.c_str(), "w");
        if (!f) { fprintf(stderr, "ERROR: cannot open %s for writing\n", dump_path.c_str()); return EXIT_FAILURE; }
        fprintf(f, "{\"thresholds\": [%d, %d, %d], \"counts\": [", thr[0], thr[1], thr[2]);
        for (int i = 0; i < 1536; ++i) fprintf(f, "%s%llu", i ? ", " : "", (unsigned long long)cnt[i]);
        fprintf(f, "]}\n");
        fclose(f);
    }

    std::mt19937_64 gen(seed);
    for (int c = 0; c < 3; ++c) {
        uint64_t* neg = cnt + (c * 2 + 0) * 256;
        uint64_t* pos = cnt + (c * 2 + 1) * 256;
        if (c == 2)  // every unmethylated CHH sample is kept with probability 0.1 (:556)
            for (int i = 0; i < 256; ++i)
                if (neg[i]) neg[i] = std::binomial_distribution<uint64_t>(neg[i], 0.1)(gen);
        uint64_t total[2] = {0, 0};
        for (int i = 0; i < 256; ++i) { total[0] += neg[i]; total[1] += pos[i]; }
        for (int l = 1; l >= 0; --l) {  // over_sampling_eval_samples (:350-440): positives first, as the messages come
            uint64_t* h = l ? pos : neg;
            if (total[l] > 0 && total[l] < kTarget) {
                fprintf(stderr, "Original %s %s samples: %llu\n", cn[c], l ? "positive" : "negative", (unsigned long long)total[l]);
                const uint64_t x = 2 * kTarget / total[l] * 2;
                for (int i = 0; i < 256; ++i) h[i] *= x;
                total[l] *= x;
                fprintf(stderr, "Over-sampled %s %s samples: %llu\n", cn[c], l ? "positive" : "negative", (unsigned long long)total[l]);
            }
        }
        if (total[0] == 0 || total[1] == 0) continue;
        fprintf(stderr, "%s positive samples: %llu, negative samples: %llu\n", cn[c], (unsigned long long)total[1], (unsigned long long)total[0]);
        // kTarget samples without replacement from a multiset given by its histogram: distinct ranks (Floyd), rank -> bin
        auto draw = [&](const uint64_t* h, uint64_t n, std::vector<uint8_t>& out) {
            std::vector<uint64_t> cum(257, 0);
            for (int i = 0; i < 256; ++i) cum[(size_t)i + 1] = cum[(size_t)i] + h[i];
            std::vector<uint64_t> picks;
            picks.reserve(kTarget);
            std::unordered_set<uint64_t> seen;
            seen.reserve(2 * kTarget);
            for (uint64_t j = n - kTarget; j < n; ++j) {
                const uint64_t t = std::uniform_int_distribution<uint64_t>(0, j)(gen);
                const uint64_t v = seen.insert(t).second ? t : j;
                if (v == j && t != j) seen.insert(j);
                picks.push_back(v);
            }
            std::shuffle(picks.begin(), picks.end(), gen);
            out.clear();
            for (uint64_t r : picks) out.push_back((uint8_t)(std::upper_bound(cum.begin(), cum.end(), r) - cum.begin() - 1));
        };
        std::vector<uint8_t> sp, sn;
        for (int i = 0; i < 5; ++i) {  // s_dump_samples (:580-611)
            const std::string path = prefix + "." + cn[c] + "." + std::to_string(i);
            FILE* f = fopen(path.c_str(), "w");
            if (!f) { fprintf(stderr, "ERROR: cannot open %s for writing\n", path.c_str()); return EXIT_FAILURE; }
            draw(pos, total[1], sp);
            draw(neg, total[0], sn);
            for (uint8_t v : sp) fprintf(f, "1\t%d\t%g\n", v >= thr[c] ? 1 : 0, 1.0 * v / 255);
            for (uint8_t v : sn) fprintf(f, "0\t%d\t%g\n", v >= thr[c] ? 1 : 0, 1.0 * v / 255);
            fclose(f);
        }
    }
    return 0;
}

// fastats REF.fa : names, lengths and a checksum of the loaded reference as one JSON object (loader tests; no GPU)
int cmd_fastats(int argc, char** argv) {
    if (argc != 3) return EXIT_FAILURE;
    Fasta fa;
    std::string err;
    if (!load_fasta(argv[2], fa, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return EXIT_FAILURE; }
    printf("{\"seqs\": [");
    size_t off = 0;
    for (size_t i = 0; i < fa.names.size(); ++i) {
        uint64_t h = 1469598103934665603ull;  // FNV-1a over the upper-cased bases
        for (int64_t k = 0; k < fa.length[i]; ++k) h = (h ^ (uint8_t)fa.bases[off + (size_t)k]) * 1099511628211ull;
        off += (size_t)fa.length[i];
        printf("%s{\"name\": \"%s\", \"length\": %lld, \"fnv1a\": \"%016llx\"}", i ? ", " : "", fa.names[i].c_str(),
               (long long)fa.length[i], (unsigned long long)h);
    }
    printf("]}\n");
    return 0;
}

// test seam: the threshold resolver of `pileup` / `eval` on histograms from stdin (n, then n x 3 x 256 counts -> n lines
// "cpg chg chh"); tests/golden/pileup_thresholds.json holds the reference's own answers for the same input format
int cmd_thresholds(int, char**) {
    int n = 0;
    if (scanf("%d", &n) != 1) return EXIT_FAILURE;
    for (int c = 0; c < n; ++c) {
        uint64_t bins[3][256];
        for (auto& b : bins)
            for (auto& v : b) {
                unsigned long long x = 0;
                if (scanf("%llu", &x) != 1) return EXIT_FAILURE;
                v = x;
            }
        uint64_t samples = 0;
        printf("%d %d %d\n", resolve_threshold(bins[0], &samples), resolve_threshold(bins[1], &samples), resolve_threshold(bins[2], &samples));
    }
    return 0;
}

namespace {
// s_resolve_scaled_prob_threshold (pileup.cpp:355-436, eval.cpp:213-305): the three thresholds with the reference's messages
void report_thresholds(const uint64_t* bins, uint8_t thr[3]) {
    static const char* cn[3] = {"CpG", "CHG", "CHH"};
    for (int c = 0; c < 3; ++c) {
        uint64_t samples = 0;
        const int t = resolve_threshold(bins + 256 * c, &samples);
        fprintf(stderr, "%s samples: %llu\n", cn[c], (unsigned long long)samples);
        const uint64_t* a = bins + 256 * c;  // the fallback branch: window narrower than 50 bins or < 10000 samples
        int st = 20, en = 256 - 20;
        while (st < 256 && a[st] < 10) ++st;
        while (en && a[en - 1] < 10) --en;
        const bool fallback = samples < 10000 || en - st < 50;
        if (fallback) fprintf(stderr, "Not enough samples for inferring scaled probability threshold, set it to 128\n");
        else fprintf(stderr, "%s scaled probability threshold: %d\n", cn[c], t);
        thr[c] = (uint8_t)t;
    }
}
}  // namespace

int cmd_pileup(int argc, char** argv) {
    PileupOptions o;
    int i = 2;
    for (; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h") { pileup_usage(argv[0]); return 0; }
        if (a.size() < 2 || a[0] != '-') break;
        if (i + 1 >= argc) { pileup_usage(argv[0]); return EXIT_FAILURE; }
        if (a == "-q") o.min_mapq = atoi(argv[++i]);
        else if (a == "-f") o.min_pi = atof(argv[++i]);
        else if (a == "-t") o.threads = std::max(1, atoi(argv[++i]));
        else if (a == "-d") o.device = atoi(argv[++i]);
        else if (a == "-b") o.read_batch = std::max(1, atoi(argv[++i]));
        else { fprintf(stderr, "ERROR: unrecognised option %s", a.c_str()); pileup_usage(argv[0]); return EXIT_FAILURE; }
    }
    if (argc - i != 3) { pileup_usage(argv[0]); return EXIT_FAILURE; }
    o.ref = argv[i];
    o.bam = argv[i + 1];
    o.prefix = argv[i + 2];
    fprintf(stderr, "\n\n====================> Parameters:\nmin-mapQ: %d\nmin-identity: %g\nCPU threads: %d\n"
                    "Genomic reference: %s\nmod-bam: %s\noutput prefix: %s\n\n\n",
            o.min_mapq, o.min_pi, o.threads, o.ref.c_str(), o.bam.c_str(), o.prefix.c_str());

    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_start = clk::now();
    double t_read = 0, t_parse = 0, t_submit = 0, t_gpu = 0;
    BgzfReader in(o.bam, o.threads);
    BamHeader hdr;
    std::string err;
    if (!in.ok() || !read_header(in, hdr, err)) { fprintf(stderr, "ERROR: %s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    if (!mapped_and_sorted(hdr)) return 1;

    Fasta fa;
    if (!load_fasta(o.ref, fa, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return EXIT_FAILURE; }
    fprintf(stderr, "Load %zu sequences (%zu bases) from %s\n", fa.names.size(), fa.bases.size(), o.ref.c_str());
    std::vector<int> tid2sid(hdr.refs.size(), -2);  // resolved at first use, as HbnDatabase::seq_name2id

    hm_pileup_t* pe = nullptr;
    if (hm_pileup_create(&pe, o.device) != HM_OK) { fprintf(stderr, "ERROR: %s\n", hm_pileup_last_error(nullptr)); return EXIT_FAILURE; }
    auto die = [&](const std::string& what) {
        fprintf(stderr, "ERROR: %s: %s\n", what.c_str(), hm_pileup_last_error(pe));
        hm_pileup_destroy(pe);
        return EXIT_FAILURE;
    };
    hm_pileup_set_option(pe, "min_mapq", o.min_mapq);
    hm_pileup_set_option(pe, "min_pi", o.min_pi);
    if (fa.names.empty()) { fprintf(stderr, "ERROR: no sequence in %s\n", o.ref.c_str()); hm_pileup_destroy(pe); return EXIT_FAILURE; }
    if (hm_pileup_set_reference(pe, (int32_t)fa.names.size(), fa.length.data(), fa.bases.data()) != HM_OK) return die("reference");

    // Two batches in flight: a producer thread inflates + parses batch k+1 (parse_mods over `threads` workers) while
    // this thread stages batch k and runs the GPU.
    struct Batch {
        std::vector<BamRecord> recs;
        std::vector<std::vector<BaseMod>> mods;
        std::vector<std::string> perr;
        int n = 0;
        bool more = true;
        std::string err;
        double t_read = 0, t_parse = 0;
    };
    Batch bb[2];
    for (Batch& b : bb) {
        b.recs.resize((size_t)o.read_batch);
        b.mods.resize((size_t)o.read_batch);
        b.perr.resize((size_t)o.read_batch);
    }
    auto produce = [&](Batch& b) {
        auto t0 = clk::now();
        b.n = 0;
        b.err.clear();
        while (b.n < o.read_batch && (b.more = read_record(in, b.recs[(size_t)b.n], b.err))) ++b.n;
        auto t1 = clk::now();
        parallel_run(b.n, o.threads, [&](int k) {
            b.mods[(size_t)k].clear();
            b.perr[(size_t)k].clear();
            if (!parse_mods(b.recs[(size_t)k], b.mods[(size_t)k], b.perr[(size_t)k])) b.mods[(size_t)k].clear();
        });
        b.t_read = secs(t0, t1);
        b.t_parse = secs(t1, clk::now());
    };
    uint64_t n_records = 0;
    int cur = 0;
    produce(bb[0]);
    std::vector<uint32_t> cig;
    while (true) {
        Batch& b = bb[cur];
        t_read += b.t_read;
        t_parse += b.t_parse;
        if (!b.err.empty()) { fprintf(stderr, "ERROR: Could not read BAM record: %s\n", b.err.c_str()); hm_pileup_destroy(pe); return EXIT_FAILURE; }
        std::thread producer;
        if (b.more) producer = std::thread(produce, std::ref(bb[cur ^ 1]));
        auto fail_out = [&]() { if (producer.joinable()) producer.join(); hm_pileup_destroy(pe); return EXIT_FAILURE; };
        auto t2 = clk::now();
        for (int k = 0; k < b.n; ++k) {
            const BamRecord& r = b.recs[(size_t)k];
            const uint64_t order = n_records++;
            if (!b.perr[(size_t)k].empty()) {
                fprintf(stderr, "ERROR at parsing read %s\n%s\n", reinterpret_cast<const char*>(r.data.data() + 32), b.perr[(size_t)k].c_str());
                return fail_out();
            }
            if (b.mods[(size_t)k].empty() || (r.flag() & 4)) continue;
            const int tid = r.ref_id();
            if (tid < 0 || tid >= (int)hdr.refs.size()) { fprintf(stderr, "ERROR: mapped record without a reference id\n"); return fail_out(); }
            if (tid2sid[(size_t)tid] == -2) tid2sid[(size_t)tid] = fa.find(hdr.refs[(size_t)tid].first);
            if (tid2sid[(size_t)tid] < 0) {
                fprintf(stderr, "ERROR: Sequence name %s does not exist\n", hdr.refs[(size_t)tid].first.c_str());
                return fail_out();
            }
            real_cigar(r, cig);
            const int rc = hm_pileup_submit_read(pe, (uint32_t)order, r.flag(), tid2sid[(size_t)tid], r.pos(), r.mapq(), r.l_qseq(),
                                                 r.seq4(), (int32_t)cig.size(), cig.data(), (int64_t)b.mods[(size_t)k].size(),
                                                 b.mods[(size_t)k].data());
            if (rc < 0) {
                fprintf(stderr, "ERROR: read %s: %s\n", reinterpret_cast<const char*>(r.data.data() + 32), hm_pileup_last_error(pe));
                return fail_out();
            }
        }
        auto t3 = clk::now();
        t_submit += secs(t2, t3);
        const int rrc = hm_pileup_run(pe);
        t_gpu += secs(t3, clk::now());
        if (producer.joinable()) producer.join();
        if (rrc != HM_OK) return die("projection");
        if (!b.more) break;
        cur ^= 1;
    }
    const auto t_loop = clk::now();

    static uint64_t bins[768];
    if (hm_pileup_histograms(pe, bins) != HM_OK) return die("histograms");
    uint8_t thr[3];
    static const char* cn[3] = {"CpG", "CHG", "CHH"};
    report_thresholds(bins, thr);
    if (hm_pileup_count(pe, thr) != HM_OK) return die("count");

    FILE* out[3];
    for (int c = 0; c < 3; ++c) {
        const std::string path = o.prefix + "." + cn[c] + ".cov.bed";
        out[c] = fopen(path.c_str(), "w");
        if (!out[c]) { fprintf(stderr, "ERROR: cannot open %s for writing\n", path.c_str()); hm_pileup_destroy(pe); return EXIT_FAILURE; }
    }
    std::vector<hm_locus_t> loci;
    const int fmt_threads = std::max(1, o.threads);
    std::vector<std::string> text((size_t)fmt_threads * 3);
    int64_t off = 0;
    for (size_t s = 0; s < fa.names.size(); ++s) {
        const int64_t lo = off, hi = off + fa.length[s];
        off = hi;
        int64_t n = hm_pileup_fetch_loci(pe, nullptr, nullptr, nullptr, 0, lo, hi, nullptr, 0);
        if (n < 0) return die("loci");
        if (n == 0) continue;
        loci.resize((size_t)n);
        n = hm_pileup_fetch_loci(pe, nullptr, nullptr, nullptr, 0, lo, hi, loci.data(), n);
        if (n < 0) return die("loci");
        // rows are formatted by `fmt_threads` workers over contiguous slices and written slice by slice (pileup.cpp:562-590)
        parallel_run(fmt_threads, fmt_threads, [&](int w) {
            for (int c = 0; c < 3; ++c) text[(size_t)w * 3 + c].clear();
            const size_t a = (size_t)n * w / fmt_threads, b = (size_t)n * (w + 1) / fmt_threads;
            char row[256];
            for (size_t i = a; i < b; ++i) {
                const hm_locus_t& l = loci[i];
                const int64_t k = l.gpos - lo;
                const double freq = 100.0 * l.pcov / (l.pcov + l.ncov);
                const int len = snprintf(row, sizeof row, "\t%lld\t%lld\t%g\t%d\t%d\n", (long long)k, (long long)k + 1, freq, l.pcov, l.ncov);
                std::string& t = text[(size_t)w * 3 + (l.motif < 3 ? l.motif : 2)];
                t += fa.names[s];
                t.append(row, (size_t)len);
            }
        });
        for (int c = 0; c < 3; ++c)
            for (int w = 0; w < fmt_threads; ++w) {
                const std::string& t = text[(size_t)w * 3 + c];
                if (!t.empty()) fwrite(t.data(), 1, t.size(), out[c]);
            }
    }
    for (FILE* f : out) fclose(f);
    hm_pileup_destroy(pe);
    fprintf(stderr, "## %llu records in %.2f s: [producer thread: BAM read %.2f s, MM/ML parse %.2f s] overlapped with "
                    "[staging %.2f s, GPU projection %.2f s]; thresholds + count + BED %.2f s\n",
            (unsigned long long)n_records, secs(t_start, clk::now()), t_read, t_parse, t_submit, t_gpu, secs(t_loop, clk::now()));
    return 0;
}
