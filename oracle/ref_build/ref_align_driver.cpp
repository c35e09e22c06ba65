// Driver (ours) around the REFERENCE's alignment projection classes; see Makefile for what is compiled.
// usage : ref_align <reference.fa> < records
//         ref_align -t <repeat> <reference.fa> < records   timing mode: no per-record output, one summary line
//                   "timing records <n> columns <c> repeat <r> seconds <s>" for BamQuerySequence::init +
//                   BamMapInfo::init + the three extract_*_mapped_samples calls (the CPU baseline of the projection)
// stdin : one record per line "<flag> <tid> <pos> <CIGAR> <SEQ>"  (SEQ over ACGTN, as stored in the BAM record)
// stdout: per record
//           "aln <qdir> <qb> <qe> <sid> <sb> <se> <as_size> <pi>"   BamMapInfo::init   (src/corelib/bam_info.cpp:373-439)
//           "qas <string>" "sas <string>" "qpos ..." "spos ..."     cigar_to_alignment (src/corelib/bam_info.cpp:262-371)
//           "cpg|chg|chh <n> <qoff>:<soff> ..."                     extract_*_mapped_samples (src/corelib/5mc_motif_finder.cpp)
//         or "unmapped" when BamMapInfo::init refuses the record.
#include <corelib/5mc_motif_finder.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void dump(const char* tag, const std::vector<MotifMappedHitInfo>& v) {
    printf("%s %zu", tag, v.size());
    for (auto& m : v) printf(" %d:%d", m.qoff, m.soff);
    printf("\n");
}

struct Rec {
    std::vector<uint8_t> data;
    bam1_t b;
};

static int timing_main(int repeat, const char* fasta);

int main(int argc, char* argv[]) {
    if (argc == 4 && strcmp(argv[1], "-t") == 0) return timing_main(atoi(argv[2]), argv[3]);
    if (argc != 2) return 2;
    HbnDatabase db(argv[1]);
    std::vector<char*> names;
    for (int i = 0; i < db.num_seqs(); ++i) names.push_back(const_cast<char*>(db.seq_name(i)));
    sam_hdr_t hdr;
    memset(&hdr, 0, sizeof hdr);
    hdr.n_targets = db.num_seqs();
    hdr.target_name = names.data();

    BamQuerySequence query;
    BamMapInfo align;
    MethylationContext ctx;
    std::vector<MotifMappedHitInfo> hits;
    static char line[1 << 22];
    while (fgets(line, sizeof line, stdin)) {
        int flag = 0, tid = 0, pos = 0, off = 0;
        static char cig[1 << 20];
        if (sscanf(line, "%d %d %d %s %n", &flag, &tid, &pos, cig, &off) < 4) continue;
        const char* s = line + off;
        int L = (int)strlen(s);
        while (L && (s[L - 1] == '\n' || s[L - 1] == '\r')) --L;
        std::vector<uint32_t> ops;
        for (const char* p = cig; *p;) {
            char* e;
            long n = strtol(p, &e, 10);
            const char* k = strchr(BAM_CIGAR_STR, *e);
            ops.push_back((uint32_t)n << BAM_CIGAR_SHIFT | (uint32_t)(k - BAM_CIGAR_STR));
            p = e + 1;
        }
        // in-memory bam1_t: qname | cigar | 4-bit seq | qual   (src/htslib/sam.h:267-325)
        std::vector<uint8_t> data(4 + 4 * ops.size() + (L + 1) / 2 + L, 0);
        data[0] = 'q';
        memcpy(data.data() + 4, ops.data(), 4 * ops.size());
        uint8_t* seq = data.data() + 4 + 4 * ops.size();
        for (int i = 0; i < L; ++i) {
            int c = s[i] == 'A' ? 1 : s[i] == 'C' ? 2 : s[i] == 'G' ? 4 : s[i] == 'T' ? 8 : 15;
            seq[i >> 1] |= (i & 1) ? c : (c << 4);
        }
        memset(seq + (L + 1) / 2, 0xff, L);
        bam1_t b;
        memset(&b, 0, sizeof b);
        b.core.l_qname = 4;
        b.core.l_qseq = L;
        b.core.flag = (uint16_t)flag;
        b.core.tid = tid;
        b.core.pos = pos;
        b.core.n_cigar = (uint32_t)ops.size();
        b.data = data.data();
        b.l_data = (int)data.size();
        b.m_data = (uint32_t)data.size();
        query.init(&b);
        if (!align.init(&hdr, &b, &db, &query)) {
            printf("unmapped\n");
            continue;
        }
        printf("aln %d %d %d %d %d %d %d %.17g\n", align.qdir, align.qb, align.qe, align.sid, align.sb, align.se,
               align.as_size, align.pi);
        printf("qas %.*s\nsas %.*s\n", align.as_size, align.qas, align.as_size, align.sas);
        printf("qpos");
        for (int i = 0; i < align.as_size; ++i) printf(" %d", align.qas_pos[i]);
        printf("\nspos");
        for (int i = 0; i < align.as_size; ++i) printf(" %d", align.sas_pos[i]);
        printf("\n");
        extract_cpg_mapped_samples(&db, query, align, hits); dump("cpg", hits);
        extract_chg_mapped_samples(&db, ctx, query, align, hits); dump("chg", hits);
        extract_chh_mapped_samples(&db, ctx, query, align, hits); dump("chh", hits);
    }
    return 0;
}

// reads every record into an in-memory bam1_t first, then times the reference's projection code over them
static int timing_main(int repeat, const char* fasta) {
    HbnDatabase db(fasta);
    std::vector<char*> names;
    for (int i = 0; i < db.num_seqs(); ++i) names.push_back(const_cast<char*>(db.seq_name(i)));
    sam_hdr_t hdr;
    memset(&hdr, 0, sizeof hdr);
    hdr.n_targets = db.num_seqs();
    hdr.target_name = names.data();
    std::vector<Rec*> recs;
    static char line[1 << 22];
    static char cig[1 << 20];
    while (fgets(line, sizeof line, stdin)) {
        int flag = 0, tid = 0, pos = 0, off = 0;
        if (sscanf(line, "%d %d %d %s %n", &flag, &tid, &pos, cig, &off) < 4) continue;
        const char* s = line + off;
        int L = (int)strlen(s);
        while (L && (s[L - 1] == '\n' || s[L - 1] == '\r')) --L;
        std::vector<uint32_t> ops;
        for (const char* p = cig; *p;) {
            char* e;
            long n = strtol(p, &e, 10);
            const char* k = strchr(BAM_CIGAR_STR, *e);
            ops.push_back((uint32_t)n << BAM_CIGAR_SHIFT | (uint32_t)(k - BAM_CIGAR_STR));
            p = e + 1;
        }
        Rec* r = new Rec;
        r->data.assign(4 + 4 * ops.size() + (L + 1) / 2 + L, 0);
        r->data[0] = 'q';
        memcpy(r->data.data() + 4, ops.data(), 4 * ops.size());
        uint8_t* seq = r->data.data() + 4 + 4 * ops.size();
        for (int i = 0; i < L; ++i) {
            int c = s[i] == 'A' ? 1 : s[i] == 'C' ? 2 : s[i] == 'G' ? 4 : s[i] == 'T' ? 8 : 15;
            seq[i >> 1] |= (i & 1) ? c : (c << 4);
        }
        memset(seq + (L + 1) / 2, 0xff, L);
        memset(&r->b, 0, sizeof r->b);
        r->b.core.l_qname = 4;
        r->b.core.l_qseq = L;
        r->b.core.flag = (uint16_t)flag;
        r->b.core.tid = tid;
        r->b.core.pos = pos;
        r->b.core.n_cigar = (uint32_t)ops.size();
        r->b.data = r->data.data();
        r->b.l_data = (int)r->data.size();
        r->b.m_data = (uint32_t)r->data.size();
        recs.push_back(r);
    }
    BamQuerySequence query;
    BamMapInfo align;
    MethylationContext ctx;
    std::vector<MotifMappedHitInfo> hits;
    long long columns = 0, samples = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < repeat; ++it)
        for (Rec* r : recs) {
            query.init(&r->b);
            if (!align.init(&hdr, &r->b, &db, &query)) continue;
            columns += align.as_size;
            extract_cpg_mapped_samples(&db, query, align, hits); samples += (long long)hits.size();
            extract_chg_mapped_samples(&db, ctx, query, align, hits); samples += (long long)hits.size();
            extract_chh_mapped_samples(&db, ctx, query, align, hits); samples += (long long)hits.size();
        }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("timing records %zu columns %lld samples %lld repeat %d seconds %.6f\n", recs.size(), columns, samples, repeat, secs);
    return 0;
}
