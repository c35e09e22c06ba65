// Driver (ours) around the REFERENCE's scanner classes; see Makefile for what is compiled.
// stdin : one record per line "<flag> <SEQ>"  (SEQ over ACGTN, as stored in the BAM record)
// stdout: per record three lines "cpg|chg|chh <n> <off>..." in the reference's emission order,
//         preceded by "fwd <forward-strand sequence>" as BamQuerySequence::init derives it.
#include <app/hifimeth/eval_kmer_features.hpp>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static void dump(const char* tag, const ns_mods::EvalKmerFeaturesGenerator& g) {
    printf("%s %d", tag, g.M_num_samples);
    for (int i = 0; i < g.M_num_samples; ++i) printf(" %d", g.M_sample_offsets[i]);
    printf("\n");
}

int main() {
    ns_mods::EvalKmerFeaturesGenerator g;
    static char line[1 << 22];
    while (fgets(line, sizeof line, stdin)) {
        int flag = 0, pos = 0;
        if (sscanf(line, "%d %n", &flag, &pos) < 1) continue;
        const char* s = line + pos;
        int L = (int)strlen(s);
        while (L && (s[L - 1] == '\n' || s[L - 1] == '\r')) --L;
        // in-memory bam1_t: qname | (no cigar) | 4-bit seq | qual   (src/htslib/sam.h:267-325)
        std::vector<uint8_t> data(2 + (L + 1) / 2 + L, 0);
        data[0] = 'q';
        uint8_t* seq = data.data() + 2;
        for (int i = 0; i < L; ++i) {
            int c = s[i] == 'A' ? 1 : s[i] == 'C' ? 2 : s[i] == 'G' ? 4 : s[i] == 'T' ? 8 : 15;
            seq[i >> 1] |= (i & 1) ? c : (c << 4);
        }
        memset(seq + (L + 1) / 2, 0xff, L);
        bam1_t b;
        memset(&b, 0, sizeof b);
        b.core.l_qname = 2;
        b.core.l_qseq = L;
        b.core.flag = (uint16_t)flag;
        b.data = data.data();
        b.l_data = (int)data.size();
        b.m_data = (uint32_t)data.size();
        g.M_query.init(&b);
        printf("fwd %.*s\n", L, g.M_query.fwd_rqs);
        g.extract_cpg_samples(); dump("cpg", g);
        g.extract_chg_samples(); dump("chg", g);
        g.extract_chh_samples(); dump("chh", g);
    }
    return 0;
}
