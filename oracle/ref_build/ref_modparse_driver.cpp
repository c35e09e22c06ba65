// Driver for the REFERENCE's own MM/ML parser core: s_parse_one_mod_list and its helpers (src/corelib/bam_mod_parser.cpp:
// 36-229) are static functions, so bam_mod_parser.cpp is compiled IN PLACE (included from where it lies under /root/reference;
// nothing is copied).  They read the record only through header macros (bam_get_qname, bam_get_seq, bam_seqi, bam_is_rev)
// and BamQuerySequence::get_bam_fwd_strand_base (src/corelib/bam_info.cpp:224-232, linked from its own source).  The two
// functions that fetch the tags out of the aux block (s_extract_bam_mod_scaled_probs, extract_bam_base_mods: bam_aux_get,
// bam_aux2Z, bam_auxB2i -- the htslib LIBRARY, not in this image) are unreferenced and dropped by --gc-sections; the driver
// hands the MM string and the ML bytes over directly and splits MM at ';' exactly as extract_bam_base_mods:276-285 does.
//   stdin : n, then per record: flag, SEQ as stored (ASCII), MM string, number of ML bytes and the bytes
//   stdout: per record: number of mods, then "qoff observed_strand unmod_base code scaled_prob" per mod
#include "corelib/bam_mod_parser.cpp"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

int main() {
    int n = 0;
    if (scanf("%d", &n) != 1) return 1;
    for (int r = 0; r < n; ++r) {
        int flag = 0, nml = 0;
        static char seq[1 << 20], mm[1 << 20];
        if (scanf("%d %1048575s %1048575s %d", &flag, seq, mm, &nml) != 4) return 2;
        std::vector<uint8_t> ml((size_t)nml);
        for (auto& v : ml) {
            int x = 0;
            if (scanf("%d", &x) != 1) return 3;
            v = (uint8_t)x;
        }
        // a BAM record with a name and the packed SEQ (sam.h layout: qname | cigar | seq | qual | aux)
        const int L = (int)strlen(seq);
        const char name[] = "rec";
        std::vector<uint8_t> data(4 + (size_t)(L + 1) / 2 + (size_t)L, 0);
        memcpy(data.data(), name, 4);
        for (int i = 0; i < L; ++i) {
            const char* p = strchr("=ACMGRSVTWYHKDBN", seq[i]);
            const int code = p ? (int)(p - "=ACMGRSVTWYHKDBN") : 15;
            data[4 + i / 2] |= (uint8_t)(code << ((~i & 1) << 2));
        }
        bam1_t b;
        memset(&b, 0, sizeof(b));
        b.core.flag = (uint16_t)flag;
        b.core.l_qname = 4;
        b.core.n_cigar = 0;
        b.core.l_qseq = L;
        b.data = data.data();
        b.l_data = (int)data.size();
        b.m_data = (uint32_t)data.size();
        std::vector<BaseModInfo> mods;
        const int mmsl = (int)strlen(mm);
        int prob_idx = 0, i = 0;
        while (i < mmsl) {  // extract_bam_base_mods:276-285
            int j = i + 1;
            while (j < mmsl && mm[j] != ';') ++j;
            ++j;
            s_parse_one_mod_list(&b, mm + i, j - i, ml.data(), nml, prob_idx, mods);
            i = j;
        }
        printf("%zu\n", mods.size());
        for (auto& m : mods) printf("%d %d %c %c %d\n", m.qoff, (int)m.observed_strand, m.unmod_base, m.code, (int)m.scaled_prob);
    }
    return 0;
}
