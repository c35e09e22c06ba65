// Driver for the REFERENCE's own threshold resolver: s_resolve_scaled_prob_threshold (src/app/hifimeth/pileup.cpp:355-436)
// is a static function of pileup.cpp, so that translation unit is compiled IN PLACE (included from where it lies under
// /root/reference; nothing is copied) and called from here.  Everything else in it (sam_read1, sam_hdr_find_tag_hd, ...:
// htslib, whose library is not in this image) is unreferenced from main() and dropped by --gc-sections; no symbol is faked.
//   stdin : n, then n cases of 3 x 256 counts (CpG, CHG, CHH histograms of scaled probabilities)
//   stdout: n lines "cpg chg chh" (the thresholds)
#include "app/hifimeth/pileup.cpp"

#include <cstdio>

int main() {
    int n = 0;
    if (scanf("%d", &n) != 1) return 1;
    for (int c = 0; c < n; ++c) {
        static size_t bins[3][256];
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 256; ++i) {
                unsigned long long v = 0;
                if (scanf("%llu", &v) != 1) return 2;
                bins[k][i] = (size_t)v;
            }
        u8 t0 = 0, t1 = 0, t2 = 0;
        ns_pileup::s_resolve_scaled_prob_threshold(bins[0], bins[1], bins[2], t0, t1, t2);
        printf("%d %d %d\n", (int)t0, (int)t1, (int)t2);
    }
    return 0;
}
