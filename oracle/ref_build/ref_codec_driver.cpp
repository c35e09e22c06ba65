// Driver for the REFERENCE's own kinetics codec: the codev1 decode table built by BamKinetics::BamKinetics()
// (src/corelib/bam_info.cpp:568-576) and the lossy frame-count encoder s_encode_signal_value (:455-478), a static function,
// so bam_info.cpp is compiled IN PLACE (included from where it lies under /root/reference; nothing is copied).  The accessors
// that call the htslib library (bam_aux_get, ...) are unreferenced from main() and dropped by --gc-sections; nothing is faked.
//   stdout: line 1 = the 256 decoded values; line 2 = encode(s) for s = 0 .. 1199; line 3 = encode(s) for s in {2000, 4095, 65535}
#include "corelib/bam_info.cpp"

#include <cstdio>

int main() {
    BamKinetics k;
    const int* t = k.codev1_table();
    for (int i = 0; i < 256; ++i) printf("%d%c", t[i], i == 255 ? '\n' : ' ');
    for (int s = 0; s < 1200; ++s) printf("%d%c", s_encode_signal_value(s), s == 1199 ? '\n' : ' ');
    printf("%d %d %d\n", s_encode_signal_value(2000), s_encode_signal_value(4095), s_encode_signal_value(65535));
    return 0;
}
