// Driver for the REFERENCE's own logits -> ML byte step: s_logits_to_methy_probs (src/app/hifimeth/mod_batch.cpp:46-64) is a
// static function of mod_batch.cpp, so that translation unit is compiled IN PLACE (included from where it lies under
// /root/reference; nothing is copied).  The rest of it (ModBatch: OpenVINO InferRequest, whose library is not in this image;
// only its headers are vendored) is unreferenced from main() and dropped by --gc-sections; no symbol is faked.
//   stdin : n, then n pairs of float bit patterns (hex)      stdout: n scaled_prob bytes, one per line
#include "app/hifimeth/mod_batch.cpp"

#include <cstdio>
#include <cstring>
#include <vector>

int main() {
    int n = 0;
    if (scanf("%d", &n) != 1) return 1;
    std::vector<float> lg(2 * (size_t)n);
    for (auto& v : lg) {
        unsigned u = 0;
        if (scanf("%x", &u) != 1) return 2;
        memcpy(&v, &u, 4);
    }
    std::vector<MolMethyCall> calls((size_t)n);
    ns_mods::s_logits_to_methy_probs(lg.data(), (size_t)n, calls.data());
    for (int i = 0; i < n; ++i) printf("%d\n", (int)calls[i].scaled_prob);
    return 0;
}
