/*
 * hm_oracle.c -- CPU restatement of the hifimeth `call` hot path (see hm_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product path.
 * Plain C (gcc), OpenMP over sites for the CNN so it can double as the timed CPU baseline.
 */
#include "hm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * a1: BamQuerySequence::init  (src/corelib/bam_info.cpp:169-222, base decode :100-167)
 * ---------------------------------------------------------------------------------------- */
static int nib_to_ascii(int c) { /* bam_info.cpp:100-121 ; anything else is HBN_ERR -> abort */
    switch (c) {
    case 1: return 'A';
    case 2: return 'C';
    case 4: return 'G';
    case 8: return 'T';
    case 15: return 'N';
    default: return -1;
    }
}

static int complement(int c) { /* bam_info.cpp:146-167 */
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    default: return 'A';
    }
}

int hmo_decode_read(const hmo_read_t* rd, char* fwd) {
    const int L = rd->l_qseq;
    for (int i = 0; i < L; ++i) {
        int nib = (rd->seq4[i >> 1] >> ((~i & 1) << 2)) & 0xf; /* sam.h bam_seqi */
        int c = nib_to_ascii(nib);
        if (c < 0) return -1;
        if (rd->flag & 16) /* stored sequence is the reverse strand: fwd = revcomp (bam_info.cpp:180-192) */
            fwd[L - 1 - i] = (char)complement(c);
        else
            fwd[i] = (char)c;
    }
    return 0;
}

/* IUPACNA_TO_BLASTNA restricted to what BAM can hold (src/corelib/hbn_aux.cpp:46-54) */
static inline int base_code(int c) {
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 14; /* N */
    }
}

/* ------------------------------------------------------------------------------------------
 * a2: kinetics codec  (src/corelib/bam_info.cpp:443-478, 520-570)
 * ---------------------------------------------------------------------------------------- */
void hmo_codev1_table(int32_t* t) { /* BamKinetics ctor, bam_info.cpp:562-570 */
    int p = 0;
    for (int i = 0; i < 64; ++i) t[p++] = i;
    for (int i = 64; i < 128; ++i) t[p++] = (i - 64) * 2 + 64;
    for (int i = 128; i < 192; ++i) t[p++] = (i - 128) * 4 + 192;
    for (int i = 192; i < 256; ++i) t[p++] = (i - 192) * 8 + 448;
}

int hmo_encode_frames(int s) { /* s_encode_signal_value, bam_info.cpp:455-478 */
    if (s > HMO_MAX_KINETIC) s = HMO_MAX_KINETIC;
    if (s >= 448) return (s - 448) / 8 + 192;
    if (s >= 192) return (s - 192) / 4 + 128;
    if (s >= 64) return (s - 64) / 2 + 64;
    return s;
}

int hmo_kinetic_code(const void* arr, int width, int idx) { /* BamKinetics::ipd / pw, bam_info.cpp:520-548 */
    if (width == 1) return ((const uint8_t*)arr)[idx];
    return hmo_encode_frames(((const uint16_t*)arr)[idx]);
}

/* ------------------------------------------------------------------------------------------
 * a3: motif tables  (src/corelib/5mc_context.cpp:3-54, 5mc_context.hpp:118-126)
 * ---------------------------------------------------------------------------------------- */
static int motif_hash(const char* s) { /* extract_motif_hash_value, 3-mer */
    int h = 0;
    for (int i = 0; i < 3; ++i) {
        int c = base_code(s[i]);
        if (c > 3) return 64;
        h = (h << 2) | c;
    }
    return h;
}

static const char* kFwdCHH[9] = {"CAA", "CCA", "CTA", "CAC", "CCC", "CTC", "CAT", "CCT", "CTT"};
static const char* kRevCHH[9] = {"TTG", "TGG", "TAG", "GTG", "GGG", "GAG", "ATG", "AGG", "AAG"};

static void chh_tables(uint8_t* fwd, uint8_t* rev) {
    memset(fwd, 255, 65);
    memset(rev, 255, 65);
    for (int i = 0; i < 9; ++i) {
        fwd[motif_hash(kFwdCHH[i])] = (uint8_t)i;
        rev[motif_hash(kRevCHH[i])] = (uint8_t)i;
    }
}

/* ------------------------------------------------------------------------------------------
 * a4-a6: site scanners  (src/app/hifimeth/eval_kmer_features.cpp:67-126)
 * ---------------------------------------------------------------------------------------- */
static int is3(const char* s, const char* m) { /* strncasecmp(s, m, 3) == 0 ; seq is upper-case here */
    for (int i = 0; i < 3; ++i) {
        int a = s[i], b = m[i];
        if (a >= 'a' && a <= 'z') a -= 32;
        if (a != b) return 0;
    }
    return 1;
}

int hmo_scan(const char* s, int L, int ctx, int32_t* out) {
    int n = 0;
    if (ctx == HMO_CPG) { /* :89-102 */
        for (int i = 0; i <= L - 2; ++i)
            if (s[i] == 'C' && s[i + 1] == 'G') out[n++] = i;
    } else if (ctx == HMO_CHG) { /* :104-126 */
        for (int i = 0; i <= L - 3; ++i)
            if (is3(s + i, "CCG") || is3(s + i, "CAG") || is3(s + i, "CTG")) out[n++] = i;
    } else { /* :67-87 -- note the non-monotonic emission: rev hit at i yields i+2 */
        uint8_t fwd[65], rev[65];
        chh_tables(fwd, rev);
        for (int i = 0; i <= L - 3; ++i) {
            int h = motif_hash(s + i);
            if (fwd[h] != 255) { out[n++] = i; continue; }
            if (rev[h] != 255) { out[n++] = i + 2; continue; }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * a7: window builder  (src/app/hifimeth/eval_kmer_features.cpp:9-65)
 * ---------------------------------------------------------------------------------------- */
void hmo_window(const hmo_read_t* rd, const char* fwd, int qoff, float* out, int* strand_out) {
    const int L = rd->l_qseq;
    const int HK = HMO_KMER / 2;
    int32_t dec[256];
    hmo_codev1_table(dec);
    int strand, off;
    if (fwd[qoff] == 'C') { strand = HMO_FWD; off = qoff; }
    else { strand = HMO_REV; off = L - 1 - qoff; } /* reference asserts fwd[qoff]=='G' here (:30) */
    const int qfrom = off >= HK ? off - HK : 0;
    const int qto = off + HK + 1 <= L ? off + HK + 1 : L;
    int fi = HK > off ? (HK - off) * HMO_FEATS : 0;
    memset(out, 0, sizeof(float) * HMO_WIN_FLOATS);
    /* K(s): FWD -> (fi, fp), REV -> (ri, rp), indexed exactly as stored in the tag (bam_info.cpp:520-548) */
    const void* ipd_s = strand == HMO_FWD ? rd->fi : rd->ri;
    const void* pw_s = strand == HMO_FWD ? rd->fp : rd->rp;
    const void* ipd_o = strand == HMO_FWD ? rd->ri : rd->fi;
    const void* pw_o = strand == HMO_FWD ? rd->rp : rd->fp;
    const int wi_s = strand == HMO_FWD ? rd->fi_w : rd->ri_w, wp_s = strand == HMO_FWD ? rd->fp_w : rd->rp_w;
    const int wi_o = strand == HMO_FWD ? rd->ri_w : rd->fi_w, wp_o = strand == HMO_FWD ? rd->rp_w : rd->fp_w;
    for (int i = qfrom; i < qto; ++i) {
        /* seq = fwd_qs or rev_qs ; rev[i] = comp(fwd[L-1-i]) (bam_info.cpp:193-206) */
        int b = strand == HMO_FWD ? base_code(fwd[i]) : base_code(complement(fwd[L - 1 - i]));
        for (int k = 0; k < 4; ++k) out[fi++] = (b == k) ? 1.0f : 0.0f; /* N: reference reads out of bounds; defined as 0 here */
        float v;
        v = (float)dec[hmo_kinetic_code(ipd_s, wi_s, i)]; v /= HMO_MAX_KINETIC; out[fi++] = v;
        v = (float)dec[hmo_kinetic_code(pw_s, wp_s, i)]; v /= HMO_MAX_KINETIC; out[fi++] = v;
        v = (float)dec[hmo_kinetic_code(ipd_o, wi_o, L - 1 - i)]; v /= HMO_MAX_KINETIC; out[fi++] = v;
        v = (float)dec[hmo_kinetic_code(pw_o, wp_o, L - 1 - i)]; v /= HMO_MAX_KINETIC; out[fi++] = v;
    }
    *strand_out = strand;
}

/* ------------------------------------------------------------------------------------------
 * a9: the CNN  (training/model_cnn.py:8-85 as exported to models/*.onnx)
 * ---------------------------------------------------------------------------------------- */
#define NCONV 8
static const int kChan[NCONV + 1] = {8, 128, 128, 128, 96, 96, 96, 64, 64};

struct hmo_model {
    int k[NCONV];
    int len[NCONV + 1]; /* 401, 197|196, ... , 2 */
    float bn_scale[8], bn_shift[8];
    float bn_gamma[8], bn_beta[8], bn_mean[8], bn_var[8], bn_eps;
    float* wt[NCONV]; /* [tap][cin][cout] */
    float* bias[NCONV];
    float* fc1_wt; /* [in=128][out=256] */
    float fc1_b[256];
    float fc2_w[2 * 256]; /* [out][in] */
    float fc2_b[2];
};

static int rd_f32(FILE* f, float* dst, size_t n) { return fread(dst, sizeof(float), n, f) == n ? 0 : -1; }

hmo_model_t* hmo_model_load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    char magic[4];
    int32_t hdr[4], chans[9], kern[8];
    hmo_model_t* m = (hmo_model_t*)calloc(1, sizeof *m);
    int ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "HMW1", 4) == 0 && fread(hdr, 4, 4, f) == 4 &&
             fread(chans, 4, 9, f) == 9 && fread(kern, 4, 8, f) == 8 && fread(&m->bn_eps, 4, 1, f) == 1;
    ok = ok && hdr[0] == HMO_KMER && hdr[1] == HMO_FEATS && hdr[3] == NCONV;
    for (int i = 0; ok && i <= NCONV; ++i) ok = chans[i] == kChan[i];
    if (ok) ok = !rd_f32(f, m->bn_gamma, 8) && !rd_f32(f, m->bn_beta, 8) && !rd_f32(f, m->bn_mean, 8) && !rd_f32(f, m->bn_var, 8);
    m->len[0] = HMO_KMER;
    for (int i = 0; ok && i < NCONV; ++i) {
        const int co = kChan[i + 1], ci = kChan[i], k = kern[i];
        m->k[i] = k;
        m->len[i + 1] = (m->len[i] + 2 - k) / 2 + 1;
        float* w = (float*)malloc(sizeof(float) * co * ci * k);
        m->wt[i] = (float*)malloc(sizeof(float) * co * ci * k);
        m->bias[i] = (float*)malloc(sizeof(float) * co);
        ok = !rd_f32(f, w, (size_t)co * ci * k) && !rd_f32(f, m->bias[i], co);
        if (ok)
            for (int o = 0; o < co; ++o)
                for (int c = 0; c < ci; ++c)
                    for (int t = 0; t < k; ++t) m->wt[i][((size_t)t * ci + c) * co + o] = w[((size_t)o * ci + c) * k + t];
        free(w);
    }
    if (ok) {
        float* w = (float*)malloc(sizeof(float) * 256 * 128);
        m->fc1_wt = (float*)malloc(sizeof(float) * 256 * 128);
        ok = !rd_f32(f, w, 256 * 128) && !rd_f32(f, m->fc1_b, 256) && !rd_f32(f, m->fc2_w, 512) && !rd_f32(f, m->fc2_b, 2);
        for (int o = 0; ok && o < 256; ++o)
            for (int i = 0; i < 128; ++i) m->fc1_wt[i * 256 + o] = w[o * 128 + i];
        free(w);
    }
    fclose(f);
    if (!ok) { hmo_model_free(m); return NULL; }
    return m;
}

void hmo_model_free(hmo_model_t* m) {
    if (!m) return;
    for (int i = 0; i < NCONV; ++i) { free(m->wt[i]); free(m->bias[i]); }
    free(m->fc1_wt);
    free(m);
}

int hmo_model_k1(const hmo_model_t* m) { return m->k[0]; }

/* Conv1d(stride 2, padding 1) + bias + ReLU, channels-last in/out.  PB positions x 32 couts per register block
 * (AVX-512: 6 x 2 zmm accumulators, AVX2: 2 x 4 ymm); rows in the padding read a zero row instead of branching.
 * Every output still accumulates bias, then tap 0 channel 0, 1, ... in the reference order: the blocking decides which
 * outputs share a pass over the weights, not the order of anyone's additions, so results do not depend on it. */
#if defined(__AVX512F__)
enum { CONV_PB = 6 };
#else
enum { CONV_PB = 2 };
#endif
static void conv_s2_relu(const float* in, int Lin, int Cin, const float* wt, const float* bias, int k, int Cout,
                         float* out, int Lout) {
    enum { PB = CONV_PB, CB = 32 };
    static const float zero_row[128] = {0};
    for (int p0 = 0; p0 < Lout; p0 += PB) {
        const int np = Lout - p0 < PB ? Lout - p0 : PB;
        for (int c0 = 0; c0 < Cout; c0 += CB) {
            float acc[PB][CB];
            for (int j = 0; j < PB; ++j)
                for (int o = 0; o < CB; ++o) acc[j][o] = bias[c0 + o];
            for (int t = 0; t < k; ++t) {
                const float* x[PB];
                for (int j = 0; j < PB; ++j) {
                    const int row = 2 * (p0 + j) - 1 + t; /* padding = 1 */
                    x[j] = (j < np && row >= 0 && row < Lin) ? in + (size_t)row * Cin : zero_row;
                }
                const float* w = wt + (size_t)t * Cin * Cout + c0;
                for (int c = 0; c < Cin; ++c, w += Cout) {
#pragma GCC unroll 8
                    for (int j = 0; j < PB; ++j) {
                        const float xv = x[j][c];
#pragma omp simd
                        for (int o = 0; o < CB; ++o) acc[j][o] += xv * w[o];
                    }
                }
            }
            for (int j = 0; j < np; ++j)
                for (int o = 0; o < CB; ++o) out[(size_t)(p0 + j) * Cout + c0 + o] = acc[j][o] > 0.f ? acc[j][o] : 0.f;
        }
    }
}

/* one site; a, b: scratch of >= 197*128 floats each. If layer in 1..8, copies that conv's output to dump. */
static void cnn_one(const hmo_model_t* m, const float* win, float* a, float* b, float* logits, int layer, float* dump) {
    /* bn0 over the 8 feature channels, eval mode (ONNX BatchNormalization, eps from the graph) */
    for (int r = 0; r < HMO_KMER; ++r)
        for (int c = 0; c < 8; ++c)
            a[r * 8 + c] = (win[r * 8 + c] - m->bn_mean[c]) / sqrtf(m->bn_var[c] + m->bn_eps) * m->bn_gamma[c] + m->bn_beta[c];
    float *src = a, *dst = b;
    for (int i = 0; i < NCONV; ++i) {
        conv_s2_relu(src, m->len[i], kChan[i], m->wt[i], m->bias[i], m->k[i], kChan[i + 1], dst, m->len[i + 1]);
        if (layer == i + 1 && dump) memcpy(dump, dst, sizeof(float) * m->len[i + 1] * kChan[i + 1]);
        float* t = src; src = dst; dst = t;
    }
    /* flatten of [C=64][L=2] is channel-major: idx = c*2 + l (model_cnn.py:79) ; src is [l][c] */
    float h[256];
    for (int o = 0; o < 256; ++o) h[o] = m->fc1_b[o];
    for (int c = 0; c < 64; ++c)
        for (int l = 0; l < 2; ++l) {
            const float xv = src[l * 64 + c];
            const float* w = m->fc1_wt + (size_t)(c * 2 + l) * 256;
            for (int o = 0; o < 256; ++o) h[o] += xv * w[o];
        }
    for (int o = 0; o < 256; ++o) h[o] = h[o] > 0.f ? h[o] : 0.f;
    for (int j = 0; j < 2; ++j) {
        float s = m->fc2_b[j];
        for (int o = 0; o < 256; ++o) s += h[o] * m->fc2_w[j * 256 + o];
        logits[j] = s;
    }
}

void hmo_cnn_logits(const hmo_model_t* m, const float* windows, int n, float* logits, int nthreads) {
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* a = (float*)malloc(sizeof(float) * 197 * 128 * 2);
        float* b = a + 197 * 128;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int i = 0; i < n; ++i) cnn_one(m, windows + (size_t)i * HMO_WIN_FLOATS, a, b, logits + 2 * (size_t)i, 0, NULL);
        free(a);
    }
}

int hmo_cnn_layer(const hmo_model_t* m, const float* window, int layer, float* out) {
    if (layer < 1 || layer > NCONV) return -1;
    float* a = (float*)malloc(sizeof(float) * 197 * 128 * 2);
    float lg[2];
    cnn_one(m, window, a, a + 197 * 128, lg, layer, out);
    free(a);
    return m->len[layer] * kChan[layer];
}

/* ------------------------------------------------------------------------------------------
 * a10: s_logits_to_methy_probs  (src/app/hifimeth/mod_batch.cpp:46-64)
 * ---------------------------------------------------------------------------------------- */
void hmo_softmax(const float* logits, int n, float* p, uint8_t* ml) {
    for (int i = 0; i < n; ++i) {
        const float v0 = logits[2 * i], v1 = logits[2 * i + 1];
        const float mx = v0 > v1 ? v0 : v1;
        const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
        const float p1 = e1 / (e0 + e1);
        int v = (int)(255 * p1);
        if (v > 255) v = 255;
        if (p) p[i] = p1;
        if (ml) ml[i] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------------------------------
 * per-read driver  (src/app/hifimeth/mod_main.cpp:180-212 ; batching of 32 is irrelevant to results)
 * ---------------------------------------------------------------------------------------- */
int hmo_call_read(hmo_model_t* const models[3], int ctx_mask, const hmo_read_t* rd, int min_len, int cap,
                  int32_t* qoff, uint8_t* strand, uint8_t* ctx, float* p, uint8_t* ml, int nthreads) {
    const int L = rd->l_qseq;
    if (L < min_len) return 0; /* mod_main.cpp:189-192 */
    char* fwd = (char*)malloc((size_t)L + 1);
    if (hmo_decode_read(rd, fwd) < 0) { free(fwd); return -1; }
    int32_t* offs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    int total = 0;
    for (int c = 0; c < 3; ++c) {
        if (!(ctx_mask >> c & 1)) continue;
        const int n = hmo_scan(fwd, L, c, offs);
        if (total + n > cap) { total = -1; break; }
        float* win = (float*)malloc(sizeof(float) * HMO_WIN_FLOATS * (size_t)(n ? n : 1));
        float* lg = (float*)malloc(sizeof(float) * 2 * (size_t)(n ? n : 1));
        for (int i = 0; i < n; ++i) {
            int s;
            hmo_window(rd, fwd, offs[i], win + (size_t)i * HMO_WIN_FLOATS, &s);
            qoff[total + i] = offs[i];
            strand[total + i] = (uint8_t)s;
            ctx[total + i] = (uint8_t)c;
        }
        hmo_cnn_logits(models[c], win, n, lg, nthreads);
        hmo_softmax(lg, n, p + total, ml + total);
        free(win);
        free(lg);
        total += n;
    }
    free(offs);
    free(fwd);
    return total;
}
