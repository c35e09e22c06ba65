/*
 * hm_oracle.h -- CPU restatement of the hifimeth `call` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (hifimeth_amd/ + libhifimeth_hip.so) never links or calls it.
 *
 * Parity status: PINNED.
 *   - site scanner: bit-exact against the reference's own scanner objects
 *     compiled from /root/reference (oracle/ref_build -> oracle/_ref/ref_scan)
 *     and against the golden site lists generated from that binary
 *     (tests/golden/scan_*.json, generator tools/make_golden.py);
 *   - window builder: against windows produced by the reference's Python
 *     training-time assembler (training/sample_dataset.py:84-139) imported in
 *     the build container (tests/golden/windows_*.npz);
 *   - CNN: against logits of the reference's shipped TorchScript models
 *     models/CpG.pt / CHH.pt run with torch.jit on CPU and against
 *     training/model_cnn.py:DNAModNet with seeded random weights
 *     (tests/golden/cnn_*.npz).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef HM_ORACLE_H
#define HM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMO_KMER 401
#define HMO_FEATS 8
#define HMO_WIN_FLOATS (HMO_KMER * HMO_FEATS)
#define HMO_MAX_KINETIC 952 /* src/corelib/bam_info.hpp:108 */

enum { HMO_CPG = 0, HMO_CHG = 1, HMO_CHH = 2 };
enum { HMO_FWD = 0, HMO_REV = 1 }; /* src/corelib/hbn_aux.hpp:60-63 */

/* One unaligned HiFi read as it sits in a BAM record (src/htslib/sam.h:267-325). */
typedef struct {
    int32_t l_qseq;
    int32_t flag;        /* only bit 0x10 matters (bam_info.cpp:180) */
    const uint8_t* seq4; /* 4-bit packed bases, high nibble first */
    const void* fi;      /* forward IPD, as stored in the tag      */
    const void* fp;      /* forward PW                              */
    const void* ri;      /* reverse IPD                             */
    const void* rp;      /* reverse PW                              */
    int32_t fi_w, fp_w, ri_w, rp_w; /* 1 = B:C codev1 bytes, 2 = B:S u16 frames */
} hmo_read_t;

typedef struct hmo_model hmo_model_t;

/* bam_info.cpp:100-222 : nibbles -> forward-strand ASCII. returns 0, or -1 on an illegal nibble */
int hmo_decode_read(const hmo_read_t* rd, char* fwd_ascii);

/* bam_info.cpp:443-453 (table), :455-478 (lossy u16 -> code) */
void hmo_codev1_table(int32_t* tbl256);
int hmo_encode_frames(int s);
/* code 0..255 of element idx of one stored kinetics array (bam_info.cpp:520-548) */
int hmo_kinetic_code(const void* arr, int width, int idx);

/* eval_kmer_features.cpp:67-126 : site offsets in the reference's emission order. returns count */
int hmo_scan(const char* fwd_ascii, int L, int ctx, int32_t* offsets);

/* eval_kmer_features.cpp:9-65 : the 401x8 fp32 window of the site at forward offset qoff */
void hmo_window(const hmo_read_t* rd, const char* fwd_ascii, int qoff, float* out, int* strand);

/* weights: flat .hmw container written by hifimeth_amd/onnx_weights.py from models/*.onnx */
hmo_model_t* hmo_model_load(const char* hmw_path);
void hmo_model_free(hmo_model_t* m);
int hmo_model_k1(const hmo_model_t* m);

/* training/model_cnn.py:75-85 as exported to models/*.onnx : windows [n][401][8] -> logits [n][2].
 * layer_out (optional, may be NULL): if non-NULL and n == 1, receives the post-ReLU
 * channels-last activations of conv `dump_layer` (1..8) -- used by the GPU debug tests. */
void hmo_cnn_logits(const hmo_model_t* m, const float* windows, int n, float* logits, int nthreads);
int hmo_cnn_layer(const hmo_model_t* m, const float* window, int layer, float* out);

/* mod_batch.cpp:46-64 : logits -> p = softmax[1] (float) and ML byte */
void hmo_softmax(const float* logits, int n, float* p, uint8_t* ml);

/* mod_main.cpp:180-212 for one read and the enabled contexts (bit c of ctx_mask).
 * Output in the reference's emission order (CpG, CHG, CHH; scanner order inside).
 * returns the number of sites, 0 when the read is skipped (l_qseq < min_len), -1 on error. */
int hmo_call_read(hmo_model_t* const models[3], int ctx_mask, const hmo_read_t* rd, int min_len,
                  int cap, int32_t* qoff, uint8_t* strand, uint8_t* ctx, float* p, uint8_t* ml,
                  int nthreads);

#ifdef __cplusplus
}
#endif
#endif
