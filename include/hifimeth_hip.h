/*
 * hifimeth_hip.h -- C ABI of libhifimeth_hip.so, the MI355X (gfx950) engine for the
 * `hifimeth call` hot path: per-read CpG/CHG/CHH site scan -> 401x8 kinetics window -> CNN ->
 * per-site 5mC probability.
 *
 * The reference has no plugin/FFI layer; this library sits where three C++ classes of the
 * reference sit inside its worker thread (src/app/hifimeth/mod_main.cpp:145-262):
 *
 *   ns_mods::ModModels                  (mod_main.cpp:18-99)         -> hm_create / hm_destroy
 *   ns_mods::EvalKmerFeaturesGenerator  (eval_kmer_features.hpp:13-49)
 *       init(bam1_t*)                                                -> hm_submit_read
 *       extract_{cpg,chg,chh}_samples()                              -> hm_run, hm_scan_sites
 *       get_next_sample_features()                                   -> hm_windows
 *   ns_mods::ModBatch                   (mod_batch.hpp:12-43)
 *       call_mods_for_one_read / call_current_batch                  -> hm_run, hm_cnn_logits
 *       results appended to std::vector<MolMethyCall>                -> hm_fetch / hm_drain
 *
 * Plain pointers and sizes only; every call returns >= 0 on success and a negative HM_E* code
 * on failure (the reference abort()s instead: src/corelib/hbn_aux.hpp:100-104), with the message
 * available from hm_last_error().
 *
 * Two ways to drive an engine:
 *  - the synchronous calls (hm_submit_read ... hm_fetch): ONE implicit batch, one host thread; what the parity tests use;
 *  - the batch pipeline (hm_batch_*): N slots per engine, each with pinned staging memory and its own copy stream.
 *    Staging of different batches may run on different host threads (the reference's workers pull reads from a shared
 *    queue: src/corelib/sam_batch.hpp:38-54); hm_batch_enqueue queues H2D -> scanner -> CNN -> results and returns
 *    without any host/device synchronisation, so batch k+1 is staged and uploaded while batch k computes (the pinned
 *    non-blocking staging of the reference's GPU variant, src/app-gpu/hifimeth-gpu/5mc_call_gpu.cpp:309-334,367).
 */
#ifndef HIFIMETH_HIP_H
#define HIFIMETH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM_CTX_CPG 0
#define HM_CTX_CHG 1
#define HM_CTX_CHH 2
#define HM_CTX_ALL 3 /* only as the `ctx` argument of hm_num_sites */

#define HM_OK 0
#define HM_EINVAL (-1)  /* bad argument                                             */
#define HM_EMODEL (-2)  /* model file missing / ill-formed (mod_main.cpp:40-52)     */
#define HM_EDEVICE (-3) /* HIP runtime error or no gfx950 device                    */
#define HM_EDATA (-4)   /* illegal base nibble in a read (bam_info.cpp:100-121)     */
#define HM_ESTATE (-5)  /* call out of order (e.g. hm_fetch before hm_run)          */
#define HM_ENOMEM (-6)

typedef struct hm_engine hm_engine_t;

/* One call = the reference's MolMethyCall (src/corelib/5mc_motif_finder.hpp:8-14) plus the
 * context and the float probability the reference never exposes (needed for the |dp| check). */
typedef struct {
    int32_t read_id;     /* qid: the id given to hm_submit_read                           */
    int32_t qoff;        /* forward-strand offset of the cytosine (of the G for strand 1) */
    uint8_t strand;      /* 0 = FWD, 1 = REV (src/corelib/hbn_aux.hpp:60-63)              */
    uint8_t ctx;         /* HM_CTX_*                                                      */
    uint8_t scaled_prob; /* min(255, (int)(255 * p))  (mod_batch.cpp:46-64)               */
    uint8_t reserved;
    float p;             /* softmax(logits)[1]                                            */
} hm_call_t;

/* Accumulated device time per kernel class since the last hm_reset_timing (HIP events on the
 * engine's stream; only collected when option "timing" is 1). */
typedef struct {
    double prep_ms, scan_ms, emit_ms, window_ms;
    double front_ms[3], tail_ms[3];
    int64_t prep_launches, scan_launches, emit_launches, window_launches;
    int64_t front_launches[3], tail_launches[3];
    int64_t front_sites[3]; /* sites processed by the timed front launches */
    int64_t window_sites;
    double pack_ms, empty_ms; /* result packing; CNN launches whose window of the site list turned out empty */
    int64_t pack_launches, empty_launches;
    double trunk_ms[3], edge_ms[3]; /* dense trunk (conv1..conv4 over whole reads) and window-edge kernels */
    int64_t trunk_launches[3], edge_launches[3];
    int64_t trunk_positions[3]; /* (read, strand view) positions evaluated by the timed trunk launches */
    int64_t trunk_list_steps[3]; /* tiles whose conv4 ran over the listed (needed) rows only: 4 m-tiles instead of 7 (sliding-window trunk) */
    int64_t trunk_const_steps[3]; /* tiles stored as constant rows instead of computed: a read's first tile and those behind its end, where no receptive field reaches the read */
    int64_t group_bases;        /* bases per trunk read group in force (option "group_bases", or what the engine sized from free memory) */
    int64_t group_bytes;        /* device bytes the engine holds for a read group's maps, edge rows and hand-off buffers */
    int64_t tail_strip_passes;  /* passes (16 site slots each) of the strip tail kernel (tail_impl 3, CHH): MFMAs issued = passes x a pass's count */
} hm_timing_t;

/* Version of this header's structs (hm_timing_t grows round by round: 4 = round 4, 5 = + tail_strip_passes).  hm_abi_version() returns what the
 * LIBRARY was built with and hm_timing_size() its sizeof(hm_timing_t): a consumer compares both with its own before calling hm_get_timing. */
#define HM_ABI_VERSION 5
int hm_abi_version(void);
size_t hm_timing_size(void);

/* ---- lifetime ---------------------------------------------------------------------------- */
/* model_dir holds {CpG,CHG,CHH}.hmw (flat fp32 container written from the reference's
 * models/{CpG,CHG,CHH}.onnx) or the .onnx files themselves; ctx_mask bit c enables context c
 * (the reference's -c cpg,chg,chh: mod_options.cpp:61-134); device = HIP device ordinal.   */
int hm_create(hm_engine_t** out, const char* model_dir, int ctx_mask, int device);
void hm_destroy(hm_engine_t* e);
const char* hm_last_error(const hm_engine_t* e); /* e may be NULL: error of a failed hm_create */
/* options: "slots" (batches in flight of the hm_batch_* pipeline, default 3), "min_read_size" (-l, default 1000), "timing" (0/1), "sub_batch_sites" (front/tail
 * launch granularity, default 65536), "front_waves" (4 or 8 waves per front workgroup), "precision" (0 = fp32 MFMA, exact;
 * 1 = split-half fp16x3 MFMA with fp32 accumulate (default, |dp| <= 1e-4); 2 = as 1 with plain fp16 WEIGHTS in conv8 and fc1 (their
 * w_lo x_hi product and lo-plane fetches dropped): the part of BASELINE.json configs[4] that holds its bar |dp| <= 1e-3 with margin;
 * as written -- fp16 weights in every layer -- that configuration reaches 2.5e-3 (profiles/r02_term_error_table.txt) and stays closed: 3 is an error), "trunk" (2 = per context by the site density of the FIRST batch the engine is given -- counted
 * on the host when that batch is queued and then fixed for the engine's lifetime, so the calls never depend on host timing --
 * default; 1 = conv1..conv4 once per read position; 0 = once per site; every precision has both forms), "trunk_mask" (0..7: that
 * choice made by the caller, see hm_trunk_mask_for_reads), "trunk_impl" (3 = the streaming trunk as a sliding window over
 * consecutive tiles, default; 1 = streaming 4-wave trunk kernel; 2 = the same on 8 waves; 0 = the 8-wave ConvH form; byte-identical results), "edge_impl" (1 = edge2_kernel, default; 0 = round 2's
 * edge_kernel; byte-identical), "tail_impl" (3 = the strip tail for CHH (16 sites of one E4-row residue class per pass share one strip of rows in LDS; hm_tail_p.hip) and 1 for the sparse contexts, default; 1 = tail with register-resident weights; 2 = the split tail: conv5 + conv6, then conv7 .. softmax over 16 sites per pass ("tail_slice": sites per launch pair); 0 = the streaming tail; byte-identical),
 * "group_bases" (reads per trunk group; default 0 = sized when the first read is staged so that a group's buffers, 5.8 KB per base, take at most a quarter of the device's free memory, and at most 16 Mi bases), "num_cu" (workgroups of the persistent kernels), "conv3_w16" (diagnostic: conv3 of the dense trunk with plain fp16 weights -- measured ABOVE the 1e-3 bar,
 * part of no mode), "stamps" (diagnostic) */
int hm_set_option(hm_engine_t* e, const char* key, int64_t value);

/* ---- staging: the EvalKmerFeaturesGenerator::init seam ----------------------------------- */
/* Copies one read into the pinned staging slab exactly as the BAM record stores it: 4-bit
 * packed SEQ, and the fi/fp/ri/rp B-arrays with element width 1 (B:C codev1) or 2 (B:S frames).
 * A NULL array means the tag is missing.  Returns 1 if the read was accepted, 0 if it is passed
 * through uncalled (l_qseq < min_read_size or a missing tag: mod_main.cpp:189-196), < 0 on error.
 * The caller keeps ownership of all pointers.                                                */
int hm_submit_read(hm_engine_t* e, int32_t read_id, int32_t l_qseq, int32_t flag, const uint8_t* seq4,
                   const void* fi, int fi_width, const void* fp, int fp_width, const void* ri, int ri_width,
                   const void* rp, int rp_width);
int hm_clear(hm_engine_t* e); /* forget the staged / resident batch */

/* ---- execution --------------------------------------------------------------------------- */
int hm_upload(hm_engine_t* e); /* staged slab -> HBM (async on the engine stream)              */
int hm_run(hm_engine_t* e);    /* scanner + window builder + CNN over the resident batch        */
int hm_sync(hm_engine_t* e);   /* wait for everything queued; reports device-side data errors   */
int64_t hm_num_sites(hm_engine_t* e, int ctx); /* after hm_run: the reference's processed_*_samples */
/* D2H of the results of the last hm_run, ordered by (read submission order, strand, qoff) -- the
 * order build_one_mod_bam needs (mod_main.cpp:217-251).  Returns the number of calls written. */
int64_t hm_fetch(hm_engine_t* e, hm_call_t* out, int64_t cap);
/* convenience: hm_flush = hm_upload + hm_run ; hm_drain = hm_sync + hm_fetch + hm_clear */
int hm_flush(hm_engine_t* e);
int64_t hm_drain(hm_engine_t* e, hm_call_t* out, int64_t cap);

/* ---- the asynchronous batch pipeline -------------------------------------------------------- */
typedef struct hm_batch hm_batch_t;
/* A free slot in STAGING state; blocks while all "slots" are staged or in flight.  NULL on error. */
hm_batch_t* hm_batch_begin(hm_engine_t* e);
/* hm_submit_read into this batch's pinned slab; batches may be staged concurrently from different threads */
int hm_batch_submit_read(hm_batch_t* b, int32_t read_id, int32_t l_qseq, int32_t flag, const uint8_t* seq4,
                         const void* fi, int fi_width, const void* fp, int fp_width, const void* ri, int ri_width,
                         const void* rp, int rp_width);
/* Bulk form: n reads in one call, copied into the slab by `threads` host threads (the per-read layout is fixed by a
 * serial pass first, so the result is identical to n hm_batch_submit_read calls in order).  A read that would be passed
 * through uncalled is skipped exactly as there; accepted[i] (may be NULL) tells which.  Returns the number accepted.
 * All n reads are validated before the first is placed: on an error (HM_EINVAL, HM_ENOMEM = the batch would pass 2^31
 * bases) the batch is exactly as it was before the call. */
typedef struct {
    int32_t read_id, l_qseq, flag;
    uint8_t width[4];     /* element width of fi, fp, ri, rp: 1 (B:C) or 2 (B:S) */
    const uint8_t* seq4;
    const void* kin[4];   /* fi, fp, ri, rp; NULL = tag missing */
} hm_read_t;
int64_t hm_batch_submit_reads(hm_batch_t* b, const hm_read_t* reads, int64_t n, int threads, uint8_t* accepted);
/* The per-context choice the engine makes under "trunk" = 2, as a function of a sample of reads (host only, no device work,
 * only seq4 / l_qseq are read): bit c set = context c takes the dense trunk.  A front end that shards one input over several
 * engines or ranks passes the SAME sample (the head of the file) everywhere and sets the result with the "trunk_mask" option,
 * so that every shard computes with the same kernels and the merged output equals the single-process output byte for byte
 * (the reference's output is deterministic: mod_main.cpp:330-362). */
int hm_trunk_mask_for_reads(const hm_read_t* reads, int64_t n, int ctx_mask);
int64_t hm_batch_staged_bases(const hm_batch_t* b);
/* Queues the batch: async H2D on the slot's stream, scanner + CNN + result packing on the engine's compute stream (site
 * counts are consumed on the device), async D2H of the totals.  Returns at once; batches compute in queueing order. */
int hm_batch_enqueue(hm_batch_t* b);
int hm_batch_done(hm_batch_t* b); /* 1 finished, 0 still in flight, < 0 error; never blocks */
/* Waits for THIS batch only.  Returns its number of calls; with calls != NULL also brings them to the host with one
 * packed D2H and points *calls at them (pinned memory owned by the slot, valid until hm_batch_release), ordered by
 * (read submission order, strand, qoff) like hm_fetch.  HM_EDATA if a staged read held an illegal base. */
int64_t hm_batch_wait(hm_batch_t* b, const hm_call_t** calls);
int64_t hm_batch_num_sites(hm_batch_t* b, int ctx); /* waits like hm_batch_wait(b, NULL) */
int hm_batch_release(hm_batch_t* b); /* the slot may be handed out again */

/* ---- seams used by the parity tests and the feature-extraction roofline ------------------- */
/* extract_*_samples: site list of one context after hm_run, in (read, qoff) order */
int64_t hm_scan_sites(hm_engine_t* e, int ctx, int32_t* read_id, int32_t* qoff, uint8_t* strand, int64_t cap);
/* get_next_sample_features: raw 401x8 fp32 windows of sites [first, first+n) of context ctx;
 * out_host may be NULL (device-only run for timing) */
int hm_windows(hm_engine_t* e, int ctx, int64_t first, int64_t n, float* out_host);
/* ModBatch::call_current_batch on caller-supplied windows [n][401][8] (host memory) */
int hm_cnn_logits(hm_engine_t* e, int ctx, const float* windows, int64_t n, float* logits, float* p, uint8_t* ml);
/* post-ReLU channels-last activations of conv `layer` (1..8) for one window (debug / tests) */
int64_t hm_debug_layer(hm_engine_t* e, int ctx, const float* window, int layer, float* out, int64_t cap);

/* Model-file utility (no GPU needed): read <src> (.onnx in either shipped dialect, or .hmw) and write
 * the flat fp32 .hmw container.  The same reader serves hm_create, so a model_dir holding the
 * reference's own CpG.onnx / CHG.onnx / CHH.onnx (mod_main.cpp:76,85,94) works unchanged. */
int hm_convert_model(const char* src_path, const char* dst_hmw_path);

/* diagnostic (option "stamps" = 1): shader-clock cycles per phase of the front kernel, summed over
 * all waves of the launches since the option was set; returns the number of slots written */
int hm_get_stamps(hm_engine_t* e, uint64_t* out, int cap);

int hm_get_timing(hm_engine_t* e, hm_timing_t* t);
int hm_reset_timing(hm_engine_t* e);

/* ==== `hifimeth pileup`: per-locus methylation frequencies ========================
 * Replaces the body of s_genomic_methy_freq_thread + the counting loop of s_compute_methy_freq
 * (src/app/hifimeth/pileup.cpp:208-353, 514-560) and the classes they drive:
 *   BamMapInfo::init / cigar_to_alignment   (src/corelib/bam_info.cpp:262-439)     -> hm_pileup_submit_read
 *   extract_chh_mapped_samples              (src/corelib/5mc_motif_finder.cpp:104-144)
 *   CpG / CHG loops                         (pileup.cpp:292-335)                     -> hm_pileup_run
 *   3 x 256 probability histograms          (pileup.cpp:237-272)                     -> hm_pileup_histograms
 *   per-locus pcov / ncov / motif           (pileup.cpp:519-560)                     -> hm_pileup_count
 *   rows of <prefix>.<ctx>.cov.bed          (pileup.cpp:562-590)                     -> hm_pileup_fetch_loci
 * The whole genome's counters stay resident in HBM (12 B per reference base) and the projected calls wait in
 * HBM (12 B each) until the thresholds are known -- the reference spills them to a temporary file instead.
 * MM/ML parsing and BED text formatting stay on the host (hm_bam.h).                                           */
typedef struct hm_pileup hm_pileup_t;

/* BaseModInfo (src/corelib/bam_mod_parser.hpp): one (position, code) of the MM lists with its ML byte */
typedef struct {
    int32_t qoff;       /* forward-strand (original read orientation) offset */
    uint8_t strand;     /* 0 '+', 1 '-'                                      */
    char unmod_base;
    char code;          /* 'm' = 5mC; other codes only enter the histograms  */
    uint8_t prob;
} hm_mod_t;

/* one covered locus = one BED row: chrom, soff, soff+1, 100*pcov/(pcov+ncov), pcov, ncov */
typedef struct {
    int64_t gpos;       /* offset into the concatenated reference (sequence offset + soff) */
    int32_t pcov, ncov;
    uint32_t motif;     /* 0 CpG, 1 CHG, 2 CHH: class of the locus' last record in BAM order */
    uint32_t reserved;
} hm_locus_t;

int hm_pileup_create(hm_pileup_t** out, int device);
void hm_pileup_destroy(hm_pileup_t* p);
const char* hm_pileup_last_error(const hm_pileup_t* p); /* p may be NULL: error of a failed create */
/* options: "min_mapq" (-q, default 0), "min_pi" (-f, default 0.0) */
int hm_pileup_set_option(hm_pileup_t* p, const char* key, double value);
/* HbnDatabase: n_seqs upper-cased sequences back to back in `bases` (seq_len[i] bytes each).  Allocates and
 * zeroes the per-locus planes unless hm_pileup_use_planes was called before. */
int hm_pileup_set_reference(hm_pileup_t* p, int32_t n_seqs, const int64_t* seq_len, const char* bases);
/* Count into caller-owned DEVICE planes of total-reference-length elements (int32 pcov, int32 ncov, uint32 key),
 * e.g. torch tensors that a RCCL reduce-scatter will consume.  The caller zeroes them. */
int hm_pileup_use_planes(hm_pileup_t* p, void* pcov, void* ncov, void* key);
int hm_pileup_planes(hm_pileup_t* p, void** pcov, void** ncov, void** key, int64_t* n_loci);
/* One mapped record: `order` = its index in the BAM, < 2^29 (decides the motif of a locus hit by two classes), `sid` =
 * index into the reference sequences, SEQ 4-bit packed and CIGAR as the BAM record stores them, `mods` = its
 * parsed MM/ML lists.  Returns 1 if staged, 0 if the record contributes nothing (unmapped, no mods), < 0 on
 * error (illegal base nibble, alignment running past the read or the reference sequence). */
int hm_pileup_submit_read(hm_pileup_t* p, uint32_t order, int32_t flag, int32_t sid, int64_t pos, int32_t mapq,
                          int32_t l_qseq, const uint8_t* seq4, int32_t n_cigar, const uint32_t* cigar,
                          int64_t n_mods, const hm_mod_t* mods);
/* histograms + projection of the staged records; the projected calls are appended to the HBM-resident list */
int hm_pileup_run(hm_pileup_t* p);
int64_t hm_pileup_num_records(hm_pileup_t* p);
/* bins[ctx*256 + scaled_prob], accumulated over all runs; hm_pileup_add_histograms adds counts from elsewhere */
int hm_pileup_histograms(hm_pileup_t* p, uint64_t* bins768);
/* debug / tests: D2H of the projected calls (unordered): gpos, prob, motif, order */
int64_t hm_pileup_fetch_records(hm_pileup_t* p, int64_t* gpos, uint8_t* prob, uint8_t* motif, uint32_t* order,
                                int64_t cap);
/* `hifimeth eval` (src/app/hifimeth/eval.cpp:469-560): joins the resident records with per-locus truth labels --
 * labels[g] over the concatenated reference: -1 none, 0 unmethylated, 1 methylated (s_fill_chr_base_label_with_bismark,
 * eval.cpp:42-114) -- into bins[(motif * 2 + label) * 256 + scaled_prob].  The records stay resident. */
int hm_pileup_label_histograms(hm_pileup_t* p, const int8_t* labels, int64_t n_labels, uint64_t* bins1536);
/* zero-free accumulate of all resident records into the planes with the given per-context thresholds
 * (prob >= thr -> pcov else ncov; key = max(order << 2 | motif)); then drops the records */
int hm_pileup_count(hm_pileup_t* p, const uint8_t thr[3]);
/* covered loci (pcov + ncov > 0) of planes[lo, hi) in ascending order; planes NULL = the engine's own, else
 * DEVICE pointers whose element 0 is locus `plane_base`.  Returns the number of loci (may exceed cap: then
 * nothing is written). */
int64_t hm_pileup_fetch_loci(hm_pileup_t* p, const void* pcov, const void* ncov, const void* key,
                             int64_t plane_base, int64_t lo, int64_t hi, hm_locus_t* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
