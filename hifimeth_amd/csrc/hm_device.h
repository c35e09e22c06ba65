// hm_device.h -- structures shared by the host engine and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace hm {

constexpr int KMER = 401;     // models/kmer.txt
constexpr int FEATS = 8;      // eval_kmer_features.cpp:42-64
constexpr int HK = 200;
constexpr int CHUNK = 1024;   // bases per prep/scan workgroup
constexpr int NCTX = 3;

enum { CPG = 0, CHG = 1, CHH = 2, CTX_NONE = 3 };

// One read of a staged batch.  Offsets are byte offsets into the raw slab exactly as the BAM
// record stores the arrays (4-bit seq; fi/fp/ri/rp as B:C bytes or B:S u16).
struct ReadDesc {
    int64_t off_seq, off_fi, off_fp, off_ri, off_rp;
    int64_t base_off;   // first element of this read in the packed per-base arrays
    int32_t len;
    int32_t flag;
    int32_t read_id;
    uint8_t w[4];       // element width of fi, fp, ri, rp (1 or 2)
};

struct Chunk {
    int32_t read_idx;
    int32_t start;      // first forward-strand position of the chunk
};

// site of one context list; uidx = rank in the unified (read, qoff)-ordered list
struct Site {
    int32_t read_idx;
    int32_t qoff;
    int32_t uidx;
};

struct USite {
    int32_t read_idx;
    int32_t qoff;
};

// per-chunk site counters / their exclusive scans: CpG, CHG, CHH, all contexts, reverse-strand sites (CHH on a G)
constexpr int NCNT = 5;

// Where a CNN launch finds its sites.  When a batch is queued its site counts exist only on the device (no host round
// trip between the scanner and the CNN): a launch names a context and a window [off, off + cap) of that context's
// list, and every workgroup resolves the actual pointer and count from the scan kernel's totals.
// totals == nullptr: `base` is the list itself (or nullptr for caller-supplied windows) and `cap` the exact count.
struct SiteRange {
    const Site* base;       // the three per-context lists back to back
    const int32_t* totals;  // [0..2] sites per context, [3] all, [4..6] first element of each context's list
    int32_t ctx, off, cap;
    // optional sub-range [*lo, *hi) of the context's list: the sites of one read group (the scanned per-chunk counters
    // at the group's first chunk and at the first chunk behind it); nullptr = the whole list
    const int32_t* lo;
    const int32_t* hi;
};

// ---- dense trunk (hm_trunk.hip) ---------------------------------------------------------------------------------------
// conv1..conv4 of the model are evaluated ONCE per (read, strand view) position instead of once per site: with stride-2
// convolutions the activations of a site's window are samples of four dense "a trous" maps over the read,
//     c1[p] = E1[off - 201 + 2p]   c2[q] = E2[off - 203 + 4q]   c3[r] = E3[off - 207 + 8r]   c4[s] = E4[off - 215 + 16s]
//     E1[x] = relu(b1 + sum_t W1[t] X[x + t]),  E(l)[x] = relu(b + sum_t W[t] E(l-1)[x + t * 2^(l-1)])
// for every output that touches neither the window's zero padding nor another such output: all but the first and the
// last position of every layer.  `off` is the site's position in its strand view (forward: qoff; reverse: L-1-qoff, the
// read seen as its reverse complement with the strands' kinetics swapped), X the bn0-normalised feature rows (rows
// outside the read are bn0(0), exactly what the reference's zero-filled window rows become: eval_kmer_features.cpp:36-40).
struct RInfo {
    int64_t base_off;  // first element of the read in bases / kin / sctx
    int32_t len;
    int32_t map_off;   // row of view position -200 of this read in the group's maps
};

struct TrunkTile {
    int32_t read_idx;
    int32_t u0;        // first view position the tile owns (the first tile of a read starts at -200)
};

constexpr int TR_OWN = 112;   // view positions per tile
constexpr int TR_PAD = 200;   // the maps cover view positions [-200, len + 200)
constexpr int TR_SLACK = 32;  // map rows of slack behind a read's tiles: the sliding-window trunk's skewed layers run up to 28 rows past the last tile

struct TrunkMaps {
    uint16_t* e[3];     // E1..E3: [2 views][rows][hi 128 | lo 128] fp16 halves (only the rows an edge chain reads are written)
    uint16_t* e4;       // E4: [2 views][rows][hi 96 | lo 96] fp16 halves (what the tail's input planes hold)
    int64_t view_rows;  // rows per view
    const uint16_t* zeros;  // >= 16 zero bytes (source of padding for the tail's LDS-DMA gather)
    uint8_t* rowlist;   // [tile x view][3][TR_OWN]: per layer the tile rows an edge chain reads, padded with row 0 (rowlist_kernel)
};

// bn0 folded into lookup tables, computed on the host with the ONNX BatchNormalization
// formula so that device windows equal the oracle's bn0 output bit for bit.
struct BnTables {
    float hot[4];        // bn0(1.0) for the one-hot channels 0..3
    float zero[8];       // bn0(0.0) for all 8 channels (cold one-hot; rows outside the read)
    float mean[8], gamma[8], beta[8], sd[8];   // sd = sqrtf(var + eps)  (windows-from-memory mode)
    float lut[4][256];   // bn0(codev1_decode(code) / 952) for channels 4..7
    float raw_lut[256];  // codev1_decode(code) / 952 (fp32 divide, eval_kmer_features.cpp:46-60)
};

// the same tables split into fp16 hi | lo << 16 for the split-half (f16x3) front kernel
struct BnTablesH {
    uint32_t hot[4];
    uint32_t zero[8];
    uint32_t lut[4][256];
    // bn0 of a kinetics channel as one fma on the decoded frame count: value = frames * ka[c] + kb[c]
    // (= ((frames / 952) - mean) / sd * gamma + beta up to fp32 rounding; used where the window never leaves the chip)
    float ka[4], kb[4];
};

// per-context device weights: MFMA-fragment-packed conv/fc weights + biases
struct CtxWeights {
    const float* wfrag[9];   // conv1..conv8, fc1 : [n-tile][k-group][lane][4]
    const float* bias[9];
    const float* fc2_w;      // [2][256]
    const float* fc2_b;      // [2]
    const BnTables* bn;
    const uint16_t* wfrag_h[9];  // conv1..conv8, fc1 as fp16 hi/lo fragments: [n-tile][k-block of 32][plane][lane][8]
    const BnTablesH* bn_h;
    // conv1 with bn0 folded into the weights (front_kernel_h, staged-read path; hm_weights.cpp): fragments in the layout
    // of wfrag_h[0], bias incl. the folded constants, and the constants to take back out at the two output rows that touch
    // the conv padding.  The operand is then exact fp16: one-hot 0 / 1 and frame counts / 32.
    const uint16_t* c1f;
    const float* c1f_bias;  // [128]
    const float* c1f_corr;  // [2][128]: first output row, last output row
    int k1;
};

constexpr int C4_LEN = 25, C4_CH = 96;          // conv4 output = hand-off between front and tail kernels
constexpr int ACT4_FLOATS = C4_LEN * C4_CH;      // 2400 floats / site
constexpr int TAIL_SITES = 8;                    // sites stacked along M in the tail kernel
constexpr int TAIL_STRIP_HANDOVER_BYTES = 512;  // what the strip tail kernel hands to tail_fc_kernel per site (conv8's rows, TAIL_X8_HALVES)
constexpr int TAIL_X8_HALVES = 256;              // conv8's output of a site on its way to tail_fc_kernel: [hi: 2 positions x 64 channels | lo] (hm_tail_fc.hip)
constexpr int TAILP_STRIP = 144;                 // strip tail (hm_tail_p.hip): lattice rows (16 map rows apart) of E4 a pass of 16 sites shares in LDS

}  // namespace hm
