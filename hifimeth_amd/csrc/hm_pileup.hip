// hm_pileup.hip -- `hifimeth pileup` on gfx950: alignment projection of the per-read 5mC calls, the 3 x 256
// probability histograms and the per-locus methylated / unmethylated counters, all resident in HBM.
//
// What the reference does on the CPU (src/app/hifimeth/pileup.cpp:208-353, 514-560): per read it expands the CIGAR
// into two gapped strings (bam_info.cpp:262-371), walks them three times with strncmp / a 3-mer hash to find CpG,
// CHG and CHH columns where read and reference agree, looks the read offset up in a per-read table of ML bytes,
// spills {sid, soff, prob, motif} to a temporary file, and after the thresholds are known replays the file one
// chromosome at a time into two int arrays.
//
// Here: the gapped strings are never built.  A motif can only sit on consecutive aligned pairs, so the host turns
// each CIGAR into "match runs" (maximal stretches of M/=/X columns: read offset, reference offset, length) and one
// GPU thread per aligned column tests the 2- and 3-column motifs straight from the 4-bit SEQ and the reference
// bytes.  This is byte / integer work bound by HBM traffic; no LDS tiling or MFMA applies.  Kernels:
//   mods_kernel     thread per MM/ML entry : ML byte -> per-base plane (last entry wins, as the reference's
//                   sequential overwrite), context histograms through LDS
//   identity_kernel thread per column      : matches per read (only when -f > 0)
//   project_kernel  thread per column      : motif tests, plane lookup, wave-aggregated append of 12-byte records
//   count_kernel    thread per record      : atomic add into pcov / ncov, atomic max into the motif key
//   loci_*          covered loci of a range in ascending order (count per block, scan, write)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hifimeth_hip.h"

namespace {

thread_local std::string g_pileup_create_error;

struct HipErr {
    hipError_t code;
    const char* what;
};
#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) throw HipErr{_e, #expr}; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // grow to at least `bytes`, keeping the first `keep` bytes
    void reserve(size_t bytes, size_t keep = 0, hipStream_t st = nullptr, bool exact = false) {
        if (bytes <= cap) return;
        const size_t want = exact ? bytes : bytes + bytes / 2 + 256;  // exact: genome-sized buffers, allocated once
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, want));
        if (keep) {
            HIP_TRY(hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- device-side records -----------------------------------------------------------------------------------
struct PRead {
    int64_t seq4_off;   // byte offset of the 4-bit SEQ in the batch slab
    int64_t plane_off;  // offset of base 0 in the per-base mod plane
    int32_t l_qseq;
    uint32_t order;
    int32_t as_size;    // alignment columns incl. gaps (denominator of the identity)
    uint8_t rev, primary, pass, pad;
};

struct PRun {
    int64_t g0;         // reference offset (concatenated) of the run's first column
    int32_t read;
    int32_t q0;         // offset in SEQ as stored
    int32_t len;
    int32_t pad;
};

struct PMod {
    int32_t read;
    int32_t qoff;
    uint32_t bits;      // idx_in_read << 10 | is_m << 9 | unmod_is_CG << 8 | prob
};

struct PRec {           // 12 bytes
    uint32_t glo;       // gpos & 0xffffffff
    uint32_t hi;        // gpos >> 32 (8 bits) | prob << 8 | motif << 16
    uint32_t order;
};

constexpr int TPB = 256;

__device__ __forceinline__ char nib_char(uint32_t c) {  // s_decode_bam_query_base (bam_info.cpp:100-121)
    return c == 1 ? 'A' : c == 2 ? 'C' : c == 4 ? 'G' : c == 8 ? 'T' : 'N';
}
__device__ __forceinline__ char comp_char(char c) {     // completement_residue (bam_info.cpp:146-167)
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
}
__device__ __forceinline__ char stored_base(const uint8_t* __restrict__ slab, const PRead& r, int k) {
    const uint32_t b = slab[r.seq4_off + (k >> 1)];
    return nib_char((k & 1) ? (b & 15u) : (b >> 4));
}
// BamQuerySequence::fwd_rqs[k] (bam_info.cpp:169-222)
__device__ __forceinline__ char fwd_base(const uint8_t* __restrict__ slab, const PRead& r, int k) {
    return r.rev ? comp_char(stored_base(slab, r, r.l_qseq - 1 - k)) : stored_base(slab, r, k);
}
__device__ __forceinline__ bool isH(char c) { return c == 'A' || c == 'C' || c == 'T'; }
__device__ __forceinline__ bool isD(char c) { return c == 'A' || c == 'G' || c == 'T'; }

// ---- mods: plane scatter + histograms (pileup.cpp:237-284) ---------------------------------------------------
__global__ __launch_bounds__(TPB) void mods_kernel(const PMod* __restrict__ mods, int64_t n, const PRead* __restrict__ reads,
                                                    const uint8_t* __restrict__ slab, uint32_t* __restrict__ plane,
                                                    unsigned long long* __restrict__ bins) {
    __shared__ uint32_t h[3 * 256];
    for (int i = threadIdx.x; i < 768; i += TPB) h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const PMod m = mods[i];
        const PRead r = reads[m.read];
        const uint32_t prob = m.bits & 255u;
        if (m.bits & 0x200u)  // code 'm': read_mods[qoff] = prob, later entries overwrite earlier ones
            atomicMax(&plane[r.plane_off + m.qoff], (((m.bits >> 10) + 1u) << 9) | 0x100u | prob);
        if (r.primary && (m.bits & 0x100u)) {
            const int q = m.qoff, L = r.l_qseq;
            const char c0 = fwd_base(slab, r, q);
            int ctx = -1;
            if (c0 == 'C') {
                const char c1 = q + 1 < L ? fwd_base(slab, r, q + 1) : 'N';
                const char c2 = q + 2 < L ? fwd_base(slab, r, q + 2) : 'N';
                if (q + 1 < L && c1 == 'G') ctx = 0;
                else if (q + 2 < L && isH(c1) && c2 == 'G') ctx = 1;
                else if (q + 2 < L && isH(c1) && isH(c2)) ctx = 2;
            } else if (q - 2 >= 0) {  // the G of [AGT][AGT]G
                if (c0 == 'G' && isD(fwd_base(slab, r, q - 1)) && isD(fwd_base(slab, r, q - 2))) ctx = 2;
            }
            if (ctx >= 0) atomicAdd(&h[ctx * 256 + prob], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += TPB)
        if (h[i]) atomicAdd(&bins[i], (unsigned long long)h[i]);
}

// run holding global column c: largest r in [lo, hi) with col0[r] <= c
__device__ __forceinline__ int find_run(const int64_t* __restrict__ col0, int n_runs, int64_t c, int lo = 0, int hi = -1) {
    if (hi < 0) hi = n_runs;  // col0[lo] <= c < col0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (col0[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- identity: s_calc_ident_perc numerator (bam_info.cpp:11-23); gap columns never match ---------------------
__global__ __launch_bounds__(TPB) void identity_kernel(const PRun* __restrict__ runs, const int64_t* __restrict__ col0,
                                                        int n_runs, int64_t n_cols, const PRead* __restrict__ reads,
                                                        const uint8_t* __restrict__ slab, const char* __restrict__ ref,
                                                        int32_t* __restrict__ matches) {
    const int64_t c = (int64_t)blockIdx.x * TPB + threadIdx.x;
    int rd = -1;
    bool eq = false;
    if (c < n_cols) {
        const int ri = find_run(col0, n_runs, c);
        const PRun run = runs[ri];
        const int o = (int)(c - col0[ri]);
        rd = run.read;
        eq = stored_base(slab, reads[rd], run.q0 + o) == ref[run.g0 + o];
    }
    const int first = __shfl(rd, 0);
    if (__all(rd == first)) {  // the common case: one read per wavefront -> one atomic
        const unsigned long long b = __ballot(eq);
        if ((threadIdx.x & 63) == 0 && first >= 0 && b) atomicAdd(&matches[first], __popcll(b));
    } else if (eq) {
        atomicAdd(&matches[rd], 1);
    }
}

// append one record per lane with `pred` to the workgroup's LDS stage: one LDS atomic per wavefront
__device__ __forceinline__ void stage_record(bool pred, const PRec& rec, PRec* __restrict__ stage, int* __restrict__ n_stage) {
    const unsigned long long b = __ballot(pred);
    if (!b) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)b) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(n_stage, __popcll(b));
    base = __shfl(base, leader);
    if (pred) stage[base + __popcll(b & ((1ull << lane) - 1ull))] = rec;
}

// ---- projection (pileup.cpp:286-347, 5mc_motif_finder.cpp:104-144) -----------------------------------------
// A workgroup walks PTILE consecutive columns, stages its records in LDS and appends them to the global list with
// ONE atomic (same-address atomics from every wavefront serialise in L2 and dominated an earlier version).
constexpr int PTILE = 1024;

__global__ __launch_bounds__(TPB) void project_kernel(const PRun* __restrict__ runs, const int64_t* __restrict__ col0,
                                                       int n_runs, int64_t n_cols, const PRead* __restrict__ reads,
                                                       const uint8_t* __restrict__ slab, const char* __restrict__ ref,
                                                       const uint32_t* __restrict__ plane,
                                                       const int32_t* __restrict__ matches, double min_pi,
                                                       PRec* __restrict__ out, unsigned long long* __restrict__ counter) {
    __shared__ PRec stage[2 * PTILE];  // a column yields at most two records (CpG + the reverse-strand CGG of CHG)
    __shared__ int n_stage;
    __shared__ int run_range[2];  // runs touched by this tile: the per-column search starts from here
    __shared__ unsigned long long out_base;
    if (threadIdx.x == 0) n_stage = 0;
    if (threadIdx.x < 2) {
        const int64_t c = min((int64_t)blockIdx.x * PTILE + (threadIdx.x ? PTILE - 1 : 0), n_cols - 1);
        run_range[threadIdx.x] = find_run(col0, n_runs, c) + (int)threadIdx.x;
    }
    __syncthreads();
    const int run_lo = run_range[0], run_hi = run_range[1];
    for (int it = 0; it < PTILE / TPB; ++it) {
        const int64_t c = (int64_t)blockIdx.x * PTILE + it * TPB + threadIdx.x;
        bool e0 = false, e1 = false;
        PRec r0{}, r1{};
        if (c < n_cols) {
            const int ri = find_run(col0, n_runs, c, run_lo, run_hi);
            const PRun run = runs[ri];
            const int o = (int)(c - col0[ri]);
            const int rem = run.len - o;
            const PRead rd = reads[run.read];
            bool live = rd.pass && rem >= 2;
            if (live && min_pi > 0.0) live = !(100.0 * matches[run.read] / rd.as_size < min_pi);
            if (live) {
                const int qp = run.q0 + o, L = rd.l_qseq;
                const int64_t g = run.g0 + o;
                const char q0 = stored_base(slab, rd, qp), q1 = stored_base(slab, rd, qp + 1);
                const char s0 = ref[g], s1 = ref[g + 1];
                char q2 = '-', s2 = '*';
                if (rem >= 3) { q2 = stored_base(slab, rd, qp + 2); s2 = ref[g + 2]; }
                const bool eq3 = rem >= 3 && q0 == s0 && q1 == s1 && q2 == s2;
                auto look = [&](int qoff, int64_t soff, uint32_t motif, bool& e, PRec& r) {
                    const uint32_t v = plane[rd.plane_off + qoff];
                    if (v & 0x100u) {
                        e = true;
                        r.glo = (uint32_t)soff;
                        r.hi = (uint32_t)((uint64_t)soff >> 32) | ((v & 255u) << 8) | (motif << 16);
                        r.order = rd.order;
                    }
                };
                if (q0 == 'C' && q1 == 'G' && s0 == 'C' && s1 == 'G')  // CpG, recorded at the reference C
                    look(rd.rev ? L - 1 - (qp + 1) : qp, g, 0, e0, r0);
                if (eq3 && q0 == 'C' && q2 == 'G') {  // CHG: forward reads CCG/CAG/CTG, reverse reads CGG/CAG/CTG
                    const bool mid = rd.rev ? (q1 == 'G' || q1 == 'A' || q1 == 'T') : (q1 == 'C' || q1 == 'A' || q1 == 'T');
                    if (mid) look(rd.rev ? L - 1 - (qp + 2) : qp, g, 1, e1, r1);
                } else if (eq3 && q0 == 'C' && isH(q1) && isH(q2)) {  // CHH on the reference's forward strand
                    look(rd.rev ? L - 1 - qp : qp, g, 2, e1, r1);
                } else if (eq3 && isD(q0) && isD(q1) && q2 == 'G') {  // CHH on the reverse strand, recorded at the G
                    look(rd.rev ? L - 1 - (qp + 2) : qp + 2, g + 2, 2, e1, r1);
                }
            }
        }
        stage_record(e0, r0, stage, &n_stage);
        stage_record(e1, r1, stage, &n_stage);
    }
    __syncthreads();
    const int n = n_stage;
    if (threadIdx.x == 0 && n) out_base = atomicAdd(counter, (unsigned long long)n);
    __syncthreads();
    if (n) {  // 12-byte records move as a stream of dwords
        const uint32_t* src = reinterpret_cast<const uint32_t*>(stage);
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + out_base);
        for (int i = threadIdx.x; i < 3 * n; i += TPB) dst[i] = src[i];
    }
}

// ---- counting (pileup.cpp:529-557) -------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void count_kernel(const PRec* __restrict__ recs, int64_t n, uint32_t thr_packed,
                                                     int32_t* __restrict__ pcov, int32_t* __restrict__ ncov,
                                                     uint32_t* __restrict__ key) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const PRec r = recs[i];
        const int64_t g = (int64_t)r.glo | ((int64_t)(r.hi & 255u) << 32);
        const uint32_t prob = (r.hi >> 8) & 255u, motif = (r.hi >> 16) & 3u;
        const uint32_t thr = (thr_packed >> (8 * motif)) & 255u;
        atomicAdd(prob >= thr ? &pcov[g] : &ncov[g], 1);
        atomicMax(&key[g], (r.order << 2) | motif);
    }
}

// ---- resident records joined with per-locus truth labels (`hifimeth eval`, src/app/hifimeth/eval.cpp:469-560) ----
// labels[g]: -1 no truth, 0 unmethylated, 1 methylated.  bins[(motif * 2 + label) * 256 + prob] counts the records whose
// locus carries a label; workgroup-private histogram in LDS, one global atomic per non-empty bin and workgroup.
__global__ __launch_bounds__(TPB) void label_kernel(const PRec* __restrict__ recs, int64_t n, const int8_t* __restrict__ labels,
                                                     unsigned long long* __restrict__ bins) {
    __shared__ uint32_t h[1536];
    for (int i = threadIdx.x; i < 1536; i += TPB) h[i] = 0u;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const PRec r = recs[i];
        const int64_t g = (int64_t)r.glo | ((int64_t)(r.hi & 255u) << 32);
        const int lab = labels[g];
        if (lab >= 0) atomicAdd(&h[(((r.hi >> 16) & 3u) * 2u + (lab ? 1u : 0u)) * 256u + ((r.hi >> 8) & 255u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1536; i += TPB)
        if (h[i]) atomicAdd(&bins[i], (unsigned long long)h[i]);
}

// ---- covered loci of a range, ascending ---------------------------------------------------------------------
constexpr int LOCI_PER_BLOCK = 4096;  // 16 per thread

__global__ __launch_bounds__(TPB) void loci_count_kernel(const int32_t* __restrict__ pcov, const int32_t* __restrict__ ncov,
                                                          int64_t lo, int64_t hi, int32_t* __restrict__ block_counts) {
    __shared__ int wsum[TPB / 64];
    const int64_t base = lo + (int64_t)blockIdx.x * LOCI_PER_BLOCK;
    int cnt = 0;
    for (int k = 0; k < LOCI_PER_BLOCK / TPB; ++k) {
        const int64_t i = base + k * TPB + threadIdx.x;
        if (i < hi && (pcov[i] | ncov[i])) ++cnt;
    }
    for (int d = 32; d; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-workgroup exclusive scan; total -> offs[n]
__global__ __launch_bounds__(1024) void loci_scan_kernel(const int32_t* __restrict__ counts, int n,
                                                          int64_t* __restrict__ offs) {
    __shared__ int64_t part[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int64_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) {
        offs[i] = run;
        run += counts[i];
    }
    if (threadIdx.x == 1023) offs[n] = part[1023];
}

__global__ __launch_bounds__(TPB) void loci_write_kernel(const int32_t* __restrict__ pcov, const int32_t* __restrict__ ncov,
                                                          const uint32_t* __restrict__ key, int64_t plane_base,
                                                          int64_t lo, int64_t hi, const int64_t* __restrict__ offs,
                                                          hm_locus_t* __restrict__ out) {
    __shared__ int wsum[TPB / 64];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    const int64_t base = lo + (int64_t)blockIdx.x * LOCI_PER_BLOCK;
    const int64_t o0 = offs[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < LOCI_PER_BLOCK / TPB; ++k) {
        __syncthreads();
        const int64_t i = base + k * TPB + threadIdx.x;
        int32_t p = 0, n = 0;
        if (i < hi) { p = pcov[i]; n = ncov[i]; }
        const bool cov = (p | n) != 0;
        const unsigned long long b = __ballot(cov);
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int before = carry;
        for (int j = 0; j < w; ++j) before += wsum[j];
        if (cov) {
            hm_locus_t l;
            l.gpos = plane_base + i;
            l.pcov = p;
            l.ncov = n;
            l.motif = key[i] & 3u;
            l.reserved = 0;
            out[o0 + before + __popcll(b & ((1ull << lane) - 1ull))] = l;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

}  // namespace

// ================================================ host ==========================================================
struct hm_pileup {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int min_mapq = 0;
    double min_pi = 0.0;

    // reference
    std::vector<int64_t> seq_off;  // n_seqs + 1
    DevBuf d_ref;
    bool own_planes = true;
    DevBuf d_pcov, d_ncov, d_key;
    int32_t* pcov = nullptr;
    int32_t* ncov = nullptr;
    uint32_t* key = nullptr;

    // staged batch (host)
    std::vector<uint8_t> slab;
    std::vector<PRead> reads;
    std::vector<PRun> runs;
    std::vector<int64_t> col0;
    std::vector<PMod> mods;
    int64_t plane_len = 0, n_m_mods = 0;

    // device
    DevBuf d_slab, d_reads, d_runs, d_col0, d_mods, d_plane, d_matches, d_bins, d_counter, d_recs;
    DevBuf d_blk, d_offs, d_loci, d_labels, d_lbins;
    int64_t n_recs = 0;
    bool bins_ready = false;
};

namespace {

int pfail(hm_pileup* p, int code, const std::string& msg) {
    if (p) p->err = msg;
    else g_pileup_create_error = msg;
    return code;
}
int pfail_hip(hm_pileup* p, const HipErr& h) {
    return pfail(p, HM_EDEVICE, std::string("HIP error: ") + hipGetErrorString(h.code) + " at " + h.what);
}

inline int grid_for(int64_t n, int cap = 1 << 20) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + TPB - 1) / TPB, cap)); }

void ensure_bins(hm_pileup* p) {
    if (p->bins_ready) return;
    p->d_bins.reserve(768 * sizeof(unsigned long long));
    p->d_counter.reserve(sizeof(unsigned long long));
    HIP_TRY(hipMemsetAsync(p->d_bins.p, 0, 768 * sizeof(unsigned long long), p->stream));
    HIP_TRY(hipMemsetAsync(p->d_counter.p, 0, sizeof(unsigned long long), p->stream));
    p->bins_ready = true;
}

void clear_batch(hm_pileup* p) {
    p->slab.clear();
    p->reads.clear();
    p->runs.clear();
    p->col0.clear();
    p->mods.clear();
    p->plane_len = 0;
    p->n_m_mods = 0;
}

}  // namespace

extern "C" {

int hm_pileup_create(hm_pileup_t** out, int device) {
    if (!out) return pfail(nullptr, HM_EINVAL, "hm_pileup_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return pfail(nullptr, HM_EDEVICE, "no HIP device: the pileup kernels need a gfx950 GPU (there is no CPU fallback)");
    if (device < 0 || device >= n) return pfail(nullptr, HM_EINVAL, "device ordinal out of range");
    hm_pileup* p = new hm_pileup;
    p->device = device;
    try {
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (!strstr(prop.gcnArchName, "gfx950")) {
            const std::string a = prop.gcnArchName;
            delete p;
            return pfail(nullptr, HM_EDEVICE, "device is " + a + ", this library holds gfx950 code only");
        }
        HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    } catch (const HipErr& h) {
        const int rc = pfail_hip(nullptr, h);
        delete p;
        return rc;
    }
    *out = p;
    return HM_OK;
}

void hm_pileup_destroy(hm_pileup_t* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (DevBuf* b : {&p->d_ref, &p->d_pcov, &p->d_ncov, &p->d_key, &p->d_slab, &p->d_reads, &p->d_runs, &p->d_col0,
                      &p->d_mods, &p->d_plane, &p->d_matches, &p->d_bins, &p->d_counter, &p->d_recs, &p->d_blk,
                      &p->d_offs, &p->d_loci})
        b->release();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

const char* hm_pileup_last_error(const hm_pileup_t* p) { return p ? p->err.c_str() : g_pileup_create_error.c_str(); }

int hm_pileup_set_option(hm_pileup_t* p, const char* key, double value) {
    if (!p || !key) return HM_EINVAL;
    const std::string k = key;
    if (k == "min_mapq") p->min_mapq = (int)value;
    else if (k == "min_pi") p->min_pi = value;
    else return pfail(p, HM_EINVAL, "unknown option " + k);
    return HM_OK;
}

int hm_pileup_use_planes(hm_pileup_t* p, void* pcov, void* ncov, void* key) {
    if (!p || !pcov || !ncov || !key) return HM_EINVAL;
    p->own_planes = false;
    p->pcov = static_cast<int32_t*>(pcov);
    p->ncov = static_cast<int32_t*>(ncov);
    p->key = static_cast<uint32_t*>(key);
    return HM_OK;
}

int hm_pileup_set_reference(hm_pileup_t* p, int32_t n_seqs, const int64_t* seq_len, const char* bases) {
    if (!p || n_seqs <= 0 || !seq_len || !bases) return pfail(p, HM_EINVAL, "hm_pileup_set_reference: bad argument");
    p->seq_off.assign(1, 0);
    for (int i = 0; i < n_seqs; ++i) {
        if (seq_len[i] < 0) return pfail(p, HM_EINVAL, "negative sequence length");
        p->seq_off.push_back(p->seq_off.back() + seq_len[i]);
    }
    const int64_t total = p->seq_off.back();
    if (total >= (int64_t(1) << 40)) return pfail(p, HM_EINVAL, "reference longer than 2^40 bases");
    try {
        HIP_TRY(hipSetDevice(p->device));
        p->d_ref.reserve((size_t)total + 4, 0, nullptr, true);
        HIP_TRY(hipMemcpyAsync(p->d_ref.p, bases, (size_t)total, hipMemcpyHostToDevice, p->stream));
        if (p->own_planes) {
            const size_t bytes = (size_t)std::max<int64_t>(total, 1) * 4;
            p->d_pcov.reserve(bytes, 0, nullptr, true);
            p->d_ncov.reserve(bytes, 0, nullptr, true);
            p->d_key.reserve(bytes, 0, nullptr, true);
            p->pcov = p->d_pcov.as<int32_t>();
            p->ncov = p->d_ncov.as<int32_t>();
            p->key = p->d_key.as<uint32_t>();
            HIP_TRY(hipMemsetAsync(p->pcov, 0, bytes, p->stream));
            HIP_TRY(hipMemsetAsync(p->ncov, 0, bytes, p->stream));
            HIP_TRY(hipMemsetAsync(p->key, 0, bytes, p->stream));
        }
        HIP_TRY(hipStreamSynchronize(p->stream));
    } catch (const HipErr& h) {
        return pfail_hip(p, h);
    }
    return HM_OK;
}

int hm_pileup_planes(hm_pileup_t* p, void** pcov, void** ncov, void** key, int64_t* n_loci) {
    if (!p || p->seq_off.empty()) return pfail(p, HM_ESTATE, "hm_pileup_planes before hm_pileup_set_reference");
    if (pcov) *pcov = p->pcov;
    if (ncov) *ncov = p->ncov;
    if (key) *key = p->key;
    if (n_loci) *n_loci = p->seq_off.back();
    return HM_OK;
}

int hm_pileup_submit_read(hm_pileup_t* p, uint32_t order, int32_t flag, int32_t sid, int64_t pos, int32_t mapq,
                          int32_t l_qseq, const uint8_t* seq4, int32_t n_cigar, const uint32_t* cigar, int64_t n_mods,
                          const hm_mod_t* mods) {
    if (!p) return HM_EINVAL;
    if (p->seq_off.empty()) return pfail(p, HM_ESTATE, "hm_pileup_submit_read before hm_pileup_set_reference");
    if (n_mods <= 0 || (flag & 4)) return 0;  // pileup.cpp:233-235
    if (n_mods >= (int64_t(1) << 22)) return pfail(p, HM_EINVAL, "more than 2^22 modification entries in one read");
    if (l_qseq < 0 || !seq4 || n_cigar < 0 || (n_cigar && !cigar) || !mods) return pfail(p, HM_EINVAL, "hm_pileup_submit_read: bad argument");
    // key = order << 2 | motif must stay below 2^31: the multi-GPU path max-reduces the key plane as int32
    // (hifimeth_amd/pileup.py: reduce_scatter_planes), where a set top bit would lose against an empty locus
    if (order >= (1u << 29)) return pfail(p, HM_EINVAL, "record order must be < 2^29");
    if (l_qseq >= (1 << 22)) return pfail(p, HM_EINVAL, "reads of 2^22 bases or more are not supported");
    const int n_seqs = (int)p->seq_off.size() - 1;
    if (sid < 0 || sid >= n_seqs) return pfail(p, HM_EINVAL, "sequence index out of range");
    const int64_t ssize = p->seq_off[sid + 1] - p->seq_off[sid];
    if (pos < 0 || pos > ssize) return pfail(p, HM_EDATA, "alignment position outside the reference sequence");
    {   // s_decode_bam_query_base accepts the nibbles 1, 2, 4, 8, 15 only (bam_info.cpp:100-121); two per byte
        static const auto ok = [] {
            std::array<uint8_t, 256> t{};
            auto good = [](int c) { return c == 1 || c == 2 || c == 4 || c == 8 || c == 15; };
            for (int b = 0; b < 256; ++b) t[(size_t)b] = good(b >> 4) && good(b & 15);
            return t;
        }();
        const int full = l_qseq >> 1;
        int bad = -1;
        for (int i = 0; i < full; ++i)
            if (!ok[seq4[i]]) { bad = i; break; }
        if (bad < 0 && (l_qseq & 1) && !ok[(seq4[full] & 0xf0) | 1]) bad = full;
        if (bad >= 0) {
            const int hi = seq4[bad] >> 4, lo = seq4[bad] & 15;
            const bool hi_bad = !(hi == 1 || hi == 2 || hi == 4 || hi == 8 || hi == 15);
            return pfail(p, HM_EDATA, "Illegal BAM base encoded value " + std::to_string(hi_bad ? hi : lo));
        }
    }
    // cigar_to_alignment (bam_info.cpp:262-371) without the strings: match runs + column count
    const size_t runs_before = p->runs.size();
    int opi = 0;
    int64_t qi = -1, si = -1;
    if (n_cigar > 0) {
        const int op0 = cigar[0] & 15;
        if (op0 == 4) { qi = (int64_t)(cigar[0] >> 4) - 1; opi = 1; }
        else if (op0 == 5) opi = 1;
    }
    int64_t as_size = 0;
    bool open = false;  // the previous column-producing op was a match-type op
    for (; opi < n_cigar; ++opi) {
        const int op = cigar[opi] & 15;
        const int64_t num = cigar[opi] >> 4;
        if (op == 0 || op == 7 || op == 8) {  // M = X
            if (num == 0) continue;
            if (open) p->runs.back().len += (int32_t)num;
            else p->runs.push_back(PRun{p->seq_off[sid] + pos + si + 1, (int32_t)p->reads.size(), (int32_t)(qi + 1), (int32_t)num, 0});
            open = true;
            qi += num; si += num; as_size += num;
        } else if (op == 1) {  // I
            qi += num; as_size += num;
            if (num) open = false;
        } else if (op == 2 || op == 3) {  // D N
            si += num; as_size += num;
            if (num) open = false;
        } else if (op == 4 || op == 5 || op == 6) {  // S H P: no columns
        } else {
            p->runs.resize(runs_before);
            return pfail(p, HM_EDATA, "Unrecognised CIGAR operation");
        }
        if (qi >= l_qseq || pos + si >= ssize) {
            p->runs.resize(runs_before);
            return pfail(p, HM_EDATA, qi >= l_qseq ? "CIGAR consumes more bases than SEQ holds"
                                                    : "alignment runs past the end of the reference sequence");
        }
    }
    if (as_size >= (int64_t(1) << 31)) { p->runs.resize(runs_before); return pfail(p, HM_EINVAL, "alignment too long"); }
    PRead r{};
    r.seq4_off = (int64_t)p->slab.size();
    r.plane_off = p->plane_len;
    r.l_qseq = l_qseq;
    r.order = order;
    r.as_size = (int32_t)as_size;
    r.rev = (flag & 16) ? 1 : 0;
    r.primary = (flag & 0x900) ? 0 : 1;
    r.pass = mapq >= p->min_mapq ? 1 : 0;
    const int32_t ri = (int32_t)p->reads.size();
    const size_t mods_before = p->mods.size();
    const int64_t m_before = p->n_m_mods;
    for (int64_t i = 0; i < n_mods; ++i) {
        const hm_mod_t& m = mods[i];
        if (m.qoff < 0 || m.qoff >= l_qseq) {  // leave the staged batch as it was
            p->runs.resize(runs_before);
            p->mods.resize(mods_before);
            p->n_m_mods = m_before;
            return pfail(p, HM_EDATA, "modification offset outside the read");
        }
        const bool is_m = m.code == 'm';
        const bool cg = m.unmod_base == 'C' || m.unmod_base == 'G';
        if (!is_m && !cg) continue;
        p->mods.push_back(PMod{ri, m.qoff, ((uint32_t)i << 10) | (is_m ? 0x200u : 0u) | (cg ? 0x100u : 0u) | m.prob});
        if (is_m) ++p->n_m_mods;
    }
    p->slab.insert(p->slab.end(), seq4, seq4 + (l_qseq + 1) / 2);
    p->plane_len += l_qseq;
    p->reads.push_back(r);
    return 1;
}

int hm_pileup_run(hm_pileup_t* p) {
    if (!p) return HM_EINVAL;
    if (p->reads.empty()) return HM_OK;
    try {
        HIP_TRY(hipSetDevice(p->device));
        ensure_bins(p);
        hipStream_t st = p->stream;
        const int n_reads = (int)p->reads.size(), n_runs = (int)p->runs.size();
        p->col0.resize((size_t)n_runs + 1);
        int64_t cols = 0;
        for (int i = 0; i < n_runs; ++i) { p->col0[i] = cols; cols += p->runs[i].len; }
        p->col0[n_runs] = cols;
        const int64_t n_mods = (int64_t)p->mods.size();

        p->d_slab.reserve(p->slab.size() + 4);
        p->d_reads.reserve(sizeof(PRead) * (size_t)n_reads);
        p->d_runs.reserve(sizeof(PRun) * (size_t)std::max(n_runs, 1));
        p->d_col0.reserve(sizeof(int64_t) * ((size_t)n_runs + 1));
        p->d_mods.reserve(sizeof(PMod) * (size_t)std::max<int64_t>(n_mods, 1));
        p->d_plane.reserve(4 * (size_t)std::max<int64_t>(p->plane_len, 1));
        p->d_matches.reserve(4 * (size_t)n_reads);
        p->d_recs.reserve(sizeof(PRec) * (size_t)(p->n_recs + p->n_m_mods + 1), sizeof(PRec) * (size_t)p->n_recs, st);

        HIP_TRY(hipMemcpyAsync(p->d_slab.p, p->slab.data(), p->slab.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(p->d_reads.p, p->reads.data(), sizeof(PRead) * (size_t)n_reads, hipMemcpyHostToDevice, st));
        if (n_runs) HIP_TRY(hipMemcpyAsync(p->d_runs.p, p->runs.data(), sizeof(PRun) * (size_t)n_runs, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(p->d_col0.p, p->col0.data(), sizeof(int64_t) * ((size_t)n_runs + 1), hipMemcpyHostToDevice, st));
        if (n_mods) HIP_TRY(hipMemcpyAsync(p->d_mods.p, p->mods.data(), sizeof(PMod) * (size_t)n_mods, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(p->d_plane.p, 0, 4 * (size_t)std::max<int64_t>(p->plane_len, 1), st));
        HIP_TRY(hipMemsetAsync(p->d_matches.p, 0, 4 * (size_t)n_reads, st));

        if (n_mods)  // <= 1024 workgroups: each flushes up to 768 histogram bins with same-address global atomics (~10 ns each)
            hipLaunchKernelGGL(mods_kernel, dim3(grid_for(n_mods, 1024)), dim3(TPB), 0, st, p->d_mods.as<PMod>(), n_mods,
                               p->d_reads.as<PRead>(), p->d_slab.as<uint8_t>(), p->d_plane.as<uint32_t>(),
                               p->d_bins.as<unsigned long long>());
        if (cols > 0) {
            const int64_t blocks = (cols + TPB - 1) / TPB, pblocks = (cols + PTILE - 1) / PTILE;
            if (blocks >= (int64_t(1) << 31)) return pfail(p, HM_EINVAL, "batch too large: submit fewer records per hm_pileup_run");
            if (p->min_pi > 0.0)
                hipLaunchKernelGGL(identity_kernel, dim3((unsigned)blocks), dim3(TPB), 0, st, p->d_runs.as<PRun>(),
                                   p->d_col0.as<int64_t>(), n_runs, cols, p->d_reads.as<PRead>(), p->d_slab.as<uint8_t>(),
                                   p->d_ref.as<char>(), p->d_matches.as<int32_t>());
            hipLaunchKernelGGL(project_kernel, dim3((unsigned)pblocks), dim3(TPB), 0, st, p->d_runs.as<PRun>(),
                               p->d_col0.as<int64_t>(), n_runs, cols, p->d_reads.as<PRead>(), p->d_slab.as<uint8_t>(),
                               p->d_ref.as<char>(), p->d_plane.as<uint32_t>(), p->d_matches.as<int32_t>(), p->min_pi,
                               p->d_recs.as<PRec>(), p->d_counter.as<unsigned long long>());
        }
        HIP_TRY(hipGetLastError());
        unsigned long long n = 0;
        HIP_TRY(hipMemcpyAsync(&n, p->d_counter.p, sizeof n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        p->n_recs = (int64_t)n;
    } catch (const HipErr& h) {
        return pfail_hip(p, h);
    }
    clear_batch(p);
    return HM_OK;
}

int64_t hm_pileup_num_records(hm_pileup_t* p) { return p ? p->n_recs : HM_EINVAL; }

int hm_pileup_histograms(hm_pileup_t* p, uint64_t* bins768) {
    if (!p || !bins768) return HM_EINVAL;
    try {
        HIP_TRY(hipSetDevice(p->device));
        ensure_bins(p);
        HIP_TRY(hipMemcpyAsync(bins768, p->d_bins.p, 768 * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    } catch (const HipErr& h) {
        return pfail_hip(p, h);
    }
    return HM_OK;
}

int64_t hm_pileup_fetch_records(hm_pileup_t* p, int64_t* gpos, uint8_t* prob, uint8_t* motif, uint32_t* order, int64_t cap) {
    if (!p) return HM_EINVAL;
    if (cap < p->n_recs) return p->n_recs;
    std::vector<PRec> h((size_t)p->n_recs);
    try {
        HIP_TRY(hipSetDevice(p->device));
        if (p->n_recs) HIP_TRY(hipMemcpyAsync(h.data(), p->d_recs.p, sizeof(PRec) * h.size(), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    } catch (const HipErr& e) {
        return pfail_hip(p, e);
    }
    for (size_t i = 0; i < h.size(); ++i) {
        if (gpos) gpos[i] = (int64_t)h[i].glo | ((int64_t)(h[i].hi & 255u) << 32);
        if (prob) prob[i] = (uint8_t)(h[i].hi >> 8);
        if (motif) motif[i] = (uint8_t)((h[i].hi >> 16) & 3u);
        if (order) order[i] = h[i].order;
    }
    return p->n_recs;
}

int hm_pileup_label_histograms(hm_pileup_t* p, const int8_t* labels, int64_t n_labels, uint64_t* bins1536) {
    if (!p || !labels || !bins1536) return HM_EINVAL;
    if (p->seq_off.empty()) return pfail(p, HM_ESTATE, "hm_pileup_label_histograms before hm_pileup_set_reference");
    if (n_labels != p->seq_off.back()) return pfail(p, HM_EINVAL, "hm_pileup_label_histograms: one label per reference base expected");
    try {
        HIP_TRY(hipSetDevice(p->device));
        p->d_labels.reserve((size_t)n_labels + 1);
        p->d_lbins.reserve(1536 * sizeof(unsigned long long));
        HIP_TRY(hipMemcpyAsync(p->d_labels.p, labels, (size_t)n_labels, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemsetAsync(p->d_lbins.p, 0, 1536 * sizeof(unsigned long long), p->stream));
        if (p->n_recs) {
            hipLaunchKernelGGL(label_kernel, dim3(grid_for(p->n_recs, 1024)), dim3(TPB), 0, p->stream, p->d_recs.as<PRec>(), p->n_recs,
                               p->d_labels.as<int8_t>(), p->d_lbins.as<unsigned long long>());
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(bins1536, p->d_lbins.p, 1536 * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    } catch (const HipErr& h) {
        return pfail_hip(p, h);
    }
    return HM_OK;
}

int hm_pileup_count(hm_pileup_t* p, const uint8_t thr[3]) {
    if (!p || !thr) return HM_EINVAL;
    if (!p->pcov) return pfail(p, HM_ESTATE, "hm_pileup_count before hm_pileup_set_reference / hm_pileup_use_planes");
    try {
        HIP_TRY(hipSetDevice(p->device));
        ensure_bins(p);
        if (p->n_recs) {
            const uint32_t packed = thr[0] | ((uint32_t)thr[1] << 8) | ((uint32_t)thr[2] << 16);
            hipLaunchKernelGGL(count_kernel, dim3(grid_for(p->n_recs, 1 << 16)), dim3(TPB), 0, p->stream, p->d_recs.as<PRec>(),
                               p->n_recs, packed, p->pcov, p->ncov, p->key);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemsetAsync(p->d_counter.p, 0, sizeof(unsigned long long), p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->n_recs = 0;
    } catch (const HipErr& h) {
        return pfail_hip(p, h);
    }
    return HM_OK;
}

int64_t hm_pileup_fetch_loci(hm_pileup_t* p, const void* pcov, const void* ncov, const void* key, int64_t plane_base,
                             int64_t lo, int64_t hi, hm_locus_t* out, int64_t cap) {
    if (!p || lo < 0 || hi < lo) return pfail(p, HM_EINVAL, "hm_pileup_fetch_loci: bad range");
    const int32_t* pc = pcov ? static_cast<const int32_t*>(pcov) : p->pcov;
    const int32_t* nc = ncov ? static_cast<const int32_t*>(ncov) : p->ncov;
    const uint32_t* ky = key ? static_cast<const uint32_t*>(key) : p->key;
    if (!pcov) plane_base = 0;
    if (!pc || !nc || !ky) return pfail(p, HM_ESTATE, "no planes");
    if (hi == lo) return 0;
    const int64_t nblk = (hi - lo + LOCI_PER_BLOCK - 1) / LOCI_PER_BLOCK;
    if (nblk >= (int64_t(1) << 31)) return pfail(p, HM_EINVAL, "range too large: fetch per sequence");
    try {
        HIP_TRY(hipSetDevice(p->device));
        hipStream_t st = p->stream;
        p->d_blk.reserve(4 * (size_t)nblk);
        p->d_offs.reserve(8 * ((size_t)nblk + 1));
        hipLaunchKernelGGL(loci_count_kernel, dim3((unsigned)nblk), dim3(TPB), 0, st, pc, nc, lo, hi, p->d_blk.as<int32_t>());
        hipLaunchKernelGGL(loci_scan_kernel, dim3(1), dim3(1024), 0, st, p->d_blk.as<int32_t>(), (int)nblk, p->d_offs.as<int64_t>());
        int64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, p->d_offs.as<int64_t>() + nblk, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (total > cap || !out || total == 0) return total;
        p->d_loci.reserve(sizeof(hm_locus_t) * (size_t)total);
        hipLaunchKernelGGL(loci_write_kernel, dim3((unsigned)nblk), dim3(TPB), 0, st, pc, nc, ky, plane_base, lo, hi,
                           p->d_offs.as<int64_t>(), p->d_loci.as<hm_locus_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, p->d_loci.p, sizeof(hm_locus_t) * (size_t)total, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return total;
    } catch (const HipErr& h) {
        return pfail_hip(p, h);
    }
}

}  // extern "C"
