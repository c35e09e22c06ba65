// hm_kernels.hip -- gfx950 (CDNA4) kernels of the 5mC calling hot path.
//
//   prep / scan / emit : site scanner + kinetics packer  (reference: src/app/hifimeth/
//                        eval_kmer_features.cpp:67-126, src/corelib/bam_info.cpp:169-222,443-603)
//   windows            : 401x8 window builder            (eval_kmer_features.cpp:9-65)
//   front / tail       : the CNN of training/model_cnn.py:8-85 as shipped in models/*.onnx,
//                        fp32 MFMA (v_mfma_f32_16x16x4_f32), activations tiled in LDS
//                        (replaces ov::InferRequest::infer(), mod_batch.cpp:66-75) and
//                        the softmax -> ML byte of mod_batch.cpp:46-64.
//
// Written for wave64 / gfx950 only.
#include "hm_conv32.h"
#include "hm_stamp.h"

#include <algorithm>

namespace hm {

typedef f32x4_t f32x4;

// =================================================================================================
// feature extraction
// =================================================================================================

// BAM nibble -> code A0 C1 G2 T3, N (15) -> 4; anything else is illegal (bam_info.cpp:100-121 aborts)
__device__ __forceinline__ int nib_to_code(int nib, int& bad) {
    switch (nib) {
    case 1: return 0;
    case 2: return 1;
    case 4: return 2;
    case 8: return 3;
    case 15: return 4;
    default: bad = 1; return 4;
    }
}

// forward-strand code of position j; positions outside the read give 4 (breaks every motif)
__device__ __forceinline__ int fwd_code(const uint8_t* __restrict__ raw, const ReadDesc& rd, int j, int& bad) {
    if (j < 0 || j >= rd.len) return 4;
    const int rev = rd.flag & 16;  // stored sequence is the reverse strand (bam_info.cpp:180-192)
    const int idx = rev ? rd.len - 1 - j : j;
    const int byte = raw[rd.off_seq + (idx >> 1)];
    const int nib = (idx & 1) ? (byte & 15) : (byte >> 4);
    int c = nib_to_code(nib, bad);
    if (rev && c < 4) c = 3 - c;
    return c;
}

// s_encode_signal_value (bam_info.cpp:455-478): u16 frame count -> codev1 byte
__device__ __forceinline__ int encode_frames(int s) {
    s = s > 952 ? 952 : s;
    if (s >= 448) return (s - 448) / 8 + 192;
    if (s >= 192) return (s - 192) / 4 + 128;
    if (s >= 64) return (s - 64) / 2 + 64;
    return s;
}

__device__ __forceinline__ uint32_t kin_code(const uint8_t* __restrict__ raw, int64_t off, int w, int idx) {
    if (w == 1) return raw[off + idx];
    const uint32_t v = raw[off + 2 * (int64_t)idx] | ((uint32_t)raw[off + 2 * (int64_t)idx + 1] << 8);
    return (uint32_t)encode_frames((int)v);
}

// context of the cytosine (forward C, or reverse-strand C seen as forward G) at the middle code.
// CpG : C G                      forward only   (eval_kmer_features.cpp:89-102)
// CHG : C [ACT] G                forward only   (:104-126)
// CHH : C [ACT] [ACT] forward ; [AGT] [AGT] G -> site on the G, reverse strand (:67-87)
__device__ __forceinline__ int classify(int cm2, int cm1, int c0, int c1, int c2) {
    if (c0 == 1) {
        if (c1 == 2) return CPG;
        const bool h1 = (c1 == 0) | (c1 == 1) | (c1 == 3);
        if (!h1) return CTX_NONE;
        if (c2 == 2) return CHG;
        if ((c2 == 0) | (c2 == 1) | (c2 == 3)) return CHH;
        return CTX_NONE;
    }
    if (c0 == 2) {
        const bool d1 = (cm1 == 0) | (cm1 == 2) | (cm1 == 3);
        const bool d2 = (cm2 == 0) | (cm2 == 2) | (cm2 == 3);
        return (d1 & d2) ? CHH : CTX_NONE;
    }
    return CTX_NONE;
}

constexpr int PREP_THREADS = 256;
constexpr int PER_THREAD = CHUNK / PREP_THREADS;  // 4 consecutive positions per thread
static_assert(PER_THREAD == 4, "prep kernels assume 4 positions per thread");

__global__ __launch_bounds__(PREP_THREADS) void prep_kernel(const uint8_t* __restrict__ raw,
                                                              const ReadDesc* __restrict__ reads,
                                                              const Chunk* __restrict__ chunks, int ctx_mask,
                                                              uint8_t* __restrict__ bases,
                                                              uint32_t* __restrict__ kin,
                                                              uint8_t* __restrict__ sctx,
                                                              int32_t* __restrict__ chunk_counts,
                                                              int32_t* __restrict__ err) {
    __shared__ int cnt[4];  // CpG, CHG, CHH, reverse-strand sites
    const Chunk ch = chunks[blockIdx.x];
    const ReadDesc rd = reads[ch.read_idx];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int j0 = ch.start + PER_THREAD * threadIdx.x;
    int bad = 0;
    int c[PER_THREAD + 4];
#pragma unroll
    for (int t = 0; t < PER_THREAD + 4; ++t) c[t] = fwd_code(raw, rd, j0 - 2 + t, bad);
    int my[4] = {0, 0, 0, 0};
    if (j0 < rd.len) {
        uint32_t bpack = 0, cpack = 0;
        uint32_t kq[PER_THREAD];
#pragma unroll
        for (int t = 0; t < PER_THREAD; ++t) {
            const int j = j0 + t;
            kq[t] = 0;
            if (j < rd.len) {
                bpack |= (uint32_t)c[t + 2] << (8 * t);
                // forward-coordinate packing: byte0 fi[j], byte1 fp[j], byte2 ri[L-1-j], byte3 rp[L-1-j]
                // (kinetics are indexed exactly as stored in the tag: bam_info.cpp:520-548)
                const int jr = rd.len - 1 - j;
                kq[t] = kin_code(raw, rd.off_fi, rd.w[0], j) | (kin_code(raw, rd.off_fp, rd.w[1], j) << 8) |
                        (kin_code(raw, rd.off_ri, rd.w[2], jr) << 16) | (kin_code(raw, rd.off_rp, rd.w[3], jr) << 24);
                int cls = classify(c[t], c[t + 1], c[t + 2], c[t + 3], c[t + 4]);
                if (cls != CTX_NONE && !((ctx_mask >> cls) & 1)) cls = CTX_NONE;
                if (cls != CTX_NONE) {
                    ++my[cls];
                    if (c[t + 2] == 2) ++my[3];  // forward G: the cytosine sits on the reverse strand
                }
                cpack |= (uint32_t)(cls | (c[t + 2] == 2 ? 4 : 0)) << (8 * t);   // context | strand << 2
            } else {
                cpack |= (uint32_t)CTX_NONE << (8 * t);
            }
        }
        const int64_t g = rd.base_off + j0;  // base_off and chunk starts are multiples of 4
        if (j0 + PER_THREAD <= rd.len) {
            *reinterpret_cast<uint32_t*>(bases + g) = bpack;
            *reinterpret_cast<uint32_t*>(sctx + g) = cpack;
            *reinterpret_cast<uint4*>(kin + g) = make_uint4(kq[0], kq[1], kq[2], kq[3]);
        } else {
#pragma unroll
            for (int t = 0; t < PER_THREAD; ++t)
                if (j0 + t < rd.len) {
                    bases[g + t] = (uint8_t)c[t + 2];
                    sctx[g + t] = (uint8_t)(cpack >> (8 * t));
                    kin[g + t] = kq[t];
                }
        }
    }
    if (my[0]) atomicAdd(&cnt[0], my[0]);
    if (my[1]) atomicAdd(&cnt[1], my[1]);
    if (my[2]) atomicAdd(&cnt[2], my[2]);
    if (my[3]) atomicAdd(&cnt[3], my[3]);
    if (bad) atomicOr(err, 1);
    __syncthreads();
    if (threadIdx.x < NCNT) {
        const int t = threadIdx.x;
        const int v = t < 3 ? cnt[t] : t == 3 ? cnt[0] + cnt[1] + cnt[2] : cnt[3];
        chunk_counts[NCNT * blockIdx.x + t] = v;
    }
}

// single-workgroup exclusive scan of the NCNT per-chunk counters; row n_chunks of `offs` holds the totals (the emit
// kernel reads "first chunk of the next read" there for the last read)
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(const int32_t* __restrict__ counts, int n_chunks,
                                                             int32_t* __restrict__ offs,
                                                             int32_t* __restrict__ totals) {
    __shared__ int part[NCNT][SCAN_THREADS];
    const int per = (n_chunks + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(n_chunks, (int)threadIdx.x * per), hi = min(n_chunks, lo + per);
    int s[NCNT];
#pragma unroll
    for (int c = 0; c < NCNT; ++c) s[c] = 0;
    for (int i = lo; i < hi; ++i)
#pragma unroll
        for (int c = 0; c < NCNT; ++c) s[c] += counts[NCNT * i + c];
#pragma unroll
    for (int c = 0; c < NCNT; ++c) part[c][threadIdx.x] = s[c];
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {
        int v[NCNT];
#pragma unroll
        for (int c = 0; c < NCNT; ++c) v[c] = threadIdx.x >= d ? part[c][threadIdx.x - d] : 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCNT; ++c) part[c][threadIdx.x] += v[c];
        __syncthreads();
    }
    int run[NCNT];
#pragma unroll
    for (int c = 0; c < NCNT; ++c) run[c] = part[c][threadIdx.x] - s[c];  // exclusive prefix of this thread's range
    for (int i = lo; i < hi; ++i)
#pragma unroll
        for (int c = 0; c < NCNT; ++c) {
            offs[NCNT * i + c] = run[c];
            run[c] += counts[NCNT * i + c];
        }
    if (threadIdx.x == SCAN_THREADS - 1) {
        const int t0 = part[0][threadIdx.x], t1 = part[1][threadIdx.x], t2 = part[2][threadIdx.x];
#pragma unroll
        for (int c = 0; c < NCNT; ++c) offs[NCNT * n_chunks + c] = part[c][threadIdx.x];
        totals[0] = t0;
        totals[1] = t1;
        totals[2] = t2;
        totals[3] = part[3][threadIdx.x];
        totals[4] = 0;        // ctx_base: the three context lists are laid out back to back
        totals[5] = t0;
        totals[6] = t0 + t1;
        totals[7] = part[4][threadIdx.x];  // reverse-strand sites
        for (int i = 8; i < 16; ++i) totals[i] = 0;  // per context: [8..10] trunk steps whose conv4 ran over listed rows only, [12..14] constant steps (hm_trunk3.hip)
    }
}

// wave64 inclusive prefix sum
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

__global__ __launch_bounds__(PREP_THREADS) void emit_kernel(const ReadDesc* __restrict__ reads,
                                                              const Chunk* __restrict__ chunks, int ctx_mask,
                                                              const uint8_t* __restrict__ bases,
                                                              const int32_t* __restrict__ chunk_offs,
                                                              const int32_t* __restrict__ totals,
                                                              USite* __restrict__ usites, uint8_t* __restrict__ utag,
                                                              Site* __restrict__ csites, int32_t* __restrict__ opos) {
    __shared__ int wave_tot[4][PREP_THREADS / 64];
    const Chunk ch = chunks[blockIdx.x];
    const ReadDesc rd = reads[ch.read_idx];
    const int j0 = ch.start + PER_THREAD * threadIdx.x;
    int c[PER_THREAD + 4];
#pragma unroll
    for (int t = 0; t < PER_THREAD + 4; ++t) {
        const int j = j0 - 2 + t;
        c[t] = (j >= 0 && j < rd.len) ? bases[rd.base_off + j] : 4;
    }
    int cls[PER_THREAD];
    int my[4] = {0, 0, 0, 0};  // CpG, CHG, CHH, reverse-strand sites
#pragma unroll
    for (int t = 0; t < PER_THREAD; ++t) {
        int k = CTX_NONE;
        if (j0 + t < rd.len) {
            k = classify(c[t], c[t + 1], c[t + 2], c[t + 3], c[t + 4]);
            if (k != CTX_NONE && !((ctx_mask >> k) & 1)) k = CTX_NONE;
        }
        cls[t] = k;
        if (k != CTX_NONE) {
            ++my[k];
            if (c[t + 2] == 2) ++my[3];
        }
    }
    // wavefront-level scans, then a 4-wave carry through LDS
    int incl[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        incl[k] = wave_incl_scan(my[k]);
        if (lane == 63) wave_tot[k][wave] = incl[k];
    }
    __syncthreads();
    const int32_t* co = chunk_offs + NCNT * blockIdx.x;
    int rank[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int carry = 0;
        for (int w = 0; w < wave; ++w) carry += wave_tot[k][w];
        rank[k] = co[k] + carry + incl[k] - my[k];
    }
    // Output order of the calls (mod_main.cpp:217-251): per read the forward-strand calls by qoff, then the
    // reverse-strand ones.  A read's chunks are consecutive, so its range in the unified list and its number of
    // reverse-strand sites follow from the scanned counters of its first chunk and of the chunk after its last.
    const int k0 = blockIdx.x - ch.start / CHUNK, k1 = k0 + (rd.len + CHUNK - 1) / CHUNK;
    const int read_first = chunk_offs[NCNT * k0 + 3], rev0 = chunk_offs[NCNT * k0 + 4];
    const int n_fwd = (chunk_offs[NCNT * k1 + 3] - read_first) - (chunk_offs[NCNT * k1 + 4] - rev0);
    int rev_before = co[4] - rev0 + incl[3] - my[3];  // reverse-strand sites of this read in front of this thread's
    for (int w = 0; w < wave; ++w) rev_before += wave_tot[3][w];
    // rank in the unified list: sites of all contexts in position order. chunk_offs[..][3] was
    // scanned from the per-chunk sum, thread-local order is position order, so the unified rank is
    // the chunk's unified offset plus the number of sites (any context) before this one in the chunk.
    int urank = co[3] + (rank[0] - co[0]) + (rank[1] - co[1]) + (rank[2] - co[2]);
#pragma unroll
    for (int t = 0; t < PER_THREAD; ++t) {
        const int k = cls[t];
        if (k == CTX_NONE) continue;
        const int q = j0 + t;
        const int strand = c[t + 2] == 2 ? 1 : 0;  // forward G = cytosine on the reverse strand
        usites[urank] = USite{ch.read_idx, q};
        utag[urank] = (uint8_t)(k | (strand << 2));
        csites[totals[4 + k] + rank[k]] = Site{ch.read_idx, q, urank};
        opos[urank] = strand ? read_first + n_fwd + rev_before : read_first + (urank - read_first - rev_before);
        rev_before += strand;
        ++rank[k];
        ++urank;
    }
}

// results -> hm_call_t records in output order: one 16-byte store per site into the packed array that the engine
// copies to the host with a single D2H (layout of hm_call_t: include/hifimeth_hip.h)
__global__ __launch_bounds__(256) void pack_kernel(const USite* __restrict__ usites, const uint8_t* __restrict__ utag,
                                                    const int32_t* __restrict__ opos, const float* __restrict__ prob,
                                                    const uint8_t* __restrict__ ml, const ReadDesc* __restrict__ reads,
                                                    const int32_t* __restrict__ totals, uint4* __restrict__ calls) {
    const int n = totals[3];
    for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
        const USite us = usites[u];
        const uint32_t tag = utag[u];
        const uint32_t word = (tag >> 2) | ((tag & 3u) << 8) | ((uint32_t)ml[u] << 16);  // strand, ctx, scaled_prob, reserved
        calls[opos[u]] = make_uint4((uint32_t)reads[us.read_idx].read_id, (uint32_t)us.qoff, word, __float_as_uint(prob[u]));
    }
}

// materialised 401x8 fp32 windows: one thread per window ROW (32 contiguous bytes), rows of consecutive sites
// back to back, so a wave writes 2 KB contiguous; the 1 KB decode table sits in LDS (12 832 B / site out).
// (One float4 per lane -- perfectly contiguous wave stores, but twice the index arithmetic -- measured slower:
//  3.4 vs 4.5 TB/s.  Non-temporal stores halve the rate: 2.3 TB/s.)
__global__ __launch_bounds__(256) void window_kernel(const Site* __restrict__ sites, int n,
                                                      const ReadDesc* __restrict__ reads,
                                                      const uint8_t* __restrict__ bases,
                                                      const uint32_t* __restrict__ kin,
                                                      const BnTables* __restrict__ bn, float* __restrict__ out) {
    __shared__ float lut[256];
    lut[threadIdx.x] = bn->raw_lut[threadIdx.x];
    __syncthreads();
    const int64_t total = (int64_t)n * KMER;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int s = (int)(g / KMER), r = (int)(g - (int64_t)s * KMER);
        const Site st = sites[s];
        const int L = reads[st.read_idx].len;
        const int64_t bo = reads[st.read_idx].base_off;
        const int rev = bases[bo + st.qoff] == 2;
        const int j = rev ? st.qoff + HK - r : st.qoff - HK + r;
        float4 hot = make_float4(0.f, 0.f, 0.f, 0.f), kv = hot;
        if (j >= 0 && j < L) {
            int b = bases[bo + j];
            uint32_t k = kin[bo + j];
            if (rev) {
                if (b < 4) b = 3 - b;
                k = (k >> 16) | (k << 16);  // own strand first: (ri, rp, fi, fp)
            }
            hot = make_float4(b == 0 ? 1.f : 0.f, b == 1 ? 1.f : 0.f, b == 2 ? 1.f : 0.f, b == 3 ? 1.f : 0.f);
            kv = make_float4(lut[k & 255], lut[(k >> 8) & 255], lut[(k >> 16) & 255], lut[k >> 24]);
        }
        float4* dst = reinterpret_cast<float4*>(out + g * FEATS);
        dst[0] = hot;
        dst[1] = kv;
    }
}

// ------------------------------------------------------------------------------------------------
// front kernel: window -> bn0 -> conv1 .. conv4   (one site per workgroup pass, all in LDS)
// ------------------------------------------------------------------------------------------------
template <int K1>
struct Geo {
    static constexpr int L1 = (KMER + 2 - K1) / 2 + 1;  // 197 (k=11) / 196 (k=13)
    static constexpr int L2 = (L1 - 1) / 2 + 1;         // 99 / 98
    static constexpr int L3 = (L2 - 1) / 2 + 1;         // 50 / 49
    static constexpr int L4 = (L3 - 1) / 2 + 1;         // 25 / 25
    static_assert(L4 == C4_LEN, "conv4 length");
    static constexpr int KT1 = (K1 * FEATS + 15) / 16 * 2;  // taps incl. zero-weight K padding: 12 / 14
    static constexpr int WRS = 12;                           // window row stride (floats, 16-B aligned rows)
    static constexpr int WROWS = 2 * (L1 - 1) + KT1;         // physical window rows touched by conv1
    static constexpr int RS = 132;                           // row stride of the 128-channel activations
    static constexpr int A1 = (L1 + 2) * RS, A2 = (L2 + 2) * RS, A3 = (L3 + 2) * RS, WIN = WROWS * WRS;
    static constexpr int BUFA = A1 > A3 ? A1 : A3;
    static constexpr int BUFB = WIN > A2 ? WIN : A2;
    static constexpr int LDS_FLOATS = BUFA + BUFB;
};

template <int K1, bool RAW, int NW, bool STAMP = false>
__global__ __launch_bounds__(NW * 64) void front_kernel(SiteRange sr,
                                                     const ReadDesc* __restrict__ reads,
                                                     const uint8_t* __restrict__ bases,
                                                     const uint32_t* __restrict__ kin,
                                                     const float* __restrict__ windows, CtxWeights W,
                                                     float* __restrict__ act4, float* __restrict__ dbg, int dbg_layer,
                                                     unsigned long long* __restrict__ stamps) {
    using G = Geo<K1>;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    __shared__ __attribute__((aligned(16))) float smem[G::LDS_FLOATS];
    float* bufA = smem;
    float* bufB = smem + G::BUFA;
    const BnTables* __restrict__ bn = W.bn;

    unsigned long long tacc[N_STAMP];
    unsigned long long tprev = 0;
    if (STAMP) {
#pragma unroll
        for (int i = 0; i < N_STAMP; ++i) tacc[i] = 0;
    }
    auto mk = [&](int ph) __attribute__((always_inline)) {
        if (STAMP) {
            const unsigned long long t_ = hm_stamp();
            tacc[ph] += t_ - tprev;
            tprev = t_;
        }
    };
#define HM_MARK(PH)                               \
    if (STAMP) {                                  \
        const unsigned long long t_ = hm_stamp(); \
        tacc[PH] += t_ - tprev;                   \
        tprev = t_;                               \
    }
    // window rows of site s -> bufB with bn0 applied, by threads [t0, t0 + nt) of the workgroup.
    // physical row pr holds window row pr-1; rows outside [0,401) are the conv's zero padding (and the
    // zero-weight K padding of conv1).
    auto build_window = [&](const int s, const int t, const int nt) __attribute__((always_inline)) {
        if (RAW) {
            const Site st = sites[s];
            const int L = reads[st.read_idx].len;
            const int64_t bo = reads[st.read_idx].base_off;
            const int rev = bases[bo + st.qoff] == 2;
            for (int pr = t; pr < G::WROWS; pr += nt) {
                const int w = pr - 1;
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (w >= 0 && w < KMER) {
                    const int j = rev ? st.qoff + HK - w : st.qoff - HK + w;
                    if (j < 0 || j >= L) {  // outside the read: the reference zero-fills BEFORE bn0
                        lo = make_float4(bn->zero[0], bn->zero[1], bn->zero[2], bn->zero[3]);
                        hi = make_float4(bn->zero[4], bn->zero[5], bn->zero[6], bn->zero[7]);
                    } else {
                        int b = bases[bo + j];
                        uint32_t k = kin[bo + j];
                        if (rev) {
                            if (b < 4) b = 3 - b;
                            k = (k >> 16) | (k << 16);  // own strand first: (ri, rp, fi, fp)
                        }
                        lo = make_float4(b == 0 ? bn->hot[0] : bn->zero[0], b == 1 ? bn->hot[1] : bn->zero[1],
                                         b == 2 ? bn->hot[2] : bn->zero[2], b == 3 ? bn->hot[3] : bn->zero[3]);
                        hi = make_float4(bn->lut[0][k & 255], bn->lut[1][(k >> 8) & 255], bn->lut[2][(k >> 16) & 255],
                                         bn->lut[3][k >> 24]);
                    }
                }
                *reinterpret_cast<float4*>(bufB + pr * G::WRS) = lo;
                *reinterpret_cast<float4*>(bufB + pr * G::WRS + 4) = hi;
            }
        } else {
            const float* __restrict__ src = windows + (size_t)s * (KMER * FEATS);
            for (int i = t; i < G::WROWS * 8; i += nt) {
                const int pr = i >> 3, c = i & 7, w = pr - 1;
                float v = 0.f;
                if (w >= 0 && w < KMER) v = (src[w * 8 + c] - bn->mean[c]) / bn->sd[c] * bn->gamma[c] + bn->beta[c];
                bufB[pr * G::WRS + c] = v;
            }
        }
    };
    // waves that do not take part in conv4 (8-wave build) prepare the next site's window meanwhile
    constexpr int C4_WAVES = 4;
    constexpr bool SPARE = NW > C4_WAVES;

    if ((int)blockIdx.x < n_sites) build_window(blockIdx.x, threadIdx.x, NW * 64);
    for (int s = blockIdx.x; s < n_sites; s += gridDim.x) {
        if (STAMP) tprev = hm_stamp();
        HM_MARK(0)
        __syncthreads();  // window of site s complete; previous conv4 done with bufA
        HM_MARK(1)

        // conv1: window (bufB) -> bufA
        Conv<NW, 1, 8, G::KT1, 128, G::L1, G::WRS, 0, 0, 1, NW, 2, G::L1 % 16>::run(
            bufB, W.wfrag[0], EpiLds<G::L1, G::RS, 0>{bufA, W.bias[0]}, [&](int k) __attribute__((always_inline)) { mk(2 + k); });
        HM_MARK(4)
        zero_pad_rows<1, G::L1, 128, G::RS, 0>(bufA);
        __syncthreads();
        HM_MARK(5)
        if (dbg && dbg_layer == 1 && s == 0) dump_lds<G::L1, 128, G::RS>(bufA, dbg);

        // conv2: bufA -> bufB
        Conv<NW, 1, 128, 3, 128, G::L2, G::RS, 0, 0, 1, NW, 2, G::L2 % 16>::run(
            bufA, W.wfrag[1], EpiLds<G::L2, G::RS, 0>{bufB, W.bias[1]}, [&](int k) __attribute__((always_inline)) { mk(6 + k); });
        HM_MARK(8)
        zero_pad_rows<1, G::L2, 128, G::RS, 0>(bufB);
        __syncthreads();
        HM_MARK(9)
        if (dbg && dbg_layer == 2 && s == 0) dump_lds<G::L2, 128, G::RS>(bufB, dbg);

        // conv3: bufB -> bufA
        Conv<NW, 1, 128, 3, 128, G::L3, G::RS, 0, 0, 1, NW, 3, G::L3 % 16>::run(
            bufB, W.wfrag[2], EpiLds<G::L3, G::RS, 0>{bufA, W.bias[2]}, [&](int k) __attribute__((always_inline)) { mk(10 + k); });
        HM_MARK(12)
        zero_pad_rows<1, G::L3, 128, G::RS, 0>(bufA);
        __syncthreads();
        HM_MARK(13)
        if (dbg && dbg_layer == 3 && s == 0) dump_lds<G::L3, 128, G::RS>(bufA, dbg);

        // conv4: bufA -> act4[s] in HBM (hand-off to the tail kernel), on the first 4 waves (one per SIMD,
        // 3 tiles each); bufB is free now, so the next site's window is built alongside.
        Conv<NW, 1, 128, 3, C4_CH, G::L4, G::RS, 0, 0, 2, 2, 6>::run(
            bufA, W.wfrag[3], EpiGlobal<C4_CH>{act4 + (size_t)s * ACT4_FLOATS, W.bias[3]}, [&](int k) __attribute__((always_inline)) { mk(14 + k); });
        HM_MARK(16)
        const int sn = s + gridDim.x;
        if (sn < n_sites) {
            if (SPARE) {
                if ((int)threadIdx.x >= C4_WAVES * 64) build_window(sn, threadIdx.x - C4_WAVES * 64, (NW - C4_WAVES) * 64);
            } else {
                build_window(sn, threadIdx.x, NW * 64);
            }
        }
        HM_MARK(17)
    }
#undef HM_MARK
    if (STAMP && (threadIdx.x & 63) == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * N_STAMP;
#pragma unroll
        for (int i = 0; i < N_STAMP; ++i) o[i] = tacc[i];
    }
}

// ------------------------------------------------------------------------------------------------
// tail kernel: conv5 .. conv8, fc1, fc2, softmax, 8 sites stacked along M per workgroup pass
// ------------------------------------------------------------------------------------------------
struct TailGeo {
    static constexpr int S = TAIL_SITES;
    static constexpr int L4 = 25, L5 = 13, L6 = 7, L7 = 4, L8 = 2;
    static constexpr int RS96 = 100, RS64 = 68, HRS = 260;
    static constexpr int IN_SS = (L4 + 2) * RS96, C5_SS = (L5 + 2) * RS96, C6_SS = (L6 + 2) * RS96;
    static constexpr int C7_SS = (L7 + 2) * RS64, C8_SS = (L8 + 2) * RS64;
    static constexpr int BUF0 = S * IN_SS;  // also holds conv6 / conv8 outputs
    static constexpr int BUF1 = S * C5_SS;  // also holds conv7 output and the fc1 activations
    static_assert(S * C6_SS <= BUF0 && S * C8_SS <= BUF0 && S * C7_SS <= BUF1 && S * HRS <= BUF1, "tail LDS plan");
    static constexpr int LDS_FLOATS = BUF0 + BUF1;
};

// e4 != nullptr (dense trunk path, hm_trunk_f32.hip): a site's conv4 rows are gathered where they were computed -- rows
// 1..23 from the fp32 E4 map (row e4row[site] + 16 s), rows 0 and 24 from the edge kernel -- instead of read from act4.
__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ act4, SiteRange sr, CtxWeights W,
                                                    float* __restrict__ logits,
                                                    float* __restrict__ prob, uint8_t* __restrict__ ml,
                                                    float* __restrict__ dbg, int dbg_layer,
                                                    const float* __restrict__ e4, const float* __restrict__ edge4,
                                                    const int32_t* __restrict__ e4row) {
    using T = TailGeo;
    constexpr int S = T::S;
    const Site* sites;
    const int n_sites = resolve_sites(sr, sites);
    __shared__ __attribute__((aligned(16))) float smem[T::LDS_FLOATS];
    float* buf0 = smem;
    float* buf1 = smem + T::BUF0;

    for (int g = blockIdx.x; g * S < n_sites; g += gridDim.x) {
        const int s0 = g * S;
        const int nv = min(S, n_sites - s0);
        // act4 [site][25][96] -> buf0 rows 1..25 ; rows 0 and 26 are the conv padding
        for (int i = threadIdx.x; i < S * (ACT4_FLOATS / 4); i += 256) {
            const int site = i / (ACT4_FLOATS / 4), rem = (i - site * (ACT4_FLOATS / 4)) * 4;
            const int pos = rem / C4_CH, c = rem - pos * C4_CH;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (site < nv) {
                const float* src = act4 + (size_t)(s0 + site) * ACT4_FLOATS + rem;
                if (e4)
                    src = pos == 0           ? edge4 + (size_t)(s0 + site) * (2 * C4_CH) + c
                          : pos == C4_LEN - 1 ? edge4 + (size_t)(s0 + site) * (2 * C4_CH) + C4_CH + c
                                              : e4 + ((size_t)e4row[s0 + site] + 16 * pos) * C4_CH + c;
                v = *reinterpret_cast<const float4*>(src);
            }
            float* d = buf0 + site * T::IN_SS + (pos + 1) * T::RS96 + c;
            d[0] = v.x;
            d[1] = v.y;
            d[2] = v.z;
            d[3] = v.w;
        }
        zero_pad_rows<S, T::L4, 96, T::RS96, T::IN_SS>(buf0);
        __syncthreads();

        Conv<4, S, 96, 3, 96, T::L5, T::RS96, T::IN_SS, 0, 2, 2>::run(
            buf0, W.wfrag[4], EpiLds<T::L5, T::RS96, T::C5_SS>{buf1, W.bias[4]});
        zero_pad_rows<S, T::L5, 96, T::RS96, T::C5_SS>(buf1);
        __syncthreads();
        if (dbg && dbg_layer == 5 && g == 0) dump_lds<T::L5, 96, T::RS96>(buf1, dbg);

        Conv<4, S, 96, 3, 96, T::L6, T::RS96, T::C5_SS, 0, 2, 2>::run(
            buf1, W.wfrag[5], EpiLds<T::L6, T::RS96, T::C6_SS>{buf0, W.bias[5]});
        zero_pad_rows<S, T::L6, 96, T::RS96, T::C6_SS>(buf0);
        __syncthreads();
        if (dbg && dbg_layer == 6 && g == 0) dump_lds<T::L6, 96, T::RS96>(buf0, dbg);

        Conv<4, S, 96, 3, 64, T::L7, T::RS96, T::C6_SS, 0, 2, 2>::run(
            buf0, W.wfrag[6], EpiLds<T::L7, T::RS64, T::C7_SS>{buf1, W.bias[6]});
        zero_pad_rows<S, T::L7, 64, T::RS64, T::C7_SS>(buf1);
        __syncthreads();
        if (dbg && dbg_layer == 7 && g == 0) dump_lds<T::L7, 64, T::RS64>(buf1, dbg);

        Conv<4, S, 64, 3, 64, T::L8, T::RS64, T::C7_SS, 0, 1, 4>::run(
            buf1, W.wfrag[7], EpiLds<T::L8, T::RS64, T::C8_SS>{buf0, W.bias[7]});
        __syncthreads();
        if (dbg && dbg_layer == 8 && g == 0) dump_lds<T::L8, 64, T::RS64>(buf0, dbg);

        // fc1 as a 2-tap "conv" over the two positions of conv8's output: A[site][l*64 + c] = rows 1,2.
        // (weights are packed in that k order from fc1.weight[o][c*2 + l], the channel-major flatten
        //  of model_cnn.py:79).  Output h[site][256] with row stride HRS: (p+1)*ORS with ORS = 0.
        Conv<4, S, 64, 2, 256, 1, T::RS64, T::C8_SS, 1, 1, 4>::run(
            buf0, W.wfrag[8], EpiLds<1, 0, T::HRS>{buf1, W.bias[8]});
        __syncthreads();

        // fc2 + softmax (mod_batch.cpp:46-64): 16 lanes per site = 2 outputs x 8 partial sums
        if (threadIdx.x < S * 16) {
            const int site = threadIdx.x >> 4, o = (threadIdx.x >> 3) & 1, part = threadIdx.x & 7;
            const float* h = buf1 + site * T::HRS + part * 32;
            const float* w2 = W.fc2_w + o * 256 + part * 32;
            float sum = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) sum = fmaf(h[k], w2[k], sum);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 1, 64);
            sum += W.fc2_b[o];
            const float other = __shfl_xor(sum, 8, 64);
            if ((threadIdx.x & 15) == 0 && site < nv) {
                const float v0 = sum, v1 = other;
                const float mx = fmaxf(v0, v1);
                const float e0 = expf(v0 - mx), e1 = expf(v1 - mx);
                const float p1 = e1 / (e0 + e1);
                int q = (int)(255 * p1);
                q = q > 255 ? 255 : q;
                const int dst = sites ? sites[s0 + site].uidx : s0 + site;
                logits[2 * (size_t)dst] = v0;
                logits[2 * (size_t)dst + 1] = v1;
                prob[dst] = p1;
                ml[dst] = (uint8_t)q;
            }
        }
        __syncthreads();
    }
}

// =================================================================================================
// launchers
// =================================================================================================
void launch_prep(hipStream_t st, const uint8_t* raw, const ReadDesc* reads, const Chunk* chunks, int n_chunks,
                 int ctx_mask, uint8_t* bases, uint32_t* kin, uint8_t* sctx, int32_t* chunk_counts, int32_t* err) {
    if (n_chunks <= 0) return;
    hipLaunchKernelGGL(prep_kernel, dim3(n_chunks), dim3(PREP_THREADS), 0, st, raw, reads, chunks, ctx_mask, bases, kin,
                       sctx, chunk_counts, err);
}

void launch_scan(hipStream_t st, const int32_t* chunk_counts, int n_chunks, int32_t* chunk_offs, int32_t* totals) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, chunk_counts, n_chunks, chunk_offs, totals);
}

void launch_emit(hipStream_t st, const ReadDesc* reads, const Chunk* chunks, int n_chunks, int ctx_mask,
                 const uint8_t* bases, const int32_t* chunk_offs, const int32_t* totals, USite* usites,
                 uint8_t* utag, Site* csites, int32_t* opos) {
    if (n_chunks <= 0) return;
    hipLaunchKernelGGL(emit_kernel, dim3(n_chunks), dim3(PREP_THREADS), 0, st, reads, chunks, ctx_mask, bases,
                       chunk_offs, totals, usites, utag, csites, opos);
}

void launch_pack(hipStream_t st, const USite* usites, const uint8_t* utag, const int32_t* opos, const float* prob,
                 const uint8_t* ml, const ReadDesc* reads, const int32_t* totals, void* calls, int grid) {
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, st, usites, utag, opos, prob, ml, reads, totals,
                       reinterpret_cast<uint4*>(calls));
}

void launch_windows(hipStream_t st, const Site* sites, int n, const ReadDesc* reads, const uint8_t* bases,
                    const uint32_t* kin, const BnTables* bn, float* out, int grid) {
    if (n <= 0) return;
    const int64_t blocks = ((int64_t)n * KMER + 255) / 256;
    hipLaunchKernelGGL(window_kernel, dim3((unsigned)std::min<int64_t>(blocks, grid)), dim3(256), 0, st, sites, n, reads,
                       bases, kin, bn, out);
}

// number of workgroups of a persistent CNN launch: the site count is only known exactly when it is given directly
static int cnn_grid(const SiteRange& sr, int per_group, int grid) {
    if (sr.totals) return grid;
    return std::max(1, std::min((sr.cap + per_group - 1) / per_group, grid));
}

void launch_front(hipStream_t st, int k1, const SiteRange& sr, const ReadDesc* reads, const uint8_t* bases,
                  const uint32_t* kin, const float* windows, const CtxWeights& w, float* act4, int grid, float* dbg,
                  int dbg_layer, int waves, unsigned long long* stamps) {
    if (sr.cap <= 0) return;
    const dim3 g(cnn_grid(sr, 1, grid));
    const bool raw = windows == nullptr;
#define HM_FRONT(K1, RAW, NW, ST)                                                                                  \
    hipLaunchKernelGGL((front_kernel<K1, RAW, NW, ST>), g, dim3(NW * 64), 0, st, sr, reads, bases, kin, windows, w, \
                       act4, dbg, dbg_layer, stamps)
    if (stamps && raw && waves == 8) {  // diagnostic build of the production configuration
        if (k1 == 11) HM_FRONT(11, true, 8, true); else HM_FRONT(13, true, 8, true);
    } else if (waves == 8) {
        if (k1 == 11) { if (raw) HM_FRONT(11, true, 8, false); else HM_FRONT(11, false, 8, false); }
        else { if (raw) HM_FRONT(13, true, 8, false); else HM_FRONT(13, false, 8, false); }
    } else {
        if (k1 == 11) { if (raw) HM_FRONT(11, true, 4, false); else HM_FRONT(11, false, 4, false); }
        else { if (raw) HM_FRONT(13, true, 4, false); else HM_FRONT(13, false, 4, false); }
    }
#undef HM_FRONT
}

int front_stamp_slots() { return N_STAMP; }

void launch_tail(hipStream_t st, const float* act4, const SiteRange& sr, const CtxWeights& w, float* logits,
                 float* p, uint8_t* ml, int grid, float* dbg, int dbg_layer) {
    if (sr.cap <= 0) return;
    hipLaunchKernelGGL(tail_kernel, dim3(cnn_grid(sr, TAIL_SITES, grid)), dim3(256), 0, st, act4, sr, w, logits, p, ml, dbg,
                       dbg_layer, nullptr, nullptr, nullptr);
}

void launch_tail_gather_f32(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const float* edge4,
                            const int32_t* e4row, float* logits, float* p, uint8_t* ml, int grid) {
    if (sr.cap <= 0) return;
    hipLaunchKernelGGL(tail_kernel, dim3(cnn_grid(sr, TAIL_SITES, grid)), dim3(256), 0, st, nullptr, sr, w, logits, p, ml, nullptr,
                       0, reinterpret_cast<const float*>(maps.e4), edge4, e4row);
}

size_t front_lds_bytes(int k1) { return sizeof(float) * (k1 == 11 ? Geo<11>::LDS_FLOATS : Geo<13>::LDS_FLOATS); }
size_t tail_lds_bytes() { return sizeof(float) * TailGeo::LDS_FLOATS; }

}  // namespace hm
