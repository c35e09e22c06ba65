// hm_engine.cpp -- host side of libhifimeth_hip.so: model loading, read staging, batch
// execution and the C ABI of include/hifimeth_hip.h.  Mirrors what ModModels, ModBatch and the
// worker loop do around the hot path in the reference (src/app/hifimeth/mod_main.cpp:18-262).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hifimeth_hip.h"
#include "hm_device.h"
#include "hm_kernels.h"
#include "hm_weights.h"

using namespace hm;

namespace {

thread_local std::string g_create_error;

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
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIP_TRY(hipMalloc(&p, want));
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

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t cap = 0, size = 0;
    void ensure(size_t extra) {
        if (size + extra <= cap) return;
        size_t want = std::max<size_t>((size + extra) * 2, size_t(64) << 20);
        uint8_t* q = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&q), want, hipHostMallocDefault));
        if (size) memcpy(q, p, size);
        if (p) (void)hipHostFree(p);
        p = q;
        cap = want;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = size = 0;
    }
};

enum Kind { K_PREP, K_SCAN, K_EMIT, K_WINDOW, K_FRONT0, K_FRONT1, K_FRONT2, K_TAIL0, K_TAIL1, K_TAIL2 };

struct TimedSpan {
    int kind;
    int64_t sites;
    hipEvent_t a, b;
};

struct DeviceModel {
    bool loaded = false;
    int k1 = 0;
    DevBuf params;  // all fragment-packed weights, biases, fc2, bn tables in one allocation
    CtxWeights w{};
};

}  // namespace

struct hm_engine {
    int device = 0;
    int num_cu = 256;
    int ctx_mask = 7;
    int min_read_size = 1000;  // mod_options.cpp:10
    int64_t sub_batch = 65536;
    int front_waves = 8;
    int precision = 1;  // 1 = split-half f16x3 MFMA with fp32 accumulate (default), 0 = fp32 MFMA
    bool stamps_on = false;
    std::vector<unsigned long long> stamp_sum;
    bool timing = false;
    hipStream_t stream = nullptr;
    DeviceModel model[3];
    std::string err;

    // staged batch (host)
    PinnedBuf slab;
    std::vector<ReadDesc> reads;
    std::vector<Chunk> chunks;
    int64_t total_bases = 0;  // padded to a multiple of 4 per read
    bool uploaded = false, ran = false, synced = false;

    // device
    DevBuf d_raw, d_reads, d_chunks, d_bases, d_kin, d_counts, d_offs, d_totals, d_err;
    DevBuf d_usites, d_utag, d_csites, d_logits, d_p, d_ml, d_act4, d_win, d_dbg, d_stamps;
    int32_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t* h_totals = nullptr;  // pinned
    int32_t* h_err = nullptr;     // pinned

    // timing
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> pool;
    hm_timing_t acc{};
};

namespace {

int fail(hm_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    else g_create_error = msg;
    return code;
}

int fail_hip(hm_engine* e, const HipErr& h) {
    return fail(e, HM_EDEVICE, std::string("HIP error: ") + hipGetErrorString(h.code) + " at " + h.what);
}

hipEvent_t get_event(hm_engine* e) {
    if (!e->pool.empty()) {
        hipEvent_t ev = e->pool.back();
        e->pool.pop_back();
        return ev;
    }
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    return ev;
}

struct Span {
    hm_engine* e;
    TimedSpan ts{};
    bool on;
    Span(hm_engine* eng, int kind, int64_t sites) : e(eng), on(eng->timing) {
        if (!on) return;
        ts.kind = kind;
        ts.sites = sites;
        ts.a = get_event(e);
        ts.b = get_event(e);
        HIP_TRY(hipEventRecord(ts.a, e->stream));
    }
    void end() {
        if (!on) return;
        HIP_TRY(hipEventRecord(ts.b, e->stream));
        e->spans.push_back(ts);
        on = false;
    }
};

void collect_timing(hm_engine* e) {
    for (auto& s : e->spans) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        hm_timing_t& t = e->acc;
        switch (s.kind) {
        case K_PREP: t.prep_ms += ms; ++t.prep_launches; break;
        case K_SCAN: t.scan_ms += ms; ++t.scan_launches; break;
        case K_EMIT: t.emit_ms += ms; ++t.emit_launches; break;
        case K_WINDOW: t.window_ms += ms; ++t.window_launches; t.window_sites += s.sites; break;
        case K_FRONT0: case K_FRONT1: case K_FRONT2:
            t.front_ms[s.kind - K_FRONT0] += ms; ++t.front_launches[s.kind - K_FRONT0];
            t.front_sites[s.kind - K_FRONT0] += s.sites; break;
        default:
            t.tail_ms[s.kind - K_TAIL0] += ms; ++t.tail_launches[s.kind - K_TAIL0]; break;
        }
        e->pool.push_back(s.a);
        e->pool.push_back(s.b);
    }
    e->spans.clear();
}

// ---- model -> device ----------------------------------------------------------------------
void upload_model(hm_engine* e, int ctx, const HostModel& hmw) {
    DeviceModel& dm = e->model[ctx];
    PackedModel pk = pack_model(hmw);
    dm.params.reserve(pk.blob.size() * sizeof(float));
    HIP_TRY(hipMemcpy(dm.params.p, pk.blob.data(), pk.blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* base = dm.params.as<float>();
    for (int i = 0; i < 9; ++i) {
        dm.w.wfrag[i] = base + pk.wfrag_off[i];
        dm.w.bias[i] = base + pk.bias_off[i];
    }
    dm.w.fc2_w = base + pk.fc2_w_off;
    dm.w.fc2_b = base + pk.fc2_b_off;
    dm.w.bn = reinterpret_cast<const BnTables*>(base + pk.bn_off);
    for (int i = 0; i < 9; ++i) dm.w.wfrag_h[i] = reinterpret_cast<const uint16_t*>(base + pk.wfrag_h_off[i]);
    dm.w.bn_h = reinterpret_cast<const BnTablesH*>(base + pk.bn_h_off);
    dm.w.c1f = reinterpret_cast<const uint16_t*>(base + pk.c1f_off);
    dm.w.c1f_bias = base + pk.c1f_bias_off;
    dm.w.c1f_corr = base + pk.c1f_corr_off;
    dm.w.k1 = hmw.k1;
    dm.k1 = hmw.k1;
    dm.loaded = true;
}

size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

void ensure_site_buffers(hm_engine* e, int64_t n) {
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    e->d_logits.reserve(nn * 2 * sizeof(float));
    e->d_p.reserve(nn * sizeof(float));
    e->d_ml.reserve(nn);
}

// diagnostic: sum the per-wave phase stamps of the front launch that was just queued
void accumulate_stamps(hm_engine* e) {
    const int ns = front_stamp_slots();
    const size_t n = (size_t)e->num_cu * 8 * ns;
    std::vector<unsigned long long> h(n);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(h.data(), e->d_stamps.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(e->d_stamps.p, 0, n * sizeof(unsigned long long)));
    e->stamp_sum.resize((size_t)ns * 8, 0);
    for (size_t i = 0; i < n; ++i) e->stamp_sum[i % ((size_t)ns * 8)] += h[i];  // [wave in workgroup][slot]
}

// front + tail over `n` sites of one context, in sub-batches that bound the act4 hand-off buffer
void run_cnn(hm_engine* e, int ctx, const Site* sites, const float* windows, int64_t n, float* dbg, int dbg_layer) {
    const DeviceModel& dm = e->model[ctx];
    const int64_t sb = e->sub_batch;
    e->d_act4.reserve((size_t)std::min<int64_t>(std::max<int64_t>(n, 1), sb) * ACT4_FLOATS * sizeof(float));
    for (int64_t off = 0; off < n; off += sb) {
        const int m = (int)std::min<int64_t>(sb, n - off);
        const Site* s_off = sites ? sites + off : nullptr;
        const float* w_off = windows ? windows + (size_t)off * KMER * FEATS : nullptr;
        {
            Span sp(e, K_FRONT0 + ctx, m);
            if (e->precision >= 1)
                launch_front_h(e->stream, dm.k1, s_off, m, e->d_reads.as<ReadDesc>(), e->d_bases.as<uint8_t>(),
                               e->d_kin.as<uint32_t>(), w_off, dm.w, e->d_act4.as<float>(), e->num_cu, dbg, dbg_layer,
                               e->stamps_on ? e->d_stamps.as<unsigned long long>() : nullptr, e->precision == 2);
            else
                launch_front(e->stream, dm.k1, s_off, m, e->d_reads.as<ReadDesc>(), e->d_bases.as<uint8_t>(),
                             e->d_kin.as<uint32_t>(), w_off, dm.w, e->d_act4.as<float>(), e->num_cu, dbg, dbg_layer,
                             e->front_waves, e->stamps_on ? e->d_stamps.as<unsigned long long>() : nullptr);
            if (e->stamps_on) accumulate_stamps(e);
            sp.end();
        }
        {
            Span sp(e, K_TAIL0 + ctx, m);
            // results land at sites[i].uidx for staged reads, at off + i for caller-supplied windows
            float* lg = e->d_logits.as<float>() + (sites ? 0 : 2 * off);
            float* pp = e->d_p.as<float>() + (sites ? 0 : off);
            uint8_t* mm = e->d_ml.as<uint8_t>() + (sites ? 0 : off);
            if (e->precision >= 1)
                launch_tail_h(e->stream, e->d_act4.as<float>(), m, dm.w, s_off, lg, pp, mm, e->num_cu, dbg, dbg_layer,
                              e->precision == 2);
            else
                launch_tail(e->stream, e->d_act4.as<float>(), m, dm.w, s_off, lg, pp, mm, e->num_cu, dbg, dbg_layer);
            sp.end();
        }
    }
    HIP_TRY(hipGetLastError());
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int hm_create(hm_engine_t** out, const char* model_dir, int ctx_mask, int device) {
    if (!out || !model_dir || !(ctx_mask & 7)) return fail(nullptr, HM_EINVAL, "hm_create: bad argument");
    *out = nullptr;
    hm_engine* e = new hm_engine();
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
            delete e;
            return fail(nullptr, HM_EDEVICE, "hm_create: no usable HIP device (this library has no CPU fallback)");
        }
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        e->device = device;
        e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        e->ctx_mask = ctx_mask & 7;
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_totals), 8 * sizeof(int32_t), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_err), sizeof(int32_t), hipHostMallocDefault));
        static const char* names[3] = {"CpG", "CHG", "CHH"};  // mod_main.cpp:76,85,94
        for (int c = 0; c < 3; ++c) {
            if (!(e->ctx_mask >> c & 1)) continue;
            HostModel hmw;
            std::string msg;
            if (!load_model_dir(model_dir, names[c], hmw, msg)) {
                hm_destroy(e);
                return fail(nullptr, HM_EMODEL, msg);
            }
            upload_model(e, c, hmw);
        }
        e->d_totals.reserve(8 * sizeof(int32_t));
        e->d_err.reserve(sizeof(int32_t));
        HIP_TRY(hipMemset(e->d_err.p, 0, sizeof(int32_t)));
    } catch (const HipErr& h) {
        int rc = fail_hip(nullptr, h);
        hm_destroy(e);
        return rc;
    }
    *out = e;
    return HM_OK;
}

void hm_destroy(hm_engine_t* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& m : e->model) m.params.release();
    for (DevBuf* b : {&e->d_raw, &e->d_reads, &e->d_chunks, &e->d_bases, &e->d_kin, &e->d_counts, &e->d_offs,
                      &e->d_totals, &e->d_err, &e->d_usites, &e->d_utag, &e->d_csites, &e->d_logits, &e->d_p,
                      &e->d_ml, &e->d_act4, &e->d_win, &e->d_dbg, &e->d_stamps})
        b->release();
    e->slab.release();
    if (e->h_totals) (void)hipHostFree(e->h_totals);
    if (e->h_err) (void)hipHostFree(e->h_err);
    for (auto& s : e->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

const char* hm_last_error(const hm_engine_t* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int hm_set_option(hm_engine_t* e, const char* key, int64_t value) {
    if (!e || !key) return HM_EINVAL;
    const std::string k(key);
    if (k == "min_read_size") e->min_read_size = (int)value;
    else if (k == "timing") e->timing = value != 0;
    else if (k == "precision") {
        if (value < 0 || value > 2) return fail(e, HM_EINVAL, "precision must be 0 (fp32), 1 (f16x3 split) or 2 (fp16 weights)");
        e->precision = (int)value;
    } else if (k == "stamps") {
        e->stamps_on = value != 0;
        if (e->stamps_on) {
            try {
                const size_t bytes = (size_t)e->num_cu * 8 * front_stamp_slots() * sizeof(unsigned long long);
                e->d_stamps.reserve(bytes);
                HIP_TRY(hipMemset(e->d_stamps.p, 0, bytes));
                e->stamp_sum.assign((size_t)front_stamp_slots() * 8, 0);
            } catch (const HipErr& h) {
                return fail_hip(e, h);
            }
        }
    } else if (k == "front_waves") {
        if (value != 4 && value != 8) return fail(e, HM_EINVAL, "front_waves must be 4 or 8");
        e->front_waves = (int)value;
    } else if (k == "sub_batch_sites") {
        if (value < TAIL_SITES) return fail(e, HM_EINVAL, "sub_batch_sites too small");
        e->sub_batch = value / TAIL_SITES * TAIL_SITES;
    } else return fail(e, HM_EINVAL, "unknown option " + k);
    return HM_OK;
}

int hm_submit_read(hm_engine_t* e, int32_t read_id, int32_t l_qseq, int32_t flag, const uint8_t* seq4,
                   const void* fi, int fi_w, const void* fp, int fp_w, const void* ri, int ri_w, const void* rp,
                   int rp_w) {
    if (!e || l_qseq < 0) return HM_EINVAL;
    if (e->uploaded) return fail(e, HM_ESTATE, "hm_submit_read: batch already uploaded; hm_clear first");
    if (l_qseq < e->min_read_size) return 0;        // mod_main.cpp:189-192
    if (!fi || !fp || !ri || !rp) return 0;          // BamKinetics::init false (bam_info.cpp:572-603)
    if (!seq4) return fail(e, HM_EINVAL, "hm_submit_read: seq4 is NULL");
    // site ranks and per-base offsets inside one batch are 32-bit on the device: keep a batch under 2^31 bases
    if (e->total_bases + (int64_t)l_qseq + 4 >= (int64_t(1) << 31))
        return fail(e, HM_ENOMEM, "hm_submit_read: batch would exceed 2^31 bases; hm_flush / hm_drain first");
    const int w[4] = {fi_w, fp_w, ri_w, rp_w};
    for (int k = 0; k < 4; ++k)
        if (w[k] != 1 && w[k] != 2) return fail(e, HM_EINVAL, "kinetics element width must be 1 (B:C) or 2 (B:S)");
    try {
        const size_t L = (size_t)l_qseq;
        const size_t need = align16((L + 1) / 2) + align16(L * fi_w) + align16(L * fp_w) + align16(L * ri_w) + align16(L * rp_w);
        e->slab.ensure(need);
        ReadDesc rd{};
        auto put = [&](const void* src, size_t bytes) {
            const int64_t off = (int64_t)e->slab.size;
            memcpy(e->slab.p + off, src, bytes);
            e->slab.size += align16(bytes);
            return off;
        };
        rd.off_seq = put(seq4, (L + 1) / 2);
        rd.off_fi = put(fi, L * fi_w);
        rd.off_fp = put(fp, L * fp_w);
        rd.off_ri = put(ri, L * ri_w);
        rd.off_rp = put(rp, L * rp_w);
        rd.base_off = e->total_bases;
        rd.len = l_qseq;
        rd.flag = flag;
        rd.read_id = read_id;
        for (int k = 0; k < 4; ++k) rd.w[k] = (uint8_t)w[k];
        const int ridx = (int)e->reads.size();
        e->reads.push_back(rd);
        for (int st = 0; st < l_qseq; st += CHUNK) e->chunks.push_back(Chunk{ridx, st});
        e->total_bases += (int64_t)((L + 3) & ~size_t(3));
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return 1;
}

int hm_clear(hm_engine_t* e) {
    if (!e) return HM_EINVAL;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    e->slab.size = 0;
    e->reads.clear();
    e->chunks.clear();
    e->total_bases = 0;
    e->uploaded = e->ran = e->synced = false;
    memset(e->totals, 0, sizeof e->totals);
    return HM_OK;
}

int hm_upload(hm_engine_t* e) {
    if (!e) return HM_EINVAL;
    try {
        HIP_TRY(hipSetDevice(e->device));
        const size_t nr = e->reads.size(), nc = e->chunks.size();
        e->d_raw.reserve(std::max<size_t>(e->slab.size, 16));
        e->d_reads.reserve(std::max<size_t>(nr, 1) * sizeof(ReadDesc));
        e->d_chunks.reserve(std::max<size_t>(nc, 1) * sizeof(Chunk));
        const size_t tb = (size_t)std::max<int64_t>(e->total_bases, 4);
        e->d_bases.reserve(tb);
        e->d_kin.reserve(tb * sizeof(uint32_t));
        e->d_counts.reserve(std::max<size_t>(nc, 1) * 4 * sizeof(int32_t));
        e->d_offs.reserve(std::max<size_t>(nc, 1) * 4 * sizeof(int32_t));
        // every forward position carries at most one site over all contexts
        e->d_usites.reserve(tb * sizeof(USite));
        e->d_utag.reserve(tb);
        e->d_csites.reserve(tb * sizeof(Site));
        if (e->slab.size) HIP_TRY(hipMemcpyAsync(e->d_raw.p, e->slab.p, e->slab.size, hipMemcpyHostToDevice, e->stream));
        if (nr) HIP_TRY(hipMemcpyAsync(e->d_reads.p, e->reads.data(), nr * sizeof(ReadDesc), hipMemcpyHostToDevice, e->stream));
        if (nc) HIP_TRY(hipMemcpyAsync(e->d_chunks.p, e->chunks.data(), nc * sizeof(Chunk), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));  // reads/chunks vectors are pageable: keep it simple and safe
        e->uploaded = true;
        e->ran = e->synced = false;
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return HM_OK;
}

int hm_run(hm_engine_t* e) {
    if (!e) return HM_EINVAL;
    if (!e->uploaded) return fail(e, HM_ESTATE, "hm_run: nothing uploaded");
    try {
        HIP_TRY(hipSetDevice(e->device));
        const int nc = (int)e->chunks.size();
        {
            Span sp(e, K_PREP, 0);
            launch_prep(e->stream, e->d_raw.as<uint8_t>(), e->d_reads.as<ReadDesc>(), e->d_chunks.as<Chunk>(), nc,
                        e->ctx_mask, e->d_bases.as<uint8_t>(), e->d_kin.as<uint32_t>(), e->d_counts.as<int32_t>(),
                        e->d_err.as<int32_t>());
            sp.end();
        }
        {
            Span sp(e, K_SCAN, 0);
            launch_scan(e->stream, e->d_counts.as<int32_t>(), nc, e->d_offs.as<int32_t>(), e->d_totals.as<int32_t>());
            sp.end();
        }
        {
            Span sp(e, K_EMIT, 0);
            launch_emit(e->stream, e->d_reads.as<ReadDesc>(), e->d_chunks.as<Chunk>(), nc, e->ctx_mask,
                        e->d_bases.as<uint8_t>(), e->d_offs.as<int32_t>(), e->d_totals.as<int32_t>(),
                        e->d_usites.as<USite>(), e->d_utag.as<uint8_t>(), e->d_csites.as<Site>());
            sp.end();
        }
        HIP_TRY(hipGetLastError());
        // the site counts size the CNN launches: one small D2H + sync per batch
        HIP_TRY(hipMemcpyAsync(e->h_totals, e->d_totals.p, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        memcpy(e->totals, e->h_totals, sizeof e->totals);
        ensure_site_buffers(e, e->totals[3]);
        for (int c = 0; c < 3; ++c) {
            if (!(e->ctx_mask >> c & 1) || e->totals[c] == 0) continue;
            run_cnn(e, c, e->d_csites.as<Site>() + e->totals[4 + c], nullptr, e->totals[c], nullptr, 0);
        }
        HIP_TRY(hipMemcpyAsync(e->h_err, e->d_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        e->ran = true;
        e->synced = false;
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return HM_OK;
}

int hm_sync(hm_engine_t* e) {
    if (!e) return HM_EINVAL;
    try {
        HIP_TRY(hipSetDevice(e->device));
        HIP_TRY(hipStreamSynchronize(e->stream));
        collect_timing(e);
        e->synced = true;
        if (e->ran && *e->h_err) {
            *e->h_err = 0;
            HIP_TRY(hipMemset(e->d_err.p, 0, sizeof(int32_t)));
            return fail(e, HM_EDATA, "illegal BAM base encoded value in a staged read");
        }
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return HM_OK;
}

int64_t hm_num_sites(hm_engine_t* e, int ctx) {
    if (!e || ctx < 0 || ctx > 3) return HM_EINVAL;
    if (!e->ran) return fail(e, HM_ESTATE, "hm_num_sites: hm_run first");
    return e->totals[ctx];
}

int64_t hm_fetch(hm_engine_t* e, hm_call_t* out, int64_t cap) {
    if (!e || (!out && cap > 0)) return HM_EINVAL;
    if (!e->ran) return fail(e, HM_ESTATE, "hm_fetch: hm_run first");
    int rc = hm_sync(e);
    if (rc < 0) return rc;
    const int64_t n = e->totals[3];
    if (n > cap) return fail(e, HM_EINVAL, "hm_fetch: output capacity too small");
    if (n == 0) return 0;
    try {
        std::vector<USite> us((size_t)n);
        std::vector<uint8_t> tag((size_t)n), ml((size_t)n);
        std::vector<float> p((size_t)n);
        HIP_TRY(hipMemcpy(us.data(), e->d_usites.p, (size_t)n * sizeof(USite), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tag.data(), e->d_utag.p, (size_t)n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ml.data(), e->d_ml.p, (size_t)n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(p.data(), e->d_p.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        // unified list is ordered by (read, qoff); per read emit FWD calls then REV calls, each by qoff
        // (the order build_one_mod_bam consumes: mod_main.cpp:217-251)
        int64_t o = 0, i = 0;
        while (i < n) {
            int64_t j = i;
            while (j < n && us[(size_t)j].read_idx == us[(size_t)i].read_idx) ++j;
            const int32_t rid = e->reads[(size_t)us[(size_t)i].read_idx].read_id;
            for (int strand = 0; strand < 2; ++strand)
                for (int64_t k = i; k < j; ++k) {
                    if ((tag[(size_t)k] >> 2) != strand) continue;
                    hm_call_t& c = out[o++];
                    c.read_id = rid;
                    c.qoff = us[(size_t)k].qoff;
                    c.strand = (uint8_t)strand;
                    c.ctx = tag[(size_t)k] & 3;
                    c.scaled_prob = ml[(size_t)k];
                    c.reserved = 0;
                    c.p = p[(size_t)k];
                }
            i = j;
        }
        return o;
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
}

int hm_flush(hm_engine_t* e) {
    int rc = hm_upload(e);
    return rc < 0 ? rc : hm_run(e);
}

int64_t hm_drain(hm_engine_t* e, hm_call_t* out, int64_t cap) {
    const int64_t n = hm_fetch(e, out, cap);
    if (n >= 0) hm_clear(e);
    return n;
}

int64_t hm_scan_sites(hm_engine_t* e, int ctx, int32_t* read_id, int32_t* qoff, uint8_t* strand, int64_t cap) {
    if (!e || ctx < 0 || ctx > 2) return HM_EINVAL;
    if (!e->ran) return fail(e, HM_ESTATE, "hm_scan_sites: hm_run first");
    int rc = hm_sync(e);
    if (rc < 0) return rc;
    const int64_t n = e->totals[ctx];
    if (n > cap) return fail(e, HM_EINVAL, "hm_scan_sites: output capacity too small");
    if (n == 0) return 0;
    try {
        std::vector<Site> s((size_t)n);
        std::vector<uint8_t> tag((size_t)e->totals[3]);
        HIP_TRY(hipMemcpy(s.data(), e->d_csites.as<Site>() + e->totals[4 + ctx], (size_t)n * sizeof(Site), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tag.data(), e->d_utag.p, tag.size(), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) {
            if (read_id) read_id[i] = e->reads[(size_t)s[(size_t)i].read_idx].read_id;
            if (qoff) qoff[i] = s[(size_t)i].qoff;
            if (strand) strand[i] = tag[(size_t)s[(size_t)i].uidx] >> 2;
        }
        return n;
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
}

int hm_windows(hm_engine_t* e, int ctx, int64_t first, int64_t n, float* out_host) {
    if (!e || ctx < 0 || ctx > 2 || first < 0 || n < 0) return HM_EINVAL;
    if (!e->ran) return fail(e, HM_ESTATE, "hm_windows: hm_run first");
    if (first + n > e->totals[ctx]) return fail(e, HM_EINVAL, "hm_windows: site range out of bounds");
    if (n == 0) return HM_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        const size_t bytes = (size_t)n * KMER * FEATS * sizeof(float);
        e->d_win.reserve(bytes);
        Span sp(e, K_WINDOW, n);
        launch_windows(e->stream, e->d_csites.as<Site>() + e->totals[4 + ctx] + first, (int)n, e->d_reads.as<ReadDesc>(),
                       e->d_bases.as<uint8_t>(), e->d_kin.as<uint32_t>(), e->model[ctx].w.bn, e->d_win.as<float>(),
                       e->num_cu * 8);
        sp.end();
        HIP_TRY(hipGetLastError());
        if (out_host) HIP_TRY(hipMemcpyAsync(out_host, e->d_win.p, bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        collect_timing(e);
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return HM_OK;
}

int hm_cnn_logits(hm_engine_t* e, int ctx, const float* windows, int64_t n, float* logits, float* p, uint8_t* ml) {
    if (!e || ctx < 0 || ctx > 2 || n < 0 || (!windows && n)) return HM_EINVAL;
    if (!e->model[ctx].loaded) return fail(e, HM_EINVAL, "hm_cnn_logits: context not enabled");
    if (n == 0) return HM_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        const size_t bytes = (size_t)n * KMER * FEATS * sizeof(float);
        e->d_win.reserve(bytes);
        ensure_site_buffers(e, n);
        HIP_TRY(hipMemcpyAsync(e->d_win.p, windows, bytes, hipMemcpyHostToDevice, e->stream));
        run_cnn(e, ctx, nullptr, e->d_win.as<float>(), n, nullptr, 0);
        if (logits) HIP_TRY(hipMemcpyAsync(logits, e->d_logits.p, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        if (p) HIP_TRY(hipMemcpyAsync(p, e->d_p.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        if (ml) HIP_TRY(hipMemcpyAsync(ml, e->d_ml.p, (size_t)n, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        collect_timing(e);
        e->ran = false;  // result buffers no longer hold the staged batch's calls
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return HM_OK;
}

int64_t hm_debug_layer(hm_engine_t* e, int ctx, const float* window, int layer, float* out, int64_t cap) {
    if (!e || ctx < 0 || ctx > 2 || !window || !out || layer < 1 || layer > 8) return HM_EINVAL;
    if (!e->model[ctx].loaded) return fail(e, HM_EINVAL, "hm_debug_layer: context not enabled");
    static const int chans[9] = {8, 128, 128, 128, 96, 96, 96, 64, 64};
    int len = KMER;
    const int k1 = e->model[ctx].k1;
    for (int i = 1; i <= layer; ++i) len = (len + 2 - (i == 1 ? k1 : 3)) / 2 + 1;
    const int64_t nf = (int64_t)len * chans[layer];
    if (nf > cap) return fail(e, HM_EINVAL, "hm_debug_layer: output capacity too small");
    try {
        HIP_TRY(hipSetDevice(e->device));
        const size_t bytes = (size_t)KMER * FEATS * sizeof(float);
        e->d_win.reserve(bytes);
        e->d_dbg.reserve((size_t)nf * sizeof(float));
        ensure_site_buffers(e, 1);
        HIP_TRY(hipMemcpyAsync(e->d_win.p, window, bytes, hipMemcpyHostToDevice, e->stream));
        run_cnn(e, ctx, nullptr, e->d_win.as<float>(), 1, e->d_dbg.as<float>(), layer);
        // conv4 is the front->tail hand-off and already sits in HBM
        const void* src = layer == 4 ? e->d_act4.p : e->d_dbg.p;
        HIP_TRY(hipMemcpyAsync(out, src, (size_t)nf * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        collect_timing(e);
        e->ran = false;
    } catch (const HipErr& h) {
        return fail_hip(e, h);
    }
    return nf;
}

int hm_convert_model(const char* src_path, const char* dst_hmw_path) {
    if (!src_path || !dst_hmw_path) return fail(nullptr, HM_EINVAL, "hm_convert_model: bad argument");
    const std::string src(src_path);
    HostModel m;
    std::string msg;
    const bool is_hmw = src.size() > 4 && src.compare(src.size() - 4, 4, ".hmw") == 0;
    if (!(is_hmw ? load_hmw(src, m, msg) : load_onnx(src, m, msg))) return fail(nullptr, HM_EMODEL, msg);
    if (!save_hmw(m, dst_hmw_path, msg)) return fail(nullptr, HM_EMODEL, msg);
    return HM_OK;
}

int hm_get_stamps(hm_engine_t* e, uint64_t* out, int cap) {
    if (!e || !out) return HM_EINVAL;
    const int n = (int)std::min<size_t>(e->stamp_sum.size(), (size_t)std::max(cap, 0));
    for (int i = 0; i < n; ++i) out[i] = e->stamp_sum[(size_t)i];
    return n;
}

int hm_get_timing(hm_engine_t* e, hm_timing_t* t) {
    if (!e || !t) return HM_EINVAL;
    *t = e->acc;
    return HM_OK;
}

int hm_reset_timing(hm_engine_t* e) {
    if (!e) return HM_EINVAL;
    memset(&e->acc, 0, sizeof e->acc);
    return HM_OK;
}

}  // extern "C"
