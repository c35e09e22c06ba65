// hm_weights.cpp -- see hm_weights.h
#include "hm_weights.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>

namespace hm {

const int kChannels[9] = {8, 128, 128, 128, 96, 96, 96, 64, 64};

static bool read_file(const std::string& path, std::vector<uint8_t>& buf) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    buf.resize((size_t)n);
    f.read(reinterpret_cast<char*>(buf.data()), n);
    return (bool)f;
}

static bool finish_geometry(HostModel& m, const std::string& path, std::string& err) {
    // the checks of s_load_one_model (mod_main.cpp:40-52) restated for the weights we can run
    if (m.k1 != 11 && m.k1 != 13) {
        err = "Ill-formated model " + path + ": first conv kernel must be 11 or 13";
        return false;
    }
    for (int i = 0; i < 8; ++i) {
        const size_t want = (size_t)kChannels[i + 1] * kChannels[i] * m.kernel[i];
        if (m.kernel[i] != (i == 0 ? m.k1 : 3) || m.conv_w[i].size() != want || m.conv_b[i].size() != (size_t)kChannels[i + 1]) {
            err = "Ill-formated model " + path + ": unexpected conv geometry";
            return false;
        }
    }
    if (m.fc1_w.size() != 256 * 128 || m.fc1_b.size() != 256 || m.fc2_w.size() != 512 || m.fc2_b.size() != 2) {
        err = "Ill-formated model " + path + ": unexpected FC geometry";
        return false;
    }
    return true;
}

bool load_hmw(const std::string& path, HostModel& m, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) {
        err = "cannot read " + path;
        return false;
    }
    size_t pos = 0;
    auto take = [&](void* dst, size_t n) {
        if (pos + n > buf.size()) return false;
        memcpy(dst, buf.data() + pos, n);
        pos += n;
        return true;
    };
    char magic[4];
    int32_t hdr[4], chans[9], kern[8];
    if (!take(magic, 4) || memcmp(magic, "HMW1", 4) || !take(hdr, 16) || !take(chans, 36) || !take(kern, 32) || !take(&m.bn_eps, 4)) {
        err = path + ": not an HMW1 file";
        return false;
    }
    if (hdr[0] != KMER || hdr[1] != FEATS || hdr[3] != 8 || memcmp(chans, kChannels, sizeof chans)) {
        err = "Ill-formated model " + path + ": Kmer/Feature geometry mismatch";
        return false;
    }
    m.k1 = hdr[2];
    bool ok = take(m.bn_gamma, 32) && take(m.bn_beta, 32) && take(m.bn_mean, 32) && take(m.bn_var, 32);
    for (int i = 0; ok && i < 8; ++i) {
        m.kernel[i] = kern[i];
        if (kern[i] < 1 || kern[i] > 64) { ok = false; break; }
        m.conv_w[i].resize((size_t)chans[i + 1] * chans[i] * kern[i]);
        m.conv_b[i].resize((size_t)chans[i + 1]);
        ok = take(m.conv_w[i].data(), m.conv_w[i].size() * 4) && take(m.conv_b[i].data(), m.conv_b[i].size() * 4);
    }
    m.fc1_w.resize(256 * 128);
    m.fc1_b.resize(256);
    m.fc2_w.resize(512);
    m.fc2_b.resize(2);
    ok = ok && take(m.fc1_w.data(), m.fc1_w.size() * 4) && take(m.fc1_b.data(), 1024) && take(m.fc2_w.data(), 2048) && take(m.fc2_b.data(), 8);
    if (!ok || pos != buf.size()) {
        err = path + ": truncated or oversized HMW1 file";
        return false;
    }
    return finish_geometry(m, path, err);
}

// ------------------------------------------------------------------------------------------------
// ONNX (protobuf wire format) reader for the two dialects the reference ships:
// CpG/CHG: initializers + Gemm(transB=1); CHH: Constant nodes + MatMul/Add.
// ------------------------------------------------------------------------------------------------
namespace {

struct Span {
    const uint8_t* p = nullptr;
    size_t n = 0;
};

struct Field {
    int no, wt;
    uint64_t v;
    Span s;
};

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    explicit Reader(Span s) : p(s.p), end(s.p + s.n) {}
    uint64_t varint() {
        uint64_t out = 0;
        int shift = 0;
        while (p < end) {
            const uint8_t b = *p++;
            out |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return out;
            shift += 7;
            if (shift > 63) break;
        }
        ok = false;
        return 0;
    }
    bool next(Field& f) {
        if (p >= end || !ok) return false;
        const uint64_t key = varint();
        f.no = (int)(key >> 3);
        f.wt = (int)(key & 7);
        f.v = 0;
        f.s = Span{};
        if (f.wt == 0) f.v = varint();
        else if (f.wt == 1) { f.s = Span{p, 8}; p += 8; }
        else if (f.wt == 5) { f.s = Span{p, 4}; p += 4; }
        else if (f.wt == 2) {
            const uint64_t n = varint();
            if (n > (uint64_t)(end - p)) { ok = false; return false; }
            f.s = Span{p, (size_t)n};
            p += n;
        } else ok = false;
        if (p > end) ok = false;
        return ok;
    }
};

struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool is_f32 = false;
};

void packed_ints(const Field& f, std::vector<int64_t>& out) {
    if (f.wt == 0) { out.push_back((int64_t)f.v); return; }
    Reader r(f.s);
    while (r.p < r.end && r.ok) out.push_back((int64_t)r.varint());
}

std::string str(Span s) { return std::string(reinterpret_cast<const char*>(s.p), s.n); }

Tensor parse_tensor(Span s, std::string& name) {
    Tensor t;
    Reader r(s);
    Field f;
    Span raw;
    std::vector<float> fl;
    int dtype = 0;
    while (r.next(f)) {
        if (f.no == 1) packed_ints(f, t.dims);
        else if (f.no == 2) dtype = (int)f.v;
        else if (f.no == 4) {
            if (f.wt == 2) { fl.resize(f.s.n / 4); memcpy(fl.data(), f.s.p, fl.size() * 4); }
            else if (f.wt == 5) { float v; memcpy(&v, f.s.p, 4); fl.push_back(v); }
        } else if (f.no == 8) name = str(f.s);
        else if (f.no == 9) raw = f.s;
    }
    t.is_f32 = dtype == 1;
    if (t.is_f32) {
        if (raw.p) { t.data.resize(raw.n / 4); memcpy(t.data.data(), raw.p, t.data.size() * 4); }
        else t.data = fl;
    }
    return t;
}

struct Node {
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, std::vector<int64_t>> ints;
    std::map<std::string, float> floats;
    Tensor tensor;
    bool has_tensor = false;
};

Node parse_node(Span s) {
    Node nd;
    Reader r(s);
    Field f;
    while (r.next(f)) {
        if (f.no == 1) nd.in.push_back(str(f.s));
        else if (f.no == 2) nd.out.push_back(str(f.s));
        else if (f.no == 4) nd.op = str(f.s);
        else if (f.no == 5) {
            Reader a(f.s);
            Field g;
            std::string an;
            while (a.next(g)) {
                if (g.no == 1) an = str(g.s);
                else if (g.no == 2) { float v; memcpy(&v, g.s.p, 4); nd.floats[an] = v; }
                else if (g.no == 3) nd.ints[an].push_back((int64_t)g.v);
                else if (g.no == 5) { std::string tn; nd.tensor = parse_tensor(g.s, tn); nd.has_tensor = true; }
                else if (g.no == 8) packed_ints(g, nd.ints[an]);
            }
        }
    }
    return nd;
}

bool ints_are(const Node& n, const char* key, std::initializer_list<int64_t> want, bool default_ok) {
    auto it = n.ints.find(key);
    if (it == n.ints.end()) return default_ok;
    return it->second == std::vector<int64_t>(want);
}

}  // namespace

bool load_onnx(const std::string& path, HostModel& m, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) {
        err = "cannot read " + path;
        return false;
    }
    Span graph;
    {
        Reader r(Span{buf.data(), buf.size()});
        Field f;
        while (r.next(f))
            if (f.no == 7 && f.wt == 2) graph = f.s;
        if (!r.ok || !graph.p) {
            err = "Ill-formated model " + path + ": no graph";
            return false;
        }
    }
    std::map<std::string, Tensor> inits;
    std::vector<Node> nodes;
    {
        Reader r(graph);
        Field f;
        while (r.next(f)) {
            if (f.no == 1 && f.wt == 2) nodes.push_back(parse_node(f.s));
            else if (f.no == 5 && f.wt == 2) {
                std::string name;
                Tensor t = parse_tensor(f.s, name);
                if (t.is_f32) inits[name] = std::move(t);
            }
        }
        if (!r.ok) {
            err = "Ill-formated model " + path + ": truncated graph";
            return false;
        }
    }
    for (auto& nd : nodes)
        if (nd.op == "Constant" && nd.has_tensor && nd.tensor.is_f32 && !nd.out.empty()) inits[nd.out[0]] = nd.tensor;
    auto get = [&](const std::string& name) -> const Tensor* {
        auto it = inits.find(name);
        return it == inits.end() ? nullptr : &it->second;
    };

    std::vector<const Node*> bns, convs, gemms, mms, adds;
    for (auto& nd : nodes) {
        if (nd.op == "BatchNormalization") bns.push_back(&nd);
        else if (nd.op == "Conv") convs.push_back(&nd);
        else if (nd.op == "Gemm") gemms.push_back(&nd);
        else if (nd.op == "MatMul") mms.push_back(&nd);
        else if (nd.op == "Add") adds.push_back(&nd);
    }
    if (bns.size() != 1 || convs.size() != 8 || bns[0]->in.size() < 5) {
        err = "Ill-formated model " + path + ": expected 1 BatchNormalization and 8 Conv nodes";
        return false;
    }
    const Tensor* bt[4];
    for (int i = 0; i < 4; ++i) {
        bt[i] = get(bns[0]->in[i + 1]);
        if (!bt[i] || bt[i]->data.size() != 8) {
            err = "Ill-formated model " + path + ": bn0 must have 8 channels (Batch, Kmer, Features=8)";
            return false;
        }
    }
    memcpy(m.bn_gamma, bt[0]->data.data(), 32);
    memcpy(m.bn_beta, bt[1]->data.data(), 32);
    memcpy(m.bn_mean, bt[2]->data.data(), 32);
    memcpy(m.bn_var, bt[3]->data.data(), 32);
    auto eps = bns[0]->floats.find("epsilon");
    m.bn_eps = eps == bns[0]->floats.end() ? 1e-5f : eps->second;
    for (int i = 0; i < 8; ++i) {
        const Node& c = *convs[i];
        const Tensor* w = c.in.size() > 1 ? get(c.in[1]) : nullptr;
        if (!w || w->dims.size() != 3 || w->dims[0] != kChannels[i + 1] || w->dims[1] != kChannels[i]) {
            err = "Ill-formated model " + path + ": conv weight shape";
            return false;
        }
        if (!ints_are(c, "strides", {2}, false) || !ints_are(c, "pads", {1, 1}, false) ||
            !ints_are(c, "dilations", {1}, true) || !ints_are(c, "group", {1}, true)) {
            err = "Ill-formated model " + path + ": conv must be stride 2, pad 1, no dilation/groups";
            return false;
        }
        m.kernel[i] = (int)w->dims[2];
        m.conv_w[i] = w->data;
        const Tensor* b = c.in.size() > 2 ? get(c.in[2]) : nullptr;
        m.conv_b[i] = b ? b->data : std::vector<float>((size_t)kChannels[i + 1], 0.f);
    }
    m.k1 = m.kernel[0];
    std::vector<float> fw[2], fb[2];
    if (gemms.size() == 2) {
        for (int i = 0; i < 2; ++i) {
            const Node& g = *gemms[i];
            const Tensor* w = g.in.size() > 1 ? get(g.in[1]) : nullptr;
            const Tensor* b = g.in.size() > 2 ? get(g.in[2]) : nullptr;
            if (!w || !b || w->dims.size() != 2 || !ints_are(g, "transB", {1}, false)) {
                err = "Ill-formated model " + path + ": Gemm(transB=1) with bias expected";
                return false;
            }
            fw[i] = w->data;  // [out][in]
            fb[i] = b->data;
        }
    } else if (mms.size() == 2) {
        for (int i = 0; i < 2; ++i) {
            const Node& mm = *mms[i];
            const Tensor* w = mm.in.size() > 1 ? get(mm.in[1]) : nullptr;  // [in][out]
            const Tensor* b = nullptr;
            for (auto* a : adds)
                for (size_t k = 0; k < a->in.size(); ++k)
                    if (a->in[k] == mm.out[0] && a->in.size() == 2) b = get(a->in[1 - k]);
            if (!w || !b || w->dims.size() != 2) {
                err = "Ill-formated model " + path + ": MatMul + Add expected";
                return false;
            }
            const int64_t in = w->dims[0], out = w->dims[1];
            fw[i].resize((size_t)(in * out));
            for (int64_t r = 0; r < in; ++r)
                for (int64_t c = 0; c < out; ++c) fw[i][(size_t)(c * in + r)] = w->data[(size_t)(r * out + c)];
            fb[i] = b->data;
        }
    } else {
        err = "Ill-formated model " + path + ": two fully-connected layers expected";
        return false;
    }
    m.fc1_w = fw[0];
    m.fc1_b = fb[0];
    m.fc2_w = fw[1];
    m.fc2_b = fb[1];
    return finish_geometry(m, path, err);
}

bool save_hmw(const HostModel& m, const std::string& path, std::string& err) {
    std::ofstream f(path, std::ios::binary);
    if (!f) {
        err = "cannot write " + path;
        return false;
    }
    auto put = [&](const void* p, size_t n) { f.write(reinterpret_cast<const char*>(p), (std::streamsize)n); };
    const int32_t hdr[4] = {KMER, FEATS, m.k1, 8};
    int32_t kern[8];
    for (int i = 0; i < 8; ++i) kern[i] = m.kernel[i];
    put("HMW1", 4);
    put(hdr, sizeof hdr);
    put(kChannels, 9 * sizeof(int32_t));
    put(kern, sizeof kern);
    put(&m.bn_eps, 4);
    put(m.bn_gamma, 32);
    put(m.bn_beta, 32);
    put(m.bn_mean, 32);
    put(m.bn_var, 32);
    for (int i = 0; i < 8; ++i) {
        put(m.conv_w[i].data(), m.conv_w[i].size() * 4);
        put(m.conv_b[i].data(), m.conv_b[i].size() * 4);
    }
    put(m.fc1_w.data(), m.fc1_w.size() * 4);
    put(m.fc1_b.data(), m.fc1_b.size() * 4);
    put(m.fc2_w.data(), m.fc2_w.size() * 4);
    put(m.fc2_b.data(), m.fc2_b.size() * 4);
    if (!f) {
        err = "write error on " + path;
        return false;
    }
    return true;
}

bool load_model_dir(const char* dir, const char* name, HostModel& out, std::string& err) {
    const std::string base = std::string(dir) + "/" + name;
    if (std::ifstream(base + ".hmw", std::ios::binary)) return load_hmw(base + ".hmw", out, err);
    if (std::ifstream(base + ".onnx", std::ios::binary)) return load_onnx(base + ".onnx", out, err);
    err = "model not found: " + base + ".{hmw,onnx}";
    return false;
}

// ------------------------------------------------------------------------------------------------
// fragment packing for v_mfma_f32_16x16x4_f32 (see the layout note above Conv in hm_kernels.hip)
// ------------------------------------------------------------------------------------------------
template <class WK>
static void pack_layer(std::vector<float>& blob, int K_real, int K_pad, int COUT, WK wk) {
    const int NT = COUT / 16, KG = K_pad / 16;
    for (int nt = 0; nt < NT; ++nt)
        for (int kg = 0; kg < KG; ++kg)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const int kk = kg * 16 + (lane >> 4) * 4 + s;  // k permutation: lane group l>>4 owns 4 consecutive K
                    const int co = nt * 16 + (lane & 15);
                    blob.push_back(kk < K_real ? wk(kk, co) : 0.f);
                }
}

static void align_blob(std::vector<float>& blob) {
    while (blob.size() % 64) blob.push_back(0.f);  // 256-byte alignment of every section
}

PackedModel pack_model(const HostModel& m) {
    PackedModel pk;
    std::vector<float>& b = pk.blob;
    for (int i = 0; i < 8; ++i) {
        const int cin = kChannels[i], cout = kChannels[i + 1], k = m.kernel[i];
        const int K_real = k * cin;
        const int K_pad = (K_real + 15) / 16 * 16;
        align_blob(b);
        pk.wfrag_off[i] = b.size();
        const std::vector<float>& w = m.conv_w[i];
        pack_layer(b, K_real, K_pad, cout, [&](int kk, int co) {
            const int tap = kk / cin, c = kk % cin;  // kk = tap*CIN + c
            return w[((size_t)co * cin + c) * k + tap];
        });
        align_blob(b);
        pk.bias_off[i] = b.size();
        b.insert(b.end(), m.conv_b[i].begin(), m.conv_b[i].end());
    }
    // fc1: k order (l, c) over conv8's [l][c] output; flatten index of the model is c*2 + l
    align_blob(b);
    pk.wfrag_off[8] = b.size();
    pack_layer(b, 128, 128, 256, [&](int kk, int co) {
        const int l = kk / 64, c = kk % 64;
        return m.fc1_w[(size_t)co * 128 + c * 2 + l];
    });
    align_blob(b);
    pk.bias_off[8] = b.size();
    b.insert(b.end(), m.fc1_b.begin(), m.fc1_b.end());
    align_blob(b);
    pk.fc2_w_off = b.size();
    b.insert(b.end(), m.fc2_w.begin(), m.fc2_w.end());
    align_blob(b);
    pk.fc2_b_off = b.size();
    b.insert(b.end(), m.fc2_b.begin(), m.fc2_b.end());

    // bn0 tables: ONNX BatchNormalization y = (x - mean) / sqrt(var + eps) * gamma + beta in fp32
    BnTables bn;
    int dec[256];
    {
        int p = 0;  // codev1 table, bam_info.cpp:562-570
        for (int i = 0; i < 64; ++i) dec[p++] = i;
        for (int i = 64; i < 128; ++i) dec[p++] = (i - 64) * 2 + 64;
        for (int i = 128; i < 192; ++i) dec[p++] = (i - 128) * 4 + 192;
        for (int i = 192; i < 256; ++i) dec[p++] = (i - 192) * 8 + 448;
    }
    auto bnf = [&](int c, float x) {
        volatile float t = (x - m.bn_mean[c]) / sqrtf(m.bn_var[c] + m.bn_eps);
        volatile float u = t * m.bn_gamma[c];
        return (float)(u + m.bn_beta[c]);
    };
    for (int c = 0; c < 8; ++c) {
        bn.zero[c] = bnf(c, 0.f);
        bn.mean[c] = m.bn_mean[c];
        bn.gamma[c] = m.bn_gamma[c];
        bn.beta[c] = m.bn_beta[c];
        bn.sd[c] = sqrtf(m.bn_var[c] + m.bn_eps);
        if (c < 4) bn.hot[c] = bnf(c, 1.f);
    }
    for (int t = 0; t < 256; ++t) {
        float v = (float)dec[t];
        v /= 952.0f;  // eval_kmer_features.cpp:46-60: fp32 divide by MAX_KINETIC_VALUE
        bn.raw_lut[t] = v;
        for (int c = 0; c < 4; ++c) bn.lut[c][t] = bnf(4 + c, v);
    }
    align_blob(b);
    pk.bn_off = b.size();
    const size_t nf = (sizeof(BnTables) + 3) / 4;
    b.resize(b.size() + nf);
    memcpy(b.data() + pk.bn_off, &bn, sizeof bn);

    // ---- split-half (f16x3) data: x = hi + lo with hi = fp16(x), lo = fp16(x - hi) ----------------------
    auto split = [](float x) -> uint32_t {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        uint16_t hb, lb;
        memcpy(&hb, &h, 2);
        memcpy(&lb, &l, 2);
        if (const char* mb = getenv("HM_XP_WLO_MASK_BITS")) {   // experiment (see hm_convh.h HM_XP_LO_BITS): round to nearest at 10 - n mantissa bits
            const int n = atoi(mb);
            if (n > 0) lb = (uint16_t)((lb + (1u << (n - 1))) & (0xffffu << n));
        }
        return (uint32_t)hb | ((uint32_t)lb << 16);
    };
    for (int i = 0; i < 9; ++i) {  // conv1..conv8, fc1: [n-tile][k-block of 32][plane][lane][8 halves]
        const bool fc = i == 8;
        const int cin = fc ? 64 : kChannels[i], cout = fc ? 256 : kChannels[i + 1], k = fc ? 2 : m.kernel[i];
        const int K_real = k * cin, KB = (K_real + 31) / 32, NT = cout / 16;
        align_blob(b);
        pk.wfrag_h_off[i] = b.size();
        std::vector<uint16_t> hw((size_t)NT * KB * 2 * 64 * 8);
        const std::vector<float>& w = fc ? m.fc1_w : m.conv_w[i];
        size_t o = 0;
        for (int nt = 0; nt < NT; ++nt)
            for (int kb = 0; kb < KB; ++kb)
                for (int plane = 0; plane < 2; ++plane)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int kk = kb * 32 + 8 * (lane >> 4) + j;  // kk = tap*CIN + c
                            const int co = nt * 16 + (lane & 15);
                            uint32_t hl = 0;
                            // conv: W[co][c][tap]; fc1: fc1_w[co][c*2 + l] with kk = l*64 + c (same index formula)
                            if (kk < K_real) hl = split(w[((size_t)co * cin + kk % cin) * k + kk / cin]);
                            hw[o++] = (uint16_t)(plane ? hl >> 16 : hl & 0xffffu);
                        }
        const size_t nfl = (hw.size() * 2 + 3) / 4;
        b.resize(b.size() + nfl);
        memcpy(b.data() + pk.wfrag_h_off[i], hw.data(), hw.size() * 2);
    }
    BnTablesH bh;
    for (int c = 0; c < 8; ++c) bh.zero[c] = split(bn.zero[c]);
    for (int c = 0; c < 4; ++c) {
        bh.hot[c] = split(bn.hot[c]);
        for (int t = 0; t < 256; ++t) bh.lut[c][t] = split(bn.lut[c][t]);
        const double sd = sqrt((double)m.bn_var[4 + c] + m.bn_eps), g = m.bn_gamma[4 + c];
        bh.ka[c] = (float)(g / (sd * 952.0));
        bh.kb[c] = (float)(m.bn_beta[4 + c] - m.bn_mean[4 + c] * g / sd);
    }
    align_blob(b);
    pk.bn_h_off = b.size();
    b.resize(b.size() + (sizeof(BnTablesH) + 3) / 4);
    memcpy(b.data() + pk.bn_h_off, &bh, sizeof bh);

    // ---- conv1 with bn0 folded into the weights ------------------------------------------------------------------
    // bn0 is affine per channel.  One-hot channel c: value = zero[c] + (hot[c] - zero[c]) * onehot, onehot in {0, 1}.
    // Kinetics channel c: value = kb[c] + ka[c] * frames with frames = decoded codev1 byte, an INTEGER in 0..952.  Both
    // operands are EXACT in fp16 (no lo plane: two MFMAs per product instead of three); the weights absorb the slopes,
    // the constants go into the bias.  The conv's own zero padding (window rows -1 and 401, 0 AFTER bn0) must not
    // receive the constant: the two output rows that reach it get their share taken back (c1f_corr).
    // The operand being exact, the hi and lo halves of the weights are STACKED along K as 2*k1 tap slots over the same
    // window rows (slot t < k1: w_hi of tap t; slot k1 + t: w_lo of tap t): ceil(2*k1 / 4) k-blocks with one product each.
    // Fragment layout: [n-tile][k-block of 4 slots][lane][8 = channels of slot 4*kb + (lane >> 4)].
    {
        const int k1 = m.kernel[0], L1 = (401 + 2 - k1) / 2 + 1, KB = (2 * k1 + 3) / 4;
        const std::vector<float>& w = m.conv_w[0];  // [128][8][k1]
        auto W = [&](int co, int c, int t) { return w[((size_t)co * 8 + c) * k1 + t]; };
        // the frame counts reach the MFMA as frames / 32 (still exact in fp16: a power-of-two scale) and the weights take the
        // x32: ka ~ 2^-5, and weights that small would leave their fp16 lo halves in the subnormal range (4 bits instead of 11)
        auto slope = [&](int c) { return c < 4 ? bn.hot[c] - bn.zero[c] : bh.ka[c - 4] * 32.0f; };
        auto konst = [&](int c) { return c < 4 ? (double)bn.zero[c] : (double)bh.kb[c - 4]; };
        std::vector<uint16_t> hw((size_t)8 * KB * 64 * 8);
        size_t o = 0;
        for (int nt = 0; nt < 8; ++nt)
            for (int kb = 0; kb < KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int slot = 4 * kb + (lane >> 4), plane = slot >= k1, tap = slot - plane * k1;
                        const int co = nt * 16 + (lane & 15);
                        uint32_t hl = 0;
                        if (slot < 2 * k1) hl = split(W(co, j, tap) * slope(j));
                        hw[o++] = (uint16_t)(plane ? hl >> 16 : hl & 0xffffu);
                    }
        align_blob(b);
        pk.c1f_off = b.size();
        b.resize(b.size() + (hw.size() * 2 + 3) / 4);
        memcpy(b.data() + pk.c1f_off, hw.data(), hw.size() * 2);
        align_blob(b);
        pk.c1f_bias_off = b.size();
        for (int co = 0; co < 128; ++co) {
            double s = m.conv_b[0][co];
            for (int t = 0; t < k1; ++t)
                for (int c = 0; c < 8; ++c) s += (double)W(co, c, t) * konst(c);
            b.push_back((float)s);
        }
        align_blob(b);
        pk.c1f_corr_off = b.size();
        for (int row = 0; row < 2; ++row)
            for (int co = 0; co < 128; ++co) {
                double s = 0;
                for (int t = 0; t < k1; ++t) {
                    const int wrow = 2 * (row ? L1 - 1 : 0) + t - 1;  // window row reached by tap t
                    if (wrow >= 0 && wrow < 401) continue;
                    for (int c = 0; c < 8; ++c) s += (double)W(co, c, t) * konst(c);
                }
                b.push_back((float)s);
            }
    }
    return pk;
}

}  // namespace hm
