// hifimeth_call.cpp -- `hifimeth-hip call`: BAM in -> GPU 5mC calling -> BAM out with MM/ML/MN tags.
// Keeps the reference's `hifimeth call` surface (src/app/hifimeth/mod_options.cpp:61-181, mod_main.cpp:303-412):
//     hifimeth-hip call [-m dir] [-l 1000] [-s 32] [-b 10000] [-k] [-c cpg,chg,chh] [-t N] [-d 0,1,..] BAM MOD-BAM
// Reads keep their input order; reads shorter than -l or without complete kinetics are passed through with the
// kinetics / old MM / ML tags stripped, exactly as the reference does.
// Two extra sub-commands need no GPU and exist for the CPU test-suite:
//     hifimeth-hip bamcopy IN.bam OUT.bam             (BGZF/BAM round trip)
//     hifimeth-hip tagtest IN.bam CALLS.bin OUT.bam   (apply hm_call_t records, read_id = record index)
//     hifimeth-hip pileup [OPTIONS] REF.fa MOD.bam PREFIX   (hifimeth_pileup.cpp)
//     hifimeth-hip corr [-c N] A.cov.bed B.cov.bed          (Pearson r of two pileup outputs, hifimeth_pileup.cpp)
//     hifimeth-hip modstats IN.bam                    (MM/ML parser + per-context histograms + adaptive thresholds)
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/hifimeth_hip.h"
#include "hm_bam.h"

using namespace hmbam;

namespace {

const char* kName = "hifimeth-hip";
const char* kVersion = "0.1.0";

struct Options {
    std::string model_dir;
    int min_read_size = 1000;  // mod_options.cpp:10-17
    int sample_batch = 32;     // accepted for compatibility; the GPU batches whole read slabs
    int read_batch = 10000;
    bool keep_kinetics = false;
    int ctx_mask = 7;
    int threads = 0;
    int level = 6;
    int precision = 1;
    std::vector<int> devices{0};
    std::string in, out;
};

void usage() {
    fprintf(stderr,
            "USAGE:\n  %s call [OPTIONS] BAM MOD-BAM\n\nOPTIONS:\n"
            "  -m <dir>     model directory holding {CpG,CHG,CHH}.onnx or .hmw (default: <exe_dir>/../weights)\n"
            "  -l <int>     minimum read length to call (default 1000)\n"
            "  -s <int>     sample batch size (accepted for compatibility, unused)\n"
            "  -b <int>     reads per batch (default 10000)\n"
            "  -k           keep the kinetics tags fi/ri/fp/rp in the output\n"
            "  -c <list>    contexts to call: cpg,chg,chh (default all)\n"
            "  -t <int>     host threads for BGZF inflate/deflate (default: all, max 16)\n"
            "  -d <list>    GPU ordinals, e.g. 0,1,2,3 (default 0)\n"
            "  -p <0|1|2>   arithmetic: 1 = split-half fp16x3 MFMA + fp32 accumulate (default), 0 = fp32 MFMA,\n"
            "               2 = fp16 weights for conv2..conv8 (|dp| <= 1e-3 mode)\n"
            "  -z <0-9>     output compression level (default 6)\n",
            kName);
}

bool parse_ctx(const char* arg, int& mask) {
    mask = 0;
    std::string s(arg), tok;
    for (size_t i = 0; i <= s.size(); ++i) {
        if (i == s.size() || s[i] == ',') {
            for (auto& c : tok) c = (char)toupper(c);
            if (tok == "CPG") mask |= 1;
            else if (tok == "CHG") mask |= 2;
            else if (tok == "CHH") mask |= 4;
            else return false;
            tok.clear();
        } else tok += s[i];
    }
    return mask != 0;
}

std::string exe_dir() {
    char buf[4096];
    ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string p(buf);
    return p.substr(0, p.find_last_of('/'));
}

bool parse(int argc, char** argv, Options& o) {
    int i = 2;
    for (; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int& dst) {
            if (i + 1 >= argc) return false;
            dst = atoi(argv[++i]);
            return true;
        };
        if (a == "-m") {
            if (i + 1 >= argc) return false;
            o.model_dir = argv[++i];
        } else if (a == "-l") { if (!need(o.min_read_size)) return false; }
        else if (a == "-s") { if (!need(o.sample_batch)) return false; }
        else if (a == "-b") { if (!need(o.read_batch)) return false; }
        else if (a == "-t") { if (!need(o.threads)) return false; }
        else if (a == "-z") { if (!need(o.level)) return false; }
        else if (a == "-p") { if (!need(o.precision)) return false; }
        else if (a == "-k") o.keep_kinetics = true;
        else if (a == "-c") {
            if (i + 1 >= argc || !parse_ctx(argv[++i], o.ctx_mask)) {
                fprintf(stderr, "Illegal argument to option '-c'\n");
                return false;
            }
        } else if (a == "-d") {
            if (i + 1 >= argc) return false;
            o.devices.clear();
            std::string s(argv[++i]), tok;
            for (size_t k = 0; k <= s.size(); ++k) {
                if (k == s.size() || s[k] == ',') {
                    if (!tok.empty()) o.devices.push_back(atoi(tok.c_str()));
                    tok.clear();
                } else tok += s[k];
            }
            if (o.devices.empty()) return false;
        } else if (a == "-h" || a == "-v") return false;
        else if (a[0] == '-' && a.size() > 1) {
            fprintf(stderr, "Unrecognised option '%s'\n", a.c_str());
            return false;
        } else break;
    }
    if (argc - i != 2) return false;
    o.in = argv[i];
    o.out = argv[i + 1];
    if (o.model_dir.empty()) o.model_dir = exe_dir() + "/../weights";
    if (o.threads <= 0) o.threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (o.read_batch < 1 || o.min_read_size < 0 || o.level < 0 || o.level > 9 || o.precision < 0 || o.precision > 2) return false;
    return true;
}

void add_pg(BamHeader& h, int argc, char** argv) {  // mod_main.cpp:101-117
    std::string line = std::string("@PG\tID:") + kName + "\tPN:" + kName + "\tVN:" + kVersion + "\tCL:" + argv[0];
    for (int i = 1; i < argc; ++i) line += std::string(" ") + argv[i];
    if (!h.text.empty() && h.text.back() != '\n') h.text += '\n';
    h.text += line + "\n";
}

struct Slot {
    hm_engine_t* eng = nullptr;
    std::vector<BamRecord> recs;
    bool active = false;
};

int cmd_call(int argc, char** argv) {
    Options o;
    if (!parse(argc, argv, o)) {
        usage();
        return EXIT_FAILURE;
    }
    const auto t0 = std::chrono::steady_clock::now();
    BgzfReader in(o.in, o.threads);
    if (!in.ok()) { fprintf(stderr, "[%s] %s\n", kName, in.error().c_str()); return EXIT_FAILURE; }
    BamHeader hdr;
    std::string err;
    if (!read_header(in, hdr, err)) { fprintf(stderr, "[%s] %s: %s\n", kName, o.in.c_str(), err.c_str()); return EXIT_FAILURE; }
    BgzfWriter out(o.out, o.threads, o.level);
    if (!out.ok()) { fprintf(stderr, "[%s] %s\n", kName, out.error().c_str()); return EXIT_FAILURE; }
    add_pg(hdr, argc, argv);
    write_header(out, hdr);

    // three engines per device: one batch being staged, one on the GPU, one being tagged / written
    std::vector<Slot> slots(o.devices.size() * 3);
    for (size_t s = 0; s < slots.size(); ++s) {
        const int dev = o.devices[s % o.devices.size()];
        if (hm_create(&slots[s].eng, o.model_dir.c_str(), o.ctx_mask, dev) < 0) {
            fprintf(stderr, "[%s] device %d: %s\n", kName, dev, hm_last_error(nullptr));
            return EXIT_FAILURE;
        }
        hm_set_option(slots[s].eng, "min_read_size", o.min_read_size);
        hm_set_option(slots[s].eng, "precision", o.precision);
    }
    size_t all_reads = 0, all_bases = 0, all_ctx[3] = {0, 0, 0};
    std::vector<hm_call_t> calls;
    std::atomic<bool> failed{false};

    auto finish = [&](Slot& sl) {
        hm_engine_t* e = sl.eng;
        if (hm_sync(e) < 0) { fprintf(stderr, "[%s] %s\n", kName, hm_last_error(e)); failed = true; return; }
        const int64_t n = hm_num_sites(e, HM_CTX_ALL);
        for (int c = 0; c < 3; ++c) all_ctx[c] += (size_t)std::max<int64_t>(0, hm_num_sites(e, c));
        calls.resize((size_t)std::max<int64_t>(n, 0));
        const int64_t got = hm_drain(e, calls.data(), (int64_t)calls.size());
        if (got < 0) { fprintf(stderr, "[%s] %s\n", kName, hm_last_error(e)); failed = true; return; }
        // calls are grouped by read in submission order: find every read's range, then build the tags of all reads
        // in parallel (the MM deltas walk every base of the read), then write in order
        std::vector<size_t> first(sl.recs.size() + 1, 0);
        size_t ci = 0;
        for (size_t i = 0; i < sl.recs.size(); ++i) {
            first[i] = ci;
            while (ci < (size_t)got && calls[ci].read_id == (int32_t)i) ++ci;
        }
        first[sl.recs.size()] = ci;
        std::vector<std::string> errs(sl.recs.size());
        parallel_run((int)sl.recs.size(), o.threads, [&](int i) {
            apply_calls(sl.recs[(size_t)i], calls.data() + first[(size_t)i], first[(size_t)i + 1] - first[(size_t)i],
                        o.keep_kinetics, errs[(size_t)i]);
        });
        for (size_t i = 0; i < sl.recs.size(); ++i) {
            if (!errs[i].empty()) {
                fprintf(stderr, "[%s] read %zu: %s\n", kName, all_reads + i, errs[i].c_str());
                failed = true;
                return;
            }
            all_bases += (size_t)sl.recs[i].l_qseq();
            write_record(out, sl.recs[i]);
        }
        all_reads += sl.recs.size();
        fprintf(stderr, "[%s] %zu reads done\n", kName, all_reads);
        sl.recs.clear();
    };

    // A producer thread inflates and parses batch k+1 while this thread stages batch k, collects an older batch from
    // its engine, builds its tags and deflates it (the reader is touched by the producer only, the writer by this thread).
    struct Batch {
        std::vector<BamRecord> recs;
        bool eof = false;
        std::string err;
    };
    Batch nb[2];
    auto produce = [&](Batch& bt) {
        bt.recs.clear();
        bt.err.clear();
        while ((int)bt.recs.size() < o.read_batch) {
            BamRecord r;
            if (!read_record(in, r, bt.err)) { bt.eof = true; break; }
            bt.recs.push_back(std::move(r));
        }
    };
    // consumer thread: collects finished batches in submission order (hm_sync + hm_drain), builds the tags, deflates
    std::mutex mu;
    std::condition_variable cv;
    std::deque<size_t> ready;
    bool no_more = false;
    std::thread writer([&]() {
        for (;;) {
            size_t idx;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !ready.empty() || no_more; });
                if (ready.empty()) return;
                idx = ready.front();
                ready.pop_front();
            }
            if (!failed) finish(slots[idx]);
            {
                std::lock_guard<std::mutex> lk(mu);
                slots[idx].active = false;
            }
            cv.notify_all();
        }
    });
    size_t b = 0;
    int cur = 0;
    produce(nb[0]);
    while (!failed) {
        Batch& bt = nb[cur];
        if (!bt.err.empty()) { fprintf(stderr, "[%s] %s: %s\n", kName, o.in.c_str(), bt.err.c_str()); failed = true; break; }
        std::thread producer;
        if (!bt.eof) producer = std::thread(produce, std::ref(nb[cur ^ 1]));
        if (!bt.recs.empty()) {
            Slot& sl = slots[b % slots.size()];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !sl.active; });  // the consumer is done with this engine's previous batch
            }
            if (!failed) {
                sl.recs = std::move(bt.recs);
                bt.recs.clear();
                for (size_t i = 0; i < sl.recs.size(); ++i) {
                    const BamRecord& r = sl.recs[i];
                    const KineticsView kv = kinetics_of(r);
                    const int rc = hm_submit_read(sl.eng, (int32_t)i, r.l_qseq(), r.flag(), r.seq4(), kv.arr[0], kv.width[0],
                                                  kv.arr[1], kv.width[1], kv.arr[2], kv.width[2], kv.arr[3], kv.width[3]);
                    if (rc < 0) { fprintf(stderr, "[%s] %s\n", kName, hm_last_error(sl.eng)); failed = true; break; }
                }
                if (!failed && hm_flush(sl.eng) < 0) { fprintf(stderr, "[%s] %s\n", kName, hm_last_error(sl.eng)); failed = true; }
                if (!failed) {
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        sl.active = true;
                        ready.push_back(b % slots.size());
                    }
                    cv.notify_all();
                    ++b;
                }
            }
        }
        if (producer.joinable()) producer.join();
        if (bt.eof) break;
        cur ^= 1;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        no_more = true;
    }
    cv.notify_all();
    writer.join();
    for (auto& sl : slots) hm_destroy(sl.eng);
    if (failed) return EXIT_FAILURE;
    if (!out.close()) { fprintf(stderr, "[%s] %s: %s\n", kName, o.out.c_str(), out.error().c_str()); return EXIT_FAILURE; }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "******** Final stats:\n  ## Reads: %zu\n  ## Bases: %zu\n", all_reads, all_bases);
    static const char* cn[3] = {"CpG", "CHG", "CHH"};
    for (int c = 0; c < 3; ++c)
        if (all_ctx[c]) fprintf(stderr, "  ## %s samples: %zu\n", cn[c], all_ctx[c]);
    fprintf(stderr, "  ## Wall time: %.2f s (%.0f sites/s end to end)\n", sec, (double)(all_ctx[0] + all_ctx[1] + all_ctx[2]) / sec);
    return 0;
}

int cmd_bamcopy(int argc, char** argv) {
    if (argc != 4) { usage(); return EXIT_FAILURE; }
    BgzfReader in(argv[2], 4);
    BamHeader h;
    std::string err;
    if (!in.ok() || !read_header(in, h, err)) { fprintf(stderr, "%s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    BgzfWriter out(argv[3], 4, 6);
    if (!out.ok()) return EXIT_FAILURE;
    write_header(out, h);
    BamRecord r;
    while (read_record(in, r, err)) write_record(out, r);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
    return out.close() ? 0 : EXIT_FAILURE;
}

int cmd_tagtest(int argc, char** argv) {
    if (argc < 5) { usage(); return EXIT_FAILURE; }
    const bool keep = argc > 5 && std::string(argv[5]) == "-k";
    BgzfReader in(argv[2], 2);
    BamHeader h;
    std::string err;
    if (!in.ok() || !read_header(in, h, err)) { fprintf(stderr, "%s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    std::vector<hm_call_t> calls;
    if (FILE* f = fopen(argv[3], "rb")) {
        hm_call_t c;
        while (fread(&c, sizeof c, 1, f) == 1) calls.push_back(c);
        fclose(f);
    } else return EXIT_FAILURE;
    BgzfWriter out(argv[4], 2, 6);
    if (!out.ok()) return EXIT_FAILURE;
    add_pg(h, argc, argv);
    write_header(out, h);
    BamRecord r;
    size_t ci = 0;
    for (int32_t i = 0; read_record(in, r, err); ++i) {
        size_t cj = ci;
        while (cj < calls.size() && calls[cj].read_id == i) ++cj;
        if (!apply_calls(r, calls.data() + ci, cj - ci, keep, err)) { fprintf(stderr, "read %d: %s\n", i, err.c_str()); return EXIT_FAILURE; }
        ci = cj;
        write_record(out, r);
    }
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
    return out.close() ? 0 : EXIT_FAILURE;
}

// modstats IN.bam : the alignment-free first pass of `hifimeth pileup` (src/app/hifimeth/pileup.cpp:237-272,355-436):
// parse MM/ML, histogram the 5mC probabilities of primary reads per context, resolve the adaptive thresholds.
// Prints one JSON object.
int cmd_modstats(int argc, char** argv) {
    if (argc != 3) { usage(); return EXIT_FAILURE; }
    BgzfReader in(argv[2], 8);
    BamHeader h;
    std::string err;
    if (!in.ok() || !read_header(in, h, err)) { fprintf(stderr, "%s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    static uint64_t bins[3][256];
    memset(bins, 0, sizeof bins);
    BamRecord r;
    std::vector<BaseMod> mods;
    uint64_t reads = 0, with_mods = 0, calls = 0;
    while (read_record(in, r, err)) {
        ++reads;
        if (!parse_mods(r, mods, err)) { fprintf(stderr, "read %llu: %s\n", (unsigned long long)reads - 1, err.c_str()); return EXIT_FAILURE; }
        if (mods.empty()) continue;
        ++with_mods;
        if (r.flag() & 0x900) continue;  // primary records only (pileup.cpp:237)
        for (const BaseMod& m : mods) {
            if (m.unmod_base != 'C' && m.unmod_base != 'G') continue;
            const int c = mod_context(r, m.qoff);
            if (c < 0) continue;
            ++bins[c][m.prob];
            ++calls;
        }
    }
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
    static const char* cn[3] = {"CpG", "CHG", "CHH"};
    printf("{\"reads\": %llu, \"reads_with_mods\": %llu, \"calls\": %llu", (unsigned long long)reads,
           (unsigned long long)with_mods, (unsigned long long)calls);
    for (int c = 0; c < 3; ++c) {
        uint64_t n = 0;
        const int thr = resolve_threshold(bins[c], &n);
        printf(", \"%s\": {\"threshold\": %d, \"samples_in_window\": %llu, \"bins\": [", cn[c], thr, (unsigned long long)n);
        for (int i = 0; i < 256; ++i) printf("%s%llu", i ? "," : "", (unsigned long long)bins[c][i]);
        printf("]}");
    }
    printf("}\n");
    return 0;
}

}  // namespace

int cmd_pileup(int argc, char** argv);   // hifimeth_pileup.cpp
int cmd_fastats(int argc, char** argv);  // hifimeth_pileup.cpp
int cmd_corr(int argc, char** argv);     // hifimeth_pileup.cpp
int cmd_cov2bed(int argc, char** argv);  // hifimeth_pileup.cpp
int cmd_sample(int argc, char** argv);   // hifimeth_pileup.cpp
int cmd_eval(int argc, char** argv);     // hifimeth_pileup.cpp

int main(int argc, char** argv) {
    if (argc < 2) { usage(); return EXIT_FAILURE; }
    const std::string cmd = argv[1];
    if (cmd == "call") return cmd_call(argc, argv);
    if (cmd == "bamcopy") return cmd_bamcopy(argc, argv);
    if (cmd == "tagtest") return cmd_tagtest(argc, argv);
    if (cmd == "modstats") return cmd_modstats(argc, argv);
    if (cmd == "pileup") return cmd_pileup(argc, argv);
    if (cmd == "fastats") return cmd_fastats(argc, argv);
    if (cmd == "corr") return cmd_corr(argc, argv);
    if (cmd == "cov2bed") return cmd_cov2bed(argc, argv);
    if (cmd == "sample") return cmd_sample(argc, argv);
    if (cmd == "eval") return cmd_eval(argc, argv);
    usage();
    return EXIT_FAILURE;
}
