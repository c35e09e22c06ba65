// hifimeth_call.cpp -- `hifimeth-hip call`: BAM in -> GPU 5mC calling -> BAM out with MM/ML/MN tags.
// Keeps the reference's `hifimeth call` surface (src/app/hifimeth/mod_options.cpp:61-181, mod_main.cpp:303-412):
//     hifimeth-hip call [-m dir] [-l 1000] [-s 32] [-b 10000] [-k] [-c cpg,chg,chh] [-t N] [-d 0,1,..] BAM MOD-BAM
// -b keeps the reference's meaning and default (reads per outer batch); it does NOT set the granularity of the GPU pipeline:
// batches are cut into engine slabs of <= 6 Mi bases (-S), so the default flags run the pipeline at full depth.
// Reads keep their input order; reads shorter than -l or without complete kinetics are passed through with the
// kinetics / old MM / ML tags stripped, exactly as the reference does.
// Two extra sub-commands need no GPU and exist for the CPU test-suite:
//     hifimeth-hip bamcopy [-R r/w | -Q queue [-C n]] IN.bam OUT.bam    (BGZF/BAM round trip of this process's parts of the input)
//     hifimeth-hip merge OUT.bam N                    (joins OUT.bam.shard0..N-1 written by N ranks of call / bamcopy -R)
//     hifimeth-hip stagebench [-t N] [-R r/w] IN.bam  (host side of call without a GPU: inflate + parse + stage; JSON rate)
//     hifimeth-hip tagtest IN.bam CALLS.bin OUT.bam   (apply hm_call_t records, read_id = record index)
//     hifimeth-hip pileup [OPTIONS] REF.fa MOD.bam PREFIX   (hifimeth_pileup.cpp)
//     hifimeth-hip corr [-c N] A.cov.bed B.cov.bed          (Pearson r of two pileup outputs, hifimeth_pileup.cpp)
//     hifimeth-hip modstats IN.bam                    (MM/ML parser + per-context histograms + adaptive thresholds)
//     hifimeth-hip modlist IN.bam                     (the MM/ML parser's output per record: test seam against the
//                                                      reference parser's fixture, tests/golden/modparse.json)
//     hifimeth-hip thresholds < HISTOGRAMS            (the threshold resolver alone: tests/golden/pileup_thresholds.json)
#include <fcntl.h>
#include <sched.h>
#include <sys/file.h>
#include <sys/stat.h>
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
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/hifimeth_hip.h"
#include "hm_bam.h"

using namespace hmbam;

namespace {

const char* kName = "hifimeth-hip";
const char* kVersion = "0.1.0";

struct Shard {  // -R r/w: this process handles the r-th of w parts of the input (see open_shard)
    int rank = 0, world = 1;
};

struct Options {
    std::string model_dir;
    int min_read_size = 1000;  // mod_options.cpp:10-17
    int sample_batch = 32;     // accepted for compatibility; the GPU batches whole read slabs
    int read_batch = 10000;
    bool keep_kinetics = false;
    int ctx_mask = 7;
    int threads = 0;
    int level = 6;
    bool ld_out = false;  // -Z: deflate the output with libdeflate (same records, other compressed bytes; ~2.5x less CPU per block)
    int precision = 1;
    // reads are handed to the engine in slabs of at most this many bases, whatever -b says: -b is the reference's outer
    // batch (mod_options.cpp:13), the slab is the granularity of THIS pipeline (decode | stage | GPU | tag + write overlap
    // slab by slab).  6 Mi bases = three trunk groups, ~25 ms of device time; measured end to end on a 1.26 GB file: 2 Mi 2.7 s,
    // 4 - 8 Mi 2.55 s, 12 Mi 2.8 s, 24 Mi 3.1 s (smaller: launch tails; larger: the pipeline fills and drains slab by slab).
    int64_t slab_bases = int64_t(6) << 20;
    int trunk = -1;  // -1: chosen per context from the head of the input file (the same on every rank); 0 / 1: forced
    std::vector<int> devices{0};
    std::string in, out;
    bool help = false;
    Shard shard;
    // -Q <file> [-C <n>]: the input is cut into n parts (by BGZF offset, like -R) that the ranks of a job PULL from a shared
    // counter in <file> -- the work queue of the reference's workers (src/corelib/sam_batch.hpp:38-54: whoever is free takes
    // the next reads) stretched over processes: a rank whose reads are site-rich or long takes fewer parts
    std::string queue;
    int chunks = 0;
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
            "  -t <int>     host threads for BGZF inflate/deflate and tag building (default: all this process is granted)\n"
            "  -d <list>    GPU ordinals, e.g. 0,1,2,3 (default 0)\n"
            "  -p <0|1|2>   arithmetic: 1 = split-half fp16x3 MFMA + fp32 accumulate (default, |dp| <= 1e-4), 0 = fp32 MFMA,\n"
            "               2 = as 1 with plain fp16 weights in conv8 and fc1 (|dp| <= 1e-3)\n"
            "  -R <r/w>     this process is rank r of w: call only the r-th part of BAM (split by BGZF offset) and write\n"
            "               MOD-BAM.shard<r>; `%s merge MOD-BAM w` joins the shards in input order\n"
            "  -Q <file>    pull the parts of BAM from the counter in <file>, shared by all ranks of the job, instead of taking the\n"
            "               fixed part -R names; every part k is written to MOD-BAM.shard<k>; `%s merge MOD-BAM n` joins them\n"
            "  -C <int>     number of parts for -Q (default: one per 256 MB of BAM)\n"
            "  -z <0-9>     output compression level (default 6)\n"
            "  -Z           deflate the output with libdeflate where the system has it (BGZF input is inflated with it anyway): the same\n"
            "               records in fewer CPU seconds, but not the bytes zlib writes at that level\n"
            "  -S <int>     bases per engine slab (pipeline granularity, default 6291456; results do not depend on it)\n"
            "  -T <0|1>     conv1..conv4 once per site (0) / once per read position (1); default: per context, from the site\n"
            "               density of the head of BAM\n",
            kName, kName, kName);
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

// host threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a container that is granted
// 16 of a node's 256 hardware threads must not start 256 workers); the reference's default is the physical core count
// (mod_options.cpp:73,129-131)
int default_threads() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long long period = 0;
        if (fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min<long long>(n, std::max<long long>(1, atoll(quota) / period));
        fclose(f);
    }
    return n;
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
        else if (a == "-Z") o.ld_out = true;
        else if (a == "-p") { if (!need(o.precision)) return false; }
        else if (a == "-T") { if (!need(o.trunk)) return false; }
        else if (a == "-S") {
            if (i + 1 >= argc) return false;
            o.slab_bases = atoll(argv[++i]);
        }
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
        } else if (a == "-R") {
            if (i + 1 >= argc || sscanf(argv[++i], "%d/%d", &o.shard.rank, &o.shard.world) != 2 || o.shard.world < 1 ||
                o.shard.rank < 0 || o.shard.rank >= o.shard.world) {
                fprintf(stderr, "Illegal argument to option '-R' (rank/world expected)\n");
                return false;
            }
        } else if (a == "-Q") {
            if (i + 1 >= argc) return false;
            o.queue = argv[++i];
        } else if (a == "-C") { if (!need(o.chunks)) return false; }
        else if (a == "-h") {
            o.help = true;
            return false;
        } else if (a == "-v") {
            fprintf(stderr, "%s\n", kVersion);
            o.help = true;
            return false;
        }
        else if (a[0] == '-' && a.size() > 1) {
            fprintf(stderr, "Unrecognised option '%s'\n", a.c_str());
            return false;
        } else break;
    }
    if (argc - i != 2) return false;
    o.in = argv[i];
    o.out = argv[i + 1];
    if (o.model_dir.empty()) o.model_dir = exe_dir() + "/../weights";
    if (o.threads <= 0) o.threads = default_threads();
    if (o.read_batch < 1 || o.min_read_size < 0 || o.level < 0 || o.level > 9 || o.precision < 0 || o.precision > 2) return false;
    if (o.slab_bases < 1 || o.trunk < -1 || o.trunk > 1 || o.chunks < 0) return false;
    return true;
}

void add_pg(BamHeader& h, int argc, char** argv) {  // mod_main.cpp:101-117
    std::string line = std::string("@PG\tID:") + kName + "\tPN:" + kName + "\tVN:" + kVersion + "\tCL:" + argv[0];
    for (int i = 1; i < argc; ++i) line += std::string(" ") + argv[i];
    if (!h.text.empty() && h.text.back() != '\n') h.text += '\n';
    h.text += line + "\n";
}

// ---- read sharding over processes ---------------------------------------------------------------------------------------
// `-R r/w`: this process is rank r of w.  The input is split by COMPRESSED bytes at BGZF block boundaries, so a rank
// inflates only its own part of the file (the reference's single reader, src/corelib/sam_batch.hpp:12-54, is what limits
// it to one node's worth of decode).  A record belongs to the rank whose byte range holds the block its first byte is in;
// ranks > 0 find their first record with find_record_start.  Every rank writes <out>.shard<r>; `merge` joins them in rank
// order, which is input order (mod_main.cpp:352-362 writes in input order).
bool parse_shard(const char* arg, Shard& sh) {
    return sscanf(arg, "%d/%d", &sh.rank, &sh.world) == 2 && sh.world >= 1 && sh.rank >= 0 && sh.rank < sh.world;
}

// (a run that is not sharded writes OUT itself; a queue run always writes OUT.shard<k>, also when the queue has a single part)
std::string shard_path(const std::string& out, const Shard& sh, bool queued = false) {
    return sh.world == 1 && !queued ? out : out + ".shard" + std::to_string(sh.rank);
}

// ---- a work queue over processes ------------------------------------------------------------------------------------------
// The counter is a small text file; a claim is read-increment-write under an exclusive flock.  Ranks of one node (the
// reference is a single-node program; so is a one-process-per-GPU job) share the file system, nothing else is needed.
// Returns the claimed part, or -1 with `err` set when the counter cannot be reached (an unwritable directory, a file system
// without flock): that is a failure of the run, not an empty queue.
int claim_chunk(const std::string& path, std::string& err) {
    const int fd = open(path.c_str(), O_RDWR | O_CREAT, 0644);
    if (fd < 0) {
        err = "work queue " + path + ": " + strerror(errno);
        return -1;
    }
    int k = -1;
    if (flock(fd, LOCK_EX) == 0) {
        char buf[32] = {0};
        const ssize_t n = pread(fd, buf, sizeof buf - 1, 0);
        k = n > 0 ? atoi(buf) : 0;
        const int len = snprintf(buf, sizeof buf, "%d\n", k + 1);
        if (n < 0 || pwrite(fd, buf, (size_t)len, 0) != len || ftruncate(fd, len) != 0) {
            err = "work queue " + path + ": " + strerror(errno);
            k = -1;
        }
        flock(fd, LOCK_UN);
    } else {
        err = "work queue " + path + ": flock: " + strerror(errno);
    }
    close(fd);
    return k;
}

int default_chunks(const std::string& bam) {
    struct stat st;
    if (stat(bam.c_str(), &st) != 0) return 1;
    return (int)std::max<int64_t>(1, ((int64_t)st.st_size + (int64_t(256) << 20) - 1) / (int64_t(256) << 20));
}

// The parts of the input this process handles, one after the other: the one part -R names, or whatever it can claim from
// the queue until the counter passes the number of parts.
struct ShardSource {
    Shard fixed;
    std::string queue;
    int chunks = 0;
    bool given = false;
    std::string err;  // set when the queue could not be reached: the caller must fail, not finish
    bool queued() const { return !queue.empty(); }
    bool next(Shard& sh) {
        if (queue.empty()) {
            if (given) return false;
            given = true;
            sh = fixed;
            return true;
        }
        const int k = claim_chunk(queue, err);
        if (k < 0 || k >= chunks) return false;
        sh = Shard{k, chunks};
        return true;
    }
};

// What open_shard learns about a file once per PROCESS: a queue run opens many parts of the same input, and the scan of the BGZF
// block offsets (one seek and two short reads per block) costs as much as reading the file's index would -- per part it would be
// (blocks in the file) x (parts per rank) system calls, more than the GPU work of a large input.
struct ShardIndex {
    bool valid = false;
    std::vector<int64_t> offs;  // compressed offset of every BGZF block
    int64_t fsize = 0, first_rec_block = 0;
    BamHeader hdr;
    int scans = 0;  // how often the file's blocks were scanned (bamcopy reports it: tests/test_dist_gloo.py)
};

// reads the header (every rank needs the reference count), then positions `in` at the shard's first record;
// end_off = compressed offset at which the next rank's records start.  `idx` carries the header and the block offsets from one
// part to the next (filled by the first call of the process).
bool open_shard(BgzfReader& in, const std::string& path, const Shard& sh, BamHeader& hdr, int64_t& end_off, std::string& err,
                ShardIndex* idx = nullptr) {
    ShardIndex local;
    ShardIndex& ix = idx ? *idx : local;
    end_off = INT64_MAX;
    bool at_first = false;  // the reader stands right behind the header
    if (!ix.valid) {
        at_first = true;
        if (in.block_offset() != 0 && !in.seek_block(0)) { err = in.error(); return false; }
        ix.hdr = BamHeader();
        if (!read_header(in, ix.hdr, err)) return false;
        ix.first_rec_block = in.block_offset();
        if (sh.world > 1) {
            if (!scan_bgzf_blocks(path, ix.offs, ix.fsize, err)) return false;
            ++ix.scans;
        }
        ix.valid = true;
        hdr = ix.hdr;
        if (sh.world == 1) return true;
    } else {
        hdr = ix.hdr;
        if (sh.world == 1) {  // (the whole file again)
            if (!in.seek_block(0)) { err = in.error(); return false; }
            BamHeader h2;
            return read_header(in, h2, err);
        }
        if (ix.offs.empty()) {
            if (!scan_bgzf_blocks(path, ix.offs, ix.fsize, err)) return false;
            ++ix.scans;
        }
    }
    const int64_t first_rec_block = ix.first_rec_block;
    auto bound = [&](int k) -> int64_t {
        if (k >= sh.world) return ix.fsize;
        const int64_t target = ix.fsize / sh.world * k;
        auto it = std::lower_bound(ix.offs.begin(), ix.offs.end(), target);
        const int64_t b = it == ix.offs.end() ? ix.fsize : *it;
        return std::max(b, first_rec_block);  // ranges that would start inside the header collapse onto the first record
    };
    const int64_t start = bound(sh.rank);
    end_off = bound(sh.rank + 1);
    if (start >= end_off) {  // empty shard
        end_off = -1;
        return true;
    }
    if (start == first_rec_block) {  // begins right behind the header
        if (at_first) return true;  // (the reader is already there)
        if (!in.seek_block(0)) { err = in.error(); return false; }
        BamHeader h2;
        return read_header(in, h2, err);
    }
    if (!in.seek_block(start)) { err = in.error(); return false; }
    if (!find_record_start(in, (int)hdr.refs.size(), err)) {
        if (!err.empty()) return false;
        end_off = -1;  // no record starts in this range
    }
    return true;
}

// Which contexts take the dense trunk (engine option "trunk_mask"): decided from the first reads of the FILE -- not of this
// rank's shard -- so that every rank of a sharded run, and the single-process run, compute with the same kernels and the merged
// output is byte-identical (the two kernel paths agree only to fp32 re-association).
int head_trunk_mask(const Options& o, std::string& err) {
    BgzfReader in(o.in, std::min(o.threads, 4));
    BamHeader hdr;
    if (!in.ok() || !read_header(in, hdr, err)) {
        if (err.empty()) err = in.error();
        return -1;
    }
    std::vector<BamRecord> recs;
    std::vector<hm_read_t> reads;
    int64_t bases = 0;
    while (bases < (int64_t(4) << 20)) {
        BamRecord r;
        if (!read_record(in, r, err)) break;
        bases += r.l_qseq();
        recs.push_back(std::move(r));
    }
    if (!err.empty()) return -1;
    // only the reads the engine will call: it counts densities over accepted reads, and so must this estimate (reads below -l and
    // reads without the four kinetics arrays are passed through uncalled)
    for (const BamRecord& r : recs) {
        if (r.l_qseq() < o.min_read_size) continue;
        const KineticsView kv = kinetics_of(r);
        if (!kv.arr[0] || !kv.arr[1] || !kv.arr[2] || !kv.arr[3]) continue;
        hm_read_t d{};
        d.l_qseq = r.l_qseq();
        d.seq4 = r.seq4();
        reads.push_back(d);
    }
    return hm_trunk_mask_for_reads(reads.data(), (int64_t)reads.size(), o.ctx_mask);
}

struct Job {
    hm_batch_t* batch = nullptr;
    std::vector<BamRecord> recs;
    size_t dev = 0;
    // the part of the input this job belongs to; a job with open_part set (and no batch) starts the part's output file
    Shard part;
    bool open_part = false;
    BamHeader hdr;  // open_part of part 0 only: the header to write
};

int cmd_call(int argc, char** argv) {
    Options o;
    if (!parse(argc, argv, o)) {
        usage();
        return o.help ? 0 : EXIT_FAILURE;  // -h / -v exit 0 like the reference (mod_options.cpp:62-71)
    }
    const auto t0 = std::chrono::steady_clock::now();
    // HM_CLI_TIMING=1: where the command's fixed costs go (seconds since start, on stderr)
    const bool clk_on = getenv("HM_CLI_TIMING") != nullptr;
    auto clk = [&](const char* what) {
        if (clk_on) fprintf(stderr, "[%s] t=%.3f s: %s\n", kName, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what);
    };
    if (o.ld_out) bam_use_libdeflate_compress(true);
    BgzfReader in(o.in, o.threads);
    if (!in.ok()) { fprintf(stderr, "[%s] %s\n", kName, in.error().c_str()); return EXIT_FAILURE; }
    std::string err;
    ShardSource src{o.shard, o.queue, o.queue.empty() ? 0 : (o.chunks > 0 ? o.chunks : default_chunks(o.in))};

    // one engine per device, three batch slots each: one being staged, one on the GPU, one being tagged / written
    std::vector<hm_engine_t*> eng(o.devices.size(), nullptr);
    for (size_t d = 0; d < eng.size(); ++d) {
        if (d == 0 && (hm_abi_version() != HM_ABI_VERSION || hm_timing_size() != sizeof(hm_timing_t))) {
            fprintf(stderr, "[hifimeth-hip] libhifimeth_hip.so was built from another include/hifimeth_hip.h (ABI %d, this program %d)\n",
                    hm_abi_version(), HM_ABI_VERSION);
            return EXIT_FAILURE;
        }
        if (hm_create(&eng[d], o.model_dir.c_str(), o.ctx_mask, o.devices[d]) < 0) {
            fprintf(stderr, "[%s] device %d: %s\n", kName, o.devices[d], hm_last_error(nullptr));
            return EXIT_FAILURE;
        }
        hm_set_option(eng[d], "min_read_size", o.min_read_size);
        hm_set_option(eng[d], "precision", o.precision);
        hm_set_option(eng[d], "slots", 3);
        // (trunk read groups: the engine's own default -- sized from free device memory, the same as bench.py runs; a group never
        //  exceeds the slab it is cut from, and the maps are allocated for the largest group actually seen)
    }
    {
        int tmask = o.trunk == 0 ? 0 : o.trunk == 1 ? o.ctx_mask : head_trunk_mask(o, err);
        if (tmask < 0) { fprintf(stderr, "[%s] %s: %s\n", kName, o.in.c_str(), err.c_str()); return EXIT_FAILURE; }
        for (auto* e : eng) hm_set_option(e, "trunk_mask", tmask);
    }
    clk("engines created, kernel paths chosen from the head of the input");
    size_t all_reads = 0, all_bases = 0, all_ctx[3] = {0, 0, 0}, all_parts = 0;
    std::atomic<bool> failed{false};

    // A producer thread inflates and parses batch k+1 while this thread stages batch k into a free slot of the least
    // loaded device (a pull queue: a device takes work whenever one of its slots frees up); a consumer thread collects
    // finished batches in submission order, builds the tags on the host threads and deflates.  The pipeline runs on across
    // the parts of the input this process takes: the consumer switches output files when a job of the next part arrives.
    struct Batch {
        std::vector<BamRecord> recs;
        bool eof = false;
        std::string err;
    };
    Batch nb[2];
    int64_t end_off = 0;
    auto produce = [&](Batch& bt) {
        bt.recs.clear();
        bt.err.clear();
        bt.eof = false;
        int64_t bases = 0;
        while ((int)bt.recs.size() < o.read_batch && bases < o.slab_bases) {
            if (end_off < 0 || in.block_offset() >= end_off) { bt.eof = true; break; }  // the next record is another part's
            BamRecord r;
            if (!read_record(in, r, bt.err)) { bt.eof = true; break; }
            bases += r.l_qseq();
            bt.recs.push_back(std::move(r));
        }
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> ready;
    std::vector<int> inflight(eng.size(), 0);
    bool no_more = false;
    std::unique_ptr<BgzfWriter> out;
    std::string out_path;

    auto close_out = [&]() {
        if (out && !out->close()) { fprintf(stderr, "[%s] %s: %s\n", kName, out_path.c_str(), out->error().c_str()); failed = true; }
        out.reset();
    };
    auto finish = [&](Job& jb) {
        if (jb.open_part) {  // the first job of a part: its output file (and, for the first part of the input, the header)
            close_out();
            out_path = shard_path(o.out, jb.part, !o.queue.empty());
            out.reset(new BgzfWriter(out_path, o.threads, o.level));
            if (!out->ok()) { fprintf(stderr, "[%s] %s\n", kName, out->error().c_str()); failed = true; return; }
            if (jb.part.rank == 0) write_header(*out, jb.hdr);
            ++all_parts;
            return;
        }
        const hm_call_t* calls = nullptr;
        const int64_t got = hm_batch_wait(jb.batch, &calls);
        if (got < 0) { fprintf(stderr, "[%s] %s\n", kName, hm_last_error(eng[jb.dev])); failed = true; return; }
        for (int c = 0; c < 3; ++c) all_ctx[c] += (size_t)std::max<int64_t>(0, hm_batch_num_sites(jb.batch, c));
        // calls are grouped by read in submission order: find every read's range, then build the tags of all reads
        // in parallel (the MM deltas walk every base of the read), then write in order
        std::vector<size_t> first(jb.recs.size() + 1, 0);
        size_t ci = 0;
        for (size_t i = 0; i < jb.recs.size(); ++i) {
            first[i] = ci;
            while (ci < (size_t)got && calls[ci].read_id == (int32_t)i) ++ci;
        }
        first[jb.recs.size()] = ci;
        std::vector<std::string> errs(jb.recs.size());
        parallel_run((int)jb.recs.size(), o.threads, [&](int i) {
            apply_calls(jb.recs[(size_t)i], calls + first[(size_t)i], first[(size_t)i + 1] - first[(size_t)i], o.keep_kinetics,
                        errs[(size_t)i]);
        });
        hm_batch_release(jb.batch);  // the pinned result view is no longer needed: the slot can take the next batch
        jb.batch = nullptr;
        for (size_t i = 0; i < jb.recs.size(); ++i) {
            if (!errs[i].empty()) {
                fprintf(stderr, "[%s] read %zu: %s\n", kName, all_reads + i, errs[i].c_str());
                failed = true;
                return;
            }
            all_bases += (size_t)jb.recs[i].l_qseq();
            write_record(*out, jb.recs[i]);
        }
        if (all_reads == 0) clk("first slab called, tagged and written");
        all_reads += jb.recs.size();
        fprintf(stderr, "[%s] %zu reads done\n", kName, all_reads);
    };
    std::thread writer([&]() {
        for (;;) {
            Job jb;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !ready.empty() || no_more; });
                if (ready.empty()) return;
                jb = std::move(ready.front());
                ready.pop_front();
            }
            if (!failed) finish(jb);
            if (jb.batch) hm_batch_release(jb.batch);
            if (!jb.open_part) {
                std::lock_guard<std::mutex> lk(mu);
                --inflight[jb.dev];
            }
            cv.notify_all();
        }
    });
    auto push = [&](Job&& jb) {
        {
            std::lock_guard<std::mutex> lk(mu);
            ready.push_back(std::move(jb));
        }
        cv.notify_all();
    };
    auto launch = [&](Job&& jb) {
        static std::atomic<int> n_launch{0};
        const int my_launch = n_launch++;
        if (my_launch == 0) clk("first slab read, parsed and staged");
        const int rc_enq = hm_batch_enqueue(jb.batch);
        if (my_launch == 0) clk("first slab queued on the device (group buffers allocated)");
        if (rc_enq < 0) {
            fprintf(stderr, "[%s] %s\n", kName, hm_last_error(eng[jb.dev]));
            failed = true;
            hm_batch_release(jb.batch);
            std::lock_guard<std::mutex> lk(mu);
            --inflight[jb.dev];
            return;
        }
        push(std::move(jb));
    };
    auto begin_job = [&](Job& jb) {
        {   // the device with the fewest batches in flight; wait while every slot everywhere is taken
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return *std::min_element(inflight.begin(), inflight.end()) < 3; });
            jb.dev = (size_t)(std::min_element(inflight.begin(), inflight.end()) - inflight.begin());
            ++inflight[jb.dev];
        }
        jb.batch = hm_batch_begin(eng[jb.dev]);
        if (!jb.batch) {
            fprintf(stderr, "[%s] %s\n", kName, hm_last_error(eng[jb.dev]));
            failed = true;
            std::lock_guard<std::mutex> lk(mu);
            --inflight[jb.dev];
        }
    };

    Shard part;
    ShardIndex sidx;
    while (!failed && src.next(part)) {
        BamHeader hdr;
        if (!open_shard(in, o.in, part, hdr, end_off, err, &sidx)) { fprintf(stderr, "[%s] %s: %s\n", kName, o.in.c_str(), err.c_str()); failed = true; break; }
        {
            Job op;
            op.part = part;
            op.open_part = true;
            if (part.rank == 0) {
                add_pg(hdr, argc, argv);
                op.hdr = hdr;
            }
            push(std::move(op));
        }
        int cur = 0;
        produce(nb[0]);
        while (!failed) {
            Batch& bt = nb[cur];
            if (!bt.err.empty()) { fprintf(stderr, "[%s] %s: %s\n", kName, o.in.c_str(), bt.err.c_str()); failed = true; break; }
            std::thread producer;
            if (!bt.eof) producer = std::thread(produce, std::ref(nb[cur ^ 1]));
            if (!bt.recs.empty()) {
                Job jb;
                jb.part = part;
                begin_job(jb);
                for (size_t i = 0; i < bt.recs.size() && !failed; ++i) {
                    BamRecord& r = bt.recs[i];
                    const KineticsView kv = kinetics_of(r);
                    int rc = hm_batch_submit_read(jb.batch, (int32_t)jb.recs.size(), r.l_qseq(), r.flag(), r.seq4(), kv.arr[0], kv.width[0],
                                                  kv.arr[1], kv.width[1], kv.arr[2], kv.width[2], kv.arr[3], kv.width[3]);
                    if (rc == HM_ENOMEM) {  // the slot is full (2^31 bases): queue it and go on in a fresh one
                        launch(std::move(jb));
                        jb = Job();
                        jb.part = part;
                        begin_job(jb);
                        if (failed) break;
                        rc = hm_batch_submit_read(jb.batch, 0, r.l_qseq(), r.flag(), r.seq4(), kv.arr[0], kv.width[0], kv.arr[1],
                                                  kv.width[1], kv.arr[2], kv.width[2], kv.arr[3], kv.width[3]);
                    }
                    if (rc < 0) { fprintf(stderr, "[%s] %s\n", kName, hm_last_error(eng[jb.dev])); failed = true; break; }
                    jb.recs.push_back(std::move(r));
                }
                bt.recs.clear();
                if (!failed) launch(std::move(jb));
                else if (jb.batch) {
                    hm_batch_release(jb.batch);
                    std::lock_guard<std::mutex> lk(mu);
                    --inflight[jb.dev];
                }
            }
            if (producer.joinable()) producer.join();
            if (bt.eof) break;
            cur ^= 1;
        }
    }
    if (!src.err.empty()) {  // the queue could not be reached: not the same as "no part left"
        fprintf(stderr, "[%s] %s\n", kName, src.err.c_str());
        failed = true;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        no_more = true;
    }
    cv.notify_all();
    clk("last slab queued");
    writer.join();
    close_out();
    clk("output closed");
    for (auto* e : eng) hm_destroy(e);
    clk("engines destroyed");
    if (failed) return EXIT_FAILURE;
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "******** Final stats:\n  ## Reads: %zu\n  ## Bases: %zu\n", all_reads, all_bases);
    static const char* cn[3] = {"CpG", "CHG", "CHH"};
    for (int c = 0; c < 3; ++c)
        if (all_ctx[c]) fprintf(stderr, "  ## %s samples: %zu\n", cn[c], all_ctx[c]);
    // which library produced / consumed the compressed bytes (-Z falls back to zlib silently where libdeflate is missing)
    fprintf(stderr, "  ## BGZF codec: inflate %s, deflate %s\n", bam_have_libdeflate() ? "libdeflate" : "zlib",
            o.ld_out && bam_have_libdeflate() ? "libdeflate" : o.ld_out ? "zlib (-Z asked, libdeflate not available)" : "zlib");
    if (!o.queue.empty()) fprintf(stderr, "  ## Parts taken from the queue: %zu of %d\n", all_parts, src.chunks);
    fprintf(stderr, "  ## Wall time: %.2f s (%.0f sites/s end to end)\n", sec, (double)(all_ctx[0] + all_ctx[1] + all_ctx[2]) / sec);
    return 0;
}

// bamcopy [-R r/w | -Q queue [-C n]] IN.bam OUT.bam : BGZF/BAM round trip of the parts of the input this process takes
int cmd_bamcopy(int argc, char** argv) {
    ShardSource src;
    int a = 2;
    while (a + 1 < argc && argv[a][0] == '-') {
        const std::string f = argv[a];
        if (f == "-R") { if (!parse_shard(argv[a + 1], src.fixed)) { usage(); return EXIT_FAILURE; } }
        else if (f == "-Q") src.queue = argv[a + 1];
        else if (f == "-C") src.chunks = atoi(argv[a + 1]);
        else { usage(); return EXIT_FAILURE; }
        a += 2;
    }
    if (argc - a != 2) { usage(); return EXIT_FAILURE; }
    if (!src.queue.empty() && src.chunks <= 0) src.chunks = default_chunks(argv[a]);
    BgzfReader in(argv[a], 4);
    if (!in.ok()) { fprintf(stderr, "%s\n", in.error().c_str()); return EXIT_FAILURE; }
    Shard sh;
    ShardIndex sidx;
    size_t parts = 0, n = 0;
    while (src.next(sh)) {
        BamHeader h;
        std::string err;
        int64_t end_off = 0;
        if (!open_shard(in, argv[a], sh, h, end_off, err, &sidx)) { fprintf(stderr, "%s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
        BgzfWriter out(shard_path(argv[a + 1], sh, src.queued()), 4, 6);
        if (!out.ok()) return EXIT_FAILURE;
        if (sh.rank == 0) write_header(out, h);
        BamRecord r;
        while (end_off >= 0 && in.block_offset() < end_off && read_record(in, r, err)) {
            write_record(out, r);
            ++n;
        }
        if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
        if (!out.close()) return EXIT_FAILURE;
        ++parts;
    }
    if (!src.err.empty()) { fprintf(stderr, "[%s] %s\n", kName, src.err.c_str()); return EXIT_FAILURE; }
    if (src.queue.empty()) fprintf(stderr, "[%s] bamcopy: rank %d/%d wrote %zu records\n", kName, sh.rank, sh.world, n);
    else fprintf(stderr, "[%s] bamcopy: took %zu of %d parts from the queue, wrote %zu records, block scans %d\n", kName, parts, src.chunks, n, sidx.scans);
    return 0;
}

// merge OUT.bam N : joins OUT.bam.shard0 .. shard<N-1> in rank order.  A BGZF file is a series of independent gzip members,
// so the shards are concatenated as they are, minus the 28-byte end-of-file marker of all but the last.
int cmd_merge(int argc, char** argv) {
    if (argc != 4) { usage(); return EXIT_FAILURE; }
    const std::string out = argv[2];
    const int n = atoi(argv[3]);
    if (n < 1) { usage(); return EXIT_FAILURE; }
    FILE* fo = fopen(out.c_str(), "wb");
    if (!fo) { fprintf(stderr, "[%s] cannot create %s\n", kName, out.c_str()); return EXIT_FAILURE; }
    std::vector<char> buf(size_t(8) << 20);
    for (int r = 0; r < n; ++r) {
        const std::string sp = out + ".shard" + std::to_string(r);
        FILE* fi = fopen(sp.c_str(), "rb");
        if (!fi) { fprintf(stderr, "[%s] cannot open %s\n", kName, sp.c_str()); fclose(fo); return EXIT_FAILURE; }
        fseeko(fi, 0, SEEK_END);
        int64_t left = (int64_t)ftello(fi) - (r + 1 < n ? 28 : 0);
        fseeko(fi, 0, SEEK_SET);
        while (left > 0) {
            const size_t want = (size_t)std::min<int64_t>(left, (int64_t)buf.size());
            if (fread(buf.data(), 1, want, fi) != want || fwrite(buf.data(), 1, want, fo) != want) {
                fprintf(stderr, "[%s] i/o error on %s\n", kName, sp.c_str());
                fclose(fi);
                fclose(fo);
                return EXIT_FAILURE;
            }
            left -= (int64_t)want;
        }
        fclose(fi);
    }
    if (fclose(fo) != 0) return EXIT_FAILURE;
    for (int r = 0; r < n; ++r) remove((out + ".shard" + std::to_string(r)).c_str());
    return 0;
}

// stagebench [-t N] [-R r/w] IN.bam : the host side of `call` without a GPU -- BGZF inflate, record parsing and the copy
// of SEQ + kinetics into a staging slab exactly as hm_submit_read lays it out (plain memory instead of pinned).  Prints the
// rate one rank's host side sustains: what has to exceed the GPU's appetite for the device to stay busy.
int cmd_stagebench(int argc, char** argv) {
    int threads = (int)std::max(1u, std::thread::hardware_concurrency());
    Shard sh;
    int a = 2;
    while (a + 1 < argc && argv[a][0] == '-') {
        if (std::string(argv[a]) == "-t") threads = std::max(1, atoi(argv[a + 1]));
        else if (std::string(argv[a]) == "-R") { if (!parse_shard(argv[a + 1], sh)) { usage(); return EXIT_FAILURE; } }
        else { usage(); return EXIT_FAILURE; }
        a += 2;
    }
    if (argc - a != 1) { usage(); return EXIT_FAILURE; }
    // the staging slab exists before the clock starts (the engine's pinned slabs are allocated once and reused)
    const size_t slab_bytes = size_t(256) << 20;
    uint8_t* slab = static_cast<uint8_t*>(malloc(slab_bytes));
    if (!slab) return EXIT_FAILURE;
    memset(slab, 0, slab_bytes);
    const auto t0 = std::chrono::steady_clock::now();
    BgzfReader in(argv[a], threads);
    BamHeader h;
    std::string err;
    int64_t end_off = 0;
    if (!in.ok() || !open_shard(in, argv[a], sh, h, end_off, err)) { fprintf(stderr, "%s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    const int64_t first_off = in.block_offset();
    size_t fill = 0, reads = 0, staged = 0, bases = 0, bytes_staged = 0;
    BamRecord r;
    auto put = [&](const void* src, size_t n) {
        const size_t al = (n + 15) & ~size_t(15);
        if (fill + al > slab_bytes) fill = 0;  // a full slab would be handed to the device here
        memcpy(slab + fill, src, n);
        fill += al;
        bytes_staged += n;
    };
    while (end_off >= 0 && in.block_offset() < end_off && read_record(in, r, err)) {
        ++reads;
        const KineticsView kv = kinetics_of(r);
        const size_t L = (size_t)r.l_qseq();
        if (L < 1000 || !kv.arr[0] || !kv.arr[1] || !kv.arr[2] || !kv.arr[3]) continue;
        put(r.seq4(), (L + 1) / 2);
        for (int k = 0; k < 4; ++k) put(kv.arr[k], L * (size_t)kv.width[k]);
        ++staged;
        bases += L;
    }
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int64_t comp = in.block_offset() - first_off;
    printf("{\"rank\": %d, \"world\": %d, \"threads\": %d, \"reads\": %zu, \"staged_reads\": %zu, \"bases\": %zu, \"seconds\": %.3f, "
           "\"bam_MB_per_s\": %.1f, \"staged_MB_per_s\": %.1f, \"Mbases_per_s\": %.1f}\n",
           sh.rank, sh.world, threads, reads, staged, bases, sec, (double)comp / sec / 1e6, (double)bytes_staged / sec / 1e6,
           (double)bases / sec / 1e6);
    return 0;
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

// test seam: the MM/ML parser's output per record, "n" then n lines "qoff strand unmod_base code prob" (the format of the
// reference-parser driver behind tests/golden/modparse.json)
int cmd_modlist(int argc, char** argv) {
    if (argc != 3) { usage(); return EXIT_FAILURE; }
    BgzfReader in(argv[2], 2);
    BamHeader h;
    std::string err;
    if (!in.ok() || !read_header(in, h, err)) { fprintf(stderr, "%s%s\n", in.error().c_str(), err.c_str()); return EXIT_FAILURE; }
    BamRecord r;
    std::vector<BaseMod> mods;
    uint64_t reads = 0;
    while (read_record(in, r, err)) {
        ++reads;
        if (!parse_mods(r, mods, err)) { fprintf(stderr, "read %llu: %s\n", (unsigned long long)reads - 1, err.c_str()); return EXIT_FAILURE; }
        printf("%zu\n", mods.size());
        for (const BaseMod& m : mods) printf("%d %d %c %c %d\n", m.qoff, (int)m.strand, m.unmod_base, m.code, (int)m.prob);
    }
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
    return 0;
}

}  // namespace

int cmd_pileup(int argc, char** argv);   // hifimeth_pileup.cpp
int cmd_fastats(int argc, char** argv);  // hifimeth_pileup.cpp
int cmd_thresholds(int argc, char** argv);
int cmd_corr(int argc, char** argv);     // hifimeth_pileup.cpp
int cmd_cov2bed(int argc, char** argv);  // hifimeth_pileup.cpp
int cmd_sample(int argc, char** argv);   // hifimeth_pileup.cpp
int cmd_eval(int argc, char** argv);     // hifimeth_pileup.cpp

int main(int argc, char** argv) {
    if (argc < 2) { usage(); return EXIT_FAILURE; }
    const std::string cmd = argv[1];
    if (cmd == "call") return cmd_call(argc, argv);
    if (cmd == "bamcopy") return cmd_bamcopy(argc, argv);
    if (cmd == "merge") return cmd_merge(argc, argv);
    if (cmd == "stagebench") return cmd_stagebench(argc, argv);
    if (cmd == "tagtest") return cmd_tagtest(argc, argv);
    if (cmd == "modstats") return cmd_modstats(argc, argv);
    if (cmd == "pileup") return cmd_pileup(argc, argv);
    if (cmd == "fastats") return cmd_fastats(argc, argv);
    if (cmd == "thresholds") return cmd_thresholds(argc, argv);
    if (cmd == "modlist") return cmd_modlist(argc, argv);
    if (cmd == "corr") return cmd_corr(argc, argv);
    if (cmd == "cov2bed") return cmd_cov2bed(argc, argv);
    if (cmd == "sample") return cmd_sample(argc, argv);
    if (cmd == "eval") return cmd_eval(argc, argv);
    usage();
    return EXIT_FAILURE;
}
