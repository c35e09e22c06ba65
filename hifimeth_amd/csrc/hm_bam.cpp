// hm_bam.cpp -- see hm_bam.h
#include "hm_bam.h"

#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_set>

// ---- libdeflate, when the image has it (dlopen of libdeflate.so.0: no build-time dependency) ----------------------------------------
// BGZF blocks are small independent raw-deflate streams: libdeflate inflates them ~2-3x faster than zlib and checks CRC-32 ~5x faster
// (the reference's reader is htslib, which uses libdeflate the same way when built with it; sam_batch.hpp:12-23).  Inflate and CRC
// give the same bytes whichever library runs; DEFLATE output differs between the two, so the writer keeps zlib unless asked
// (hm_bam_use_libdeflate_compress), and output files stay byte-identical to earlier rounds' at the same -z.
namespace {
struct LibDeflate {
    void* h = nullptr;
    void* (*alloc_d)() = nullptr;
    int (*inflate)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_d)(void*) = nullptr;
    void* (*alloc_c)(int) = nullptr;
    size_t (*deflate)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*bound)(void*, size_t) = nullptr;
    void (*free_c)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    LibDeflate() {
        if (getenv("HM_NO_LIBDEFLATE")) return;
        for (const char* n : {"libdeflate.so.0", "libdeflate.so"}) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        alloc_d = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        inflate = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        free_d = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        alloc_c = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
        deflate = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
        bound = (size_t (*)(void*, size_t))dlsym(h, "libdeflate_deflate_compress_bound");
        free_c = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        if (!alloc_d || !inflate || !free_d || !alloc_c || !deflate || !bound || !free_c || !crc) h = nullptr;
    }
    bool ok() const { return h != nullptr; }
};
const LibDeflate& libdeflate() {
    static const LibDeflate ld;
    return ld;
}
// one decompressor / compressor per pool worker, for the life of the process (they are not thread-safe, and allocating one per
// 64 KB block costs more than the block)
struct TlsD {
    void* d = nullptr;
    ~TlsD() { if (d) libdeflate().free_d(d); }
};
struct TlsC {
    void* c = nullptr;
    int level = -1;
    ~TlsC() { if (c) libdeflate().free_c(c); }
};
std::atomic<bool> g_ld_compress{false};
uint32_t crc_of(const uint8_t* p, size_t n) {
    const LibDeflate& ld = libdeflate();
    return ld.ok() ? ld.crc(0, p, n) : (uint32_t)crc32(0L, p, (uInt)n);
}
// raw-deflate stream -> exactly `isize` bytes; false on any error
bool inflate_block(const uint8_t* in, size_t n_in, uint8_t* out, size_t isize) {
    const LibDeflate& ld = libdeflate();
    if (ld.ok()) {
        thread_local TlsD t;
        if (!t.d) t.d = ld.alloc_d();
        if (t.d) {
            size_t got = 0;
            return ld.inflate(t.d, in, n_in, out, isize, &got) == 0 && got == isize;
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t*>(in);
    zs.avail_in = (uInt)n_in;
    zs.next_out = out;
    zs.avail_out = (uInt)isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.total_out == isize;
}
}  // namespace

namespace hmbam {
bool bam_have_libdeflate() { return libdeflate().ok(); }
void bam_use_libdeflate_compress(bool on) { g_ld_compress = on && libdeflate().ok(); }
}  // namespace hmbam

namespace hmbam {

namespace {

// Worker pool of the process (advisor r4 #4): parallel_for used to start `threads` std::threads per call -- once per 256-block BGZF
// batch, per record batch, per output flush -- so every thread_local (de)compressor lived for one batch.  The workers below are
// started on first need, grow to the largest `threads` any caller asked for and live until the process ends (the pool object is
// never destroyed: no join order to get wrong against the HIP runtime's own exit handlers), so a worker's libdeflate state is
// allocated once.  Several callers may have a job open at the same time (the reader's prefetch inflates while the writer deflates and
// the tag builder runs): workers take items from the oldest job that still has any, at most `threads` of them on one job.  The caller
// only waits; a call made from inside a worker runs its items inline (no nesting deadlock).
class Pool {
  public:
    static Pool& get() {
        static Pool* p = new Pool;
        return *p;
    }
    void run(int n, int threads, const std::function<void(int)>& f) {
        threads = std::max(1, std::min(threads, n));
        if (threads == 1 || in_worker_) {
            for (int i = 0; i < n; ++i) f(i);
            return;
        }
        Job job;
        job.f = &f;
        job.n = n;
        job.cap = threads;
        std::unique_lock<std::mutex> lk(mu_);
        while ((int)workers_.size() < std::min(threads, kMaxWorkers)) workers_.emplace_back([this] { work(); });
        jobs_.push_back(&job);
        cv_.notify_all();
        job.cv_done.wait(lk, [&] { return job.done == job.n; });
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job));
    }

  private:
    struct Job {
        const std::function<void(int)>* f = nullptr;
        int n = 0, cap = 0;
        int next = 0, done = 0, active = 0;  // all under mu_
        std::condition_variable cv_done;
    };
    static constexpr int kMaxWorkers = 256;
    static thread_local bool in_worker_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Job*> jobs_;  // oldest first
    std::vector<std::thread> workers_;

    Job* pick() {
        for (Job* j : jobs_)
            if (j->next < j->n && j->active < j->cap) return j;
        return nullptr;
    }
    void work() {
        in_worker_ = true;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            Job* j = nullptr;
            cv_.wait(lk, [&] { return (j = pick()) != nullptr; });
            ++j->active;
            // a few items per lock round trip when there are many (records), one when there are few (blocks, slices)
            while (j->next < j->n) {
                const int a = j->next, b = std::min(j->n, a + std::max(1, j->n / (j->cap * 16)));
                j->next = b;
                lk.unlock();
                for (int i = a; i < b; ++i) (*j->f)(i);
                lk.lock();
                j->done += b - a;
            }
            --j->active;
            if (j->done == j->n) j->cv_done.notify_one();
        }
    }
};
thread_local bool Pool::in_worker_ = false;

template <class F>
void parallel_for(int n, int threads, F f) {
    if (n <= 0) return;
    const std::function<void(int)> fn = std::ref(f);
    Pool::get().run(n, threads, fn);
}

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void wr16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
inline void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

constexpr size_t BGZF_MAX_PAYLOAD = 0xff00;  // uncompressed bytes per block (as htslib)
constexpr int BATCH_BLOCKS = 256;

const uint8_t kEofBlock[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43,
                               0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};

}  // namespace

void parallel_run(int n, int threads, const std::function<void(int)>& f) { parallel_for(n, threads, f); }

// ------------------------------------------------------------------------------------------------
// BGZF
// ------------------------------------------------------------------------------------------------
BgzfReader::BgzfReader(const std::string& path, int threads) : threads_(std::max(1, threads)) {
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) {
        err_ = "cannot open " + path;
        return;
    }
    start_prefetch();
}

BgzfReader::~BgzfReader() {
    if (bg_) {
        static_cast<std::thread*>(bg_)->join();
        delete static_cast<std::thread*>(bg_);
    }
    if (fp_) fclose(fp_);
}

// BSIZE of a gzip member's extra field (SAMv1 4.1: subfield 'B' 'C', SLEN 2), -1 if there is none.  Subfields are
// walked with their lengths checked against XLEN: a corrupt SLEN must not read past the field.
static int bgzf_bsize(const uint8_t* extra, int xlen) {
    for (int i = 0; i + 4 <= xlen;) {
        const int slen = rd16(extra + i + 2);
        if (i + 4 + slen > xlen) return -1;
        if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) return rd16(extra + i + 4);
        i += 4 + slen;
    }
    return -1;
}

void BgzfReader::load(Chunk& c, int64_t from) {
    c.data.clear();
    c.segs.clear();
    c.err.clear();
    c.eof = false;
    c.next_off = from;
    struct Blk {
        std::vector<uint8_t> comp;
        uint32_t crc, isize;
        size_t at;
        int64_t off;
    };
    std::vector<Blk> blocks;
    size_t total = 0;
    if (fseeko(fp_, (off_t)from, SEEK_SET) != 0) {
        c.err = "seek failed";
        return;
    }
    while ((int)blocks.size() < 2 * BATCH_BLOCKS) {
        uint8_t hdr[12];
        const size_t got = fread(hdr, 1, 12, fp_);
        if (got == 0) {
            c.eof = true;
            break;
        }
        if (got != 12 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) {
            c.err = "not a BGZF file (bad gzip member header)";
            return;
        }
        const int xlen = rd16(hdr + 10);
        uint8_t extra[65536];  // XLEN is a 16-bit field: any valid member fits
        if (fread(extra, 1, (size_t)xlen, fp_) != (size_t)xlen) {
            c.err = "truncated BGZF block";
            return;
        }
        const int bsize = bgzf_bsize(extra, xlen);
        if (bsize < 0) {
            c.err = "BGZF block without BC field";
            return;
        }
        const long clen = (long)bsize + 1 - 12 - xlen - 8;
        if (clen < 0) {
            c.err = "corrupt BGZF block size";
            return;
        }
        Blk b;
        b.comp.resize((size_t)clen);
        uint8_t tail[8];
        if (fread(b.comp.data(), 1, (size_t)clen, fp_) != (size_t)clen || fread(tail, 1, 8, fp_) != 8) {
            c.err = "truncated BGZF block";
            return;
        }
        b.crc = rd32(tail);
        b.isize = rd32(tail + 4);
        if (b.isize > 0x10000) {
            c.err = "corrupt BGZF block (ISIZE)";
            return;
        }
        b.at = total;
        b.off = c.next_off;
        total += b.isize;
        c.next_off += (int64_t)bsize + 1;
        blocks.push_back(std::move(b));
    }
    c.data.resize(total);
    std::atomic<bool> bad{false};
    parallel_for((int)blocks.size(), threads_, [&](int i) {
        Blk& b = blocks[(size_t)i];
        if (b.isize == 0) return;
        if (!inflate_block(b.comp.data(), b.comp.size(), c.data.data() + b.at, b.isize) || crc_of(c.data.data() + b.at, b.isize) != b.crc) bad = true;
    });
    if (bad) {
        c.err = "BGZF block failed to inflate (corrupt data or CRC mismatch)";
        return;
    }
    for (auto& b : blocks)
        if (b.isize) c.segs.emplace_back(b.at, b.off);
}

void BgzfReader::start_prefetch() {
    const int64_t from = nxt_.next_off;
    bg_ = new std::thread([this, from] { load(nxt_, from); });
}

bool BgzfReader::advance(bool append) {
    if (!bg_) return false;  // nothing more to read
    static_cast<std::thread*>(bg_)->join();
    delete static_cast<std::thread*>(bg_);
    bg_ = nullptr;
    if (!nxt_.err.empty()) {
        err_ = nxt_.err;
        return false;
    }
    const bool was_eof = nxt_.eof;
    if (append) {
        // keep the unread part of the current chunk in front (a record, or the peek window, spans the chunk boundary)
        size_t keep = 0;
        while (keep + 1 < cur_.segs.size() && cur_.segs[keep + 1].first <= pos_) ++keep;
        cur_.segs.erase(cur_.segs.begin(), cur_.segs.begin() + (long)keep);
        for (auto& sg : cur_.segs) sg.first = sg.first >= pos_ ? sg.first - pos_ : 0;
        if (pos_ >= cur_.data.size()) cur_.segs.clear();
        cur_.data.erase(cur_.data.begin(), cur_.data.begin() + (long)std::min(pos_, cur_.data.size()));
        pos_ = 0;
        const size_t base = cur_.data.size();
        for (auto& sg : nxt_.segs) cur_.segs.emplace_back(base + sg.first, sg.second);
        cur_.data.insert(cur_.data.end(), nxt_.data.begin(), nxt_.data.end());
        cur_.next_off = nxt_.next_off;
    } else {
        std::swap(cur_, nxt_);
        nxt_.next_off = cur_.next_off;
        pos_ = 0;
    }
    cur_.eof = was_eof;
    nxt_.next_off = cur_.next_off;
    if (!was_eof) start_prefetch();
    return true;
}

bool BgzfReader::read(void* dst, size_t n) {
    uint8_t* d = static_cast<uint8_t*>(dst);
    size_t done = 0;
    while (done < n) {
        const size_t avail = cur_.data.size() - pos_;
        if (avail == 0) {
            if (!advance(false)) {
                if (done != 0 && err_.empty()) err_ = "truncated file";
                return false;
            }
            continue;
        }
        const size_t take = std::min(avail, n - done);
        memcpy(d + done, cur_.data.data() + pos_, take);
        pos_ += take;
        done += take;
    }
    return true;
}

bool BgzfReader::seek_block(int64_t file_offset) {
    if (!fp_) return false;
    if (bg_) {
        static_cast<std::thread*>(bg_)->join();
        delete static_cast<std::thread*>(bg_);
        bg_ = nullptr;
    }
    cur_ = Chunk();
    nxt_ = Chunk();
    pos_ = 0;
    cur_.next_off = nxt_.next_off = file_offset;
    err_.clear();
    start_prefetch();
    return true;
}

int64_t BgzfReader::block_offset() {
    // the next unread byte lies in the next chunk: look there, so that empty blocks in between (an EOF marker inside a
    // concatenated file) are skipped and the answer is the block that really holds the byte
    while (pos_ >= cur_.data.size())
        if (!advance(false)) return cur_.next_off;  // end of file (an error shows up in the next read())
    int64_t off = cur_.segs.empty() ? cur_.next_off : cur_.segs.front().second;
    for (const auto& sg : cur_.segs) {
        if (sg.first > pos_) break;
        off = sg.second;
    }
    return off;
}

size_t BgzfReader::peek(const uint8_t*& p, size_t want) {
    while (cur_.data.size() - pos_ < want) {
        if (!advance(true)) break;
    }
    p = cur_.data.data() + pos_;
    return std::min(want, cur_.data.size() - pos_);
}

void BgzfReader::skip(size_t n) { pos_ = std::min(cur_.data.size(), pos_ + n); }

bool scan_bgzf_blocks(const std::string& path, std::vector<int64_t>& offsets, int64_t& file_size, std::string& err) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) {
        err = "cannot open " + path;
        return false;
    }
    offsets.clear();
    int64_t at = 0;
    std::vector<uint8_t> extra(65536);
    for (;;) {
        uint8_t hdr[12];
        if (fseeko(fp, (off_t)at, SEEK_SET) != 0) break;
        const size_t got = fread(hdr, 1, 12, fp);
        if (got == 0) break;
        // the same member walk as BgzfReader::load: any subfield order, any XLEN
        const int xlen = got == 12 ? rd16(hdr + 10) : 0;
        if (got != 12 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4) ||
            fread(extra.data(), 1, (size_t)xlen, fp) != (size_t)xlen) {
            err = "not a BGZF file (bad gzip member header)";
            fclose(fp);
            return false;
        }
        const int bsize = bgzf_bsize(extra.data(), xlen);
        if (bsize < 0 || bsize + 1 < 12 + xlen + 8) {
            err = "BGZF block without a valid BC field";
            fclose(fp);
            return false;
        }
        offsets.push_back(at);
        at += (int64_t)bsize + 1;
    }
    file_size = at;
    fclose(fp);
    return true;
}

static bool plausible_record(const uint8_t* p, size_t avail, int n_ref, size_t& total, bool& complete) {
    complete = false;
    if (avail < 36) return false;
    const uint32_t bs = rd32(p);
    if (bs < 34 || bs > (1u << 28)) return false;
    const int32_t ref = (int32_t)rd32(p + 4), pos = (int32_t)rd32(p + 8);
    const int l_name = p[12], n_cig = rd16(p + 16);
    const int32_t l_seq = (int32_t)rd32(p + 20), nref = (int32_t)rd32(p + 24), npos = (int32_t)rd32(p + 28);
    if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos < -1 || npos < -1) return false;
    if (l_name < 2 || l_seq < 0 || l_seq > (1 << 28)) return false;
    const size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (fixed > bs) return false;
    total = 4 + (size_t)bs;
    const size_t name_end = 4 + 32 + (size_t)l_name;
    if (avail < name_end) return true;  // cannot look further: not contradicted
    for (size_t i = 36; i + 1 < name_end; ++i)
        if (p[i] < 33 || p[i] > 126) return false;
    if (p[name_end - 1] != 0) return false;
    if (avail < total) return true;
    const uint8_t* a = p + 4 + fixed;
    const uint8_t* end = p + total;
    AuxField f;
    while (a < end)
        if (!next_aux(a, end, f)) return false;
    complete = a == end;
    return complete;
}

bool find_record_start(BgzfReader& in, int n_ref, std::string& err) {
    const uint8_t* p;
    const size_t avail = in.peek(p, size_t(16) << 20);
    if (!in.error().empty()) {
        err = in.error();
        return false;
    }
    for (size_t o = 0; o + 36 <= avail; ++o) {
        size_t at = o;
        int ok = 0;
        bool good = true;
        while (ok < 4 && at + 36 <= avail) {  // the candidate and up to three records behind it
            size_t total;
            bool complete;
            if (!plausible_record(p + at, avail - at, n_ref, total, complete)) { good = false; break; }
            if (!complete) break;  // ran out of buffered data: not contradicted
            ++ok;
            at += total;
        }
        if (good && ok >= 1) {
            in.skip(o);
            return true;
        }
    }
    in.skip(avail);
    return false;
}

BgzfWriter::BgzfWriter(const std::string& path, int threads, int level) : threads_(threads), level_(level) {
    fp_ = fopen(path.c_str(), "wb");
    if (!fp_) err_ = "cannot create " + path;
}

BgzfWriter::~BgzfWriter() {
    if (fp_) fclose(fp_);
}

void BgzfWriter::write(const void* src, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(src);
    buf_.insert(buf_.end(), p, p + n);
    if (buf_.size() >= BGZF_MAX_PAYLOAD * BATCH_BLOCKS) flush_blocks(false);
}

void BgzfWriter::flush_blocks(bool all) {
    if (!fp_) return;
    const size_t nfull = buf_.size() / BGZF_MAX_PAYLOAD;
    const size_t nblk = nfull + ((all && buf_.size() % BGZF_MAX_PAYLOAD) ? 1 : 0);
    if (nblk == 0) return;
    std::vector<std::vector<uint8_t>> out(nblk);
    std::atomic<bool> bad{false};
    parallel_for((int)nblk, threads_, [&](int i) {
        const size_t off = (size_t)i * BGZF_MAX_PAYLOAD;
        const size_t len = std::min(BGZF_MAX_PAYLOAD, buf_.size() - off);
        std::vector<uint8_t>& o = out[(size_t)i];
        size_t clen = 0;
        if (g_ld_compress) {
            const LibDeflate& ld = libdeflate();
            thread_local TlsC t;
            if (!t.c || t.level != level_) {
                if (t.c) ld.free_c(t.c);
                t.c = ld.alloc_c(level_ < 1 ? 1 : level_);  // (libdeflate's level 0 stores; zlib's level 0 is not used by the CLI either)
                t.level = level_;
            }
            if (!t.c) { bad = true; return; }
            o.resize(18 + ld.bound(t.c, len) + 8);
            clen = ld.deflate(t.c, buf_.data() + off, len, o.data() + 18, o.size() - 18 - 8);
            if (clen == 0 || 18 + clen + 8 > 0x10000) { bad = true; return; }
        } else {
            o.resize(18 + compressBound((uLong)len) + 8);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                bad = true;
                return;
            }
            zs.next_in = const_cast<uint8_t*>(buf_.data() + off);
            zs.avail_in = (uInt)len;
            zs.next_out = o.data() + 18;
            zs.avail_out = (uInt)(o.size() - 18 - 8);
            const int rc = deflate(&zs, Z_FINISH);
            clen = zs.total_out;
            deflateEnd(&zs);
            if (rc != Z_STREAM_END || 18 + clen + 8 > 0x10000) {
                bad = true;
                return;
            }
        }
        static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00};
        memcpy(o.data(), head, 16);
        wr16(o.data() + 16, (uint32_t)(18 + clen + 8 - 1));
        wr32(o.data() + 18 + clen, crc_of(buf_.data() + off, len));
        wr32(o.data() + 18 + clen + 4, (uint32_t)len);
        o.resize(18 + clen + 8);
    });
    if (bad) {
        err_ = "deflate failed";
        return;
    }
    for (auto& o : out)
        if (fwrite(o.data(), 1, o.size(), fp_) != o.size()) err_ = "write error";
    const size_t consumed = std::min(buf_.size(), nblk * BGZF_MAX_PAYLOAD);
    buf_.erase(buf_.begin(), buf_.begin() + (long)consumed);
}

bool BgzfWriter::close() {
    if (!fp_) return false;
    flush_blocks(true);
    if (fwrite(kEofBlock, 1, sizeof kEofBlock, fp_) != sizeof kEofBlock) err_ = "write error";
    if (fclose(fp_) != 0) err_ = "close error";
    fp_ = nullptr;
    return err_.empty();
}

// ------------------------------------------------------------------------------------------------
// BAM
// ------------------------------------------------------------------------------------------------
int32_t BamRecord::l_qseq() const { return (int32_t)rd32(data.data() + 16); }
uint16_t BamRecord::flag() const { return rd16(data.data() + 14); }
int BamRecord::n_cigar() const { return rd16(data.data() + 12); }
int32_t BamRecord::ref_id() const { return (int32_t)rd32(data.data()); }
int32_t BamRecord::pos() const { return (int32_t)rd32(data.data() + 4); }
const uint8_t* BamRecord::seq4() const { return data.data() + 32 + l_read_name() + 4 * (size_t)n_cigar(); }
size_t BamRecord::aux_offset() const {
    const size_t L = (size_t)l_qseq();
    return 32 + (size_t)l_read_name() + 4 * (size_t)n_cigar() + (L + 1) / 2 + L;
}

bool read_header(BgzfReader& in, BamHeader& h, std::string& err) {
    uint8_t b[8];
    if (!in.read(b, 8) || memcmp(b, "BAM\1", 4) != 0) {
        err = in.error().empty() ? "not a BAM file" : in.error();
        return false;
    }
    const uint32_t l_text = rd32(b + 4);
    h.text.resize(l_text);
    if (l_text && !in.read(&h.text[0], l_text)) { err = "truncated BAM header"; return false; }
    while (!h.text.empty() && h.text.back() == '\0') h.text.pop_back();
    if (!in.read(b, 4)) { err = "truncated BAM header"; return false; }
    const uint32_t n_ref = rd32(b);
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!in.read(b, 4)) { err = "truncated BAM header"; return false; }
        const uint32_t ln = rd32(b);
        std::string name(ln, '\0');
        if ((ln && !in.read(&name[0], ln)) || !in.read(b, 4)) { err = "truncated BAM header"; return false; }
        while (!name.empty() && name.back() == '\0') name.pop_back();
        h.refs.emplace_back(name, (int32_t)rd32(b));
    }
    return true;
}

void write_header(BgzfWriter& out, const BamHeader& h) {
    std::vector<uint8_t> b;
    auto put32 = [&](uint32_t v) { uint8_t t[4]; wr32(t, v); b.insert(b.end(), t, t + 4); };
    b.insert(b.end(), {'B', 'A', 'M', 1});
    put32((uint32_t)h.text.size());
    b.insert(b.end(), h.text.begin(), h.text.end());
    put32((uint32_t)h.refs.size());
    for (auto& r : h.refs) {
        put32((uint32_t)r.first.size() + 1);
        b.insert(b.end(), r.first.begin(), r.first.end());
        b.push_back(0);
        put32((uint32_t)r.second);
    }
    out.write(b.data(), b.size());
}

bool read_record(BgzfReader& in, BamRecord& r, std::string& err) {
    uint8_t b[4];
    if (!in.read(b, 4)) {
        err = in.error();
        return false;  // clean EOF when err is empty
    }
    const uint32_t bs = rd32(b);
    if (bs < 32 || bs > (1u << 30)) { err = "corrupt BAM record size"; return false; }
    r.data.resize(bs);
    if (!in.read(r.data.data(), bs)) { err = in.error().empty() ? "truncated BAM record" : in.error(); return false; }
    if (r.aux_offset() > r.data.size()) { err = "corrupt BAM record"; return false; }
    return true;
}

void write_record(BgzfWriter& out, const BamRecord& r) {
    uint8_t b[4];
    wr32(b, (uint32_t)r.data.size());
    out.write(b, 4);
    out.write(r.data.data(), r.data.size());
}

static int aux_elem_size(char t) {
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    default: return 0;
    }
}

bool next_aux(const uint8_t*& p, const uint8_t* end, AuxField& f) {
    if (end - p < 3) return false;
    const uint8_t* s = p;
    f.tag[0] = (char)s[0];
    f.tag[1] = (char)s[1];
    f.type = (char)s[2];
    f.subtype = 0;
    f.count = 0;
    s += 3;
    if (f.type == 'Z' || f.type == 'H') {
        f.payload = s;
        while (s < end && *s) ++s;
        if (s >= end) return false;
        ++s;
    } else if (f.type == 'B') {
        if (end - s < 5) return false;
        f.subtype = (char)s[0];
        f.count = rd32(s + 1);
        const int es = aux_elem_size(f.subtype);
        s += 5;
        if (es == 0 || (uint64_t)(end - s) < (uint64_t)es * f.count) return false;
        f.payload = s;
        s += (size_t)es * f.count;
    } else {
        const int es = aux_elem_size(f.type);
        if (es == 0 || end - s < es) return false;
        f.payload = s;
        s += es;
    }
    f.total = (size_t)(s - p);
    p = s;
    return true;
}

KineticsView kinetics_of(const BamRecord& r) {
    KineticsView kv{{nullptr, nullptr, nullptr, nullptr}, {1, 1, 1, 1}};
    static const char* names[4] = {"fi", "fp", "ri", "rp"};
    const uint8_t* p = r.data.data() + r.aux_offset();
    const uint8_t* end = r.data.data() + r.data.size();
    AuxField f;
    while (p < end && next_aux(p, end, f)) {
        for (int k = 0; k < 4; ++k) {
            if (f.tag[0] != names[k][0] || f.tag[1] != names[k][1] || kv.arr[k]) continue;
            // first occurrence wins (bam_aux_get); must be B:C or B:S with l_qseq elements (bam_info.cpp:443-453)
            if (f.type == 'B' && (f.subtype == 'C' || f.subtype == 'S') && f.count == (uint32_t)r.l_qseq()) {
                kv.arr[k] = f.payload;
                kv.width[k] = f.subtype == 'C' ? 1 : 2;
            } else {
                kv.arr[k] = nullptr;
            }
        }
    }
    return kv;
}

char fwd_strand_base(const BamRecord& r, int k) {
    static const char dec[16] = {'=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
    const int L = r.l_qseq();
    const uint8_t* s = r.seq4();
    if (r.flag() & 16) {  // stored reverse strand: complement of the mirrored position
        const int i = L - 1 - k;
        const int nib = (s[i >> 1] >> ((~i & 1) << 2)) & 15;
        switch (nib) {
        case 1: return 'T';
        case 2: return 'G';
        case 4: return 'C';
        case 8: return 'A';
        default: return 'N';
        }
    }
    const int nib = (s[k >> 1] >> ((~k & 1) << 2)) & 15;
    return dec[nib];
}

bool apply_calls(BamRecord& r, const hm_call_t* calls, size_t n, bool keep_kinetics, std::string& err) {
    // 1. copy the aux block without fi/ri/fp/rp (unless -k) and without MM / ML (build_mod_bam.cpp:87-109)
    const size_t aux0 = r.aux_offset();
    std::vector<uint8_t> aux;
    const uint8_t* p = r.data.data() + aux0;
    const uint8_t* end = r.data.data() + r.data.size();
    AuxField f;
    // MN = l_qseq (build_mod_bam.cpp:222-224, bam_aux_update_int): an EXISTING MN tag keeps its place among the tags ("This
    // function will not change the ordering of tags in the bam record", htslib sam.h:1844-1866) and its width when the new value
    // fits it, as an unsigned type; a new one is appended in the smallest unsigned type that holds the value
    const uint32_t L = (uint32_t)r.l_qseq();
    auto put_mn = [&](int min_size) {
        const int need = L <= 0xff ? 1 : L <= 0xffff ? 2 : 4, size = std::max(need, min_size);
        aux.insert(aux.end(), {'M', 'N', (uint8_t)(size == 1 ? 'C' : size == 2 ? 'S' : 'I')});
        for (int k = 0; k < size; ++k) aux.push_back((uint8_t)(L >> (8 * k)));
    };
    bool mn_done = false;
    while (p < end) {
        const uint8_t* start = p;
        if (!next_aux(p, end, f)) {
            err = "corrupt aux data";
            return false;
        }
        const bool kin = (f.tag[0] == 'f' || f.tag[0] == 'r') && (f.tag[1] == 'i' || f.tag[1] == 'p');
        const bool oldmod = f.tag[0] == 'M' && (f.tag[1] == 'M' || f.tag[1] == 'L');
        if ((kin && !keep_kinetics) || oldmod) continue;
        if (f.tag[0] == 'M' && f.tag[1] == 'N' && n > 0 && !mn_done) {
            const char t = (char)start[2];
            const int old = (t == 'c' || t == 'C') ? 1 : (t == 's' || t == 'S') ? 2 : (t == 'i' || t == 'I') ? 4 : 0;
            if (old == 0) {
                err = "existing MN tag is not of an integer type";
                return false;
            }
            put_mn(old);
            mn_done = true;
            continue;
        }
        aux.insert(aux.end(), start, start + f.total);
    }
    if (n > 0) {
        // 2. MM:Z  "C+m" {",delta"} ";" "G-m" {",delta"} ";"   delta = number of skipped C (G) on the forward
        //    strand since the previous call (build_mod_bam.cpp:134-168)
        std::string mm;
        size_t nf = 0;
        while (nf < n && calls[nf].strand == 0) ++nf;
        for (int strand = 0; strand < 2; ++strand) {
            mm += strand == 0 ? "C+m" : "G-m";
            const char base = strand == 0 ? 'C' : 'G';
            const hm_call_t* c = strand == 0 ? calls : calls + nf;
            const size_t m = strand == 0 ? nf : n - nf;
            int last = 0;
            for (size_t i = 0; i < m; ++i) {
                if (c[i].strand != strand || (i + 1 < m && c[i].qoff >= c[i + 1].qoff) || c[i].qoff < last ||
                    c[i].qoff >= r.l_qseq() || fwd_strand_base(r, c[i].qoff) != base) {
                    err = "calls are not strictly increasing per strand or do not sit on C/G";
                    return false;
                }
                int delta = 0;
                for (int k = last; k < c[i].qoff; ++k) delta += fwd_strand_base(r, k) == base;
                mm += ',';
                mm += std::to_string(delta);
                last = c[i].qoff + 1;
            }
            mm += ';';
        }
        aux.insert(aux.end(), {'M', 'M', 'Z'});
        aux.insert(aux.end(), mm.begin(), mm.end());
        aux.push_back(0);
        // 3. ML:B:C  forward-strand probabilities then reverse-strand ones (build_mod_bam.cpp:170-176,200)
        aux.insert(aux.end(), {'M', 'L', 'B', 'C'});
        uint8_t cnt[4];
        wr32(cnt, (uint32_t)n);
        aux.insert(aux.end(), cnt, cnt + 4);
        for (size_t i = 0; i < n; ++i) aux.push_back(calls[i].scaled_prob);
        // 4. MN (see above)
        if (!mn_done) put_mn(0);
    }
    r.data.resize(aux0);
    r.data.insert(r.data.end(), aux.begin(), aux.end());
    return true;
}

// ------------------------------------------------------------------------------------------------
// MM/ML parser, contexts, thresholds
// ------------------------------------------------------------------------------------------------
static char chebi_to_code(long c) {  // bam_mod_parser.cpp:36-77
    switch (c) {
    case 27551: return 'm';
    case 76792: return 'h';
    case 76794: return 'f';
    case 76793: return 'c';
    case 16964: return 'g';
    case 80961: return 'e';
    case 17477: return 'b';
    case 28871: return 'a';
    case 44605: return 'o';
    case 18107: return 'n';
    default: return 0;
    }
}

bool parse_mods(const BamRecord& r, std::vector<BaseMod>& mods, std::string& err) {
    mods.clear();
    const uint8_t* p = r.data.data() + r.aux_offset();
    const uint8_t* end = r.data.data() + r.data.size();
    AuxField f, mm{}, ml{};
    bool has_mm = false, has_ml = false;
    while (p < end) {
        if (!next_aux(p, end, f)) { err = "corrupt aux data"; return false; }
        if (f.tag[0] == 'M' && f.tag[1] == 'M' && !has_mm) { mm = f; has_mm = true; }
        if (f.tag[0] == 'M' && f.tag[1] == 'L' && !has_ml) { ml = f; has_ml = true; }
    }
    if (!has_ml || !has_mm) return true;  // nothing to parse (bam_mod_parser.cpp:250-256)
    if (ml.type != 'B' || mm.type != 'Z') { err = "MM must be a Z tag and ML a B array"; return false; }
    std::vector<uint8_t> probs(ml.count);
    const int es = aux_elem_size(ml.subtype);
    for (uint32_t i = 0; i < ml.count; ++i) {
        int64_t v = 0;
        const uint8_t* q = ml.payload + (size_t)i * es;
        switch (ml.subtype) {
        case 'C': v = q[0]; break;
        case 'c': v = (int8_t)q[0]; break;
        case 'S': v = rd16(q); break;
        case 's': v = (int16_t)rd16(q); break;
        case 'I': v = rd32(q); break;
        case 'i': v = (int32_t)rd32(q); break;
        default: err = "ML has a non-integer element type"; return false;
        }
        if (v < 0 || v > 255) { err = "Illegal scaled probability value, which must be in range [0, 255]"; return false; }
        probs[i] = (uint8_t)v;
    }
    if (probs.empty()) return true;
    const char* mms = reinterpret_cast<const char*>(mm.payload);
    const size_t mml = strlen(mms);
    if (mml == 0 || mms[mml - 1] != ';') { err = "The MM aux tag must end with ';'"; return false; }
    const int L = r.l_qseq();
    // forward-strand sequence, decoded once (get_bam_fwd_strand_base for every offset, bam_info.cpp:224-233)
    thread_local std::string fwd;
    fwd.resize((size_t)L);
    {
        const uint8_t* s4 = r.seq4();
        static const char dec[17] = "NACNGNNNTNNNNNNN", cmp[17] = "NTGNCNNNANNNNNNN";
        if (r.flag() & 16)
            for (int k = 0; k < L; ++k) {
                const int i = L - 1 - k;
                fwd[(size_t)k] = cmp[(s4[i >> 1] >> ((~i & 1) << 2)) & 15];
            }
        else
            for (int k = 0; k < L; ++k) fwd[(size_t)k] = dec[(s4[k >> 1] >> ((~k & 1) << 2)) & 15];
    }
    size_t pi = 0;
    for (size_t i = 0; i < mml;) {
        const char* sc = static_cast<const char*>(memchr(mms + i, ';', mml - i));
        const size_t j = (size_t)(sc - mms);
        const char* s = mms + i;          // one edit series incl. ';'
        const size_t sl = j - i + 1;
        i = j + 1;
        auto series = [&]() { return std::string(s, sl); };
        if (sl < 4) { err = "Corrupted edit series " + series(); return false; }
        const char ub = s[0];
        if (!strchr("CGTAUN", ub) || (s[1] != '+' && s[1] != '-')) { err = "Unrecognised base or strand in edit series " + series(); return false; }
        const uint8_t strand = s[1] == '+' ? 0 : 1;
        char codes[16];
        int n_codes = 0;
        size_t si = 2;
        if (isdigit((unsigned char)s[2])) {
            long c = 0;
            while (si < sl && isdigit((unsigned char)s[si])) c = c * 10 + (s[si++] - '0');
            const char code = chebi_to_code(c);
            if (!code) { err = "Unrecognised ChEBI code in edit series " + series(); return false; }
            codes[n_codes++] = code;
        } else {
            for (; si < sl && s[si] != ',' && s[si] != ';'; ++si)
                if (s[si] != '.' && s[si] != '?') {
                    if (n_codes == 16) { err = "Too many modification codes in edit series " + series(); return false; }
                    codes[n_codes++] = s[si];
                }
        }
        int qoff = 0;
        while (si < sl && s[si] != ';') {  // bam_mod_parser.cpp:184-229, one delta at a time
            if (s[si] != ',') { err = "Illegal character in edit series " + series(); return false; }
            ++si;
            if (si >= sl || !isdigit((unsigned char)s[si])) { err = "Illegal character in edit series " + series(); return false; }
            long d = 0;
            while (si < sl && isdigit((unsigned char)s[si])) d = d * 10 + (s[si++] - '0');
            long cnt = 0;
            while (cnt < d) {
                if (qoff >= L) { err = "edit series runs past the read end: " + series(); return false; }
                if (fwd[(size_t)qoff] == ub) ++cnt;
                ++qoff;
            }
            while (qoff < L && fwd[(size_t)qoff] != ub) ++qoff;
            if (qoff >= L) { err = "edit series runs past the read end: " + series(); return false; }
            for (int c = 0; c < n_codes; ++c) {
                if (pi >= probs.size()) { err = "ML is shorter than the MM edit lists"; return false; }
                mods.push_back(BaseMod{qoff, strand, ub, codes[c], probs[pi++]});
            }
            ++qoff;
        }
    }
    return true;
}

int mod_context(const BamRecord& r, int q) {
    const int L = r.l_qseq();
    auto b = [&](int k) { return (k >= 0 && k < L) ? fwd_strand_base(r, k) : 'N'; };
    auto H = [](char c) { return c == 'A' || c == 'C' || c == 'T'; };
    auto D = [](char c) { return c == 'A' || c == 'G' || c == 'T'; };
    const char c0 = b(q);
    if (c0 == 'C') {
        if (b(q + 1) == 'G') return 0;
        if (q + 2 < L && H(b(q + 1)) && b(q + 2) == 'G') return 1;
        if (q + 2 < L && H(b(q + 1)) && H(b(q + 2))) return 2;
        return -1;
    }
    if (c0 == 'G') return (q - 2 >= 0 && D(b(q - 1)) && D(b(q - 2))) ? 2 : -1;
    return -1;
}

int resolve_threshold(const uint64_t* a, uint64_t* samples) {
    uint64_t sum = 0, min_cnt = ~uint64_t(0);
    int min_i = -1, st = 20, en = 256 - 20;
    while (st < 256 && a[st] < 10) ++st;
    while (en && a[en - 1] < 10) --en;
    if (en - st >= 50)
        for (int i = st; i < en; ++i) {
            sum += a[i];
            if (min_cnt > a[i]) { min_cnt = a[i]; min_i = i; }
        }
    if (samples) *samples = sum;
    return (sum < 10000 || min_i == -1) ? 128 : min_i;
}

int Fasta::find(const std::string& name) const {
    for (size_t i = 0; i < names.size(); ++i)
        if (names[i] == name) return (int)i;
    return -1;
}

bool load_fasta(const std::string& path, Fasta& fa, std::string& err) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    gzbuffer(f, 1 << 20);
    fa = Fasta();
    std::vector<char> buf(1 << 22);
    std::string line;
    bool in_seq = false;  // a header has been seen
    auto flush_line = [&]() {
        size_t a = 0, b = line.size();
        while (a < b && isspace((unsigned char)line[a])) ++a;
        while (b > a && isspace((unsigned char)line[b - 1])) --b;
        if (a == b) return;
        const char c = line[a];
        if (c == '!' || c == '#' || c == ';') return;
        bool is_id = false;
        for (size_t i = a; i < b && i - a <= 32; ++i)
            if (isdigit((unsigned char)line[i]) || line[i] == '|') { is_id = true; break; }
        if (is_id || c == '>') {
            size_t s = a + (c == '>' ? 1 : 0), e = s;
            while (e < b && !isspace((unsigned char)line[e])) ++e;
            if (e == s) { in_seq = false; return; }  // the reference drops sequences with an empty name
            fa.names.push_back(line.substr(s, e - s));
            fa.length.push_back(0);
            in_seq = true;
        } else if (in_seq) {
            const size_t at = fa.bases.size();
            fa.bases.resize(at + (b - a));
            for (size_t i = a; i < b; ++i) {
                const char ch = line[i];
                fa.bases[at + (i - a)] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
            }
            fa.length.back() += (int64_t)(b - a);
        }
    };
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
        int s = 0;
        for (int i = 0; i < n; ++i)
            if (buf[i] == '\n') {
                line.append(buf.data() + s, i - s);
                flush_line();
                line.clear();
                s = i + 1;
            }
        line.append(buf.data() + s, n - s);
    }
    if (n < 0) { err = "read error in " + path; gzclose(f); return false; }
    flush_line();
    gzclose(f);
    std::unordered_set<std::string> seen;
    for (const std::string& nm : fa.names)
        if (!seen.insert(nm).second) { err = "Duplicate sequence name " + nm; return false; }
    return true;
}

}  // namespace hmbam
