// hm_bam.h -- minimal BGZF + BAM record codec and the MM/ML/MN tag writer for the `call` front end.
// htslib is not available in this image, so the container formats are implemented directly on zlib:
//   BGZF : SAMv1 section 4.1 (gzip members with a 'BC' extra field, <= 64 KiB each, 28-byte EOF block)
//   BAM  : SAMv1 section 4.2 (header, records, aux fields)
// What the reference does with htslib at this boundary: src/corelib/sam_batch.hpp:12-54 (reader, 8 threads),
// src/app/hifimeth/mod_main.cpp:316-362 (writer), src/corelib/build_mod_bam.cpp:87-248 (tags).
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/hifimeth_hip.h"

namespace hmbam {

// run f(0..n-1) on up to `threads` host threads (work-stealing counter); used for BGZF blocks and per-read tagging
void parallel_run(int n, int threads, const std::function<void(int)>& f);

// BGZF reader.  The file is consumed in chunks of up to 512 blocks: a background thread reads the next chunk's blocks and
// inflates them on `threads` host threads -- each block straight into its final place, the places being known from the
// blocks' ISIZE fields -- while the caller parses the current chunk, so inflate overlaps record parsing and staging.
class BgzfReader {
public:
    BgzfReader(const std::string& path, int threads);
    ~BgzfReader();
    bool ok() const { return fp_ != nullptr && err_.empty(); }
    const std::string& error() const { return err_; }
    // read exactly n bytes; returns false at clean EOF (0 bytes available) or on error (error() set)
    bool read(void* dst, size_t n);
    // ---- reading a byte range of the file (one rank of a read-sharded job inflates only its own blocks) ----
    // continue at the BGZF block that starts at compressed offset `file_offset` (buffered data is dropped)
    bool seek_block(int64_t file_offset);
    // compressed offset of the (non-empty) block that holds the next unread byte; the file size once everything is consumed
    int64_t block_offset();
    // make up to `want` unread bytes available without consuming them; returns how many there are (fewer at EOF)
    size_t peek(const uint8_t*& p, size_t want);
    void skip(size_t n);  // consume n bytes that peek() has shown

private:
    struct Chunk {
        std::vector<uint8_t> data;
        std::vector<std::pair<size_t, int64_t>> segs;  // (first byte in data, compressed offset) of every non-empty block
        int64_t next_off = 0;                          // compressed offset behind the chunk's last block
        bool eof = false;
        std::string err;
    };
    void load(Chunk& c, int64_t from);  // blocks starting at compressed offset `from` -> c (runs on the background thread)
    void start_prefetch();
    bool advance(bool append);          // next chunk becomes (or is appended to) the current one
    FILE* fp_ = nullptr;
    int threads_;
    Chunk cur_, nxt_;
    size_t pos_ = 0;
    void* bg_ = nullptr;  // std::thread of the prefetch in flight
    std::string err_;
};

// compressed offsets of all BGZF blocks of a file (header hop, nothing is inflated) + the file size
bool scan_bgzf_blocks(const std::string& path, std::vector<int64_t>& offsets, int64_t& file_size, std::string& err);

// Finds the first BAM record that starts at or after the reader's current position (a block boundary somewhere inside
// the file) and consumes the bytes in front of it.  BAM has no sync marker: candidates are validated like Hadoop-BAM's
// split guesser does -- fixed fields in range, printable NUL-terminated name, sizes that add up, aux fields that walk
// exactly to the record's end -- and must be followed by further records that validate the same way.
// Returns false with err empty if there is no record start before EOF.
bool find_record_start(BgzfReader& in, int n_ref, std::string& err);

class BgzfWriter {
public:
    BgzfWriter(const std::string& path, int threads, int level);
    ~BgzfWriter();
    bool ok() const { return fp_ != nullptr && err_.empty(); }
    const std::string& error() const { return err_; }
    void write(const void* src, size_t n);
    bool close();  // flushes and appends the EOF marker block

private:
    void flush_blocks(bool all);
    FILE* fp_ = nullptr;
    int threads_, level_;
    std::vector<uint8_t> buf_;
    std::string err_;
};

struct BamHeader {
    std::string text;
    std::vector<std::pair<std::string, int32_t>> refs;
};

// One BAM record: `data` holds everything after the 4-byte block_size.
struct BamRecord {
    std::vector<uint8_t> data;
    int32_t l_qseq() const;
    uint16_t flag() const;
    int l_read_name() const { return data[8]; }
    int n_cigar() const;
    int32_t ref_id() const;
    int32_t pos() const;
    int mapq() const { return data[9]; }
    const uint8_t* cigar_bytes() const { return data.data() + 32 + l_read_name(); }  // n_cigar little-endian u32
    const uint8_t* seq4() const;
    size_t aux_offset() const;
};

struct AuxField {
    char tag[2];
    char type;
    char subtype;           // for 'B'
    uint32_t count;         // for 'B'
    const uint8_t* payload; // first data byte ('B': first element)
    size_t total;           // bytes of the whole field incl. tag and type
};

// walks the aux block; returns false (and stops) on a corrupt field
bool next_aux(const uint8_t*& p, const uint8_t* end, AuxField& f);

bool read_header(BgzfReader& in, BamHeader& h, std::string& err);
// libdeflate (dlopen'ed when the system has it) inflates and checksums BGZF blocks; the writer deflates with zlib unless asked otherwise
// (the compressed bytes differ between the two libraries, the records do not)
bool bam_have_libdeflate();
void bam_use_libdeflate_compress(bool on);
void write_header(BgzfWriter& out, const BamHeader& h);
bool read_record(BgzfReader& in, BamRecord& r, std::string& err);
void write_record(BgzfWriter& out, const BamRecord& r);

// Kinetics views of a record for hm_submit_read: NULL when the tag is missing, not a B:C / B:S array or has the
// wrong length (BamKinetics::init, bam_info.cpp:443-453,572-603).
struct KineticsView {
    const void* arr[4];  // fi, fp, ri, rp
    int width[4];
};
KineticsView kinetics_of(const BamRecord& r);

// forward-strand base of forward position k (BamQuerySequence::get_bam_fwd_strand_base, bam_info.cpp:224-233)
char fwd_strand_base(const BamRecord& r, int k);

// build_one_mod_bam (build_mod_bam.cpp:125-248): strips fi/ri/fp/rp (unless keep_kinetics) and any old MM/ML,
// then -- if there is at least one call -- appends MM:Z, ML:B:C and MN.  `calls` are this read's calls in the
// order hm_fetch returns them: FWD strand ascending qoff, then REV strand ascending qoff.
// Returns false if a call does not sit on a C (FWD) / G (REV) or the per-strand order is not strictly increasing
// (the reference hbn_assert()s these).
bool apply_calls(BamRecord& r, const hm_call_t* calls, size_t n, bool keep_kinetics, std::string& err);

// ---- MM/ML parser (inverse of apply_calls) and the alignment-free part of `pileup` -------------------------
// extract_bam_base_mods (src/corelib/bam_mod_parser.cpp:231-286): every (position, code) of the MM lists with its
// ML probability, positions in forward-strand coordinates.
using BaseMod = hm_mod_t;  // {qoff, strand (0 '+', 1 '-'), unmod_base, code, prob}
bool parse_mods(const BamRecord& r, std::vector<BaseMod>& mods, std::string& err);

// context of a 5mC call from the read sequence, as pileup does it for its histograms
// (src/app/hifimeth/pileup.cpp:237-272): 0 CpG, 1 CHG, 2 CHH, -1 none
int mod_context(const BamRecord& r, int qoff);

// s_resolve_scaled_prob_threshold (pileup.cpp:355-436) for one 256-bin histogram; `samples` = the sum it reports
int resolve_threshold(const uint64_t* bins, uint64_t* samples);

// ---- reference genome (HbnDatabase, src/corelib/hbn_seqdb.cpp:37-95) -------------------------------------------
// plain or gzip FASTA; bases upper-cased; a line is a header when it starts with '>' or holds a digit or '|' among
// its first 33 characters (s_IsSeqID); lines starting with ! # ; are skipped; the name ends at the first blank.
struct Fasta {
    std::vector<std::string> names;
    std::vector<int64_t> length;
    std::string bases;  // all sequences back to back
    int find(const std::string& name) const;  // -1 if absent
};
bool load_fasta(const std::string& path, Fasta& fa, std::string& err);

}  // namespace hmbam
