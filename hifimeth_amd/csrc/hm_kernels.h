// hm_kernels.h -- host-callable launchers of the gfx950 kernels (defined in hm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "hm_device.h"

namespace hm {

// feature extraction -------------------------------------------------------------------------
// A: decode 4-bit bases to forward-strand codes, pack the four kinetics arrays per forward
//    position, count sites per 1024-base chunk.
//    sctx[j] = context of the site at forward position j (CTX_NONE = 3: no site, or its context is masked out) | strand << 2
//    (strand 1: the base is a G, the cytosine sits on the reverse strand): a site of context c in strand view v reads c | v << 2.
void launch_prep(hipStream_t st, const uint8_t* raw, const ReadDesc* reads, const Chunk* chunks, int n_chunks,
                 int ctx_mask, uint8_t* bases, uint32_t* kin, uint8_t* sctx, int32_t* chunk_counts, int32_t* err);
// S: exclusive scan of the NCNT chunk counters -> chunk offsets [n_chunks + 1][NCNT] (last row = totals),
//    totals[12]: CpG, CHG, CHH, all, ctx_base[3], reverse-strand sites; [8..10] zeroed (the sliding-window trunk counts its listed-row steps there).
void launch_scan(hipStream_t st, const int32_t* chunk_counts, int n_chunks, int32_t* chunk_offs, int32_t* totals);
// B: emit the unified (read, qoff)-ordered site list, the per-context lists and opos[uidx] = position of the site's
//    call in the output order (per read: forward-strand calls by qoff, then reverse-strand calls; mod_main.cpp:217-251).
void launch_emit(hipStream_t st, const ReadDesc* reads, const Chunk* chunks, int n_chunks, int ctx_mask,
                 const uint8_t* bases, const int32_t* chunk_offs, const int32_t* totals, USite* usites,
                 uint8_t* utag, Site* csites, int32_t* opos);
// P: results -> packed hm_call_t records in output order (one D2H then carries a batch's results).
void launch_pack(hipStream_t st, const USite* usites, const uint8_t* utag, const int32_t* opos, const float* prob,
                 const uint8_t* ml, const ReadDesc* reads, const int32_t* totals, void* calls, int grid);
// W: materialise raw (no bn0) 401x8 fp32 windows for a site list: out[n][401][8].
void launch_windows(hipStream_t st, const Site* sites, int n, const ReadDesc* reads, const uint8_t* bases,
                    const uint32_t* kin, const BnTables* bn, float* out, int grid);

// CNN ------------------------------------------------------------------------------------------
// front: window (from staged reads, or from materialised windows when `windows` != nullptr)
//        -> bn0 -> conv1..conv4 -> act4[n][25][96]
//        The sites of a launch are named by a SiteRange (hm_device.h): counts stay on the device.
void launch_front(hipStream_t st, int k1, const SiteRange& sr, const ReadDesc* reads, const uint8_t* bases,
                  const uint32_t* kin, const float* windows, const CtxWeights& w, float* act4, int grid,
                  float* dbg, int dbg_layer, int waves, unsigned long long* stamps);
int front_stamp_slots();
// split-half (f16x3) variant of the front kernel: fp16 hi/lo operands on v_mfma_f32_16x16x32_f16, fp32 accumulate
void launch_front_h(hipStream_t st, int k1, const SiteRange& sr, const ReadDesc* reads, const uint8_t* bases,
                    const uint32_t* kin, const float* windows, const CtxWeights& w, float* act4, int grid, float* dbg,
                    int dbg_layer, unsigned long long* stamps, bool w16);
void launch_tail_h(hipStream_t st, const float* act4, const SiteRange& sr, const CtxWeights& w, float* logits,
                   float* p, uint8_t* ml, int grid, float* dbg, int dbg_layer, int w16_level);
// tail: conv5..conv8, fc1, fc2, softmax for 8 sites per workgroup pass.
// results go to index sites[i].uidx (or i when sites == nullptr).
void launch_tail(hipStream_t st, const float* act4, const SiteRange& sr, const CtxWeights& w, float* logits,
                 float* p, uint8_t* ml, int grid, float* dbg, int dbg_layer);

#ifdef __HIPCC__
// sites and count of a launch, resolved on the device (see SiteRange)
__device__ __forceinline__ int resolve_sites(const SiteRange& sr, const Site*& sites) {
    if (!sr.totals) {
        sites = sr.base;
        return sr.cap;
    }
    const int first = sr.lo ? *sr.lo : 0, last = sr.hi ? *sr.hi : sr.totals[sr.ctx];
    int n = last - first - sr.off;
    n = n < 0 ? 0 : (n > sr.cap ? sr.cap : n);
    sites = sr.base + sr.totals[4 + sr.ctx] + first + sr.off;
    return n;
}
#endif

// dense trunk path (hm_trunk.hip): conv1..conv4 once per (read, strand view) position, the two window-edge rows of
// conv4 per site, then the tail gathers its conv4 rows from the maps
void launch_trunk(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                  const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, const CtxWeights& w,
                  const TrunkMaps& maps, int grid, bool w16);
void launch_edge(hipStream_t st, int k1, const SiteRange& sr, const RInfo* rinfo, const uint8_t* bases,
                 const uint32_t* kin, const CtxWeights& w, const TrunkMaps& maps, uint16_t* edge4, int32_t* e4row, int grid,
                 bool w16);
// the same edge rows with taps read in place, zero taps skipped and map rows / weights streamed a layer ahead (hm_edge2.hip):
// bit-identical results
void launch_edge2(hipStream_t st, int k1, const SiteRange& sr, const RInfo* rinfo, const uint8_t* bases, const uint32_t* kin,
                  const CtxWeights& w, const TrunkMaps& maps, uint16_t* edge4, int32_t* e4row, int grid);
void launch_tail_gather(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* edge4,
                        const int32_t* e4row, float* logits, float* p, uint8_t* ml, int grid, int w16_level);
// the same tail with conv5..conv7's weights resident in registers (hm_tail_r.hip): bit-identical results
void launch_tail_gather_r(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* edge4,
                          const int32_t* e4row, float* logits, float* p, uint8_t* ml, int grid, bool w16 = false);   // w16: conv8 + fc1 with plain fp16 weights (precision 2)
// the tail as two kernels (hm_tail_s.hip): conv5 + conv6 (8 sites per pass, conv6's rows to `x6`), then conv7 .. softmax (16 sites per
// pass, all weights resident): bit-identical results.  `x6`: tail_split_x6_bytes(max sites of a launch) bytes, zeroed once;
// `x6_plane_halves` = tail_split_x6_plane_halves(that same maximum) for every launch into it.
size_t tail_split_x6_plane_halves(int64_t sites);
size_t tail_split_x6_bytes(int64_t sites);
void launch_tail_split(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const uint16_t* edge4,
                       const int32_t* e4row, uint16_t* x6, size_t x6_plane_halves, float* logits, float* p, uint8_t* ml, int grid);
// the STRIP tail (hm_tail_p.hip) for a dense context (CHH): the launch's sites visited in (first map row mod 16, first map row) order,
// 16 sites of one residue class per pass sharing one strip of TAILP_STRIP lattice rows in LDS; bit-identical results.
// `mark`: tail_strip_mark_bytes(view rows x views), `cnt`: tail_strip_count_bytes(same), `order` / `okey`: one int32 per site of the launch.
size_t tail_strip_mark_bytes(int64_t map_rows);
size_t tail_strip_count_bytes(int64_t map_rows);
void launch_tail_strip(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, int n_views, const uint16_t* edge4,
                       const int32_t* e4row, int32_t* mark, int32_t* cnt, int32_t* order, int32_t* okey, int32_t* odst, uint16_t* x8, float* logits, float* p,
                       uint8_t* ml, int32_t* pass_count, int grid, bool w16);   // pass_count (may be null): += the launch's passes of 16 site slots;
                                                                               // w16: precision 2 (plain fp16 weights in conv8 and fc1)
// fc1, fc2, softmax over conv8's rows (x8: TAIL_X8_HALVES per site, in list order; dst: list position -> slot in logits / p / ml) -- hm_tail_fc.hip
size_t tail_fc_x8_bytes(int64_t sites);
size_t tail_strip_handover_bytes(int64_t sites);
void launch_tail_fc(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const uint16_t* x8, const int32_t* dst, float* logits, float* p,
                    uint8_t* ml, int grid, bool w16);
void launch_trunk2(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                   const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, const CtxWeights& w,
                   const TrunkMaps& maps, int grid, bool w16, bool waves8 = false);
// the trunk as a sliding window (hm_trunk3.hip): a workgroup walks consecutive tiles and keeps every layer's right-hand rows, so that
// each layer computes 112 rows per tile instead of 144 / 144 / 128 / 112: byte-identical maps, 14 % fewer MFMAs.  `maps.rowlist` must
// hold trunk3_rowlist_bytes(n_tiles * n_views) bytes, `dump` trunk3_dump_bytes(grid); every read's map region needs 32 rows of slack at its end.
// `tcost`: the running cost of the tiles (n_tiles + 1 entries from 0; hm_engine.cpp add_read_tiles) the workgroups' runs are cut by, or null (equal counts).
size_t trunk3_rowlist_bytes(int64_t n_work);
size_t trunk3_dump_bytes(int grid);
void launch_trunk3(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                   const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, int64_t n_bases, const CtxWeights& w, const TrunkMaps& maps,
                   uint16_t* dump, int32_t* list_steps, const int32_t* tcost, int grid, bool w3_single = false);   // n_bases: bytes of sctx
// (w3_single: conv3 with plain fp16 weights -- its w_lo x_hi product dropped -- engine option precision = 2)
// the same path in strict fp32 (precision 0; hm_trunk_f32.hip): fp32 maps and edge rows, v_mfma_f32_16x16x4_f32
void launch_trunk_f32(hipStream_t st, int k1, const TrunkTile* tiles, int n_tiles, int n_views, int ctx, const RInfo* rinfo,
                      const uint8_t* bases, const uint32_t* kin, const uint8_t* sctx, const CtxWeights& w,
                      const TrunkMaps& maps, int grid);
void launch_edge_f32(hipStream_t st, int k1, const SiteRange& sr, const RInfo* rinfo, const uint8_t* bases, const uint32_t* kin,
                     const CtxWeights& w, const TrunkMaps& maps, float* edge4, int32_t* e4row, int grid);
void launch_tail_gather_f32(hipStream_t st, const SiteRange& sr, const CtxWeights& w, const TrunkMaps& maps, const float* edge4,
                            const int32_t* e4row, float* logits, float* p, uint8_t* ml, int grid);
size_t trunk_lds_bytes();

size_t front_lds_bytes(int k1);
size_t tail_lds_bytes();

}  // namespace hm
