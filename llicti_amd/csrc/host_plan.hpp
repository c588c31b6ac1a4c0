// host_plan.hpp -- the HOST logic of the whole-batch calls: container tags, the layout of a call (workspace carving, per-image geometry
// tables, stream descriptors, tile lists of the band CNN), size bounds, header parsing.  Plain C++17 with no HIP dependency -- g++
// compiles it, with host_types.hpp and cnn_pack.hpp, under AddressSanitizer / UBSan and drives it over every (B, sizes, mode) the tests
// use plus malformed input (tests/sanitize_host.sh, tests/sanitize_host.cpp).  In the HIP build it is part of llicti_hip.hip.
#pragma once
#include <algorithm>
#include <vector>

#include "cnn_pack.hpp"
#include "host_types.hpp"

// One whole-batch call's layout: the workspace carving and the per-image tables the kernels read.  The images of a call may differ in
// size (llicti_encode_images_v / llicti_decode_images_v); llicti_encode_images / llicti_decode_images are the same code with B equal sizes.
struct TileRun { size_t off = 0; int n_tiles = 0, TH = 0, gx = 0; };      // band-CNN launch of one (level, band) of a mixed-size plan: its tile list
struct Plan {
    int B = 0, ME = 0;
    int M = 0;                          // rANS streams per image (0: AC container only); images of a call may differ: the largest count
    int nstreams = 0;                   // ... and the streams of all images together (sref)
    std::vector<StreamRef> sref;        // [nstreams]: stream -> (image, stream of the image, its count, its first stream)
    int Q = 1;                          // 64-lane sub-chunks per stream step (2: wide streams of 128 lanes; 4: xwide streams of 256 lanes)
    bool uniform = true;                // every image has the size of image 0: the band CNN runs its division form, the AC container is available
    bool vec_ok = true;                 // every image's plane size and placement allow the lift's 4-pixel accesses
    long lev_maxpos[LLICTI_NLEVELS];    // largest band grid (h * w) of a level
    std::vector<long> key;              // (ME, B, tile-form tuning, H, W, rgb offset per image): what the cache compares
    size_t off_lift_part, off_acstate;
    long ac_cap_rows = 0;               // rows per image of one colour's chunk table buffer (AC decode)
    size_t off_planes, off_fplanes, off_minmax, off_status, off_params, off_params2, off_pairs, off_slots, off_slot_len, off_tables;
    size_t total = 0;
    size_t rgb_bytes = 0;               // extent of the caller's RGB buffer
    long max_plane = 0;                 // largest H * W of the batch
    std::vector<ImgGeo> img;            // [B]
    std::vector<Geom> geo;              // [level][B]
    std::vector<StageGeom> sg;          // [level * 3 + band][B]
    size_t lev_floats[LLICTI_NLEVELS];  // CNN outputs of one (level, band): 64 floats per band-grid position of every image
    std::vector<StreamDesc> desc;       // stage-major, image-minor: index (stage * B + b)
    std::vector<long> slot_off;         // AC container (uniform plans)
    std::vector<int32_t> slot_cap;
    long pair_base[LLICTI_NLEVELS * 3]; // per (lvl, band): first pair of [clr][image][n]
    size_t max_container = 0;           // of the batch's largest image
    int rslot_cap = 0;
    std::vector<long> rslot_off;        // [B*M] byte offsets into the slots region
    size_t off_rinfo, off_rstate, off_rpos, off_rtail;
    std::vector<TileRef> tiles;            // mixed-size plans: the tile lists of the 15 band-CNN launches, back to back
    TileRun run[LLICTI_NLEVELS * 3];
    // device copies (one block, see PlanBlock)
    size_t d_img = 0, d_geo = 0, d_sg = 0, d_desc = 0, d_slot_off = 0, d_slot_cap = 0, d_rslot_off = 0, d_tiles = 0, d_sref = 0, d_total = 0;
};

// AC decode has two table forms.  Few images in flight (latency bound: every stream is one serial wave and the GPU is
// mostly idle): FULL rows from cdf_table_kernel, because the search over a ready-made row is the shortest instruction
// sequence on the serial wave (B = 24: 249 ms against 293 ms).  Many images (the SIMDs' issue slots are the bound): ANCHOR
// rows -- 1/8 of the erfc work and a fifth of the HBM traffic, the bucket's 8 entries evaluated by the decoding wave
// (B = 256: 475 ms against 641 ms).  The workspace is sized for full rows below kAcAnchorBatch images and for anchor rows
// from there on; llicti_set_tuning("ac_anchor_min_batch") can only LOWER the switch point (tests run both forms).
constexpr int kAcAnchorBatch = 96;
static bool ac_use_anchors(int B, int min_batch = kAcAnchorBatch) { return B >= std::min(min_batch, kAcAnchorBatch); }

constexpr int kRansMaxStreams = 128;   // rANS streams per image: <= 32 one per segment, 64 / 128 grouped (rans_group())
constexpr int kStatusHead = 16;       // status words in front of the per-image ones (common.hpp: image_status())
// AC decode: a stage of nc symbols per stream is cut into C chunks (multiples of 64 symbols) so that the Y, Co and Cg
// streams of a band run as a three-deep pipeline on three HIP streams (see decode_batch)
static int ac_chunks(long nc) { return nc >= 32768 ? 16 : nc >= 4096 ? 8 : nc >= 1024 ? 4 : nc >= 256 ? 2 : 1; }
static long ac_chunk_rows(long nc) { const int C = ac_chunks(nc); return ((nc + C - 1) / C + 63) / 64 * 64; }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int stage_index(int lvl, int band, int clr) { return (LLICTI_NLEVELS - 1 - lvl) * 9 + band * 3 + clr; }   // scale 4..0

static int pad_int(int H, int W)
{
    int v = 0;
    for (int l = 0; l < LLICTI_NLEVELS; ++l) {
        Geom g = make_geom(1, H, W, l);
        v = 4 * v + 2 * g.padH + g.padW;       // LLICTI_nets.py:230
    }
    return v;
}

// Tile height of a band-CNN launch (16, 8 or 4 rows; one wavefront per row, so 16 / 8 / 4 wavefronts per workgroup): the form whose launch is
// shortest under a two-parameter model of the persistent grid -- rounds = ceil(tiles / workgroups that fit the chip), a round = a fixed part
// (halo rows, staging the head's weights, barrier) + a part per tile row; the constants are the measured 46 / 25 / 15 us of a band-2
// tile of 16 / 8 / 4 rows.  Full launches come out at 16 rows; launches of one to three half-empty rounds (levels 3 and 4 of a batch
// of 24) at 8; launches that cannot give every compute unit a workgroup (coarse levels of a single image) at 4.  Every form computes
// each position with the same fmaf chains: the results do not depend on it (test_band_params_bitexact_and_golden runs all three).
// count(th) = tiles of the launch in tiles of th rows (all images).  tile_rows: llicti_set_tuning("cnn_tile_rows").
struct TileForm { int TH, gx; long n_tiles; };
template <class COUNT>
static TileForm choose_tile_form(int n_cu, int tile_rows, int band, COUNT &&count)
{
    auto plan_for = [&](int th, int *gx_out, long *tiles_out) -> double {
        const long tiles = count(th);
        const int per_cu = std::max(1, std::min(4, (160 * 1024) / cnn_lds_bytes(band, th)));
        const long gx = std::max<long>(1, std::min<long>(tiles, (long)n_cu * per_cu / 4));
        *gx_out = (int)gx; *tiles_out = tiles;
        return (double)((tiles + gx - 1) / gx) * (4.7 + 2.6 * th);
    };
    TileForm f{ kTileHMax, 1, 0 };
    if (tile_rows > 0) { f.TH = tile_rows; (void)plan_for(f.TH, &f.gx, &f.n_tiles); }
    else if (tile_rows < 0) {                              // round 3's rule (A/B): 4 rows iff the 16-row tiles cannot fill the chip
        f.TH = (4 * count(kTileHMax) < n_cu) ? kTileHSmall : kTileHMax;
        (void)plan_for(f.TH, &f.gx, &f.n_tiles);
    } else {
        // Launches of one or two rounds of 16-row tiles (coarse levels, single images) are priced from a table instead: a launch's first
        // round and its later ones per (band, form), measured on single-image launches (rocprofv3, tools/single_image_trace.py) -- the linear
        // model is off exactly there: a 4-row tile of band 2 takes 21-25 us, not 15 (one wavefront per SIMD cannot keep the matrix pipe busy and
        // 120 input channels are the longest layer 0), so three rounds of them lost to ONE round of 16-row tiles at level 1 of a lone image
        // (75 against 47 us).  Band 0 fits two workgroups per compute unit: its rounds are priced as shared.
        static const double kFirstUs[3][3] = { { 31.5, 18.1, 12.5 }, { 36.4, 20.8, 14.5 }, { 46.3, 26.2, 21.4 } };      // [band][16, 8, 4 rows]
        static const double kLaterUs[3][3] = { { 31.0, 17.0, 11.0 }, { 36.4, 19.0, 11.2 }, { 46.0, 23.9, 25.2 } };
        static const double kSharedUs[3] = { 61.3, 31.2, 16.7 };                                                            // band 0, two workgroups per CU
        int gx16; long nt16;
        (void)plan_for(kTileHMax, &gx16, &nt16);
        const bool small_launch = (nt16 + gx16 - 1) / gx16 <= 2;
        double best = 0;
        int fi = 0;
        for (int th : { kTileHMax, kTileHMid, kTileHSmall }) {
            int gx_t; long nt;
            double t = plan_for(th, &gx_t, &nt);
            if (small_launch) {
                const long rounds = (nt + gx_t - 1) / gx_t;
                const bool shared = 4L * gx_t > n_cu;                      // more workgroups (4 heads) than compute units
                t = shared ? 2.0 + rounds * kSharedUs[fi] : kFirstUs[band][fi] + (rounds - 1) * kLaterUs[band][fi];
            }
            if (th == kTileHMax || t < 0.995 * best) { best = t; f.TH = th; f.gx = gx_t; f.n_tiles = nt; }    // ties go to the larger form
            ++fi;
        }
    }
    return f;
}

// ME: streams per image, | 0x100 for wide (128-lane) streams, | 0x200 for xwide (256-lane) streams -- what mode_streams() returns.
// Hs, Ws: B sizes; rgb_off: B byte offsets of the images in the caller's RGB buffer, or nullptr = tightly packed in call order.
// n_cu, tile_rows: the band CNN's tile forms of a mixed-size plan are chosen (and its tile lists written) here.
static int rans_byte0(int M, int Q);
static int rans_pad_hi(int M, int Q);

// Ms: B stream counts (rANS containers: the images of a call may have different ones -- every header carries its own -- so that larger images
// get more streams and a stage launch does not wait for its largest image), or nullptr = ME's count for every image.
static void build_plan(Plan &p, int B, const int *Hs, const int *Ws, const size_t *rgb_off, int ME, int n_cu = 256, int tile_rows = 0, bool force_ragged = false,
                       const int *Ms = nullptr)
{
    const int Q = 1 << ((ME >> 8) & 3);
    const bool autoM = (ME & 0x1000) != 0;      // LLICTI_MODE_RANS_X_AUTO: the counts are the size rule's (Mlo); the encoder picks per image in [rans_auto_min, rans_auto_hi]
    int M = ME & 0xFF;
    if (Ms && M > 0) { M = 0; for (int b = 0; b < B; ++b) M = std::max(M, Ms[b]); }
    if (autoM) M = rans_auto_hi(M);
    p.B = B; p.ME = ME; p.M = M; p.Q = Q;
    p.uniform = !force_ragged;
    for (int b = 1; b < B; ++b) if (Hs[b] != Hs[0] || Ws[b] != Ws[0]) p.uniform = false;
    {
        long pos = 0;
        for (int b = 0; b < B; ++b) { if (rgb_off && (long)rgb_off[b] != pos) p.uniform = false; pos += 3L * Hs[b] * Ws[b]; }      // (the division form of the kernels assumes tightly packed images)
    }
    p.vec_ok = true;
    p.rgb_bytes = 0;
    p.key.clear();
    p.key.reserve(3 + 4 * (size_t)B);
    p.key.push_back(ME); p.key.push_back(B); p.key.push_back(tile_rows * 2 + (force_ragged ? 1 : 0));
    auto m_of = [&](int b) -> int { return (Ms && (ME & 0xFF) > 0) ? Ms[b] : (ME & 0xFF); };
    // images: sizes, header constants, placement (mixed sizes: planes / fplanes blocks start at multiples of 64 elements; equal sizes: tightly
    // packed, [B][3][H][W] -- what the division form of the band CNN and the AC container's kernels index)
    p.img.assign(B, ImgGeo{});
    long pix = 0, rgb_pos = 0;
    p.max_plane = 0;
    for (int b = 0; b < B; ++b) {
        ImgGeo &ig = p.img[b];
        ig.H = Hs[b]; ig.W = Ws[b];
        const Geom g4 = make_geom(1, ig.H, ig.W, 4);
        ig.h4 = g4.h; ig.w4 = g4.w; ig.hdr_bytes = 17 + 3 * g4.h * g4.w;
        ig.plane = (long)ig.H * ig.W;
        ig.pix_off = pix;
        pix += p.uniform ? 3 * ig.plane : (long)align_up((size_t)(3 * ig.plane), 64);
        ig.rgb_off = rgb_off ? (long)rgb_off[b] : rgb_pos;
        rgb_pos += 3 * ig.plane;
        if ((ig.plane & 3) || (ig.rgb_off & 3)) p.vec_ok = false;
        p.rgb_bytes = std::max(p.rgb_bytes, (size_t)(ig.rgb_off + 3 * ig.plane));
        p.max_plane = std::max(p.max_plane, ig.plane);
        ig.Mlo = autoM ? m_of(b) : 0;
        ig.M = autoM ? rans_auto_hi(ig.Mlo) : m_of(b);
        ig.byte0 = ig.M ? rans_byte0(ig.M, Q) : LLICTI_NLEVELS;
        ig.padint = pad_int(ig.H, ig.W) | ((ig.M ? rans_pad_hi(ig.M, Q) : 0) << 10);      // the header's int16 pad field (xwide v4: its high bits carry the stream count; an "auto" encode writes the count it picked)
        p.key.push_back(ig.H); p.key.push_back(ig.W); p.key.push_back(ig.rgb_off); p.key.push_back(m_of(b));
    }
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    p.off_status = take(kStatusHead * sizeof(int32_t) + (size_t)B * sizeof(int32_t));   // [0]: the call's status; [kStatusHead + b]: image b's
    p.off_minmax = take((size_t)B * 4 * sizeof(int32_t));
    p.off_lift_part = take((size_t)kLiftMaxParts * 4 * sizeof(int32_t));
    p.off_planes = take((size_t)pix * sizeof(int16_t));
    p.off_fplanes = take((size_t)pix * sizeof(float));
    // levels: geometry and the placement of every image's CNN outputs
    p.geo.assign((size_t)LLICTI_NLEVELS * B, Geom{});
    for (int lvl = 0; lvl < LLICTI_NLEVELS; ++lvl) {
        size_t fl = 0;
        p.lev_maxpos[lvl] = 0;
        for (int b = 0; b < B; ++b) {
            Geom g = make_geom(B, Hs[b], Ws[b], lvl);
            p.lev_maxpos[lvl] = std::max(p.lev_maxpos[lvl], (long)g.h * g.w);
            g.pix_off = p.img[b].pix_off;
            g.par_off = (long)fl;
            fl += (size_t)g.h * g.w * LLICTI_PARAM_STRIDE;
            p.geo[(size_t)lvl * B + b] = g;
        }
        p.lev_floats[lvl] = fl;
    }
    p.off_params = take(std::max(p.lev_floats[0], 3 * p.lev_floats[1]) * sizeof(float));
    // stages: pairs, streams, AC slots
    p.sg.assign((size_t)LLICTI_NLEVELS * 3 * B, StageGeom{});
    p.desc.assign((size_t)LLICTI_NSTREAMS * B, StreamDesc{});
    if (M == 0) { p.slot_off.assign((size_t)LLICTI_NSTREAMS * B, 0); p.slot_cap.assign((size_t)LLICTI_NSTREAMS * B, 0); }
    long pair_pos = 0, slot_pos = 0;
    std::vector<size_t> container(B);
    for (int b = 0; b < B; ++b) container[b] = (size_t)p.img[b].hdr_bytes;
    for (int lvl = LLICTI_NLEVELS - 1; lvl >= 0; --lvl) {
        for (int band = 0; band < 3; ++band) {
            StageGeom *sgr = &p.sg[(size_t)(lvl * 3 + band) * B];
            long cs = 0;
            for (int b = 0; b < B; ++b) {
                sgr[b] = make_stage(p.geo[(size_t)lvl * B + b], band);
                sgr[b].pair_off = cs;
                cs += (long)sgr[b].hc * sgr[b].wc;
            }
            for (int b = 0; b < B; ++b) sgr[b].pair_cs = cs;
            p.pair_base[lvl * 3 + band] = pair_pos;
            for (int clr = 0; clr < 3; ++clr) {
                const int st = stage_index(lvl, band, clr);
                for (int b = 0; b < B; ++b) {
                    const long nc = (long)sgr[b].hc * sgr[b].wc;
                    StreamDesc &d = p.desc[(size_t)st * B + b];
                    d.pair_off = pair_pos + (long)clr * cs + sgr[b].pair_off;
                    d.n = (int)nc;
                    if (M == 0) {
                        const int cap = (int)align_up((size_t)(2 * nc + 8 + 16), 16);   // <= 16 bits per symbol + termination + zero pad
                        d.out_off = slot_pos;
                        d.cap = cap - 16;
                        p.slot_off[(size_t)st * B + b] = slot_pos;
                        p.slot_cap[(size_t)st * B + b] = cap;
                        slot_pos += cap;
                    }
                    container[b] += (size_t)(2 * nc + 8);
                }
            }
            pair_pos += 3 * cs;
        }
    }
    p.max_container = 0;
    for (int b = 0; b < B; ++b) p.max_container = std::max(p.max_container, align_up(container[b] + 64 * 45, 16));
    p.off_pairs = take((size_t)pair_pos * sizeof(uint32_t));
    p.sref.clear();
    p.nstreams = 0;
    if (M > 0) {
        // worst case of one stream: every symbol emits 16 bits; chunks are dealt round-robin, so a
        // stream gets at most ceil(nchunks / M) chunks of every stage (one capacity for all: the largest any image's streams need)
        const int L = 64 * Q;
        const int pay_bytes = Q * RansGeo<1>::kPayBytes;
        p.rslot_cap = 0;
        for (int b = 0; b < B; ++b) {
            const int Mb = p.img[b].M;
            const int Mfew = p.img[b].Mlo ? rans_auto_min(p.img[b].Mlo) : Mb;      // the fewest streams the image may end up with: the longest ones
            long syms = 0, all_syms = 0;
            for (int st = 0; st < LLICTI_NSTREAMS; ++st) {
                const long n = p.desc[(size_t)st * B + b].n;
                const long nchunks = (n + L - 1) / L;
                syms += (nchunks + Mfew - 1) / Mfew * L;
                all_syms += (n + 63) / 64 * 64;
            }
            p.rslot_cap = std::max(p.rslot_cap, (int)align_up((size_t)(2 * syms + 4 + 8 + pay_bytes + 16 + 64 + (Q == 4 ? kRansSpillMax / 8 + 8 : 0)), 64));   // + T, the 31-bit states, slack, zero pad (xwide v4: + the tail's spill and the header field)
            // container bound: the streams together hold every symbol once (<= 16 bits each, whole chunks), plus per stream T | pad, the
            // 64 final states, a table entry (M > 32) and the byte the bit region rounds up to
            p.max_container = std::max(p.max_container, align_up((size_t)p.img[b].hdr_bytes + (size_t)(2 * all_syms) + (size_t)Mb * (2 + pay_bytes + 4 + 4 + (Q == 4 ? kRansSpillMax / 8 : 0)) + 64, 16));
            p.img[b].sbase = p.nstreams;
            for (int m = 0; m < Mb; ++m) p.sref.push_back(StreamRef{ b, m, Mb, p.nstreams });
            p.nstreams += Mb;
        }
        p.rslot_off.assign((size_t)p.nstreams, 0);
        for (long i = 0; i < (long)p.nstreams; ++i) p.rslot_off[i] = (long)i * p.rslot_cap;
        slot_pos = (long)p.nstreams * p.rslot_cap;
    }
    p.off_slots = take((size_t)slot_pos);
    const size_t ns = std::max((size_t)p.nstreams, (size_t)B * 32);
    p.off_rinfo = take(ns * 2 * sizeof(int32_t));
    p.off_rstate = take(ns * 64 * Q * sizeof(uint32_t));
    p.off_rpos = take(ns * sizeof(uint32_t));
    p.off_rtail = take(ns * sizeof(uint32_t));
    p.off_slot_len = take((size_t)LLICTI_NSTREAMS * B * sizeof(int32_t));
    // AC decode (equal sizes only): one chunk buffer per colour channel -- full rows (512 x uint16) or anchor rows (kAnchorRow bytes), see ac_use_anchors()
    size_t tables_bytes = 0;
    p.ac_cap_rows = 0;
    if (M == 0) {
        for (int st = 0; st < LLICTI_NSTREAMS; ++st) p.ac_cap_rows = std::max(p.ac_cap_rows, ac_chunk_rows((long)p.desc[(size_t)st * B].n));
        tables_bytes = (size_t)3 * B * p.ac_cap_rows * (ac_use_anchors(B) ? (size_t)kAnchorRow : (size_t)512 * sizeof(uint16_t));
    }
    {   // a second, quarter-size buffer for the CNN outputs of levels >= 1 (llicti_set_tuning("enc_side_levels"): the encoder's coarse levels
        // on a side stream).  Only the encoder uses it and only the AC DECODER uses the chunk tables, so the two share their bytes.
        tables_bytes = std::max(tables_bytes, p.lev_floats[1] * sizeof(float));
        p.off_tables = take(tables_bytes);
        p.off_params2 = p.off_tables;
    }
    static_assert(kAnchorRow <= 1024, "anchor rows must fit the full-row buffer");
    p.off_acstate = take((size_t)3 * B * 8 * sizeof(uint32_t));
    p.total = o;
    // mixed sizes: the band CNN's tile lists (image-major, rows, columns: the order the division form walks)
    p.tiles.clear();
    for (int k = 0; k < LLICTI_NLEVELS * 3; ++k) p.run[k] = TileRun{};
    if (!p.uniform) {
        for (int lvl = 0; lvl < LLICTI_NLEVELS; ++lvl) {
            const Geom *gl = &p.geo[(size_t)lvl * B];
            auto count = [&](int th) -> long {
                long t = 0;
                for (int b = 0; b < B; ++b) t += (long)((gl[b].w + kTileW - 1) / kTileW) * ((gl[b].h + th - 1) / th);
                return t;
            };
            for (int band = 0; band < 3; ++band) {
                const TileForm f = choose_tile_form(n_cu, tile_rows, band, count);
                TileRun &r = p.run[lvl * 3 + band];
                r.off = p.tiles.size(); r.n_tiles = (int)std::min<long>(f.n_tiles, 0x7FFFFFFFL); r.TH = f.TH; r.gx = f.gx;
                for (int b = 0; b < B; ++b) {
                    const int tx_n = (gl[b].w + kTileW - 1) / kTileW, ty_n = (gl[b].h + f.TH - 1) / f.TH;
                    for (int ty = 0; ty < ty_n; ++ty)
                        for (int tx = 0; tx < tx_n; ++tx) p.tiles.push_back(TileRef{ b, (ty << 16) | tx });
                }
            }
        }
    }
    // the device block: every table at a 256-byte boundary
    size_t d = 0;
    auto dtake = [&](size_t bytes) { size_t r = d; d = align_up(d + bytes, 256); return r; };
    p.d_img = dtake(p.img.size() * sizeof(ImgGeo));
    p.d_geo = dtake(p.geo.size() * sizeof(Geom));
    p.d_sg = dtake(p.sg.size() * sizeof(StageGeom));
    p.d_desc = dtake(p.desc.size() * sizeof(StreamDesc));
    p.d_slot_off = dtake(p.slot_off.size() * sizeof(long));
    p.d_slot_cap = dtake(p.slot_cap.size() * sizeof(int32_t));
    p.d_rslot_off = dtake(p.rslot_off.size() * sizeof(long));
    p.d_tiles = dtake(p.tiles.size() * sizeof(TileRef));
    p.d_sref = dtake(p.sref.size() * sizeof(StreamRef));
    p.d_total = d;
}

// mode: 0 = AC container (torchac-compatible, the reference's format); 0x100 | M = rANS container (v3) with M
// streams per image, M in 1 .. 32 (one per container segment) or {64, 128} (latency modes: 2 / 4 streams per segment)
static int mode_streams(int mode)      // -> M, | 0x100 for wide streams (LLICTI_MODE_RANS_WIDE), | 0x200 for xwide streams (LLICTI_MODE_RANS_X); 0: AC container; -1: unknown
{
    if (mode == 0) return 0;
    const int M = mode & 0xFF;
    if ((mode & ~0xFF) == 0x500) return ((M >= 1 && M <= 32) || M == 64 || M == 128) ? (M | 0x200) : -1;
    if ((mode & ~0xFF) == (0x500 | 0x10000)) return (M >= 1 && M <= 32) ? (M | 0x200 | 0x1000) : -1;      // LLICTI_MODE_RANS_X_AUTO(M): encode only
    if ((mode & ~0xFF) == 0x300) return (M >= 1 && M <= 14) ? (M | 0x100) : -1;
    if ((mode & ~0xFF) != 0x100) return -1;
    if (M < 1 || (M > 32 && M != 64 && M != 128)) return -1;
    return M;
}
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// Header byte 0 of a rANS container (the AC container stores the number of scales, 5, there): bit 7 = rANS, bit 3 = format v3 or later (the
// retired v2 had it clear), bit 6 = extended, bits 5,4,2,1,0 = a 5-bit value v:  M = v + 1 (1 .. 32 streams of 64 lanes, one per segment);
// extended: v = 0, 1: 64 / 128 streams of 64 lanes (M / 32 per segment); v = 2 .. 15: v - 1 wide streams (128 lanes); v = 16: xwide streams
// (256 lanes) in the v4 layout, whose COUNT is in bits 10 .. 15 of the int16 pad field (rans_pad_hi(); those bits are zero in every other
// container, so a reader of the older formats finds a pad field that contradicts the size and refuses).  v = 17 .. 31 were the xwide tags of
// the v3 layout (rounds 4-5): retired, refused.
static int rans_byte0(int M, int Q)
{
    const int ext = (M > 32 || Q > 1) ? 1 : 0;
    const int v = Q == 4 ? 16 : Q == 2 ? M + 1 : M > 32 ? (M == 64 ? 0 : 1) : M - 1;
    return 0x88 | (ext << 6) | (((v >> 3) & 3) << 4) | (v & 7);
}
static int rans_pad_hi(int M, int Q) { return Q == 4 ? (M <= 32 ? M : M == 64 ? 33 : 34) : 0; }      // bits 10 .. 15 of the pad field
// -> M (| 0x100 for wide, | 0x200 for xwide streams) of a header's byte 0 and pad field; 0: not a rANS container this build reads
static int rans_streams_of_header(int b0, int padfield)
{
    if ((b0 & 0x88) != 0x88) return 0;
    const int v = (((b0 >> 4) & 3) << 3) | (b0 & 7), u = (padfield >> 10) & 0x3F;
    if (((b0 >> 6) & 1) && v >= 16) {
        if (v != 16 || u < 1 || u > 34) return 0;             // (v3's xwide tags, or no count)
        return (u <= 32 ? u : u == 33 ? 64 : 128) | 0x200;
    }
    if (u) return 0;
    if (!((b0 >> 6) & 1)) return v + 1;
    if (v <= 1) return 64 << v;
    return (v - 1) | 0x100;
}

static int check_dims_v(int B, const int *Hs, const int *Ws)
{
    if (B < 1 || !Hs || !Ws) return fail(LLICTI_EINVAL, "bad batch: B=%d (need B>=1 and the sizes of every image)", B);
    for (int b = 0; b < B; ++b)
        if (Hs[b] < 32 || Ws[b] < 32 || Hs[b] > 8160 || Ws[b] > 8160) return fail(LLICTI_EINVAL, "bad shape of image %d: H=%d W=%d (need 32<=H,W<=8160)", b, Hs[b], Ws[b]);
    return 0;
}
// modes: one mode (n_modes = 1) or one per image (rANS containers of one lane kind, stream counts may differ)
static size_t plan_workspace_bytes_vm(int B, const int *Hs, const int *Ws, const int *modes, int n_modes)
{
    if (check_dims_v(B, Hs, Ws) || !modes || (n_modes != 1 && n_modes != B)) return 0;
    const int ME = mode_streams(modes[0]);
    if (ME < 0) return 0;
    std::vector<int> Ms;
    if (n_modes == B && B > 1)
        for (int b = 0; b < B; ++b) {
            const int MEb = mode_streams(modes[b]);
            if (MEb < 0 || (MEb >> 8) != (ME >> 8) || ((MEb & 0xFF) == 0) != ((ME & 0xFF) == 0)) return 0;      // (one lane kind per call -- and all "auto" or none)
            Ms.push_back(MEb & 0xFF);
        }
    Plan p, q;
    build_plan(p, B, Hs, Ws, nullptr, ME, 256, 0, false, Ms.empty() ? nullptr : Ms.data());
    build_plan(q, B, Hs, Ws, nullptr, ME, 256, 0, true, Ms.empty() ? nullptr : Ms.data());      // (llicti_set_tuning("force_ragged"): image blocks at 64-element boundaries)
    return std::max(p.total, q.total);
}
static size_t plan_workspace_bytes_v(int B, const int *Hs, const int *Ws, int mode) { return plan_workspace_bytes_vm(B, Hs, Ws, &mode, 1); }
static size_t plan_workspace_bytes(int B, int H, int W, int mode)
{
    if (check_dims(B, H, W)) return 0;
    std::vector<int> Hs(B, H), Ws(B, W);
    return plan_workspace_bytes_v(B, Hs.data(), Ws.data(), mode);
}
static size_t plan_max_container_bytes(int H, int W)
{
    if (check_dims(1, H, W)) return 0;
    Plan p, q;
    build_plan(p, 1, &H, &W, nullptr, 32);     // covers the AC container and M <= 32 ...
    build_plan(q, 1, &H, &W, nullptr, kRansMaxStreams);     // ... and the many-stream latency modes (more per-stream slack)
    Plan a, w, x;
    build_plan(a, 1, &H, &W, nullptr, 0);
    build_plan(w, 1, &H, &W, nullptr, 14 | 0x100);          // ... and wide ...
    build_plan(x, 1, &H, &W, nullptr, 128 | 0x200);         // ... and xwide streams (larger state blocks)
    return std::max(std::max(std::max(p.max_container, q.max_container), std::max(w.max_container, x.max_container)), a.max_container);
}

static int plan_header_dims(const uint8_t *h, int *H, int *W)
{
    if (!h || !H || !W) return fail(LLICTI_EINVAL, "header_dims: null pointer");
    if ((h[0] & 0x88) == 0x80)
        return fail(LLICTI_EFORMAT, "header: byte 0 = 0x%02x is the retired LLICTI-rANS v2 container; this build reads and writes v3 only", h[0]);
    if (h[0] != LLICTI_NLEVELS && rans_streams_of_header(h[0], (int)(uint16_t)(h[15] | (h[16] << 8))) == 0)
        return fail(LLICTI_EFORMAT, "header: byte 0 = 0x%02x (pad field 0x%04x) is neither %d scales (AC container) nor a rANS container tag of this build "
                    "(the xwide v3 layout of rounds 4-5 is retired)", h[0], (unsigned)(h[15] | (h[16] << 8)), LLICTI_NLEVELS);
    int Hc = h[1], Wc = h[2];
    int pad = (int)(int16_t)(h[15] | (h[16] << 8));
    for (int l = LLICTI_NLEVELS - 1; l >= 0; --l) {     // _get_padHW_lev_list, LLICTI_nets.py:533-542
        const int padW = pad & 1; pad >>= 1;
        const int padH = pad & 1; pad >>= 1;
        Hc = 2 * Hc - padH;
        Wc = 2 * Wc - padW;
    }
    *H = Hc; *W = Wc;
    return LLICTI_OK;
}

