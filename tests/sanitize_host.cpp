// sanitize_host.cpp -- the HOST logic of libllicti_hip.so (llicti_amd/csrc/host_types.hpp, cnn_pack.hpp, host_plan.hpp: no HIP in them)
// compiled by g++ under AddressSanitizer + UBSan and driven over every (B, sizes, mode) the tests, the fuzzer and bench.py use, mixed-size
// batches at the sizes of the reference's own test set, malformed headers and out-of-range arguments.  Last round's heap over-read lived in
// exactly this code (the plan's slot table); the GPU cannot run a sanitizer on this pool, the host half can.  tests/sanitize_host.sh builds
// and runs it.  Every plan is checked for internal consistency too: regions inside the workspace and disjoint, tables inside their block,
// per-image placements disjoint, stream descriptors inside the pairs region, tile lists covering each band grid exactly once.
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <set>

#include "../llicti_amd/csrc/host_plan.hpp"

static long n_plans = 0, n_checks = 0;
#define REQUIRE(c)                                                                      \
    do {                                                                                \
        ++n_checks;                                                                     \
        if (!(c)) { fprintf(stderr, "FAILED %s (%s:%d)\n", #c, __FILE__, __LINE__); exit(1); } \
    } while (0)

static void check_plan(const Plan &p, int B, const int *Hs, const int *Ws, int ME, const int *Ms = nullptr)
{
    ++n_plans;
    const int Q = 1 << (ME >> 8);
    int M = ME & 0xFF;
    if (Ms && M > 0) { M = 0; for (int b = 0; b < B; ++b) M = std::max(M, Ms[b]); }
    REQUIRE(p.B == B && p.M == M && p.Q == Q);
    // streams: a count per image (its header's tag says it), a per-stream table, one slot per stream
    if (M > 0) {
        int ns = 0;
        for (int b = 0; b < B; ++b) {
            const int Mb = Ms ? Ms[b] : M;
            REQUIRE(p.img[b].M == Mb && p.img[b].sbase == ns);
            REQUIRE(rans_streams_of_header(p.img[b].byte0, p.img[b].padint) == (Mb | ((ME >> 8) << 8)));
            for (int m = 0; m < Mb; ++m) {
                const StreamRef &r = p.sref[(size_t)ns + m];
                REQUIRE(r.b == b && r.m == m && r.M == Mb && r.sbase == ns);
            }
            ns += Mb;
        }
        REQUIRE(p.nstreams == ns && (int)p.sref.size() == ns && (int)p.rslot_off.size() == ns);
        REQUIRE(p.d_sref + p.sref.size() * sizeof(StreamRef) <= p.d_total);
    } else {
        REQUIRE(p.nstreams == 0 && p.sref.empty());
        for (int b = 0; b < B; ++b) REQUIRE(p.img[b].byte0 == LLICTI_NLEVELS);
    }
    // workspace regions: ascending, inside [0, total)
    struct R { size_t off, len; };
    long pix = 0;
    for (int b = 0; b < B; ++b) pix = std::max(pix, p.img[b].pix_off + 3 * p.img[b].plane);
    std::vector<R> rs = { { p.off_status, (size_t)(16 + B) * 4 }, { p.off_minmax, (size_t)B * 16 }, { p.off_planes, (size_t)pix * 2 },
                          { p.off_fplanes, (size_t)pix * 4 }, { p.off_params, std::max(p.lev_floats[0], 3 * p.lev_floats[1]) * 4 } };
    for (size_t i = 0; i < rs.size(); ++i) {
        REQUIRE(rs[i].off + rs[i].len <= p.total);
        if (i) REQUIRE(rs[i - 1].off + rs[i - 1].len <= rs[i].off);
    }
    REQUIRE(p.off_pairs >= rs.back().off + rs.back().len && p.off_slots >= p.off_pairs && p.off_acstate < p.total);
    // images: placements disjoint, header constants
    for (int b = 0; b < B; ++b) {
        const ImgGeo &ig = p.img[b];
        REQUIRE(ig.H == Hs[b] && ig.W == Ws[b] && ig.plane == (long)Hs[b] * Ws[b]);
        REQUIRE(ig.hdr_bytes == 17 + 3 * ig.h4 * ig.w4 && ig.h4 >= 1 && ig.w4 >= 1 && ig.h4 <= 255 && ig.w4 <= 255);
        if (b) REQUIRE(p.img[b - 1].pix_off + 3 * p.img[b - 1].plane <= ig.pix_off);
        uint8_t hdr[17] = { (uint8_t)ig.byte0, (uint8_t)ig.h4, (uint8_t)ig.w4 };
        hdr[15] = (uint8_t)(ig.padint & 0xFF); hdr[16] = (uint8_t)(ig.padint >> 8);
        int H2 = 0, W2 = 0;
        REQUIRE(plan_header_dims(hdr, &H2, &W2) == 0 && H2 == ig.H && W2 == ig.W);     // the header round-trips the size
    }
    // levels and stages
    long pair_end = 0;
    for (int lvl = 0; lvl < LLICTI_NLEVELS; ++lvl) {
        size_t fl = 0;
        for (int b = 0; b < B; ++b) {
            const Geom &g = p.geo[(size_t)lvl * B + b];
            REQUIRE(g.par_off == (long)fl && g.pix_off == p.img[b].pix_off && g.lvl == lvl);
            fl += (size_t)g.h * g.w * 64;
        }
        REQUIRE(fl == p.lev_floats[lvl]);
        for (int band = 0; band < 3; ++band)
            for (int b = 0; b < B; ++b) {
                const StageGeom &sg = p.sg[(size_t)(lvl * 3 + band) * B + b];
                REQUIRE(sg.hc >= 1 && sg.wc >= 1 && sg.hc <= sg.h && sg.wc <= sg.w);
                REQUIRE(div_wc(sg, sg.hc * sg.wc - 1) == sg.hc - 1);
                for (int clr = 0; clr < 3; ++clr) {
                    const StreamDesc &d = p.desc[(size_t)stage_index(lvl, band, clr) * B + b];
                    REQUIRE(d.n == sg.hc * sg.wc);
                    REQUIRE(d.pair_off == p.pair_base[lvl * 3 + band] + clr * sg.pair_cs + sg.pair_off);
                    pair_end = std::max(pair_end, d.pair_off + d.n);
                }
            }
    }
    REQUIRE(p.off_pairs + (size_t)pair_end * 4 <= p.off_slots);
    if (M > 0) {
        REQUIRE(p.off_slots + (size_t)(p.rslot_off.back() + p.rslot_cap) <= p.off_rinfo);
        REQUIRE(p.d_rslot_off + p.rslot_off.size() * sizeof(long) <= p.d_total);
    } else {
        REQUIRE(p.slot_off.size() == (size_t)45 * B && p.off_slots + (size_t)(p.slot_off.back() + p.slot_cap.back()) <= p.off_rinfo);
    }
    for (int b = 0; b < B; ++b) REQUIRE(p.max_container >= (size_t)p.img[b].hdr_bytes);
    REQUIRE(p.d_img + p.img.size() * sizeof(ImgGeo) <= p.d_geo && p.d_geo + p.geo.size() * sizeof(Geom) <= p.d_sg &&
            p.d_sg + p.sg.size() * sizeof(StageGeom) <= p.d_desc && p.d_desc + p.desc.size() * sizeof(StreamDesc) <= p.d_slot_off &&
            p.d_tiles + p.tiles.size() * sizeof(TileRef) <= p.d_total);
    // tile lists of a mixed-size plan: every (image, tile) of every band grid exactly once, in image-major order
    if (!p.uniform)
        for (int k = 0; k < LLICTI_NLEVELS * 3; ++k) {
            const TileRun &r = p.run[k];
            const int lvl = k / 3;
            REQUIRE(r.TH == 16 || r.TH == 8 || r.TH == 4);
            REQUIRE(r.off + (size_t)r.n_tiles <= p.tiles.size() && r.gx >= 1 && r.gx <= r.n_tiles);
            size_t t = r.off;
            for (int b = 0; b < B; ++b) {
                const Geom &g = p.geo[(size_t)lvl * B + b];
                for (int ty = 0; ty < (g.h + r.TH - 1) / r.TH; ++ty)
                    for (int tx = 0; tx < (g.w + kTileW - 1) / kTileW; ++tx, ++t) REQUIRE(p.tiles[t].img == b && p.tiles[t].yx == ((ty << 16) | tx));
            }
            REQUIRE(t == r.off + (size_t)r.n_tiles);
        }
}

static const int kModes[] = { 0, 0x100 | 1, 0x100 | 8, 0x100 | 32, 0x100 | 64, 0x100 | 128, 0x300 | 1, 0x300 | 10, 0x300 | 14,
                              0x500 | 1, 0x500 | 3, 0x500 | 10, 0x500 | 16, 0x500 | 21, 0x500 | 32, 0x500 | 64, 0x500 | 128 };

int main()
{
    std::mt19937 rng(7);
    REQUIRE(selftest_div_magic() == 0);
    // container tags: every mode round-trips through header byte 0; everything else is rejected
    for (int mode : kModes) {
        const int ME = mode_streams(mode);
        REQUIRE(ME >= 0);
        if (ME) REQUIRE(rans_streams_of_header(rans_byte0(ME & 0xFF, 1 << (ME >> 8)), rans_pad_hi(ME & 0xFF, 1 << (ME >> 8)) << 10) == ME);
    }
    for (int mode : { -1, 1, 0x100, 0x100 | 33, 0x100 | 127, 0x300 | 15, 0x500, 0x500 | 33, 0x500 | 96, 0x700 | 1, 0x10000 }) REQUIRE(mode_streams(mode) < 0);
    // every (byte 0, pad field high bits) pair: either not a container of this build, or a mode that round-trips; the xwide v3 tags of rounds 4-5
    // (v = 17 .. 31) and an xwide v4 tag without its count are refused, and so are high bits in a container that is not xwide
    for (int b0 = 0; b0 < 256; ++b0)
        for (int u = 0; u < 64; ++u) {
            const int v = rans_streams_of_header(b0, (u << 10) | 0x155);
            REQUIRE(v == 0 || mode_streams(((v >> 8) == 2 ? 0x500 : (v >> 8) == 1 ? 0x300 : 0x100) | (v & 0xFF)) == v);
            const int tagv = (((b0 >> 4) & 3) << 3) | (b0 & 7);
            if ((b0 & 0xC8) == 0xC8 && tagv >= 17) REQUIRE(v == 0);
            if ((b0 & 0xC8) == 0xC8 && tagv == 16) REQUIRE((v != 0) == (u >= 1 && u <= 34));
            if (!((b0 & 0xC8) == 0xC8 && tagv >= 16) && u) REQUIRE(v == 0);
        }
    // equal-size plans: every shape of the suite, the fuzzer's range and bench.py, the format's limits, in every mode family
    const int shapes[][2] = { { 32, 32 }, { 33, 64 }, { 67, 93 }, { 64, 48 }, { 96, 160 }, { 150, 131 }, { 97, 351 }, { 256, 256 }, { 512, 768 }, { 768, 512 },
                              { 577, 768 }, { 2160, 3840 }, { 8160, 32 }, { 32, 8160 }, { 8160, 8160 }, { 4097, 4099 } };
    for (auto &sh : shapes)
        for (int mode : kModes)
            for (int B : { 1, 2, 3, 24, 32 }) {
                if ((long)B * sh[0] * sh[1] > 70L << 20) continue;
                const int ME = mode_streams(mode);
                std::vector<int> Hs(B, sh[0]), Ws(B, sh[1]);
                for (int ragged = 0; ragged < 2; ++ragged) {
                    if (ragged && ME == 0 && B > 1) continue;
                    Plan p;
                    build_plan(p, B, Hs.data(), Ws.data(), nullptr, ME, 256, 0, ragged != 0);
                    check_plan(p, B, Hs.data(), Ws.data(), ME);
                    REQUIRE(p.uniform == !ragged);
                }
                REQUIRE(plan_workspace_bytes(B, sh[0], sh[1], mode) > 0);
            }
    for (auto &sh : shapes) REQUIRE(plan_max_container_bytes(sh[0], sh[1]) >= 17);
    // mixed-size plans: random batches, every tile-form tuning, odd compute-unit counts
    for (int it = 0; it < 400; ++it) {
        const int B = 1 + (int)(rng() % 33);
        std::vector<int> Hs(B), Ws(B);
        for (int b = 0; b < B; ++b) { Hs[b] = 32 + (int)(rng() % (it % 7 == 0 ? 1500 : 300)); Ws[b] = 32 + (int)(rng() % (it % 5 == 0 ? 1500 : 300)); }
        const int mode = kModes[1 + rng() % (sizeof kModes / sizeof kModes[0] - 1)];
        const int tr[] = { 0, 16, 8, 4, -1 };
        Plan p;
        build_plan(p, B, Hs.data(), Ws.data(), nullptr, mode_streams(mode), 1 + (int)(rng() % 320), tr[rng() % 5], false);
        check_plan(p, B, Hs.data(), Ws.data(), mode_streams(mode));
        REQUIRE(plan_workspace_bytes_v(B, Hs.data(), Ws.data(), mode) >= p.total);
        if (it % 2 == 0) {                                      // a stream count per image (llicti_encode_images_vm): any count the lane kind allows
            const int ME = mode_streams(mode), kind = ME >> 8;
            std::vector<int> Ms(B), modes(B);
            for (int b = 0; b < B; ++b) {
                Ms[b] = kind == 0 ? 1 + (int)(rng() % 32) : 1 + (int)(rng() % 14);
                modes[b] = (kind == 2 ? 0x500 : kind == 1 ? 0x300 : 0x100) | Ms[b];
            }
            Plan q;
            build_plan(q, B, Hs.data(), Ws.data(), nullptr, ME, 256, 0, false, Ms.data());
            check_plan(q, B, Hs.data(), Ws.data(), ME, Ms.data());
            REQUIRE(plan_workspace_bytes_vm(B, Hs.data(), Ws.data(), modes.data(), B) >= q.total);
            if (B > 1) { modes[B - 1] = 0; REQUIRE(plan_workspace_bytes_vm(B, Hs.data(), Ws.data(), modes.data(), B) == 0); }      // the reference format among rANS modes: refused
        }
        if (it % 3 == 0) {                                      // caller-chosen RGB placement
            std::vector<size_t> off(B);
            size_t pos = 64;
            for (int b = 0; b < B; ++b) { off[b] = pos; pos += 3 * (size_t)Hs[b] * Ws[b] + rng() % 100; }
            Plan q;
            build_plan(q, B, Hs.data(), Ws.data(), off.data(), mode_streams(mode));
            check_plan(q, B, Hs.data(), Ws.data(), mode_streams(mode));
            REQUIRE(!q.uniform && q.rgb_bytes <= pos);
        }
    }
    // bad arguments: rejected, never indexed with
    {
        int Hs[3] = { 64, 16, 64 }, Ws[3] = { 64, 64, 9000 };
        REQUIRE(check_dims_v(3, Hs, Ws) != 0 && check_dims_v(0, Hs, Ws) != 0 && check_dims_v(1, nullptr, Ws) != 0);
        REQUIRE(plan_workspace_bytes_v(3, Hs, Ws, 0x500 | 10) == 0 && plan_workspace_bytes(0, 64, 64, 0) == 0 && plan_workspace_bytes(1, 64, 64, 0x100) == 0);
        REQUIRE(plan_max_container_bytes(16, 64) == 0 && plan_max_container_bytes(64, 9000) == 0);
        REQUIRE(check_dims(1, 31, 64) != 0 && check_dims(1, 64, 8161) != 0 && check_dims(1, 8160, 8160) == 0);
    }
    // headers: every byte 0, random pad fields
    for (int b0 = 0; b0 < 256; ++b0)
        for (int k = 0; k < 64; ++k) {
            uint8_t h[17];
            for (auto &v : h) v = (uint8_t)rng();
            h[0] = (uint8_t)b0;
            int H = 0, W = 0;
            const int rc = plan_header_dims(h, &H, &W);
            REQUIRE(rc == 0 || rc == LLICTI_EFORMAT);
            if (rc == 0) REQUIRE(H <= 2 * 255 * 16 && W <= 2 * 255 * 16);
        }
    REQUIRE(plan_header_dims(nullptr, nullptr, nullptr) == LLICTI_EINVAL);
    // the weight pack: every canonical element lands once in the fragment-ordered image of its head
    for (int band = 0; band < 3; ++band) {
        const int K0 = band == 0 ? 48 : band == 1 ? 72 : 120;
        std::vector<float> w0((size_t)352 * K0), b0(352), w1((size_t)352 * 88), b1(352), w2((size_t)60 * 88), b2(60), pk;
        float v = 1.0f;
        for (auto *a : { &w0, &b0, &w1, &b1, &w2, &b2 }) for (float &x : *a) x = v++;
        pack_band(K0, w0.data(), b0.data(), w1.data(), b1.data(), w2.data(), b2.data(), pk);
        REQUIRE(pk.size() == (size_t)4 * pack_floats(K0));
        std::multiset<float> got(pk.begin(), pk.end());
        for (auto *a : { &w0, &b0, &w1, &b1, &w2, &b2 }) for (float x : *a) REQUIRE(got.count(x) >= 1);
        REQUIRE(cnn_lds_bytes(band, 16) <= 160 * 1024 && cnn_lds_bytes(band, 4) < cnn_lds_bytes(band, 8));
    }
    printf("sanitize_host: %ld plans, %ld checks, clean\n", n_plans, n_checks);
    return 0;
}
