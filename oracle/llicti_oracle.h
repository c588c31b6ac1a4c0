/*
 * llicti_oracle.h -- CPU ORACLE for the LLICTI encode/decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under llicti_amd/ (the product) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * It restates, in plain C, the algorithm of the reference (kamisli-icpl/LLICTI @ 2024_08_07):
 *   graphs/models/LLICTI_nets.py:62-88      integer YCoCg-R lift / inverse lift
 *   graphs/models/LLICTI_nets.py:125-252    compress / decompres / lazyDWT / DC band
 *   graphs/models/LLICTI_nets.py:344-583    entropy layer: header, stage loop, pad / crop, symbol shifts
 *   graphs/models/LLICTI_nets.py:644-753    interpolator CNN (layer0 convs + grouped 1x1 tail), :822-825
 *   graphs/models/LLICTI_nets.py:938-983    CDF sample grid and 16-bit integerisation
 *   graphs/layers/entropy_layer_nets.py:185-204   5-component Gaussian-mixture CDF
 * and of its two third-party dependencies that are NOT in /root/reference:
 *   compressai==1.1.8  GaussianConditional._standardized_cumulative (0.5*erfc(-x/sqrt2)), LowerBound (max)
 *   torchac==0.9.3     32-bit binary arithmetic coder with 16-bit CDFs
 *
 * Floating point follows the "LLICTI-MI355X numerics spec v1" (DESIGN.md section 4): every fp32
 * operation, its order and its rounding are fixed (k-ordered fmaf chains for the convolutions, a
 * polynomial erfc with fixed coefficients, IEEE division), so the HIP kernels can reproduce the
 * oracle bit for bit.  Against the reference's PyTorch ops this agrees to ~1e-6 (params) and to
 * +-1 count on a few per cent of 16-bit table entries; the integer parts (lift, split, header,
 * symbols, coder) are exact.
 *
 * PARITY PINNING: pinned against fixtures generated from the reference-owned Python
 * (tests/golden/make_fixtures.py) for everything except the arithmetic coder's byte output:
 * torchac is absent from this image, so the coder is "parity unpinned": restated from its published
 * algorithm, and checked against an INDEPENDENT second restatement (tests/ref_ac.py, pure Python written from
 * SURVEY.md Appendix A), against three vectors worked out by hand, and for self-consistency on the reference's own
 * recorded tables (tests/test_ref_ac.py).
 */
#ifndef LLICTI_ORACLE_H
#define LLICTI_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NCH   352          /* hidden channels = 4 heads x 88 (configs/llicti_A.json chs[0]=88, mwsa separate) */
#define ORC_HEAD   88
#define ORC_NPAR   60          /* 15 sigma | 15 mu | 15 weight | 5 a | 5 b | 5 d */
#define ORC_NLEV    5
#define ORC_NSTREAM 45         /* 5 levels x 3 bands x 3 colour channels */

/* Canonical packed weights of one band network (band 0: x00->x11, 1: x00,x11->x01, 2: x00,x11,x01->x10).
 * w0 is the concatenation along K of the band's layer-0 conv weights in source order x00, x11, x01
 * (K0 = 48 / 72 / 120); a conv with a 4-wide kernel is flattened (ci, ky, kx), a 4x3 conv (ci, kx, ky),
 * so that four consecutive k walk the kernel's length-4 axis (llicti_amd/weights.py); b0 is the fp32
 * sum of their biases taken left to right. */
typedef struct {
    int K0;
    const float *w0;   /* [352][K0] */
    const float *b0;   /* [352]     */
    const float *w1;   /* [352][88] (grouped 1x1: row c reads inputs of head c/88) */
    const float *b1;   /* [352]     */
    const float *w2;   /* [60][88]  (row o reads inputs of head o/15) */
    const float *b2;   /* [60]      */
} orc_band_weights;

typedef struct { orc_band_weights band[3]; } orc_weights;

/* ---- integer colour lift (LLICTI_nets.py:62-74, :76-88, :571-582); planes are Y-127, Co, Cg ---- */
void orc_lift(const uint8_t *rgb, int H, int W, int16_t *planes, int16_t minmax[6]);
void orc_unlift(const int16_t *planes, int H, int W, uint8_t *rgb);

/* level geometry: Hl = ceil(H / 2^l); band grid h = ceil(Hl/2); padH = Hl & 1 */
void orc_level_geom(int H, int W, int lvl, int *Hl, int *Wl, int *h, int *w, int *padH, int *padW);

/* ---- interpolator CNN for one (level, band): out[i*w + j][60] on the full (padded) h x w grid ---- */
void orc_band_params(const int16_t *planes, int H, int W, int lvl, int band,
                     const orc_band_weights *bw, float *out);
/* same network on float planes [3][H][W] */
void orc_band_params_f(const float *fplanes, int H, int W, int lvl, int band,
                       const orc_band_weights *bw, float *out);

/* training / validation likelihood path (LLICTI.forward, LLICTI_nets.py:101-123; entropy_layer_nets.py:117-183) */
void orc_lift_train(const uint8_t *rgb, int H, int W, float *fplanes);
void orc_selfinfo(const float *fplanes, int H, int W, int lvl, int band, const float *params, float *out);

/* ---- one row of the integer CDF table (LLICTI_nets.py:938-983 + entropy_layer_nets.py:185-204) ----
 * par: the 60 raw CNN outputs of the position; yv/cov: the band's own Y / Co pixel as int/255
 * (cross-channel mean update, LLICTI_nets.py:389-392); row gets Lp = maxv-minv+2 uint16 entries. */
void orc_cdf_row(const float *par, int clr, float yv, float cov, int minv, int maxv, uint16_t *row);
/* float mixture CDF at one sample point, before integerisation (for tolerance checks) */
float orc_cdf_float(const float *par, int clr, float yv, float cov, float pt);
float orc_erfc(float x);        /* the spec's erfc, exported for unit tests */

/* ---- torchac-compatible arithmetic coder on explicit tables (the reference's third-party seam) ----
 * cdf: [N][Lp] uint16 (int16 reinterpreted), sym: [N] int16.  Returns bytes written (<= cap) or -1. */
long orc_ac_encode_tables(const uint16_t *cdf, int Lp, const int16_t *sym, long N, uint8_t *out, long cap);
void orc_ac_decode_tables(const uint16_t *cdf, int Lp, const uint8_t *in, long nbytes, long N, int16_t *sym);
/* encoder on (c_low, c_high) pairs (c_high up to 0x10000) */
long orc_ac_encode_pairs(const uint32_t *clow, const uint32_t *chigh, long N, uint8_t *out, long cap);

/* ---- whole image (LLICTI.compress / LLICTI.decompres) ----
 * Output container = the reference's bytestream_list flattened: 4 header segments
 * [S,h4,w4 u8] [6 x int16 min/max] [int16 padHW] [raw DC band u8 CHW] followed by the 45 streams in
 * order scale 4..0, band 0..2, clr Y,Co,Cg.  seg_len[49] receives the segment lengths.
 * full_tables != 0 materialises every Lp-entry table like the reference does (slow; cpu_baseline);
 * 0 evaluates only the entries the coder reads -- identical bytes by construction. */
long orc_encode_image(const uint8_t *rgb, int H, int W, const orc_weights *wts, int full_tables,
                      uint8_t *out, long cap, int32_t seg_len[49]);
/* returns 0 on success; rgb must hold 3*H*W bytes where H, W are derived from the header */
int orc_decode_image(const uint8_t *in, const int32_t seg_len[49], const orc_weights *wts, int full_tables,
                     uint8_t *rgb, long rgb_cap, int *H_out, int *W_out);
void orc_header_dims(const uint8_t *in, const int32_t seg_len[49], int *H_out, int *W_out);

/* per-symbol (c_low, c_high) of one stream, in stream order (cropped raster); returns symbol count */
long orc_stream_pairs(const int16_t *planes, int H, int W, const int16_t minmax[6], int lvl, int band, int clr,
                      const float *params /* [h*w][60] from orc_band_params */,
                      uint32_t *clow, uint32_t *chigh, int16_t *sym);

/* ---- rANS containers ("LLICTI-rANS v3", and "v4" for the 256-lane streams; NEW formats of this build: the reference has only torchac) ----
 * Same header segments except byte 0 and -- v4 -- the high bits of the pad field: byte 0 bit 7 = rANS, bit 3 = format v3 or later (v2 had it clear
 * and is rejected), bit 6 = extended, bits 5,4,2,1,0 = v.  Not extended: M = v + 1 streams (1 .. 32) of L = 64 lanes.  Extended, v = 0 / 1: M = 64 /
 * 128 streams of 64 lanes (latency modes: M / 32 streams per segment behind a table of their u32 lengths).  Extended, v = 2 .. 15: M = v - 1 WIDE
 * streams (1 .. 14) of L = 128 lanes.  Extended, v = 16 (byte 0 = 0xE8): XWIDE streams of L = 256 lanes in the v4 layout; their count is in bits
 * 10 .. 15 of the int16 pad field (bits 0 .. 9 are the five levels' pad flags; these six bits are zero in every other container): u = 1 .. 32 -> M = u
 * streams, one per segment; u = 33 / 34 -> 64 / 128 streams, two / four per segment behind the length table.
 * COMPATIBILITY RULE: a format revision that changes the meaning of stream bytes takes a header value no earlier reader accepts.  v = 17 .. 31 were
 * the xwide tags of rounds 4-5 (v3 layout: u16 T field in front, seeded chains inside the payload, escape); v4 retires them -- this reader refuses
 * them with -4 / LLICTI_EFORMAT -- and a round-5 reader refuses a v4 container because its pad field contradicts the image size.
 * Same CDFs, same symbols; the 45 torchac streams are replaced by M independent L-way interleaved rANS streams per image
 * (seg_len[4 .. 4+M-1], the rest 0).  Stage st (decode order) has nc symbols in cropped raster order; symbol n sits in chunk
 * n/L, lane n%L; chunk c belongs to stream c % M and is that stream's step c / M of the stage.
 *   coder      32-bit states in [2^31, 2^32), 16-bit probabilities, BIT-granular renormalisation: before coding a symbol of
 *              frequency f the encoder shifts out the n lowest bits of x, n minimal with (x >> n) < f << 16; then
 *              x = ((x / f) << 16) + x % f + c_low.  The decoder inverts: x = f * (x >> 16) + (x & 0xFFFF) - c_low, n = clz(x),
 *              x = x << n | n bits.  x / f >= 2^15: the loss per symbol is ~2^-16 of its length, like the range coder's.
 *   order      per step the decoder first decodes the L lanes' symbols, then renormalises lane-ascending; it reads the bit
 *              region DOWNWARDS (the encoder, which runs the steps backwards and the lanes descending, wrote it upwards).
 *   tail       a lane's initial state carries payload instead of nothing: the stream's last T symbols of the LAST stage
 *              (sequence positions cnt-T .. cnt-1 of its cnt symbols there) are coded by a single-state coder of the same kind,
 *              last symbol first, and its output is cut into the 31 L bits the lanes start from: lane l = 2^31 | payload bits [31 l, 31 l + 31).
 *              After the last stage the decoder is left with exactly those states, reassembles the payload and decodes the T symbols.
 *   tail, 64 / 128 lanes (v3, unchanged): the first pushed symbol starts from x = f << 15 (it codes to 2^31 + c_low without a bit); the bits go up
 *              from bit 0 of the payload (1984 / 3968 bits), the final 32-bit state on top (leading one = the payload's highest set bit).  T is
 *              maximal with 32 + bits <= 31 L (T <= 2047).  The decoder finds the tail state by its leading one, reads downwards, and must end
 *              with the tail coder's start state and no bit left (and the main region read to its last bit) -- the format's integrity check.
 *              Stream: u16 LE (T | pad << 11, bits 14 and 15 zero) | bit region, LSB first, ceil(bits / 8) bytes, pad = unused zero bits on top
 *              of its last byte | L x 31-bit final states (low 31 bits of x_l at bit 31 l; 248 / 496 bytes).
 *   tail, xwide v4 (round 6):
 *     arena    the tail coder's output is NOT cut to the payload: tail symbols are taken -- counting from the stream's end, j = 0 the last -- until,
 *              at a multiple of 32, the output has reached the payload's 7936 bits (or the stream's share of the last stage, or 8160 symbols, ends).
 *              The output ("arena", alen bits) is the payload followed by a SPILL of alen - 7936 < 512 bits, which lies at the bottom of the main
 *              bit region: the main coder's bits start above it, and the main decoder, reading down, leaves it -- its cursor ends AT the spill's
 *              length, which is how the tail decoder knows alen.  No unused payload bits where the tail has symbols to take; T is a multiple of
 *              32 (or the stream's whole share), so its field is T / 32 rounded up: 8 bits.
 *     chains   ONE OR TWO single-state coders.  With A = the number of symbol values of the image's Cg channel (max - min + 1) and n = the
 *              largest count with A^n <= 2^31 (at most 31): TWO chains iff the stream has 2 n symbols and, over its last up to 64 symbols (k of
 *              them), 2 n sum(16 - floor(log2 freq)) >= k (64 + n) -- symbols expensive enough that n raw ones are worth a 32-bit state.
 *              Two: chain A starts from 2^31 | sum sym(i) A^i (i < n), chain B from the same of sym(n + i) (symbol INDICES, raw); symbol
 *              j >= 2 n is pushed on A if j is even, on B if odd.  Arena: bits [0, 32) A's final state, A's bit fields from bit 32 UP in the
 *              decoder's reading order (last pushed first); the top 32 bits B's final state, B's fields below it, read DOWN; zeros between
 *              (only a tail that ran out of symbols leaves any: alen = 7936 then).  Checks: both states have their leading one, the cursors do
 *              not cross, the bits between them are zero, each chain ends at a seed below A^n.
 *              One: the chain starts from x = sym(0) -- the stream's last symbol itself, its index, nothing added -- and symbol j >= 1 is pushed
 *              with the same rule (n minimal with (x >> n) < f << 16), which emits NOTHING while the state is still small: the first few pushes
 *              only grow it (x < f: x + c_low).  So a chain costs what its symbols cost (v3's start state of 2^31 cost ~31 bits minus the raw
 *              seeds).  Arena: bits [0, 32) the final state (flat: it may be below 2^31 if the tail is a handful of symbols), the fields from
 *              bit 32 up in the decoder's reading order, then ONE end-marker bit -- the arena's highest set bit (a spill ends with it).  The
 *              decoder takes min(clz(x), bits left below the marker) bits after each symbol: once the bits are used up it is in the encoder's
 *              silent start and the state stays small.  Checks: clz <= 16 wherever bits were left, every bit read, the final state = sym(0) < A.
 *     stream   bit region, LSB first | L x 31-bit final states (992 bytes).  Bit region, bottom up: the spill | the main coder's bits | 8 bits
 *              ceil(T / 32) | 1 bit "one chain" | 1 end-marker bit | zeros to the byte boundary: the decoder finds the marker as the highest set
 *              bit of the region's last byte (which is never zero), takes the 9 bits below it and reads the main bits down from there.
 *              T = min(32 field, the stream's share of the last stage).
 * Cost over the ideal code length, measured on 768x512 images (tests/sim_v4.py, tests): 64 lanes ~6 bytes per stream that has symbols, 128 lanes
 * ~6.5, xwide v4 2.3-2.8 (noise) / 4.1 (natural-like, model-drawn) / 5.1 (a 1.5-bit source) of which 1.8 are the 256 lanes' 0.057 bit each (v3:
 * 3.5-4.1 / 5.1-6.6 / 9.8); an empty stream costs 250 / 498 / 994 bytes.
 * M: streams per image, | 0x100 for wide streams, | 0x200 for xwide streams; | 0x1200: xwide streams whose count the ENCODER picks from the image
 * (orc_auto_streams: M is what the image's size gives; expensive last-stage symbols -> M + ceil(M / 3), a last stage too cheap to fill M
 * payloads -> ceil(M / 2)) and writes into the header -- the container is an ordinary xwide v4 container of that count.  Returns total bytes or <0. */
int orc_auto_streams(int Mlo, const uint32_t *clow, const uint32_t *chigh, long n);
long orc_encode_image_rans(const uint8_t *rgb, int H, int W, const orc_weights *wts, int M,
                           uint8_t *out, long cap, int32_t seg_len[49]);
int orc_decode_image_rans(const uint8_t *in, const int32_t seg_len[49], const orc_weights *wts,
                          uint8_t *rgb, long rgb_cap, int *H_out, int *W_out);

void orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
