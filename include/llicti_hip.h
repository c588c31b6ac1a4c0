/*
 * llicti_hip.h -- C-ABI of the MI355X-native LLICTI encode/decode hot path (libllicti_hip.so).
 *
 * The reference (kamisli-icpl/LLICTI) has no FFI: its seam for this path is Python-method level
 * (SURVEY.md section 8b).  Every entry point below names the reference call site it replaces; the
 * ctypes binding a maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; device buffers are raw pointers obtained from the caller's allocator
 *     (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - every function returns 0 on success or a negative LLICTI_E* code; llicti_last_error() gives the
 *     message of the calling thread's last failure
 *   - nothing allocates device memory except llicti_create (status word, 1 MB of lift partials, streams and events
 *     of the AC decode pipeline), llicti_set_band_weights (weights) and a whole-batch call whose (mode, image sizes) the
 *     context has not seen lately (the plan's per-image tables, 0.1 - 0.3 MB of device + pinned host memory from a pool of
 *     blocks that are reused and never freed before llicti_destroy) -- all working memory comes from the caller-sized
 *     workspace (llicti_workspace_bytes / llicti_workspace_bytes_v)
 *   - launches are asynchronous on `stream`; functions that return host-visible results say so and
 *     synchronise the stream themselves
 *   - one context per GPU / host thread; a context is not thread-safe and its calls must not overlap on different
 *     streams (llicti_decode_images with the AC container fans out over two internal streams and joins back on `stream`;
 *     llicti_encode_images with the tuning switch "enc_side_levels" runs its coarse levels on one internal stream and joins back before
 *     the entropy coder)
 *   - every call makes the context's device current for its own duration and restores the caller's current device
 *   - calls that BLOCK the host: llicti_create / llicti_destroy / llicti_set_band_weights (device-wide synchronise:
 *     work in flight may still read the old weights), llicti_check_status and llicti_last_timing (they return
 *     host-visible results).  A whole-batch call of a new (mode, image sizes) builds its plan on the host (tens of
 *     microseconds) and enqueues ONE asynchronous upload of its tables on the call's stream: it does not synchronise the
 *     device (the cache holds 32 plans, least recently used out first; a block that leaves it waits in the pool until its
 *     last user has finished -- the reference's own test set has 119 image sizes among 500 images, interleaved)
 *   - 32 <= H, W <= 8160 (the header stores h4, w4 as uint8, LLICTI_nets.py:347).  llicti_encode_images /
 *     llicti_decode_images take B images of ONE size; llicti_encode_images_v / llicti_decode_images_v take a size per
 *     image (the reference's test loader yields images of arbitrary sizes one at a time, dataloaders/image_dl.py:40-45):
 *     same kernels, same bytes per image as a call of its own -- an image's container never depends on its batch
 *
 * Data layout in HBM
 *   rgb     uint8  [B][3][H][W]   planar
 *   planes  int16  [B][3][H][W]   Y-127, Co, Cg (YCoCg-R); the polyphase bands of every level are
 *                                 strided views of these planes: band (oi,oj) of level l, pixel (i,j)
 *                                 is planes[.., (2i+oi)<<l, (2j+oj)<<l]  (lazyDWT, LLICTI_nets.py:218-225)
 *   fplanes float  [B][3][H][W]   planes / 255 (one IEEE division, LLICTI_nets.py:143-144)
 *   params  float  [B][64][h*w]   raw CNN outputs on the band grid of one (level, band), CHANNEL-PLANAR: 4 heads x 16 planes
 *                                 (15 used), reference channel o is plane (o/15)*16 + o%15; position (i, j) at i*w + j -- a
 *                                 decoder wavefront's consecutive symbols read consecutive floats of the planes it needs
 *   tables  uint16 [B][hc*wc][stride]  integer CDF rows of one stream (stride = Lp rounded up to 8)
 */
#ifndef LLICTI_HIP_H
#define LLICTI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLICTI_OK            0
#define LLICTI_EINVAL       -1   /* bad argument (shape, pointer, level/band index) */
#define LLICTI_EHIP         -2   /* a HIP runtime call failed */
#define LLICTI_ENOWEIGHTS   -3   /* llicti_set_band_weights has not been called for every band */
#define LLICTI_ENOSPACE     -4   /* workspace or output buffer too small */
#define LLICTI_EFORMAT      -5   /* malformed container (decode) */
#define LLICTI_ENODEVICE    -6   /* no usable gfx950 device */

#define LLICTI_NLEVELS   5
#define LLICTI_NSTREAMS 45      /* 5 levels x 3 bands x 3 colour channels per image */
#define LLICTI_NSEG     49      /* 4 header segments + 45 streams (the reference's bytestream_list) */
#define LLICTI_NPARAMS  60      /* mixture parameters per position: 15 sigma | 15 mu | 15 weight | 5 a | 5 b | 5 d */
#define LLICTI_PARAM_STRIDE 64  /* channel planes per (image, level, band) in HBM: 4 heads x 16 (15 used): reference channel o is plane (o/15)*16 + o%15 */

typedef struct llicti_ctx llicti_ctx;

const char *llicti_last_error(void);
const char *llicti_version(void);

/* Context on HIP device `device`.  Fails with LLICTI_ENODEVICE when no GPU is present: there is no
 * CPU fallback. */
int llicti_create(llicti_ctx **ctx, int device);
int llicti_destroy(llicti_ctx *ctx);

/* Upload one band network in canonical packed form (host pointers, float32):
 *   w0 [352][K0] (K0 = 48 / 72 / 120), b0 [352], w1 [352][88], b1 [352], w2 [60][88], b2 [60].
 * Replaces LLICTIEntropyModel4.__init__ / load_state_dict for the eval path
 * (LLICTI_nets.py:651-675, :695-712; agents/base.py:51-76). */
int llicti_set_band_weights(llicti_ctx *ctx, int band, int K0, const float *w0, const float *b0,
                            const float *w1, const float *b1, const float *w2, const float *b2);

/* Level geometry helper (host only): Hl = ceil(H/2^l), band grid h = ceil(Hl/2), pad flag = Hl odd;
 * coded (cropped) size of `band`'s streams in hc, wc (LLICTI_nets.py:226-230, :396-397). */
int llicti_level_geom(int H, int W, int lvl, int band, int *Hl, int *Wl, int *h, int *w,
                      int *padH, int *padW, int *hc, int *wc);

/* ---- kernel-level entry points ------------------------------------------------------------------ */

/* K1-K3: uint8 RGB -> YCoCg-R planes (+ float copy) and per-image min/max of Co, Cg.
 * d_minmax: int32 [B][4] = minCo, minCg, maxCo, maxCg.
 * Replaces get_YCoCg_R_from_RGB__intOps + min()/max().item() + x/255 (LLICTI_nets.py:62-74, :137-144). */
int llicti_lift_u8(llicti_ctx *ctx, const uint8_t *d_rgb, int B, int H, int W,
                   int16_t *d_planes, float *d_fplanes, int32_t *d_minmax, void *stream);

/* K13: planes -> uint8 RGB.  Replaces get_RGB_from_YCoCg_R__intOps (LLICTI_nets.py:76-88, :174-175). */
int llicti_unlift_u8(llicti_ctx *ctx, const int16_t *d_planes, int B, int H, int W, uint8_t *d_rgb, void *stream);

/* K4+K5: interpolator CNN of one (level, band) for every position of the h x w band grid of every
 * image -> d_params [B][64][h*w] (LLICTI_PARAM_STRIDE channel planes).  fp32 MFMA, k-ordered accumulation (bit-exact to the numerics spec).
 * Replaces LLICTIEntropyModel4.get_params (LLICTI_nets.py:721-753, :822-825). */
int llicti_band_params_f32(llicti_ctx *ctx, const float *d_fplanes, int B, int H, int W, int lvl, int band,
                           float *d_params, void *stream);

/* Training / validation likelihood path (adjacent to the codec, same CNN kernel).
 * llicti_lift_train_f32: uint8 RGB -> float planes [B][3][H][W] = (Y - 127/255, Co, Cg) of the FLOAT lift with
 * torch.round (half to even), bit-exact to the reference's elementwise fp32 ops.
 * Replaces get_YCoCg_R_from_RGB + "x[:,0] -= mean_y_ycocg" (LLICTI_nets.py:40-49, :108-110).
 * llicti_selfinfo_f32: -log2 of the mixture likelihood of every target pixel of (lvl, band):
 * d_bits [B][3 (Y, Co, Cg)][h][w]; d_params from llicti_band_params_f32 on the same planes.
 * Replaces LLICTIEntropyModel4.get_self_infos (LLICTI_nets.py:862-880, :933-935) and
 * GaussianConditionalLosslessGMM.forward / _likelihood_fk (entropy_layer_nets.py:117-139, :160-183). */
int llicti_lift_train_f32(llicti_ctx *ctx, const uint8_t *d_rgb, int B, int H, int W, float *d_fplanes, void *stream);
int llicti_selfinfo_f32(llicti_ctx *ctx, const float *d_fplanes, const float *d_params, int B, int H, int W,
                        int lvl, int band, float *d_bits, void *stream);

/* K6-K8 (full table): integer CDF rows of stream (lvl, band, clr) for every coded position:
 * d_tables [B][hc*wc][row_stride] uint16, row_stride >= Lp (entries beyond Lp are 0xFFFF).
 * Replaces the cross-channel mean update + LLICTIEntropyModel4.get_cdfs(int_cdf=True)
 * (LLICTI_nets.py:385-392, :938-983; entropy_layer_nets.py:185-204). */
int llicti_cdf_u16(llicti_ctx *ctx, const int16_t *d_planes, const float *d_params, const int32_t *d_minmax,
                   int B, int H, int W, int lvl, int band, int clr, uint16_t *d_tables, int row_stride, void *stream);

/* K6-K9 (encoder form): for every coded position of (lvl, band) and each colour channel, only the two
 * table entries the coder reads, packed (c_high & 0xFFFF) << 16 | c_low (c_high == 0x10000 is stored as 0),
 * written in stream order to d_pairs[clr][B][hc*wc]. */
int llicti_cdf_pairs_u32(llicti_ctx *ctx, const int16_t *d_planes, const float *d_params, const int32_t *d_minmax,
                         int B, int H, int W, int lvl, int band, uint32_t *d_pairs, void *stream);

/* K10: torchac-compatible arithmetic ENCODER on explicit tables -- the reference's third-party seam
 * torchac.encode_int16_normalized_cdf(cdf, sym) (LLICTI_nets.py:406-407).  n_streams independent
 * streams of N symbols; d_cdf [n_streams][N][row_stride] uint16 (Lp valid entries), d_sym [n_streams][N]
 * int16; bytes of stream s go to d_out + s*out_stride (4-byte aligned, out_stride a multiple of 4 and
 * >= 2N + 8), its length to d_len[s]. */
int llicti_ac_encode_u16cdf(llicti_ctx *ctx, const uint16_t *d_cdf, int Lp, int row_stride, const int16_t *d_sym,
                            int n_streams, long N, uint8_t *d_out, long out_stride, int32_t *d_len, void *stream);

/* K11: the matching DECODER, torchac.decode_int16_normalized_cdf (LLICTI_nets.py:492-493).
 * d_in + s*in_stride holds stream s (4-byte aligned, in_stride a multiple of 4, d_len[s] <= in_stride); bytes past
 * d_len[s] are never interpreted: like torchac's reader the decoder shifts in zero bits once the stream is exhausted. */
int llicti_ac_decode_u16cdf(llicti_ctx *ctx, const uint16_t *d_cdf, int Lp, int row_stride, const uint8_t *d_in,
                            long in_stride, const int32_t *d_len, int n_streams, long N, int16_t *d_sym, void *stream);

/* ---- whole-batch entry points (LLICTI.compress / LLICTI.decompres, LLICTI_nets.py:125-179) -------
 * Containers stay in HBM.  Image b's container is the reference's bytestream_list flattened: its 49
 * segments concatenated tightly at d_out + b*out_stride, lengths in d_seg_len[b][49]; segment order
 * [S,h4,w4 u8] | 6 x int16 min/max | int16 padHW | raw DC band u8 CHW | 45 streams, scale 4..0 x band
 * x (Y,Co,Cg)  (LLICTI_nets.py:347-354, :411).  Both calls are asynchronous on `stream` (except for the first
 * call of a new shape, see "calls that BLOCK" above); device-side failures (malformed header or segment lengths,
 * stream overflow) are latched in the context and reported by llicti_check_status.  A malformed container is a
 * reported error, never a memory fault: every segment length is validated against in_stride before any byte of the
 * container is read, and no read leaves [d_in + b*in_stride, d_in + (b+1)*in_stride). */

/* mode: LLICTI_MODE_AC = the reference's container (45 torchac-algorithm streams per image, bit-exact
 * to the oracle / reference format); LLICTI_MODE_RANS*(M) = the rANS containers, NEW formats of this
 * build (BASELINE.json north_star: "torchac replaced by a HIP rANS coder"): header byte 0 = bit 7 (rANS) | bit 3 (format v3 or later; the
 * retired v2 tag has it clear and is rejected with LLICTI_EFORMAT) | bit 6 (extended) | v in bits 5,4,2,1,0 with M = v + 1; extended:
 * v = 0 / 1 = 64 / 128 streams (latency modes), v = 2 .. 15 = v - 1 wide streams (LLICTI_MODE_RANS_WIDE), v = 16 (byte 0 = 0xE8) = xwide streams
 * in the v4 layout (LLICTI_MODE_RANS_X), their COUNT in bits 10 .. 15 of the header's int16 pad field (1 .. 32 as they are, 33 / 34 = 64 / 128
 * streams; zero in every other container).  v = 17 .. 31 were the xwide tags of the v3 layout (rounds 4-5): retired, LLICTI_EFORMAT; a reader of
 * the older formats refuses a v4 container because its pad field contradicts the image size.  Then M independent
 * L-way interleaved rANS streams per image (L = 64 lanes, 128 for wide, 256 for xwide streams; segments 4 .. 4+M-1, the other stream segments
 * empty), same CDFs and symbols, decodable L*M symbols at a time.  States live in [2^31, 2^32) and renormalise bit by bit (the coder loses
 * ~2^-16 of a symbol's length, like the range coder); the L INITIAL states of a stream carry the last T symbols of the stream's last stage,
 * coded by a single-state tail coder.  64 / 128 lanes (v3): stream = u16 (T | pad << 11) | bit region | L x 31-bit final states.  xwide (v4):
 * stream = bit region | 256 x 31-bit final states; the tail coder's output is not cut to the 7,936 payload bits -- T is a multiple of 32 and
 * what exceeds the payload lies at the bottom of the bit region --; the tail is one chain that starts from the stream's last symbol itself, or
 * two seeded chains where symbols are expensive; the header field (T / 32, the one-chain flag) sits on top of the bit region under an end marker.
 * Cost over the ideal code length: about 6 bytes per 64-lane stream that has symbols, 2.5 - 5 per xwide v4 stream (the reference format's 45
 * stream terminations cost about 25 bytes per image).  Format: oracle/llicti_oracle.h, DESIGN.md section 5. */
#define LLICTI_MODE_AC        0
#define LLICTI_MODE_RANS(M)  (0x100 | (M))      /* M in 1 .. 32: one stream per segment; {64, 128}: latency modes for single / large
                                                  images, M / 32 streams per segment behind a table of their u32 lengths (+6 bytes per stream) */
#define LLICTI_MODE_RANS_WIDE(M) (0x300 | (M))  /* M in 1 .. 14 WIDE streams: 128 lanes per stream (two 64-symbol chunks per coder step, 128 x
                                                  31-bit states, about twice the tail symbols), header byte 0 = bit 6 set with v = M + 1; two decoder lanes per symbol */
#define LLICTI_MODE_RANS_X(M) (0x500 | (M))     /* M in 1 .. 32, 64, 128 XWIDE streams (v4 layout): 256 lanes per stream, header byte 0 = 0xE8, the count in the pad field's
                                                  bits 10 .. 15 (64 / 128: two / four streams per segment); ONE decoder lane per symbol, four wavefronts per stream: the fewest
                                                  vector instructions per symbol of the three (15 per 768x512 image are inside +0.001 bpp of the reference format on natural-like content) */
#define LLICTI_MODE_RANS_X_AUTO(M) (0x10500 | (M))  /* ENCODE ONLY.  xwide v4 streams whose count the ENCODER picks per image, on the device, from the image itself: M (1 .. 32) is the
                                                  count the image's size gives (llicti_amd.codec.image_streams); an image whose last stage's symbols are expensive (sum of
                                                  16 - floor(log2 freq) >= 11 per symbol: an xwide stream costs ~2.5 bytes there) gets M + ceil(M / 3) streams (at most 32), one
                                                  whose symbols are cheap (< 4: ~5 bytes per stream, long serial tails) ceil(2 M / 3), one whose last stage cannot fill the
                                                  payloads of 7,936 bits ceil(M / 2), every other M.  A pure function of the image: its
                                                  container does not depend on the batch, the device or anything coded before.  The container is an ordinary
                                                  LLICTI_MODE_RANS_X(count) container -- its header says which (llicti_header_mode) -- and is decoded as such. */

/* Bytes of device workspace the calls below need for B images of H x W in `mode` (_v: of Hs[b] x Ws[b]). */
size_t llicti_workspace_bytes(int B, int H, int W, int mode);
size_t llicti_workspace_bytes_v(int B, const int *Hs, const int *Ws, int mode);
size_t llicti_workspace_bytes_vm(int B, const int *Hs, const int *Ws, const int *modes);      /* one mode per image, see llicti_encode_images_vm */
/* Upper bound of the container size of ONE image: the minimum out_stride / in_stride. */
size_t llicti_max_container_bytes(int H, int W);

int llicti_encode_images(llicti_ctx *ctx, const uint8_t *d_rgb, int B, int H, int W, int mode,
                         void *d_workspace, size_t workspace_bytes,
                         uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream);

/* H and W must be the size the headers describe (llicti_header_dims on a host copy of the first 17
 * bytes); every image of the call has that size. */
int llicti_decode_images(llicti_ctx *ctx, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                         int B, int H, int W, int mode, void *d_workspace, size_t workspace_bytes,
                         uint8_t *d_rgb, void *stream);

/* Batches of MIXED sizes (the reference's eval loop, agents/llicti_agent.py:122-164, meets a different H x W at almost every step;
 * what it does one image at a time these calls do for B images at once).  Hs, Ws: host arrays of B sizes.  rgb_off: host array of B byte
 * offsets into d_rgb, image b's uint8 [3][Hs[b]][Ws[b]] block at d_rgb + rgb_off[b]; NULL = the blocks tightly packed in call order.
 * Containers as above: image b's at d_out + b*out_stride (out_stride >= llicti_max_container_bytes of the largest image), its 49 segment
 * lengths in d_seg_len[b].  Image b's bytes equal those of llicti_encode_images(B = 1) on that image, whatever else is in the batch.
 * rANS containers only: the reference-format container (LLICTI_MODE_AC) codes equal sizes per call (LLICTI_EINVAL otherwise). */
int llicti_encode_images_v(llicti_ctx *ctx, const uint8_t *d_rgb, const size_t *rgb_off, int B, const int *Hs, const int *Ws, int mode,
                           void *d_workspace, size_t workspace_bytes,
                           uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream);
/* ... with a container mode PER IMAGE (host array of B modes): rANS containers of one lane kind whose STREAM COUNTS may differ -- every image's header
 * carries its own count.  What it is for: a stage of the decoder takes as long as its longest stream, and in a batch of mixed sizes that is a
 * stream of the largest image; with each image's count given by its own size (llicti_amd.codec.auto_modes) the streams are about equally long,
 * and every image stays inside its own byte budget (a 768x768 image affords 20 xwide v4 streams, a 321x481 one 6).  Image b's bytes are those of
 * llicti_encode_images(B = 1) on it in modes[b]. */
int llicti_encode_images_vm(llicti_ctx *ctx, const uint8_t *d_rgb, const size_t *rgb_off, int B, const int *Hs, const int *Ws, const int *modes,
                            void *d_workspace, size_t workspace_bytes,
                            uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream);
int llicti_decode_images_vm(llicti_ctx *ctx, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                            int B, const int *Hs, const int *Ws, const int *modes, void *d_workspace, size_t workspace_bytes,
                            uint8_t *d_rgb, const size_t *rgb_off, void *stream);
/* Hs[b] x Ws[b] must be the size container b's header describes (llicti_header_dims); a mismatch flags that image (llicti_image_status). */
int llicti_decode_images_v(llicti_ctx *ctx, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                           int B, const int *Hs, const int *Ws, int mode, void *d_workspace, size_t workspace_bytes,
                           uint8_t *d_rgb, const size_t *rgb_off, void *stream);

/* Where llicti_encode_images / llicti_decode_images of B images of H x W in `mode` keep the YCoCg-R planes inside the caller's workspace
 * (byte offsets): int16 [B][3][H][W] (Y - 127, Co, Cg) and float32 [B][3][H][W] = planes / 255 -- the second is the `x_ycocg` the
 * reference's compress() returns beside the streams (LLICTI_nets.py:143-144, :159), so a caller that wants it reads it from the workspace
 * behind the encode instead of lifting the image a second time.  Valid until the next whole-batch call on that workspace. */
int llicti_workspace_planes(llicti_ctx *ctx, int B, int H, int W, int mode, size_t *off_planes, size_t *off_fplanes);
/* Where a whole-batch call on B images of Hs[b] x Ws[b] in modes[b] (llicti_encode_images_vm / llicti_decode_images_vm; the same placement as the
 * _v entry points with one mode) leaves the CNN outputs of its LAST launch -- level 0, band x10 -- of image `image`: byte offset of its
 * float32 [64][npos] block (channel planes, LLICTI_PARAM_STRIDE; position (i, j) of the h x w band grid at i * w + j) inside the caller's
 * workspace, and npos = h * w.  What it is for: holding the mixed-size form of the band CNN (tile lists, per-image geometry, the odd-edge
 * staging path) against the reference's own get_params outputs on a full-size odd shape (tests: ..._ragged_vs_reference).  Valid until the
 * next whole-batch call on that workspace. */
int llicti_workspace_params_v(llicti_ctx *ctx, int B, const int *Hs, const int *Ws, const int *modes, int image, size_t *off_params, long *npos);

/* Synchronises `stream` and returns the latched device-side status of the calls issued since the
 * last check (LLICTI_OK, LLICTI_EFORMAT, LLICTI_ENOSPACE). */
int llicti_check_status(llicti_ctx *ctx, void *stream);

/* Per-image status of the last llicti_decode_images call: h_status[b] = LLICTI_OK or LLICTI_EFORMAT for image b (a batch with one
 * malformed container decodes the others correctly; the bad image's pixels are deterministic garbage).  Synchronises `stream`.
 * The words are latched into the context at the end of the decode: the workspace may be freed or reused before this call. */
int llicti_image_status(llicti_ctx *ctx, int32_t *h_status, int n, void *stream);

/* Host-side self-test of arithmetic helpers that have no device dependency (the magic division of the stage geometry). */
int llicti_selftest(void);

/* Image size from a container's first 17 header bytes (host memory). */
int llicti_header_dims(const uint8_t *h_hdr17, int *H, int *W);
/* The mode a container's first 17 header bytes name -- LLICTI_MODE_AC or LLICTI_MODE_RANS*(count) -- i.e. what to hand to llicti_decode_images*
 * (LLICTI_EFORMAT for a header this build does not read: the retired v2 and xwide-v3 layouts). */
int llicti_header_mode(const uint8_t *h_hdr17, int *mode);

/* Device-resident timing of the last llicti_encode_images / llicti_decode_images call, measured with
 * HIP events on `stream`: ms[0] = whole call, ms[1] = sum of the band-CNN kernel launches,
 * n_launch = number of band-CNN launches.  Used by bench.py for the roofline figure. */
int llicti_last_timing(llicti_ctx *ctx, float ms[4], int *n_launch);
/* Where the last whole-batch call spent its device time (profiling on): summed event durations per kernel group --
 * cat_ms[LLICTI_PROF_CNN] band-CNN launches, [RANS_STAGE] rans_decode_stage_kernel, [RANS_TAIL] rans_tail_kernel,
 * [PAIRS] cdf_pairs_kernel, [RANS_ENC] rans_encode + pack, [AC] the range-coder / table kernels of the reference-format
 * container, [MISC] lift / unlift / header / unpack / init -- and the duration of every band-CNN launch in launch order
 * (scale 4..0 x band 0..2; up to cnn_cap entries, their number in *n_cnn).  Synchronises like llicti_last_timing. */
#define LLICTI_NPROF 7
#define LLICTI_PROF_CNN 0
#define LLICTI_PROF_RANS_STAGE 1
#define LLICTI_PROF_RANS_TAIL 2
#define LLICTI_PROF_PAIRS 3
#define LLICTI_PROF_RANS_ENC 4
#define LLICTI_PROF_AC 5
#define LLICTI_PROF_MISC 6
int llicti_last_timing_detail(llicti_ctx *ctx, float cat_ms[LLICTI_NPROF], float *cnn_launch_ms, int cnn_cap, int *n_cnn);
/* ... and the band-CNN time of that call per level (a level's launches may be split into sub-batches: their count is not fixed). */
int llicti_last_cnn_level_ms(llicti_ctx *ctx, float level_ms[LLICTI_NLEVELS]);
/* Tuning switches that never change a result.  "cnn_tile_rows" (0 = per launch, 16, 4): force the band CNN's 16-row throughput tiles
 * or its 4-row latency tiles (default: 4 rows whenever the 16-row tiles could not give every compute unit a workgroup).
 * "ac_anchor_min_batch" (default 96): from this many images per call on,
 * llicti_decode_images decodes the AC container over anchor rows (every 8th table entry from cdf_anchor_kernel, the 8
 * entries of the located bucket evaluated by the decoding wavefront) instead of full table rows; values above the
 * default are clamped to it (the workspace is sized for the default).  "force_ragged" (default 0; 1: a batch of equal sizes takes the
 * code path of a mixed-size batch too -- per-image tables, tile lists -- so that tests can run every case through both).
 * "enc_side_levels" (default 0; 1: llicti_encode_images runs levels 4..1 on an internal stream next to level 0's launches and joins it
 * before the entropy coder: -0.3 % of a step on MI355X, at the price of kernel traces whose side-queue durations include waiting). */
int llicti_set_tuning(llicti_ctx *ctx, const char *key, int value);
/* What the whole-batch calls have cost the host / device since llicti_create: "device_syncs" (hipDeviceSynchronize inside a whole-batch call),
 * "device_allocs" (hipMalloc inside one), "plan_builds" / "plan_hits" (calls whose (mode, sizes) were new / cached), "block_waits" (a new plan
 * had to wait for the last user of a pooled table block), "plans_cached", "blocks_pooled".  A warm context coding batch after batch of ever new
 * image sizes adds to plan_builds only (tests/test_hip_parity.py::test_many_sizes_no_device_sync_and_plan_reuse). */
int llicti_get_counter(llicti_ctx *ctx, const char *name, long *value);
/* enable / disable the per-kernel event timing above (off by default: it adds event records). */
int llicti_set_profiling(llicti_ctx *ctx, int enable);

#ifdef __cplusplus
}
#endif
#endif
