// canny_common.h -- shared declarations of the hipcanny device code and its host launcher.
// Hand-written for gfx950 (CDNA4, wave64).  Not a translation of the reference's CUDA kernels:
// see DESIGN.md for the layout (strips, chunks, bit planes) and for why each choice was made.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hc {

typedef uint32_t u32;
typedef uint64_t u64;

// ---- geometry of the fused path ---------------------------------------------------------------
// A wave owns a vertical STRIP of the frame: lane l holds the 4 adjacent pixels at columns
// strip*STRIP_W - 4 + 4*l .. +3 of the current row, packed in one dword.  Lanes 0 and 63 are halo
// lanes (4 px each side = 2 blur + 1 Sobel + 1 NMS), lanes 1..62 produce STRIP_W = 248 outputs.
constexpr int LANES = 64;
constexpr int PX_PER_LANE = 4;
constexpr int STRIP_W = (LANES - 2) * PX_PER_LANE;  // 248
constexpr int STRIP_HALO = PX_PER_LANE;             // 4 columns = one lane

// Bit planes: two plain bitmaps per frame, STRONG and CANDIDATE (candidate includes strong):
// bit c of a row <-> column c, rows padded to RD dwords.  A strip's 248 valid columns are 31 whole
// bytes, so each wave-row of k_front stores its 31 bytes at byte offset strip*31 of the row.
struct FrontParams {
  const uint8_t *in;       // u8 frames, pitched: mono, or interleaved BGR when bgr != 0 (stage 0 fused into the load)
  int bgr;
  size_t in_pitch;         // bytes per row   (multiple of 4)
  size_t in_frame_stride;  // bytes per frame (multiple of 4)
  u32 *sbits, *cbits;      // bit planes [frame][H][RD]
  int RD;                  // dwords per bit-plane row
  int W, H;
  int nstrips, nchunks, nframes;
  int subchunks;   // Mode R kernel: sub-chunks of 24 blur rows a wave marches through per work item
  int run_rows;    // = 24 * subchunks - 4 output rows per work item; nchunks = ceil(H / run_rows)
  int chunk_rows;          // Mode O kernel: output rows per work item (any value >= 1)
  int l2gradient;          // Mode O kernel: magnitude dx^2 + dy^2 instead of |dx| + |dy| (cv::Canny's L2gradient)
  int total_items;         // nframes * nstrips * nchunks
  // thresholds on S = sumX^2 + sumY^2 for "u8-wrapped gradient > T" (band test)
  u32 a_lo[3], a_hi[3];
  // split mode (k_blur + k_nms): the u8 blur plane between the two kernels and k_nms's own work split
  uint8_t *blur;             // [frame][strip][H][256]: one aligned 256-byte row per wave-row (bytes 4..251 = the strip's columns)
  size_t blur_frame_stride;  // >= nstrips * H * 256
  int nchunks_b, run_rows_b, total_items_b;
  // k_nms: when set, the strong pixels are also written as 255 (others 0) into this u8 map -- the provisional edge
  // map the hysteresis then only patches (W % 4 == 0: a lane stores its 4 pixels as one dword)
  uint8_t *prov_out; u32 prov_pitch; size_t prov_fs;
  // diagnostics (HC_OPT_DEBUG_TAPS): the fused kernel also stores its (fixed-up) blur rows here, plain [frame][H][pitch]
  uint8_t *dbg_blur; u32 dbg_pitch; size_t dbg_fs;
  const uint8_t *zeros;  // k_front8: >= 3 * 8192 + 32 bytes of zeros (what rows above / below the image read as); HALF form: + in_frame_stride
  // k_front8: memory that may be overwritten with anything -- where the branch-free row code stores rows that are not its
  // own: STRONG plane bytes (dump), CANDIDATE plane bytes (dump_c), provisional map (dump_p).  Plain form: 2 KiB, 2 KiB,
  // W + 8 bytes.  HALF form: each + the byte offset of half-wave B's frame (3 * H * RD * 4 / 3 * prov_fs at most).
  uint8_t *dump, *dump_c, *dump_p;
  // k_front8 / k_front8o: words the kernel zeroes before anything else (the run's hysteresis flags, worklist counts and
  // reason words: one memset kernel and one host call fewer per run); null: nothing
  u32 *zero_words; u32 zero_count;
  int half;        // k_front8: HALF form (two 240-column half-strips per wave, narrow frames); nstrips is unused then
  int one_wave;    // k_front8, mono / BGR with a provisional map: one-wave workgroups instead of four-wave ones
  int nhalf;       // HALF form: half-strips per frame = ceil(W / 240); total_items = ceil(in_frames * nhalf / 2) * nchunks (* 3 per-channel)
  // k_front8: a window that follows one with more than dense_enter half-lanes above the low threshold takes the dense path
  // (wave-wide NMS in registers), and the windows after it while they count more than dense_leave (0x7FFFFFFF: never)
  int dense_enter, dense_leave;
  u32 wrap_limit;  // S >= wrap_limit: gradient >= 256, the wrap bands apply (0xFFFFFFFF: saturating variant)
};

struct HystParams {
  u32 *sbits;
  const u32 *cbits;
  int RD, H, nframes;
  int tile_rows;   // rows per wave
  int waves;       // waves per workgroup; a workgroup tile is waves * tile_rows rows
  int nrtiles;     // row tiles per frame = ceil(H / (waves * tile_rows))
  int npanels;     // column panels per row tile = RD / 64 (a panel = 64 dwords = 2048 columns)
  u32 *flags;      // flags[k] != 0: launch k changed a tile-boundary row (another launch is needed)
  // How launches >= 1 find the tiles with work: a tile that changes a boundary row / column leaves a reason word with the
  // neighbours that look at it; wide frames also append them to the next launch's worklist.  [2] = launch parity;
  // wl_stride >= nframes * nrtiles * npanels words.
  u32 *wl_count;   // [launches + 1] wide frames: tiles on the list of (visited by) launch k; zero at the start of a run
  u32 *wl_reason;  // [2][wl_stride] per tile: 1 a tile above changed (its `top`), 2 below, 4 beside; zero at the start of a run
  u32 *wl_list;    // [2][wl_stride] wide frames: tile ids (frame * tiles per frame + tile)
  size_t wl_stride;
  int lists;       // this launch: 1 takes its tiles from the worklist (k_hyst MODE 1 / 2); 0 a workgroup per tile (MODE 0); 2 a workgroup per tile that also writes the next launch's list (MODE 3)
  int late_grid;   // worklist scheme: workgroups of launches >= 1 (0 = by the tile count, launch_hyst)
  int iter;        // index of this launch
  u32 *stats;      // optional diagnostics (3 words per launch) or null
  // fused expand: every launch also writes the 0/255 u8 rows it owns (launch 0: all rows of the tile,
  // later launches: the rows they changed), so no separate bit-plane -> u8 pass is needed
  uint8_t *out;
  size_t out_pitch, out_frame_stride;
  int W;
  int prov;        // the output already holds 255 for every strong pixel of the input planes (written by k_nms): launch 0 only rewrites rows it changes
  int first_pass;  // the planes come straight from k_front / k_pack: rows are not yet closed under the in-row fill
};

struct PackParams {  // tri-state u8 map (0/128/255) -> bit planes
  const uint8_t *in;
  size_t in_pitch, in_frame_stride;
  u32 *sbits, *cbits;
  int RD, W, H, nframes;
};

// ---- host-callable launchers (defined in canny_kernels.hip) -----------------------------------
hipError_t launch_selftest(u32 *d_result, hipStream_t s);
hipError_t check_gauss_coeffs(const float gk[25]);
hipError_t launch_front_o(const FrontParams &p, hipStream_t s);
#ifdef HC_LEGACY_FRONT  // legacy_front.hip: the round-1 front kernels of Mode R, built into libhipcanny_legacy.so only (parity tests)
hipError_t launch_front(const FrontParams &p, hipStream_t s);
hipError_t launch_blur(const FrontParams &p, hipStream_t s);
hipError_t launch_nms(const FrontParams &p, hipStream_t s);
size_t front_lds_bytes();
int front_run_rows(int subchunks);
#endif
// front8.hip: the whole front path as one kernel, 8 px per lane (strips of 496 columns, runs of 6 * windows - 4 rows)
hipError_t launch_front8(const FrontParams &p, hipStream_t s);
hipError_t launch_front8o(const FrontParams &p, hipStream_t s);  // Mode O on the same skeleton (one-channel sources)
int front8_run_rows(int windows);
int front8_strips(int W);
int front8_half_strips(int W);
// front_mx.hip: Mode R, one-channel frames, the blur and Sobel contractions as i8 MFMAs (strips of 216 columns, runs of
// 16 * blocks rows); big batches
hipError_t launch_front_mx(const FrontParams &p, hipStream_t s);
int front_mx_strips(int W);
int front_mx_run_rows(int blocks);
hipError_t launch_hyst(const HystParams &p, hipStream_t s);
// the first `rounds` launches of the workgroup-per-tile form as ONE launch with device-wide barriers between the rounds
// (small runs: at most HYST_LOOP_MAX_TILES tiles); bar: two zeroed words (arrival counter, abort flag)
constexpr int HYST_LOOP_MAX_TILES = 128;
hipError_t launch_hyst_loop(const HystParams &p, int rounds, u32 *bar, hipStream_t s);
hipError_t launch_pack(const PackParams &p, hipStream_t s);
// pitched device-to-device copy of n frames (any alignment on either side); rows, n <= 65535
hipError_t launch_copy_rows(void *dst, size_t dpitch, size_t dfs, const void *src, size_t spitch, size_t sfs, size_t row_bytes, int rows, int n, hipStream_t s);
void hyst_tile_geometry(int geom, bool beside_front, long frames_x_rows, int H, int *tile_rows, int *waves);

// plain per-stage kernels (exact, unfused): the `finalStage` taps MONO..THRESH of CannyEdge::run
hipError_t launch_gray(const uint8_t *bgr, size_t bpitch, size_t bfs, uint8_t *mono, size_t mpitch, size_t mfs, int W, int H, int n, hipStream_t s);
hipError_t launch_gauss(const uint8_t *mono, size_t mpitch, size_t mfs, uint8_t *blur, size_t bpitch, size_t bfs, int W, int H, int n, hipStream_t s);
hipError_t launch_sobel(const uint8_t *blur, size_t bpitch, size_t bfs, int16_t *sx, int16_t *sy, size_t spitch_elems, size_t sfs_elems, int W, int H, int n, hipStream_t s);
hipError_t launch_graddisp(const int16_t *sx, const int16_t *sy, size_t spitch_elems, size_t sfs_elems, uint8_t *out, size_t opitch, size_t ofs, int W, int H, int n, hipStream_t s);
hipError_t launch_nms(const int16_t *sx, const int16_t *sy, size_t spitch_elems, size_t sfs_elems, uint8_t *out, size_t opitch, size_t ofs, int W, int H, int n, int saturate, hipStream_t s);
hipError_t launch_thresh(const uint8_t *nms, size_t npitch, size_t nfs, uint8_t *out, size_t opitch, size_t ofs, int W, int H, int n, int low, int high, hipStream_t s);

}  // namespace hc
