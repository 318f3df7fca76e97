// hipcanny.hip -- host side of libhipcanny.so: the C ABI declared in include/hipcanny.h.
// Replaces the host half of the reference operator (src/cvp/cannyEdgeH.cu): allocation, upload,
// the stage switch of CannyEdge::run, the hysteresis launch loop and the output copy.
// There is no CPU fallback anywhere in this file: without a gfx950 device hc_create fails.
#include "../../include/hipcanny.h"
#include "canny_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace hc;

namespace {
// the two copy streams all HC_OPT_COPY_STREAMS contexts of a device share (created on first use, kept for the process)
constexpr int MAX_DEVICES = 64;
hipStream_t g_h2d[MAX_DEVICES] = { nullptr }, g_d2h[MAX_DEVICES] = { nullptr };
std::mutex g_copy_streams_mutex;  // contexts of different host threads may ask for them at the same time
thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
  g_err = msg;
  return code;
}
#define HIPCK(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return fail(HC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
  } while (0)

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr int MAX_HYST_LAUNCHES = 96;  // (48 until a weak edge wobbling along a tile boundary needed 52: one launch per crossing)
constexpr int FLAG_WORDS = MAX_HYST_LAUNCHES * 4;  // [0 .. MAX) launch flags, then 3 diagnostic words per launch
// d_flags continues with what must also be zero when a run starts (one memset): the worklist counts of the hysteresis
// launches, then the per-tile reason words of both launch parities (HystParams::wl_count / wl_reason)
constexpr int WL_COUNT_WORDS = 128;
static_assert(WL_COUNT_WORDS >= MAX_HYST_LAUNCHES + 1 + 2, "a count per launch, one beyond the last, and the two words of k_hyst_loop's barrier");

// Everything one in-flight fused run owns.  Two slots let run i+1's front kernel overlap run i's hysteresis (pipelined
// mode); the plain mode only uses slot 0.  Big batches rotate through two -- or three, while the hysteresis chain of a run
// is seen to outlast the front kernel of the next (finish_slot; 8K x 3: 6.9 -> 7.8 k frames/s, 8K grey 25.3 -> 27.5 k;
// where the front kernel bounds the step a third slot costs 1 %, a fourth 4 %: profiles/r03/experiments.md).
// SMALL batches use four slots, each with a hysteresis stream of its own: there a step is the latency of the
// hysteresis' chain of dependent launches (8 frames: 0.33 ms for a 0.04 ms front kernel), and chains of different runs
// share the device without noticing each other.
constexpr int NSLOT = 4;
struct Slot {
  bool complete = false;                       // every allocation below succeeded (alloc_slot)
  u32 *d_sbits = nullptr, *d_cbits = nullptr;  // bit planes [max_batch][H][RD]
  u32 *d_wl_list = nullptr;  // hysteresis worklists (HystParams::wl_list)
  size_t wl_cap = 0;         // tiles a run can have
  u32 *d_flags = nullptr, *h_flags = nullptr;
  hipEvent_t ev_front = nullptr, ev_done = nullptr;  // front kernel finished / hysteresis + expand finished
  bool pending = false;                              // convergence flag not yet checked by the host
  HystParams ph{};
  void *copy_dst = nullptr;  // caller buffer when the expand went to the internal one
  size_t copy_pitch = 0, copy_fs = 0;
  int n = 0;
  int k_launches = 0;            // hysteresis launches queued for this run
  bool prov = false;             // this run's k_nms wrote the provisional output
  hipStream_t stream = nullptr;  // stream the hysteresis of this run was queued on
  hipStream_t s_hyst = nullptr;  // this slot's hysteresis stream (pipelined mode)
  int hyst_level = 0;            // tile height level of this run's hysteresis (hc_ctx::hyst_obs index)
  int mixed_from = 0;            // > 0: launches below it ran a workgroup per tile, launch `mixed_from` wrote the first list, the rest took lists
  uintptr_t out0 = 0, out1 = 0;  // output range of this (pipelined, still pending) run: a later run into the same memory waits for it
  unsigned long long seq = 0;    // number of the pipelined run that uses the slot (hc_ctx::run_seq)
};
}  // namespace

struct hc_ctx {
  int device = 0, W = 0, H = 0, C = 1, max_batch = 1, mode = HC_MODE_R;
  int low = 10, high = 40;
  int nms_saturate = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;  // context stream (own, or the caller's)
  // internal pitched frames
  uint8_t *d_in = nullptr, *d_mono = nullptr, *d_out = nullptr;
  size_t in_pitch = 0, in_fs = 0, mono_pitch = 0, mono_fs = 0, out_pitch = 0, out_fs = 0;
  // stage-tap scratch (lazy)
  uint8_t *d_blur = nullptr, *d_nms = nullptr;
  int16_t *d_sx = nullptr, *d_sy = nullptr;
  // fused path
  Slot slot[NSLOT];
  int nslot_use = 2;  // slots the pipelined runs rotate through (4 for small batches)
  int dense_enter = 512, dense_leave = 384;  // HC_DENSE_ENTER / HC_DENSE_LEAVE (experiments)
  int pipe_slots = 0;  // HC_OPT_PIPELINE_SLOTS 2 / 3: that many slots whatever the batch size (0: by the rule)
  // timestamps of the last pipelined runs, by run number & 7: front kernel finished / hysteresis finished (watch_chain)
  hipEvent_t ring_f[8] = {}, ring_d[8] = {};
  unsigned long long ring_seq[8] = {};
  int front_wpb_mode = -1;   // HC_OPT_FRONT_WPB: -1 = by the slack of the hysteresis stream, 1 / 4 = fixed
  bool front_one = false;    // the automatic choice: one-wave workgroups for k_front8 (mono / BGR, pipelined big batches)
  float slack_ema = 0.0f;    // share of a front kernel's time by which the previous run's hysteresis chain ended before it (smoothed)
  int last_front_waves = 4;  // waves per workgroup of the most recent k_front8 launch
  int big_slots = 2;   // slots of big pipelined batches: 2, or 3 while the hysteresis chain bounds the step (finish_slot)
  int chain_bound_runs = 0, chain_light_runs = 0, chain_light_needed = 16;
  float period_ms[4] = { 0, 0, 0, 0 }, period_two = 0.0f;  // the last four steps (front kernel end to front kernel end); their mean before the trial of a third slot
  int trial_runs = -1;                                      // >= 0: runs since the third slot was taken on trial
  int retry_wait = 0, retry_backoff = 64;                   // runs until the next trial after one that did not pay (doubling)
  int chain_told = 0;  // diagnostics (HC_OPT_PIPELINE_SLOTS 20 / 21): +1 / -1 = every chain counts as ending after / before the next front kernel
  unsigned long long run_seq = 0;
  int cur = 0;
  bool pipeline = false;
  int per_channel = 0;  // 3-channel input: one edge map per channel (3 output frames per input frame)
  u32 wl_prev[MAX_HYST_LAUNCHES + 1] = { 0 };  // worklist lengths of the last finished run's launches
  size_t wl_prev_tiles = 0;                    // ... and its tile count (0: none / not a wide-frame run)
  bool hyst_lists_last = false;        // the last run used the worklist scheme
  int hyst_late_grid = 0;              // tests (HC_OPT_TEST_HYST_LATE_GRID): workgroups of the hysteresis launches >= 1
  bool hyst_loop = true;               // small runs: one looping hysteresis launch (HC_OPT_TEST_HYST_LOOP 0 turns it off)
  int hyst_obs[3] = { 0, 0, 0 };       // hysteresis launches the last runs needed with base_waves << i waves per workgroup (0: not seen)
  int hyst_obs_base = 0, hyst_obs_rows = 0;  // the base shape those observations belong to
  bool split_set = false;  // HC_OPT_FRONT_SPLIT was set by the caller
  int split = 2;        // Mode R front path: 2 = k_front8 (one kernel, 8 px per lane; default), 1 = k_blur + k_nms, 0 = the 4-px fused k_front
  int l2gradient = 0;   // Mode O: cv::Canny's L2gradient flag
  int half_mode = -1;   // HC_OPT_FRONT_HALF: -1 automatic, 0 never, 1 whenever the buffers allow it
  int dense_mode = -1;  // HC_OPT_FRONT_DENSE: -1 automatic, 0 never, 1 every window
  int mx_mode = 0;      // HC_OPT_FRONT_MX: 1 = k_front_mx whenever the run allows it (opt-in: include/hipcanny.h)
  uint8_t *d_dump = nullptr;    // k_front8's dump areas (FrontParams::dump / dump_c / dump_p), followed by its page of zeros (FrontParams::zeros)
  size_t dump_region = 0;       // 0: the plain layout (16 KiB + 32 KiB); otherwise four regions of this size (the HALF form's lane offsets reach a frame further)
  uint8_t *d_bplane = nullptr;  // split mode: u8 blur plane between the two kernels (lazy)
  size_t bplane_fs = 0, bplane_frames = 0;
  // HC_OPT_DEBUG_TAPS: copies of the bit planes as the front kernels left them, and (fused kernel) a plain blur plane
  bool debug_taps = false;
  u32 *dbg_s = nullptr, *dbg_c = nullptr;
  uint8_t *dbg_blur = nullptr;
  int dbg_frames = 0;        // output frames captured by the last run (0: nothing captured)
  bool dbg_blur_split = false, dbg_blur_valid = false;
  int RD = 0;
  int nstrips = 0, chunk = 0, hyst_launches = 6;
  // diagnostic environment variables, read ONCE at hc_create (never in the launch path): HC_HYST_DIAG (per-launch
  // counters for hc_hysteresis_stats; slows the launches), HC_HYST_GEOM (hysteresis workgroup shape, e.g. "32x8")
  bool hyst_diag = false;
  int hyst_geom = 0;
  bool hyst_launches_set = false;  // hc_set_tuning called: queue exactly that many launches
  int last_work_launches = 0, last_continued = 0;
  int last_in_staged = 0, last_out_staged = 0, last_front_form = -1;  // what the last run did with the caller's buffers / which front kernels it used
  int hyst_need_rows = 0;  // launches that found work in recent runs (continuation rounds included) x rows per tile: how far changes travelled
  u32 h_stats[3 * MAX_HYST_LAUNCHES] = { 0 };
  int uploaded = 0, last_run_n = 0;
  int last_slot = 0;          // slot of the most recent fused run
  // hc_download_begin .. hc_download_end
  uint8_t *dl_host = nullptr; size_t dl_row = 0, dl_fs = 0; int dl_n = 0;
  bool dl_stale = false;  // a host-side hysteresis continuation rewrote maps after hc_download_begin queued their copy (whichever entry point ran it)
  // HC_OPT_COPY_STREAMS: uploads / downloads on the device's shared copy streams, tied to the context stream by events
  bool copy_streams = false;
  hipEvent_t ev_up = nullptr, ev_ready = nullptr, ev_ready2 = nullptr, ev_down = nullptr;
  bool profiling = false;
  // hipEvent ring: up to EV_PER_RUN events per profiled run.  Interval i = ev[i] -> ev[i + 1] covers the reference stages
  // in RunProf::mask[i] (one kernel may cover several: its time is divided equally among them, see hc_stage_time_ms)
  static constexpr int EV_RUNS = 256;
  static constexpr int EV_PER_RUN = 8;
  struct RunProf { int nint = 0; bool after_gap = false; uint8_t mask[EV_PER_RUN - 1] = { 0 }; uint8_t kind[EV_PER_RUN - 1] = { 0 }; };
  bool prof_gap = false;  // a run went untimed since the last timed one (ring full)
  enum { K_STAGE0 = 0, K_FRONT_A = 1, K_FRONT_B = 2, K_HYST = 3 };  // grey kernel / k_blur / k_nms, the fused front kernel or the tap kernels / hysteresis
  std::vector<hipEvent_t> evpool;
  std::vector<RunProf> runprof;
  int ev_head = 0, ev_count = 0;  // runs recorded since the last collect
  float stage_ms[6] = { 0, 0, 0, 0, 0, 0 };
  unsigned stage_ran = 0;         // stages the last profiled run executed (bit per stage)
  double prof_sum[3] = { 0, 0, 0 };
  double prof_split_sum[2] = { 0, 0 };  // k_blur, k_nms (split front path only)
  long prof_split_runs = 0;
  long prof_runs = 0;
  std::vector<float> step_ms;     // end-of-run to end-of-run intervals of consecutive profiled runs (steady-state step time)
  std::vector<float> front_each;  // the front kernels' time of every profiled HYSTER run (hc_profile_get_front_each)
  unsigned long long hyst_totals[4] = { 0, 0, 0, 0 };  // runs, continued runs, launches with work, launches queued
  hipEvent_t prev_end = nullptr;  // last event of the previous profiled run (its ring slot is not reused before the next collect: at most EV_RUNS - 1 runs are in flight)
};

namespace {

// Internal frame buffers.  Rows of whole 16-byte groups are stored TIGHT (pitch = row bytes): a batch is then one contiguous
// block, and hc_upload / hc_download move it with a single 1-D DMA instead of a strided 2-D copy per frame (16 frames of
// 1080p over PCIe: 23.8 -> ~50 GB/s each way, bench.py host_fed).  Other widths keep rows padded to 256 bytes, which
// also gives the 8-px kernels the whole pixel groups they load.
int alloc_frames(uint8_t **ptr, size_t *pitch, size_t *fs, size_t row_bytes, int H, int n, size_t tight_row_bytes = 0)
{
  *pitch = (tight_row_bytes && tight_row_bytes % 16 == 0) ? tight_row_bytes : round_up(row_bytes, 256);
  *fs = *pitch * (size_t)H;
  HIPCK(hipMalloc((void **)ptr, *fs * (size_t)n));
  return HC_OK;
}

int ensure_stage_scratch(hc_ctx *c)
{
  if (c->d_blur) return HC_OK;
  const size_t n = (size_t)c->max_batch;
  HIPCK(hipMalloc((void **)&c->d_blur, c->out_fs * n));
  HIPCK(hipMalloc((void **)&c->d_nms, c->out_fs * n));
  HIPCK(hipMalloc((void **)&c->d_sx, c->out_fs * n * 2));
  HIPCK(hipMalloc((void **)&c->d_sy, c->out_fs * n * 2));
  return HC_OK;
}

int alloc_slot_parts(hc_ctx *c, Slot &s)
{
  const size_t out_frames = (size_t)c->max_batch * (c->per_channel ? 3 : 1);
  const size_t plane_bytes = sizeof(u32) * (size_t)c->RD * c->H * out_frames;
  HIPCK(hipMalloc((void **)&s.d_sbits, plane_bytes));
  HIPCK(hipMalloc((void **)&s.d_cbits, plane_bytes));
  // row padding beyond the strips' bytes is never written by the kernels and must read as 0
  HIPCK(hipMemset(s.d_sbits, 0, plane_bytes));
  HIPCK(hipMemset(s.d_cbits, 0, plane_bytes));
  // tiles of a run: at most out_frames x row tiles (16 rows or more each) x column panels
  s.wl_cap = out_frames * ((size_t)(c->H + 15) / 16 + 1) * ((c->RD + 63) / 64);
  HIPCK(hipMalloc((void **)&s.d_wl_list, sizeof(u32) * 2 * s.wl_cap));
  HIPCK(hipMalloc((void **)&s.d_flags, sizeof(u32) * (FLAG_WORDS + WL_COUNT_WORDS + 2 * s.wl_cap)));
  HIPCK(hipHostMalloc((void **)&s.h_flags, sizeof(u32) * (FLAG_WORDS + WL_COUNT_WORDS), hipHostMallocDefault));
  {
    // the hysteresis launches are few, small and dependent (latency-bound); at the highest priority their workgroups are
    // placed ahead of the next run's 30k-wave front kernel instead of behind it
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIPCK(hipStreamCreateWithPriority(&s.s_hyst, hipStreamNonBlocking, greatest));
  }
  HIPCK(hipEventCreateWithFlags(&s.ev_front, hipEventDisableTiming));  // (cross-stream waits only: the timestamps watch_chain compares are ring_f / ring_d)
  HIPCK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
  return HC_OK;
}

void free_slot(Slot &s)
{
  for (void *q : { (void *)s.d_sbits, (void *)s.d_cbits, (void *)s.d_wl_list, (void *)s.d_flags }) (void)hipFree(q);
  if (s.h_flags) (void)hipHostFree(s.h_flags);
  if (s.ev_front) (void)hipEventDestroy(s.ev_front);
  if (s.ev_done) (void)hipEventDestroy(s.ev_done);
  if (s.s_hyst) (void)hipStreamDestroy(s.s_hyst);
  s = Slot{};
}

// A slot is either complete or empty: slots 1..3 are allocated lazily inside a run, and a run that found d_sbits set
// but the stream or the flag words missing would memset a null pointer and queue its hysteresis on the null stream.
int alloc_slot(hc_ctx *c, Slot &s)
{
  if (s.complete) return HC_OK;
  const int rc = alloc_slot_parts(c, s);
  if (rc != HC_OK) {
    const std::string why = g_err;
    free_slot(s);
    return fail(rc, why);
  }
  s.complete = true;
  return HC_OK;
}

// split mode: blur plane [frames][strip][H][256 B] (see canny_kernels.hip); every byte k_nms reads is written by k_blur
int ensure_blur_plane(hc_ctx *c)
{
  const size_t frames = (size_t)c->max_batch * (c->per_channel ? 3 : 1);
  if (c->d_bplane && c->bplane_frames == frames) return HC_OK;
  if (c->d_bplane) { (void)hipFree(c->d_bplane); c->d_bplane = nullptr; }
  c->bplane_fs = (size_t)c->nstrips * c->H * 256;
  HIPCK(hipMalloc((void **)&c->d_bplane, c->bplane_fs * frames));
  c->bplane_frames = frames;
  return HC_OK;
}

// k_front8's dump areas and page of zeros.  big: sized for the HALF form, whose half-wave B reaches its frame through lane
// offsets of up to one frame stride (three bit-plane / output frames in per-channel mode)
int alloc_dump(hc_ctx *c, bool big)
{
  if (c->d_dump && (c->dump_region != 0) == big) return HC_OK;
  if (c->d_dump) { (void)hipFree(c->d_dump); c->d_dump = nullptr; }
  size_t bytes = 16384 + 32768;
  c->dump_region = 0;
  if (big) {
    const size_t plane_fs = sizeof(u32) * (size_t)c->RD * c->H;
    c->dump_region = round_up(std::max(std::max(c->in_fs, 3 * c->out_fs), 3 * plane_fs) + 32768, 4096);
    bytes = 4 * c->dump_region;
  }
  HIPCK(hipMalloc((void **)&c->d_dump, bytes));
  HIPCK(hipMemset(c->d_dump, 0, bytes));
  return HC_OK;
}

int ensure_debug_buffers(hc_ctx *c)
{
  if (c->dbg_s) return HC_OK;
  const size_t frames = (size_t)c->max_batch * (c->per_channel ? 3 : 1);
  const size_t plane_bytes = sizeof(u32) * (size_t)c->RD * c->H * frames;
  HIPCK(hipMalloc((void **)&c->dbg_s, plane_bytes));
  HIPCK(hipMalloc((void **)&c->dbg_c, plane_bytes));
  HIPCK(hipMalloc((void **)&c->dbg_blur, c->out_fs * frames));
  return HC_OK;
}

void free_debug_buffers(hc_ctx *c)
{
  for (void *q : { (void *)c->dbg_s, (void *)c->dbg_c, (void *)c->dbg_blur }) (void)hipFree(q);
  c->dbg_s = c->dbg_c = nullptr;
  c->dbg_blur = nullptr;
  c->dbg_frames = 0;
}

// rows per work item: about 16 rounds of the whole chip (8192 resident waves) when the batch allows it -- the tail of a
// launch is one work item long, measured optimum 68-135 rows at 1024 frames -- but never runs shorter than 64 rows
// (each run repeats a 4-row warm-up)
int pick_run_rows(long units, int H, int want_rows)
{
  if (want_rows > 0) return std::min(std::max(want_rows, 2), H);
  const long nch = std::min<long>(std::max<long>((16 * 8192 + units - 1) / units, 1), std::max(1, H / 64));
  return (int)((H + nch - 1) / nch);
}

bool aligned4(const void *p, size_t a, size_t b) { return (((uintptr_t)p | a | b) & 3u) == 0; }

// "stored u8 gradient > T" as thresholds on S = sumX^2+sumY^2 (gradient g = isqrt(S>>2)):
// wrapping variant: g in [256k+T+1, 256k+255] for k = 0,1,2  ->  S >= a[k] (and below 4*(256(k+1))^2);
// saturating variant: min(g,255) > T  ->  S >= a[0], never for T = 255.
void band_thresholds(int T, bool saturate, u32 a[3])
{
  for (int k = 0; k < 3; ++k) {
    const u64 g = 256ull * k + (u64)T + 1;
    a[k] = (u32)std::min<u64>(4ull * g * g, 0xFFFFFFFFull);
  }
  if (saturate && T >= 255) a[0] = 0xFFFFFFFFu;
}

int copy_frames_d2d(hc_ctx *c, hipStream_t st, void *dst, size_t dpitch, size_t dfs, const void *src, size_t spitch, size_t sfs, size_t row_bytes, int n)
{
  // a kernel, not hipMemcpy2DAsync: its row-by-row DMA took 1.5 - 2.5 ms for 512 frames of 1918 x 1079 (k_copy_rows: 0.3 ms)
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = std::min(n - f0, 65535);
    HIPCK(launch_copy_rows((uint8_t *)dst + dfs * f0, dpitch, dfs, (const uint8_t *)src + sfs * f0, spitch, sfs, row_bytes, c->H, nf, st));
  }
  return HC_OK;
}

// slots the pipelined runs of n_out output frames rotate through: by pixels (16 8K x 3 frames are a big batch)
int pipeline_slots(const hc_ctx *c, int n_out) { return c->pipe_slots ? c->pipe_slots : (long long)n_out * c->H * c->W < 500ll * 1000 * 1000 ? NSLOT : c->big_slots; }

// What the hysteresis chain of a run did to the front kernel it ran beside, from timestamps of the runs themselves
// (ring_f / ring_d: recorded behind every pipelined run's front kernel and behind its last hysteresis launch).  Called
// when run i is complete: the chain of run i-1 ran beside the front kernel of run i, and all three events involved --
// end of front i-1, end of chain i-1, end of front i -- are complete.
//  * Two or three slots for big batches?  With two, the front kernel of run i+2 waits for the hysteresis of run i: while
//    that chain is the shorter of the two nothing waits, and a third slot only lets a second chain compete for the same
//    wave slots (-1 % at 1080p).  Where the chain outlasts the front kernel (8K: 30 dependent launches over 68 row tiles
//    and 4 column panels) the front kernels sit idle for the difference, and a third slot lets the next run start on
//    time.  Three runs in a row whose chain ended after the front kernel beside it -> a third slot ON TRIAL: kept if
//    the mean step of runs 7-10 with it is 3 % shorter than the last four steps without (8K x 3: -11 %, 8K grey -8 %,
//    256 frames of 1080p -5 %), otherwise given back, next trial after 64 runs, doubling; from three back to two after
//    16 runs in a row (doubling, up to 1024) whose chain ended first.
//  * One-wave or four-wave workgroups for k_front8?  One-wave workgroups take every slot a retiring wave leaves at once:
//    the front kernel gains 2-3 %, the hysteresis stream needs 40 % longer -- good while that stream has the time
//    (1080p grey: it ends 40 % of a front kernel early; +1.5 % frames/s), bad where it has none (BGR -> grey: -5 %).
//    By the smoothed share of the front kernel's time that the chain left unused: above 25 % -> one wave (1080p grey 41 %,
//    640 x 480 32 %, 4K 27 %; BGR -> grey 5 %), and back to four below 3 % (with one-wave workgroups the same streams
//    leave 17 %, 7 %, 15 %; BGR -> grey would fall 90 % behind).
void watch_chain(hc_ctx *c, const Slot &s)
{
  if (!s.seq || s.stream == c->stream || c->nslot_use >= NSLOT) return;
  const unsigned long long i = s.seq;
  const int a = (int)((i - 1) & 7), b = (int)(i & 7);
  if (c->ring_seq[b] != i || c->ring_seq[a] != i - 1 || i < 2) return;
  float front_ms = 0.0f, lead_ms = 0.0f;  // front kernel i (end to end); end of chain i-1 -> end of front kernel i
  if (hipEventElapsedTime(&front_ms, c->ring_f[a], c->ring_f[b]) != hipSuccess || hipEventElapsedTime(&lead_ms, c->ring_d[a], c->ring_f[b]) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  if (front_ms <= 0.0f) return;
  const bool outlasts = c->chain_told ? c->chain_told > 0 : lead_ms < 0.0f;
  c->period_ms[i & 3] = front_ms;  // front kernel end to front kernel end: the step
  const float period4 = 0.25f * (c->period_ms[0] + c->period_ms[1] + c->period_ms[2] + c->period_ms[3]);
  if (c->retry_wait > 0) --c->retry_wait;
  if (!c->pipe_slots) {
    if (c->nslot_use == 2 && c->big_slots == 2) {
      c->chain_bound_runs = outlasts ? c->chain_bound_runs + 1 : 0;
      if (c->chain_bound_runs >= 3 && i >= 5 && (c->retry_wait == 0 || c->chain_told)) {  // try a third slot
        c->period_two = period4;
        c->big_slots = 3;
        c->trial_runs = 0;
        c->chain_bound_runs = c->chain_light_runs = 0;
      }
    } else if (c->nslot_use == 3 && c->trial_runs >= 0) {  // the trial: ten runs, the last four measured
      if (++c->trial_runs >= 10) {
        c->trial_runs = -1;
        const bool better = c->chain_told ? c->chain_told > 0 : period4 < 0.97f * c->period_two;
        if (!better) {
          c->big_slots = 2;
          c->retry_wait = c->retry_backoff;
          c->retry_backoff = std::min(4096, 2 * c->retry_backoff);
        }
      }
    } else if (c->nslot_use == 3) {
      c->chain_light_runs = outlasts ? 0 : c->chain_light_runs + 1;
      if (c->chain_light_runs >= c->chain_light_needed) {
        c->big_slots = 2;
        c->chain_light_needed = std::min(1024, 2 * c->chain_light_needed);
        c->chain_bound_runs = c->chain_light_runs = 0;
      }
    }
  }
  const float slack = std::max(-1.0f, std::min(1.0f, lead_ms / front_ms));
  c->slack_ema = 0.75f * c->slack_ema + 0.25f * slack;
  if (!c->front_one && c->slack_ema > 0.25f) c->front_one = true;
  else if (c->front_one && c->slack_ema < 0.03f) c->front_one = false;
}

// Completes a queued fused run: waits for it, and if its queued hysteresis launches did not reach
// the fixpoint (flag of the last one still set -- adversarial inputs only), keeps iterating, then
// redoes the expand.
int finish_slot(hc_ctx *c, Slot &s)
{
  if (!s.pending) return HC_OK;
  s.pending = false;
  s.out0 = s.out1 = 0;  // (this function only returns when the run is complete)
  hipStream_t st = s.stream;
  HIPCK(hipEventSynchronize(s.ev_done));
  watch_chain(c, s);
  const int K = s.k_launches;
  int work = 0;
  for (int k = 0; k < K; ++k) work += s.h_flags[k] != 0;
  std::memcpy(c->h_stats, s.h_flags + MAX_HYST_LAUNCHES, sizeof(c->h_stats));
  c->last_work_launches = std::min(K, work + 1);
  c->last_continued = 0;
  c->hyst_totals[0] += 1;
  c->hyst_totals[3] += (unsigned long long)K;
  // worklist lengths of this run's launches (wide frames): the next run of the same shape sizes its grids by them
  c->wl_prev_tiles = (s.ph.npanels > 1 || s.ph.lists) ? s.ph.wl_stride : 0;
  for (int k = 0; k <= MAX_HYST_LAUNCHES; ++k) c->wl_prev[k] = s.h_flags[FLAG_WORDS + k];
  const int tile = s.ph.tile_rows * s.ph.waves;
  c->hyst_need_rows = std::max(c->last_work_launches * tile, c->hyst_need_rows - 32);  // follows the content up at once, down slowly
  // launches this run needed at its tile height (queue_hyst_expand picks the next runs' height from these)
  auto observe = [&]() {
    const int lvl = s.hyst_level, L = c->last_work_launches;
    if ((s.ph.waves >> lvl) != c->hyst_obs_base || s.ph.tile_rows != c->hyst_obs_rows) return;
    const int o = c->hyst_obs[lvl];
    // the content changed: what was seen at the other heights no longer holds
    if (o && (L * 10 > o * 13 + 20 || L * 10 < o * 7 - 20)) c->hyst_obs[0] = c->hyst_obs[1] = c->hyst_obs[2] = 0;
    c->hyst_obs[lvl] = L;
  };
  if (s.h_flags[K - 1] == 0) {
    c->hyst_totals[2] += (unsigned long long)c->last_work_launches;
    observe();
    return HC_OK;
  }
  c->last_continued = 1;
  c->hyst_totals[1] += 1;
  if (c->dl_host) c->dl_stale = true;
  for (int round = 0; round < 1000000; ++round) {
    HIPCK(hipMemsetAsync(s.d_flags, 0, sizeof(u32) * (FLAG_WORDS + WL_COUNT_WORDS + 2 * s.ph.wl_stride), st));  // flags, worklist counts and reasons
    HystParams hp = s.ph;
    hp.late_grid = c->hyst_late_grid > 0 ? c->hyst_late_grid : 0;  // (not the grid of the run's last queued launch)
    hp.first_pass = 0;
    hp.stats = nullptr;
    for (int k = 0; k < K; ++k) {
      hp.iter = k;
      // the schedule the run itself used (s.ph holds the parameters of its LAST launch)
      if (s.mixed_from > 0) hp.lists = k < s.mixed_from ? 0 : k == s.mixed_from ? 2 : 1;
      HIPCK(launch_hyst(hp, st));
    }
    HIPCK(hipMemcpyAsync(s.h_flags, s.d_flags, sizeof(u32) * (FLAG_WORDS + WL_COUNT_WORDS), hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    for (int k = 0; k < K; ++k) c->last_work_launches += s.h_flags[k] != 0;
    if (s.h_flags[K - 1] == 0) break;
  }
  c->hyst_need_rows = std::max(c->hyst_need_rows, c->last_work_launches * tile);
  c->hyst_totals[2] += (unsigned long long)c->last_work_launches;
  observe();
  if (s.copy_dst)
    if (int rc = copy_frames_d2d(c, st, s.copy_dst, s.copy_pitch, s.copy_fs, s.ph.out, s.ph.out_pitch, s.ph.out_frame_stride, (size_t)c->W, s.n)) return rc;
  HIPCK(hipStreamSynchronize(st));
  return HC_OK;
}

int finish_all(hc_ctx *c)
{
  // oldest first: slot `cur` is the next to be reused
  for (int k = 0; k < c->nslot_use; ++k)
    if (int rc = finish_slot(c, c->slot[(c->cur + k) % c->nslot_use])) return rc;
  for (Slot &q : c->slot)  // (slots of the other ring size hold nothing: the ring is drained before its size changes)
    if (int rc = finish_slot(c, q)) return rc;
  return HC_OK;
}

// bit planes of slot s -> fixpoint -> u8 image, queued on `st`
// flags_zeroed: the front kernel of this run already zeroed `zeroed_words` words of s.d_flags (FrontParams::zero_words)
int queue_hyst_expand(hc_ctx *c, Slot &s, hipStream_t st, uint8_t *out, size_t out_pitch, size_t out_fs, int n, bool small_tiles, size_t zeroed_words = 0)
{
  HystParams hp{};
  hp.sbits = s.d_sbits; hp.cbits = s.d_cbits; hp.RD = c->RD; hp.H = c->H; hp.nframes = n; hp.flags = s.d_flags;
  // one workgroup per (frame, tile of waves x tile_rows rows); the geometry follows the row width
  hyst_tile_geometry(c->hyst_geom, small_tiles, (long)n * c->H, c->H, &hp.tile_rows, &hp.waves);
  // Adaptive tile height: a launch carries a change across one tile boundary, so frames whose weak edges wind through
  // many tiles need many launches.  The library remembers how many launches the runs needed with the base shape and
  // with twice / four times its waves (hyst_obs, updated by finish_slot, forgotten when the content changes): above 20
  // launches the next taller shape is tried -- and kept only if it needs fewer than 60 % of the launches.  Mode O frames:
  // 24 launches with 64-row tiles, 5 with 128 rows: taller (501 against 480 k frames/s).  BGR frames blended into grey:
  // 25 either way, their chains wind around the tile boundaries whatever the height: the small workgroups, which find
  // room beside the front kernel more easily, and the worklists (255 against 224 k frames/s with 4-wave tiles).
  // (Round 2's first rule went by rows -- launches x tile height -- alone: it kept the BGR stream on tall tiles, and
  // made the Mode O stream flip between the two shapes every few runs, each flip a host-side continuation.)
  const int base_waves = hp.waves;
  if (base_waves != c->hyst_obs_base || hp.tile_rows != c->hyst_obs_rows) {  // another base shape (batch size, plain / pipelined): start over
    c->hyst_obs_base = base_waves; c->hyst_obs_rows = hp.tile_rows;
    c->hyst_obs[0] = c->hyst_obs[1] = c->hyst_obs[2] = 0;
  }
  int lvl = 0;
  while (lvl < 2 && (base_waves << (lvl + 1)) <= 8) {
    const int cur = c->hyst_obs[lvl], nxt = c->hyst_obs[lvl + 1];
    if (cur <= 20) break;                  // unknown (0) or few enough
    if (nxt != 0 && nxt * 5 > cur * 3) {  // the taller tiles did not pay
      // (frames of several panels: the tallest then -- an 8K grey stream whose weak edge wobbles along a tile boundary
      // needs 53 launches at every height, and runs them faster on a quarter of the tiles: 21.1 against 16.5 k frames/s)
      if (c->RD > 64) while (lvl < 2 && (base_waves << (lvl + 1)) <= 8) ++lvl;
      break;
    }
    ++lvl;
  }
  hp.waves = base_waves << lvl;
  s.hyst_level = lvl;
  hp.nrtiles = (c->H + hp.tile_rows * hp.waves - 1) / (hp.tile_rows * hp.waves);
  hp.npanels = (c->RD + 63) / 64;
  // launches queued per run: the user's number, or by default enough for an edge that crosses every row tile of a
  // tall frame (later launches exit at once after convergence; beyond the queue, hc_sync continues from the host)
  // (one more than the tiles an edge can cross monotonically: the last queued launch must find nothing to do, or the host
  // continues in hc_sync -- which stalls a pipelined stream of runs)
  const int need = (c->hyst_need_rows + hp.tile_rows * hp.waves - 1) / (hp.tile_rows * hp.waves);
  int K = std::min(MAX_HYST_LAUNCHES, std::max(std::max(c->hyst_launches, hp.nrtiles + hp.npanels + 1), need + 2));
  // One or a few frames per call, not pipelined (the reference's pattern): every queued launch that finds nothing to do
  // still costs ~5 us of pure latency, so only what the last runs needed is queued, + 2; frames that need more are
  // finished by the host-side continuation (cheap here: nothing else is in flight).
  if (!small_tiles && (long)n * c->H < 128 * 1024 && c->hyst_need_rows > 0) K = std::min(K, std::max(4, need + 2));
  // ... and in a pipelined stream whose needs are known, not the worst case of an edge down the whole frame (68 row tiles
  // at 8K: 70 launches queued, 50 of them idle at ~5 us each on the hysteresis stream) but what the last runs needed, + 4;
  // a frame that needs more is finished by the continuation, and the estimate follows it at once
  if (small_tiles && c->hyst_need_rows > 0) K = std::min(K, std::max(6, need + 4));
  // (a few frames per run, pipelined: the step is the host's time to queue the run -- 75 us for ~25 API calls -- so every
  // launch that is not needed counts)
  if (small_tiles && c->hyst_need_rows > 0 && (long)n * c->H < 128 * 1024) K = std::min(K, std::max(4, need + 2));
  if (c->hyst_launches_set) K = c->hyst_launches;
  hp.out = out; hp.out_pitch = out_pitch; hp.out_frame_stride = out_fs; hp.W = c->W;
  // worklists of launches >= 1: counts and reason words live behind the launch flags and are zeroed with them
  hp.wl_stride = (size_t)n * hp.nrtiles * hp.npanels;
  if (hp.wl_stride > s.wl_cap) return fail(HC_E_ARG, "internal: hysteresis worklist capacity");
  hp.wl_count = s.d_flags + FLAG_WORDS;
  hp.wl_reason = s.d_flags + FLAG_WORDS + WL_COUNT_WORDS;
  hp.wl_list = s.d_wl_list;
  if (zeroed_words < FLAG_WORDS + WL_COUNT_WORDS + 2 * hp.wl_stride)
    HIPCK(hipMemsetAsync(s.d_flags, 0, sizeof(u32) * (FLAG_WORDS + WL_COUNT_WORDS + 2 * hp.wl_stride), st));
  // Worklists or a workgroup per tile in every launch (k_hyst)?  Lists where the step follows the hysteresis chain:
  // frames wider than one panel -- unless they are dense (the last run visited more than 60 % of the tiles in launch 1:
  // noise; camera-like frames: a third; the front kernel, which bounds such streams, loses less to a hysteresis that is
  // spread over it: 33.6 against 30.8 k frames/s on 4K noise) -- and one-panel streams whose runs need 20 launches or more
  // (BGR frames blended into grey: 25 launches, 205 -> 222 k frames/s; the 16 launches of 1080p grey frames fit inside
  // the front kernel's time, and there the lists cost 2 %).
  if (c->hyst_late_grid) hp.lists = c->hyst_late_grid > 0;
  else if (hp.npanels > 1) hp.lists = !(c->wl_prev_tiles == hp.wl_stride && (size_t)c->wl_prev[1] * 5 > hp.wl_stride * 3);
  else hp.lists = c->last_work_launches >= 20 || (c->hyst_lists_last && c->last_work_launches >= 14);
  c->hyst_lists_last = hp.lists != 0;
  hp.first_pass = 1;
  hp.prov = s.prov ? 1 : 0;
  // The other streams: a workgroup per tile for launches 0 and 1 -- which do most of the work, and whose idle workgroups
  // keep the hysteresis spread over the front kernel it runs beside -- then lists for the tail of launches that follow a
  // few long edges through the frame (launch 2 still starts every tile, and writes the first list): 1080p grey 394 -> 405 k
  // frames/s, 256 frames per run 307 -> 317 k; with the lists from launch 1 on: 400 k, from launch 4: 404 k.  (The list
  // streams above keep their lists from launch 1: BGR 259 against 252 k, 8K x 3 8.76 against 8.64 k; 4K would gain 2 %.)
  int mixed_from = (!hp.lists && !c->hyst_late_grid && small_tiles) ? 2 : 0;  // first launch of a mixed-schedule run that works from lists
  // A small run (a few frames): all K rounds in one launch, device-wide barriers between them (k_hyst_loop) -- K host
  // calls and K trips through the command processor fewer per run; a run it cannot finish (its workgroups not resident
  // together, or more rounds needed than queued) is continued by finish_slot like any other.
  // (not beside other runs: in the pipelined small batches the rounds' barriers -- ~10 us each, with the waiting workgroups
  // resident -- cost more than the launches they replace: 8 frames per run 0.147 against 0.117 ms per call; one frame per
  // call, the reference's pattern: 0.150 against 0.168 ms)
  const bool loop = c->hyst_loop && !small_tiles && !hp.lists && !c->hyst_late_grid && !c->hyst_diag && hp.npanels == 1 && c->RD == 64 && hp.wl_stride <= (size_t)HYST_LOOP_MAX_TILES
                    && ((hp.tile_rows == 16 && hp.waves == 8) || (hp.tile_rows == 32 && hp.waves == 2));
  if (loop) {
    mixed_from = 0;
    hp.iter = 0; hp.late_grid = 0; hp.stats = nullptr;
    HIPCK(launch_hyst_loop(hp, K, s.d_flags + FLAG_WORDS + WL_COUNT_WORDS - 2, st));  // (the last two count words: unused by this form, zeroed with the flags)
    hp.iter = K - 1;
  }
  for (int k = loop ? K : 0; k < K; ++k) {
    hp.iter = k;
    // worklist scheme, launches >= 1: a workgroup per list entry.  Grid: twice what the last run of this shape listed for
    // the launch (entries beyond the grid wait a launch: a dense frame would need several launches more); without such
    // a run, launch_hyst's schedule by the tile count
    hp.late_grid = c->hyst_late_grid > 0 ? c->hyst_late_grid : 0;
    if (mixed_from > 0) hp.lists = k < mixed_from ? 0 : k == mixed_from ? 2 : 1;
    if (hp.lists == 1 && !hp.late_grid && k > 0 && c->wl_prev_tiles == hp.wl_stride) hp.late_grid = (int)std::min<size_t>(hp.wl_stride, std::max<size_t>((size_t)2048, 2 * (size_t)c->wl_prev[k] + 256));
    // diagnostics cost ~3 same-address atomics per wave (hundreds of microseconds per launch): opt-in only
    hp.stats = c->hyst_diag ? s.d_flags + MAX_HYST_LAUNCHES + 3 * k : nullptr;
    HIPCK(launch_hyst(hp, st));
  }
  HIPCK(hipMemcpyAsync(s.h_flags, s.d_flags, sizeof(u32) * (FLAG_WORDS + WL_COUNT_WORDS), hipMemcpyDeviceToHost, st));
  s.pending = true;
  s.k_launches = K;
  s.ph = hp;
  s.mixed_from = mixed_from;
  s.n = n;
  s.stream = st;
  s.copy_dst = nullptr;
  return HC_OK;
}

int run_impl(hc_ctx *c, const uint8_t *in, size_t in_pitch, size_t in_fs, uint8_t *out, size_t out_pitch, size_t out_fs, int n, int stage)
{
  if (c->mode == HC_MODE_O && stage != HC_STAGE_HYSTER)
    return fail(HC_E_ARG, "mode O (cv::Canny) produces the final edge map only (cv::Canny has no intermediate outputs)");
  if (c->mode == HC_MODE_O && c->per_channel) return fail(HC_E_ARG, "HC_OPT_PER_CHANNEL applies to mode R contexts");
  if (c->per_channel && stage != HC_STAGE_HYSTER) return fail(HC_E_ARG, "per-channel mode only produces the final edge maps (HC_STAGE_HYSTER)");
  const int W = c->W, H = c->H;
  const int n_out = c->per_channel ? 3 * n : n;  // output frames (= bit-plane frames)
  const bool piped = c->pipeline && stage == HC_STAGE_HYSTER;
  if (piped) {
    // big batches rotate through two slots, small ones (fewer than 0.5 G pixels per run: the step is the latency of the
    // hysteresis chain) through four
    // (measured at 1080p: 128 frames per run 237 against 218 k frames/s with four, 256 frames 301 against 310 k)
    const int use = pipeline_slots(c, n_out);
    if (use != c->nslot_use) {
      if (int rc = finish_all(c)) return rc;
      c->nslot_use = use;
      c->cur = 0;
    }
  }
  Slot &s = c->slot[piped ? c->cur : 0];
  if (piped) {
    if (int rc = alloc_slot(c, s)) return rc;
    if (int rc = finish_slot(c, s)) return rc;  // the run that used this slot NSLOT steps ago
    s.seq = ++c->run_seq;
  } else if (int rc = finish_all(c)) return rc;
  else s.seq = 0;
  // streams: the front kernels always run on the context stream, in order with the caller's own work on it (whatever
  // it did to the input before this call, whatever it does to it afterwards); pipelined mode puts the rest on s_hyst.
  // (A separate front stream tied to the context stream by events cost a 50 us bubble per run: every cross-stream wait
  // is a round trip through the command processor.)
  hipStream_t sf = c->stream, sh = piped ? s.s_hyst : c->stream;
  // unaligned caller buffers go through the internal pitched ones
  const uint8_t *src = in;
  size_t sp = in_pitch, sfs = in_fs;
  // (mode O on 3-channel data reads whole 12-byte groups of 4 pixels: a tighter caller pitch is staged as well; so are
  // rows that do not hold whole 8-pixel groups when the 8-px front kernels are to run -- k_front8 / k_front8o load 8 or
  // 24 bytes per lane and row: tight rows of a width that is not a multiple of 8.  Round 2 fell back to the 4-px kernels
  // for those; one copy through the internal pitched buffer keeps every frame on the one-kernel path)
  c->last_in_staged = 0;
  const bool wants8 = stage == HC_STAGE_HYSTER && c->split == 2 && (c->mode == HC_MODE_R || c->C == 1);
  if (!aligned4(in, in_pitch, in_fs) || (c->mode == HC_MODE_O && c->C == 3 && in_pitch < round_up((size_t)c->W, 4) * 3)
      || (wants8 && in_pitch < round_up((size_t)c->W, 8) * (size_t)c->C)) {
    c->last_in_staged = 1;
    if (int rc = copy_frames_d2d(c, sf, c->d_in, c->in_pitch, c->in_fs, in, in_pitch, in_fs, (size_t)W * c->C, n)) return rc;
    src = c->d_in; sp = c->in_pitch; sfs = c->in_fs;
  }
  uint8_t *dst = out;
  size_t dp = out_pitch, dfs = out_fs;
  const bool out_internal = !aligned4(out, out_pitch, out_fs);
  if (out_internal) { dst = c->d_out; dp = c->out_pitch; dfs = c->out_fs; }
  c->last_out_staged = out_internal ? 1 : 0;
  c->last_front_form = -1;

  // ring full: this run goes untimed.  One slot stays free: `prev_end` still points at the last event of the run collected
  // last, and a 256th queued run would record over it
  const bool prof = c->profiling && c->ev_count < hc_ctx::EV_RUNS - 1;
  if (c->profiling && !prof) c->prof_gap = true;  // the next timed run's step interval would span this one
  hipEvent_t *ev = nullptr;
  hc_ctx::RunProf *rp = nullptr;
  if (prof) {
    const size_t slot_i = (size_t)((c->ev_head + c->ev_count) % hc_ctx::EV_RUNS);
    ev = &c->evpool[slot_i * hc_ctx::EV_PER_RUN];
    rp = &c->runprof[slot_i];
    *rp = hc_ctx::RunProf{};
    rp->after_gap = c->prof_gap;
    c->prof_gap = false;
    HIPCK(hipEventRecord(ev[0], sf));
  }
  // closes the interval that began at the previous event: it covered `mask` (bit per reference stage)
  auto mark = [&](hipStream_t st, unsigned mask, int kind) -> hipError_t {
    if (!prof || rp->nint >= hc_ctx::EV_PER_RUN - 1) return hipSuccess;
    rp->mask[rp->nint] = (uint8_t)mask;
    rp->kind[rp->nint] = (uint8_t)kind;
    rp->nint++;
    return hipEventRecord(ev[rp->nint], st);
  };
  constexpr unsigned B_MONO = 1u << HC_STAGE_MONO, B_GAUSS = 1u << HC_STAGE_GAUSSIAN, B_GRAD = 1u << HC_STAGE_GRADIENT,
                     B_NMS = 1u << HC_STAGE_NMS, B_THR = 1u << HC_STAGE_THRESH, B_HYST = 1u << HC_STAGE_HYSTER;
  // stage 0 (cannyEdgeH.cu:214-227); 1-channel input skips it (the reference's mono path is broken, SURVEY §3 ii)
  const uint8_t *mono = src;
  size_t mp = sp, mfs = sfs;
  // the fused kernel converts BGR while loading (needs whole 12-byte pixel groups inside each row)
  const bool whole_groups = sp >= round_up((size_t)W, 4) * 3;
  if (c->per_channel && !whole_groups) return fail(HC_E_ARG, "per-channel mode needs an input pitch of at least 3 * round_up(width, 4) bytes");
  const bool fuse_bgr = c->C == 3 && stage == HC_STAGE_HYSTER && whole_groups;
  if (c->C == 3 && !fuse_bgr) {
    if (stage == HC_STAGE_MONO) {
      HIPCK(launch_gray(src, sp, sfs, dst, dp, dfs, W, H, n, sf));
    } else {
      HIPCK(launch_gray(src, sp, sfs, c->d_mono, c->mono_pitch, c->mono_fs, W, H, n, sf));
      mono = c->d_mono; mp = c->mono_pitch; mfs = c->mono_fs;
    }
    HIPCK(mark(sf, B_MONO, hc_ctx::K_STAGE0));
  } else if (stage == HC_STAGE_MONO) {
    if (int rc = copy_frames_d2d(c, sf, dst, dp, dfs, src, sp, sfs, (size_t)W, n)) return rc;
    HIPCK(mark(sf, B_MONO, hc_ctx::K_STAGE0));
  }

  if (stage == HC_STAGE_HYSTER) {
    FrontParams fp{};
    fp.in = mono; fp.bgr = c->per_channel ? 2 : fuse_bgr ? 1 : 0; fp.in_pitch = mp; fp.in_frame_stride = mfs; fp.sbits = s.d_sbits; fp.cbits = s.d_cbits; fp.RD = c->RD; fp.W = W; fp.H = H;
    // Mode R, fused kernel: a wave marches through `subchunks` sub-chunks of 24 blur rows (run of 24*m - 4
    // output rows).  Longer runs amortise the 8-row warm-up; shorter runs give more work items (small batches).
    fp.nstrips = c->nstrips; fp.nframes = n_out;
#ifdef HC_LEGACY_FRONT
    int m = c->chunk ? (c->chunk + 4 + 23) / 24 : 0;
    if (m == 0) {
      m = 3;
      while (m > 1 && (long)n_out * c->nstrips * ((H + front_run_rows(m) - 1) / front_run_rows(m)) < 24576) --m;
    }
    fp.subchunks = m; fp.run_rows = front_run_rows(m);
    fp.nchunks = (H + fp.run_rows - 1) / fp.run_rows;
    fp.total_items = n_out * fp.nstrips * fp.nchunks;
#endif
    // Mode R front path: k_front8 reads whole 8-pixel groups (8 or 24 bytes per lane and row), the 4-px kernels 4-pixel groups
    const bool can8 = sp >= round_up((size_t)W, 8) * (size_t)(fuse_bgr || c->per_channel ? 3 : 1);
    // Mode O: k_front8o (form 3) for one-channel sources, the 4-px k_front_o (form -1) for 3-channel ones, for rows that
    // do not hold whole 8-pixel groups, or when HC_OPT_FRONT_SPLIT asks for a 4-px form
    // (Narrow frames: k_front8's HALF form, below.  Round 2 sent 640-column batches to k_blur + k_nms instead.)
    const int form = c->mode != HC_MODE_R ? ((c->C == 1 && c->split == 2 && can8) ? 3 : -1) : (c->split == 2 && !can8) ? 1 : c->split;  // (!can8 cannot happen any more: such rows were staged above)
    const bool split = form == 1, f8 = form == 2 || form == 3;
    c->last_front_form = form;  // (4 when k_front8 runs in its half-strip form, below)
    // Pipelined mode: k_nms / k_front_o also write the strong pixels as 255 into the output (4 px per lane: whole
    // dwords need W % 4 == 0), so that the hysteresis, which runs beside the next run's bandwidth-hungry k_blur, only
    // rewrites the 16-pixel groups it changes instead of streaming out the whole map (+8 % end to end; without the
    // overlap the extra stores of the VALU-bound kernel cost more than the hysteresis saves).
    // Not when this run's output overlaps the previous run's (a caller that keeps one output buffer): that run's
    // hysteresis may still be patching it, and a late patch would survive into this run's map.
    const uintptr_t o0 = (uintptr_t)dst, o1 = o0 + (size_t)n_out * dfs;
    bool out_overlap = false;
    if (piped) {
      // ... and that run is completed first: should its queued launches not have reached the fixpoint, its host-side
      // continuation rewrites whole maps (finish_slot) and would otherwise land on top of this run's result.  The same
      // for an older run still in flight (a caller that alternates two output buffers): it is waited for, oldest first
      // (once complete it patches nothing any more, so the shortcut stays).
      for (int k = 1; k < c->nslot_use; ++k) {
        Slot &o = c->slot[(c->cur + k) % c->nslot_use];  // k = nslot_use - 1: the previous run
        if (!(o.out0 < o1 && o0 < o.out1)) continue;
        if (k == c->nslot_use - 1) out_overlap = true;
        if (int rc = finish_slot(c, o)) return rc;
      }
    }
    s.prov = piped && !out_overlap && (f8 ? W % 8 == 0 : (W % 4 == 0 && (split || c->mode == HC_MODE_O)));
    if (piped) { s.out0 = o0; s.out1 = o1; }
    if (s.prov) { fp.prov_out = dst; fp.prov_pitch = (u32)dp; fp.prov_fs = dfs; }
    if (c->debug_taps) {
      if (int rc = ensure_debug_buffers(c)) return rc;
      if (!split) { fp.dbg_blur = c->dbg_blur; fp.dbg_pitch = (u32)c->out_pitch; fp.dbg_fs = c->out_fs; }  // out_pitch: the width if that is a multiple of 16, else padded to 256
    }
#ifdef HC_LEGACY_FRONT
    if (split) {  // k_blur + k_nms through the blur plane
      if (int rc = ensure_blur_plane(c)) return rc;
      fp.blur = c->d_bplane; fp.blur_frame_stride = c->bplane_fs;
      const int rows = pick_run_rows((long)n_out * c->nstrips, H, c->chunk);
      fp.run_rows = (rows + 1) & ~1;  // k_blur walks rows in pairs
      fp.nchunks = (H + fp.run_rows - 1) / fp.run_rows;
      fp.total_items = n_out * fp.nstrips * fp.nchunks;
      fp.run_rows_b = rows;
      fp.nchunks_b = (H + rows - 1) / rows;
      fp.total_items_b = n_out * fp.nstrips * fp.nchunks_b;

    }
#else
    if (!f8 && c->mode == HC_MODE_R) return fail(HC_E_ARG, "this library is built without the round-1 front kernels (HC_OPT_FRONT_SPLIT 1 / 0: libhipcanny_legacy.so)");
#endif
    size_t zeroed_words = 0;
    bool use_mx = false;
    if (f8) {  // strips of 496 columns, runs of 6 * windows - 4 rows
      // the 8-px kernels zero the run's hysteresis flag words on their way in: every tile shape has at least 16 rows per tile
      zeroed_words = FLAG_WORDS + WL_COUNT_WORDS + 2 * std::min(s.wl_cap, (size_t)n_out * ((size_t)(H + 15) / 16 + 1) * (size_t)((c->RD + 63) / 64));
      fp.zero_words = s.d_flags; fp.zero_count = (u32)zeroed_words;
      fp.dump = c->d_dump; fp.dump_c = c->d_dump + 2048; fp.dump_p = c->d_dump + 4096;
      fp.zeros = c->d_dump + 16384;
      fp.nstrips = front8_strips(W);
      // dense path of k_front8 (wave-wide NMS): enter above 512 half-lanes per window of 768, leave below 384.  (The batch
      // scheme costs 27 + 41 + 3.6 e instructions per row for e queued half-lanes, the dense path ~260: break-even near
      // 320 per window -- but the zero padding makes the first two rows of every frame candidates across the whole
      // width, and with 320 the window after them went dense on every natural frame: +1.2 % on the benchmark's frames.)
      fp.dense_enter = c->dense_mode == 0 ? 0x7FFFFFFF : c->dense_mode == 1 ? -1 : c->dense_enter;
      fp.dense_leave = c->dense_mode == 0 ? 0x7FFFFFFF : c->dense_mode == 1 ? -1 : c->dense_leave;
      long waves_per_chunk = (long)n_out * fp.nstrips;
      if (c->mode == HC_MODE_R && c->dump_region && c->half_mode != 0 && !(c->mx_mode == 1 && fp.bgr == 0 && !c->per_channel)) {  // (HC_OPT_FRONT_MX 1 goes first)
        // HALF form (narrow frames): the (frame, 240-column half-strip) units of a run of rows are dealt to half-waves in
        // pairs -- 640 columns: 1.5 waves instead of 2 -- when that needs fewer waves and the lane offsets fit
        const long per = c->per_channel ? 3 : 1, nh = front8_half_strips(W), pairs = ((long)n * nh + 1) / 2;
        const size_t R = c->dump_region;
        const bool fits = sfs + 32768 <= R && per * sizeof(u32) * (size_t)c->RD * H + 4096 <= R && (!s.prov || per * dfs + 16384 <= R)
                          && (unsigned long long)sfs + (unsigned long long)H * sp < (1ull << 32) && (!s.prov || (unsigned long long)per * dfs + (unsigned long long)H * dp < (1ull << 32));
        if ((pairs * per < waves_per_chunk || c->half_mode == 1) && fits) {
          fp.half = 1; fp.nhalf = (int)nh;
          fp.dump = c->d_dump; fp.dump_c = c->d_dump + R; fp.dump_p = c->d_dump + 2 * R; fp.zeros = c->d_dump + 3 * R;
          waves_per_chunk = pairs * per;
          c->last_front_form = 4;
        }
      }
      // runs of about 16 rounds of the chip for big batches (pick_run_rows); a small batch is cut into short runs instead --
      // down to 8 rows, where the 8-row warm-up doubles the work but one frame still spreads over 540 waves
      // Run length.  Every run repeats an 8-row warm-up, so long runs are cheaper -- measured optimum 110-180 rows at 1024
      // frames, provided the runs tile the frame evenly (a last run of a few rows pays the warm-up for nothing): the frame
      // is cut into round(H / 120) equal runs.  A small batch is cut into shorter runs instead, down to 8 rows, where the
      // warm-up doubles the work but one frame still spreads over 540 waves (3072 waves of this kernel are resident).
      int rows;
      if (c->chunk) rows = std::min(std::max(c->chunk, 2), H);
      else {
        const long units = waves_per_chunk;
        long nch = std::max<long>(1, (H + 60) / 120);
        if (units * nch < 3072) nch = std::min<long>((3072 + units - 1) / units, std::max(1, H / 8));  // (spread over 2048 / 1536 / 1024 waves instead: no better, profiles/r03/experiments.md)
        rows = (int)((H + nch - 1) / nch);
      }
      const int windows = std::max(1, (rows + 4 + 5) / 6);
      fp.run_rows = front8_run_rows(windows);
      fp.nchunks = (H + fp.run_rows - 1) / fp.run_rows;
      fp.total_items = (int)(waves_per_chunk * fp.nchunks);
      // k_front_mx (blur and Sobel on the matrix pipe): one-channel frames of Mode R, on request (HC_OPT_FRONT_MX)
      use_mx = c->mode == HC_MODE_R && form == 2 && fp.bgr == 0 && !fp.half && c->mx_mode == 1 && (unsigned long long)H * sp < (1ull << 32)
               && sp >= round_up((size_t)W, 4) && (!s.prov || W % 8 == 0);
      if (use_mx) {
        fp.nstrips = front_mx_strips(W);
        const long units = (long)n_out * fp.nstrips;
        // a run of n blocks covers 16 n - 4 rows and costs about one block more to start (workgroup launch, prologue: 0.2 ms
        // of a 1024-frame launch in runs of 124 rows, profiles/r04/mx_ablation.txt): the run count that needs the fewest
        // blocks in all, among those that give every wave slot of the chip (3072) six runs or more where the frame allows
        long nch;
        if (c->chunk) nch = std::max<long>(1, (H + c->chunk - 1) / c->chunk);
        else {
          const long hi = std::max<long>(1, (H + 11) / 12);
          const long lo = std::min<long>(hi, std::max<long>(1, (6 * 3072 + units - 1) / units));
          long best = -1, best_cost = 0;
          for (long k = lo; k <= std::min<long>(hi, lo + 24); ++k) {
            const long rows = (H + k - 1) / k, runs = (H + rows - 1) / rows, last = H - rows * (runs - 1);
            const long cost = ((rows + 4 + 15) / 16 + 1) * (runs - 1) + (last + 4 + 15) / 16 + 1;
            if (best < 0 || cost < best_cost) { best = k; best_cost = cost; }
          }
          nch = best;
        }
        fp.run_rows = (int)((H + nch - 1) / nch);
        fp.nchunks = (H + fp.run_rows - 1) / fp.run_rows;
        fp.total_items = (int)(units * fp.nchunks);
        c->last_front_form = 5;
      }
    }
    if (c->mode == HC_MODE_O) {
      // cv::Canny: plain thresholds on the L1 magnitude; long chunks (no LDS slab, 4-row warm-up)
      fp.a_lo[0] = (u32)c->low; fp.a_hi[0] = (u32)c->high;
      fp.l2gradient = c->l2gradient;
      if (c->l2gradient) {  // canny.cpp: thresholds capped at 32767 (hc_set_thresholds) and squared; the magnitude is dx^2 + dy^2
        fp.a_lo[0] = (u32)c->low * (u32)c->low;
        fp.a_hi[0] = (u32)c->high * (u32)c->high;
      }
      if (f8) {
        HIPCK(launch_front8o(fp, sf));  // strips and runs as set for k_front8 above
      } else {
        const long units = (long)n_out * c->nstrips;
        const int per_strip = (int)std::max<long>(1, std::min<long>((12288 + units - 1) / units, (H + 15) / 16));
        fp.chunk_rows = (H + per_strip - 1) / per_strip;
        fp.nchunks = (H + fp.chunk_rows - 1) / fp.chunk_rows;
        fp.total_items = n_out * fp.nstrips * fp.nchunks;
        if (sp < round_up((size_t)W, 4) * (size_t)c->C) return fail(HC_E_ARG, "mode O needs an input pitch of at least round_up(width, 4) * channels");
        HIPCK(launch_front_o(fp, sf));
      }
      HIPCK(mark(sf, B_GRAD | B_NMS | B_THR, hc_ctx::K_FRONT_B));  // cv::Canny has no blur stage
    } else {
      band_thresholds(c->low, c->nms_saturate != 0, fp.a_lo);
      band_thresholds(c->high, c->nms_saturate != 0, fp.a_hi);
      fp.wrap_limit = c->nms_saturate ? 0xFFFFFFFFu : 262144u;
      const unsigned b_mono = fuse_bgr ? B_MONO : 0u;  // stage 0 fused into the blur's load (per-channel mode has no grey stage)
#ifdef HC_LEGACY_FRONT
      if (split) {
        HIPCK(launch_blur(fp, sf));
        HIPCK(mark(sf, b_mono | B_GAUSS, hc_ctx::K_FRONT_A));
        HIPCK(launch_nms(fp, sf));
        HIPCK(mark(sf, B_GRAD | B_NMS | B_THR, hc_ctx::K_FRONT_B));
      } else if (!f8) {
        HIPCK(launch_front(fp, sf));
        HIPCK(mark(sf, b_mono | B_GAUSS | B_GRAD | B_NMS | B_THR, hc_ctx::K_FRONT_B));
      } else
#endif
      {
        // one-wave workgroups: pipelined big batches with the provisional map, mono / BGR (the per-channel form is three waves, one per channel)
        // (small batches, whose four chains overlap anyway: from 0.12 G pixels per run -- 64 frames of 1080p +3.5 %, 128 frames
        //  +4.5 %; 4 to 32 frames -3 to -6 %: tools/experiments/exp_small_wpb.sh)
        const bool auto_one = c->nslot_use < NSLOT ? c->front_one : (long long)n_out * W * H >= 120ll * 1000 * 1000;
        fp.one_wave = (s.prov && !c->per_channel && (c->front_wpb_mode == 1 || (c->front_wpb_mode < 0 && auto_one))) ? 1 : 0;
        c->last_front_waves = c->per_channel ? 3 : fp.one_wave ? 1 : 4;
        if (use_mx) {  // (its waves are independent too: one-wave workgroups beside the hysteresis, -3.5 % there, four-wave ones alone; HC_OPT_FRONT_WPB 1 / 4 fixes it)
          fp.one_wave = (c->front_wpb_mode == 1 || (c->front_wpb_mode < 0 && s.prov)) ? 1 : 0;
          c->last_front_waves = fp.one_wave ? 1 : 4;
          HIPCK(launch_front_mx(fp, sf));
        }
        else HIPCK(launch_front8(fp, sf));
        HIPCK(mark(sf, b_mono | B_GAUSS | B_GRAD | B_NMS | B_THR, hc_ctx::K_FRONT_B));
      }
    }
    if (c->debug_taps) {  // what the front kernels hand to the hysteresis (which updates the STRONG plane in place)
      const size_t bytes = sizeof(u32) * (size_t)c->RD * H * (size_t)n_out;
      HIPCK(hipMemcpyAsync(c->dbg_s, s.d_sbits, bytes, hipMemcpyDeviceToDevice, sf));
      HIPCK(hipMemcpyAsync(c->dbg_c, s.d_cbits, bytes, hipMemcpyDeviceToDevice, sf));
      c->dbg_frames = n_out;
      c->dbg_blur_split = split;
      c->dbg_blur_valid = c->mode == HC_MODE_R;
    }
    if (piped) {
      HIPCK(hipEventRecord(s.ev_front, sf));
      if (s.seq) {
        const int k = (int)(s.seq & 7);
        if (!c->ring_f[k]) { HIPCK(hipEventCreate(&c->ring_f[k])); HIPCK(hipEventCreate(&c->ring_d[k])); }
        HIPCK(hipEventRecord(c->ring_f[k], sf));
        c->ring_seq[k] = 0;  // (valid once the chain's end is recorded too)
      }
      HIPCK(hipStreamWaitEvent(sh, s.ev_front, 0));
    }
    if (int rc = queue_hyst_expand(c, s, sh, dst, dp, dfs, n_out, piped, zeroed_words)) return rc;
  } else if (stage > HC_STAGE_MONO) {
    if (int rc = ensure_stage_scratch(c)) return rc;
    const size_t bp = c->out_pitch, bfs = c->out_fs;  // scratch planes share the output geometry
    uint8_t *blur = stage == HC_STAGE_GAUSSIAN ? dst : c->d_blur;
    const size_t blp = stage == HC_STAGE_GAUSSIAN ? dp : bp, blfs = stage == HC_STAGE_GAUSSIAN ? dfs : bfs;
    // every plain kernel is booked on its own stage, as the reference's _endCudaTimer(stage) does (cannyEdgeH.cu:415-430)
    HIPCK(launch_gauss(mono, mp, mfs, blur, blp, blfs, W, H, n, sf));
    HIPCK(mark(sf, B_GAUSS, hc_ctx::K_FRONT_B));
    if (stage >= HC_STAGE_GRADIENT) {
      HIPCK(launch_sobel(blur, blp, blfs, c->d_sx, c->d_sy, bp, bfs, W, H, n, sf));
      if (stage == HC_STAGE_GRADIENT) HIPCK(launch_graddisp(c->d_sx, c->d_sy, bp, bfs, dst, dp, dfs, W, H, n, sf));
      HIPCK(mark(sf, B_GRAD, hc_ctx::K_FRONT_B));
      if (stage > HC_STAGE_GRADIENT) {
        uint8_t *nms = stage == HC_STAGE_NMS ? dst : c->d_nms;
        const size_t np = stage == HC_STAGE_NMS ? dp : bp, nfs = stage == HC_STAGE_NMS ? dfs : bfs;
        HIPCK(launch_nms(c->d_sx, c->d_sy, bp, bfs, nms, np, nfs, W, H, n, c->nms_saturate, sf));
        HIPCK(mark(sf, B_NMS, hc_ctx::K_FRONT_B));
        if (stage == HC_STAGE_THRESH) {
          HIPCK(launch_thresh(nms, np, nfs, dst, dp, dfs, W, H, n, c->low, c->high, sf));
          HIPCK(mark(sf, B_THR, hc_ctx::K_FRONT_B));
        }
      }
    }
  }

  if (out_internal) {
    if (int rc = copy_frames_d2d(c, sh, out, out_pitch, out_fs, c->d_out, c->out_pitch, c->out_fs, (size_t)W, n_out)) return rc;
    if (s.pending) { s.copy_dst = out; s.copy_pitch = out_pitch; s.copy_fs = out_fs; }
  }
  if (prof) {
    // the hysteresis (and the copy-out of an unaligned caller buffer) end the run; for the earlier stages the copy-out
    // belongs to the last stage that ran
    if (stage == HC_STAGE_HYSTER) HIPCK(mark(sh, B_HYST, hc_ctx::K_HYST));
    else if (out_internal) HIPCK(mark(sh, 1u << stage, hc_ctx::K_FRONT_B));
    c->ev_count++;
  }
  if (stage == HC_STAGE_HYSTER) HIPCK(hipEventRecord(s.ev_done, sh));
  if (stage == HC_STAGE_HYSTER && piped && s.seq) {
    HIPCK(hipEventRecord(c->ring_d[s.seq & 7], sh));
    c->ring_seq[s.seq & 7] = s.seq;
  }
  c->last_slot = piped ? c->cur : 0;
  if (piped) c->cur = (c->cur + 1) % c->nslot_use;
  c->last_run_n = n_out;
  return HC_OK;
}

}  // namespace

extern "C" {

const char *hc_last_error(void) { return g_err.c_str(); }
#ifdef HC_LEGACY_FRONT
const char *hc_version(void) { return "hipcanny 0.4 (gfx950) + round-1 front kernels (test build)"; }
#else
const char *hc_version(void) { return "hipcanny 0.4 (gfx950)"; }
#endif

void *hc_host_alloc(size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    fail(HC_E_HIP, "hipHostMalloc failed");
    return nullptr;
  }
  return p;
}

void hc_host_free(void *p)
{
  if (p) (void)hipHostFree(p);
}

hc_ctx *hc_create(int device, int width, int height, int channels, int max_batch, int mode)
{
  if (width <= 0 || height <= 0 || (channels != 1 && channels != 3) || max_batch <= 0 || (mode != HC_MODE_R && mode != HC_MODE_O)) {
    fail(HC_E_ARG, "hc_create: bad width/height/channels/max_batch/mode");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    fail(HC_E_NOGPU, "hc_create: no HIP device (this library has no CPU fallback)");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) { fail(HC_E_HIP, "hipSetDevice failed"); return nullptr; }
  hc_ctx *c = new hc_ctx();
  c->device = device; c->W = width; c->H = height; c->C = channels; c->max_batch = max_batch; c->mode = mode;
  if (mode == HC_MODE_O) { c->low = 50; c->high = 150; }
  c->nstrips = (width + STRIP_W - 1) / STRIP_W;
  // bit-plane row: covers every strip's 31 bytes, padded to a multiple of 64 dwords (one LDS row per wave)
  {
    const size_t need = std::max<size_t>((size_t)(width + 31) / 32, ((size_t)c->nstrips * 31 + 3) / 4);
    if (need > 256) { fail(HC_E_ARG, "hc_create: width above 8184 is not supported"); delete c; return nullptr; }
    c->RD = need <= 64 ? 64 : need <= 128 ? 128 : 256;  // 64 * NW dwords, NW in {1, 2, 4}
  }
  auto ok = [&](hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fail(HC_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return false;
  };
  bool good = ok(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate");
  c->stream = c->own_stream;
  // (a BGR row is read in 12-byte groups of 4 pixels: keep room for the ragged last group)
  good = good && alloc_frames(&c->d_in, &c->in_pitch, &c->in_fs, round_up((size_t)width, 8) * channels, height, max_batch, (size_t)width * channels) == HC_OK;
  good = good && alloc_frames(&c->d_out, &c->out_pitch, &c->out_fs, (size_t)width, height, max_batch, (size_t)width) == HC_OK;
  if (good && channels == 3) good = alloc_frames(&c->d_mono, &c->mono_pitch, &c->mono_fs, (size_t)width, height, max_batch) == HC_OK;
  good = good && alloc_slot(c, c->slot[0]) == HC_OK;
  {
    // narrow frames take k_front8's HALF form when that needs fewer waves: an odd number of half-strips (pairs across frames)
    const bool half_pays = front8_half_strips(width) % 2 == 1 || (front8_half_strips(width) + 1) / 2 < front8_strips(width);
    good = good && alloc_dump(c, half_pays) == HC_OK;
  }
  c->evpool.assign((size_t)hc_ctx::EV_RUNS * hc_ctx::EV_PER_RUN, nullptr);
  c->runprof.assign((size_t)hc_ctx::EV_RUNS, hc_ctx::RunProf{});
  for (size_t i = 0; good && i < c->evpool.size(); ++i) good = ok(hipEventCreate(&c->evpool[i]), "hipEventCreate");
  if (good) {
    // cannyEdgeH.cu:372-380: float coefficients K * (1 / 159.0f), computed in binary32 on the host
    float gk[25];
    static const int K[25] = { 2, 4, 5, 4, 2, 4, 9, 12, 9, 4, 5, 12, 15, 12, 5, 4, 9, 12, 9, 4, 2, 4, 5, 4, 2 };
    volatile float r = 1 / 159.0f;
    for (int i = 0; i < 25; ++i) { volatile float k = (float)K[i]; volatile float v = k * r; gk[i] = v; }
    good = ok(check_gauss_coeffs(gk), "Gaussian coefficient table differs from the kernels' literals");
  }
  if (!good) { hc_destroy(c); return nullptr; }
  return c;
}

void hc_destroy(hc_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void *q : { (void *)c->d_in, (void *)c->d_mono, (void *)c->d_out, (void *)c->d_blur, (void *)c->d_nms, (void *)c->d_sx, (void *)c->d_sy, (void *)c->d_bplane, (void *)c->d_dump }) (void)hipFree(q);
  for (Slot &q : c->slot) free_slot(q);
  free_debug_buffers(c);
  for (auto &e : c->evpool) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : { c->ev_up, c->ev_ready, c->ev_ready2, c->ev_down }) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ring_f) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ring_d) if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int hc_set_thresholds(hc_ctx *c, int low, int high)
{
  if (!c) return fail(HC_E_ARG, "null context");
  const int tmax = c->mode == HC_MODE_O ? 32767 : 255;  // Mode O thresholds apply to |dx|+|dy| (up to 2040)
  low = std::max(0, std::min(tmax, low));
  high = std::max(0, std::min(tmax, high));
  if (low > high) std::swap(low, high);
  c->low = low; c->high = high;
  return HC_OK;
}

int hc_get_thresholds(const hc_ctx *c, int *low, int *high)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (low) *low = c->low;
  if (high) *high = c->high;
  return HC_OK;
}

int hc_set_stream(hc_ctx *c, void *s)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (int rc = finish_all(c)) return rc;
  c->stream = (hipStream_t)s;  // 0 = the null stream itself (ordered with everything the caller queued on it)
  return HC_OK;
}

int hc_use_own_stream(hc_ctx *c)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (int rc = finish_all(c)) return rc;
  c->stream = c->own_stream;
  return HC_OK;
}

int hc_set_tuning(hc_ctx *c, int chunk_rows, int hyst_launches)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (chunk_rows < 0 || chunk_rows > 16384) return fail(HC_E_ARG, "chunk_rows must be 0 (auto) or 1..16384");
  if (hyst_launches < 0 || hyst_launches > MAX_HYST_LAUNCHES) return fail(HC_E_ARG, "hyst_launches out of range (0 = auto, 1..96)");
  if (int rc = finish_all(c)) return rc;
  c->chunk = chunk_rows;
  c->hyst_launches_set = hyst_launches != 0;
  c->hyst_launches = hyst_launches ? hyst_launches : 6;
  return HC_OK;
}

int hc_set_option(hc_ctx *c, int option, int value)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (int rc = finish_all(c)) return rc;
  if (option == HC_OPT_NMS_SATURATE) c->nms_saturate = value != 0;
  else if (option == HC_OPT_PER_CHANNEL) {
    if (c->C != 3) return fail(HC_E_ARG, "HC_OPT_PER_CHANNEL needs a 3-channel context");
    if ((value != 0) != (c->per_channel != 0)) {  // output-side buffers change size: 3 edge maps per input frame
      HIPCK(hipSetDevice(c->device));
      HIPCK(hipDeviceSynchronize());
      for (Slot &q : c->slot) free_slot(q);  // (the slots beyond the first are allocated again by the pipelined runs that need them)
      (void)hipFree(c->d_out);
      c->d_out = nullptr;
      c->per_channel = value != 0;
      free_debug_buffers(c);
      if (c->d_bplane) { (void)hipFree(c->d_bplane); c->d_bplane = nullptr; c->bplane_frames = 0; }
      if (alloc_frames(&c->d_out, &c->out_pitch, &c->out_fs, (size_t)c->W, c->H, c->max_batch * (c->per_channel ? 3 : 1), (size_t)c->W) != HC_OK) return HC_E_HIP;
      if (alloc_slot(c, c->slot[0]) != HC_OK) return HC_E_HIP;
    }
  } else if (option == HC_OPT_FRONT_SPLIT) {
    if (value < 0 || value > 2) return fail(HC_E_ARG, "HC_OPT_FRONT_SPLIT: 0 (k_front), 1 (k_blur + k_nms) or 2 (k_front8)");
#ifndef HC_LEGACY_FRONT
    if (value != 2 && c->mode == HC_MODE_R)
      return fail(HC_E_ARG, "HC_OPT_FRONT_SPLIT 1 / 0: the round-1 front kernels are not part of this library (parity tests load libhipcanny_legacy.so)");
#endif
    c->split = value;
    c->split_set = true;  // the caller's choice: no automatic switch to the 4-px pair for narrow frames
  } else if (option == HC_OPT_COPY_STREAMS) {
    if (c->dl_host) return fail(HC_E_STATE, "HC_OPT_COPY_STREAMS: a download is in flight");
    HIPCK(hipSetDevice(c->device));
    if (value && c->device < MAX_DEVICES) {
      std::lock_guard<std::mutex> lock(g_copy_streams_mutex);
      if (!g_h2d[c->device]) HIPCK(hipStreamCreateWithFlags(&g_h2d[c->device], hipStreamNonBlocking));
      if (!g_d2h[c->device]) HIPCK(hipStreamCreateWithFlags(&g_d2h[c->device], hipStreamNonBlocking));
      if (!c->ev_up) HIPCK(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
      if (!c->ev_ready) HIPCK(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
      if (!c->ev_ready2) HIPCK(hipEventCreateWithFlags(&c->ev_ready2, hipEventDisableTiming));
      if (!c->ev_down) HIPCK(hipEventCreateWithFlags(&c->ev_down, hipEventDisableTiming));
    }
    c->copy_streams = value != 0 && c->device < MAX_DEVICES;
  } else if (option == HC_OPT_FRONT_WPB) {
    if (value != -1 && value != 1 && value != 4) return fail(HC_E_ARG, "HC_OPT_FRONT_WPB: -1 (automatic), 1 or 4");
    c->front_wpb_mode = value;
  } else if (option == HC_OPT_PIPELINE_SLOTS) {
    if (value == -1 || value == 20 || value == 21) { c->pipe_slots = 0; c->chain_told = value == 20 ? 1 : value == 21 ? -1 : 0; }
    else if (value == 2 || value == 3) c->pipe_slots = value;
    else return fail(HC_E_ARG, "HC_OPT_PIPELINE_SLOTS: -1 (automatic), 2, 3, or 20 / 21 (diagnostics)");
  } else if (option == HC_OPT_FRONT_DENSE) {
    if (value < -1 || value > 1) return fail(HC_E_ARG, "HC_OPT_FRONT_DENSE: -1 (automatic), 0 (never) or 1 (every window)");
    c->dense_mode = value;
  } else if (option == HC_OPT_FRONT_MX) {
    if (value != 0 && value != 1) return fail(HC_E_ARG, "HC_OPT_FRONT_MX: 0 (never) or 1 (whenever the run allows it)");
    c->mx_mode = value;
  } else if (option == HC_OPT_TEST_HYST_LATE_GRID) {  // tests: tiny grids exercise the hand-on of worklist entries
    c->hyst_late_grid = std::max(-1, value);
  } else if (option == HC_OPT_TEST_HYST_LOOP) {
    c->hyst_loop = value != 0;
  } else if (option == HC_OPT_TEST_HYST_DIAG) {
    c->hyst_diag = value != 0;
  } else if (option == HC_OPT_TEST_HYST_GEOM) {  // rows per wave x 100 + waves per workgroup; 0: by the rule
    c->hyst_geom = std::max(0, value);
  } else if (option == HC_OPT_TEST_DENSE_ENTER) {
    c->dense_enter = std::max(0, value);
  } else if (option == HC_OPT_TEST_DENSE_LEAVE) {
    c->dense_leave = std::max(0, value);
  } else if (option == HC_OPT_FRONT_HALF) {
    if (value < -1 || value > 1) return fail(HC_E_ARG, "HC_OPT_FRONT_HALF: -1 (automatic), 0 (never) or 1 (whenever possible)");
    HIPCK(hipSetDevice(c->device));
    if (value == 1 && !c->dump_region) {  // the half-strip form needs the larger dump areas
      HIPCK(hipDeviceSynchronize());
      if (int rc = alloc_dump(c, true)) return rc;
    }
    c->half_mode = value;
  } else if (option == HC_OPT_L2_GRADIENT) {
    if (c->mode != HC_MODE_O) return fail(HC_E_ARG, "HC_OPT_L2_GRADIENT applies to mode O contexts");
    c->l2gradient = value != 0;
  } else if (option == HC_OPT_DEBUG_TAPS) {
    c->debug_taps = value != 0;
    c->dbg_frames = 0;
  } else if (option == HC_OPT_PIPELINE) {
    HIPCK(hipSetDevice(c->device));
    if (value && alloc_slot(c, c->slot[1]) != HC_OK) return HC_E_HIP;  // (the slots of the four-slot ring are allocated by the small batches that use them)
    c->pipeline = value != 0;
    c->cur = 0;
    c->nslot_use = 2;
  } else return fail(HC_E_ARG, "hc_set_option: unknown option");
  return HC_OK;
}

int hc_upload(hc_ctx *c, const uint8_t *host, size_t row_stride, size_t frame_stride, int n)
{
  if (!c || !host) return fail(HC_E_ARG, "hc_upload: null argument");
  if (n <= 0 || n > c->max_batch) return fail(HC_E_ARG, "hc_upload: nframes out of range");
  const size_t rb = (size_t)c->W * c->C;
  if (row_stride < rb) return fail(HC_E_ARG, "hc_upload: row_stride smaller than a row");
  HIPCK(hipSetDevice(c->device));
  if (int rc = finish_all(c)) return rc;
  // cannyEdgeH.cu:136/144 (cudaMemcpy2D host -> pitched device).  Tight rows on both sides: one contiguous block, one DMA
  hipStream_t cs = c->copy_streams ? g_h2d[c->device] : c->stream;
  if (c->copy_streams && hipStreamQuery(c->stream) != hipSuccess) {  // what is still queued on the context stream (it may read d_in) comes first
    HIPCK(hipEventRecord(c->ev_ready, c->stream));
    HIPCK(hipStreamWaitEvent(cs, c->ev_ready, 0));
  }
  if (row_stride == rb && c->in_pitch == rb && frame_stride == c->in_fs)
    HIPCK(hipMemcpyAsync(c->d_in, host, c->in_fs * (size_t)n, hipMemcpyHostToDevice, cs));
  else
    for (int f = 0; f < n; ++f)
      HIPCK(hipMemcpy2DAsync(c->d_in + c->in_fs * f, c->in_pitch, host + frame_stride * f, row_stride, rb, (size_t)c->H, hipMemcpyHostToDevice, cs));
  if (c->copy_streams) {
    HIPCK(hipEventRecord(c->ev_up, cs));
    HIPCK(hipStreamWaitEvent(c->stream, c->ev_up, 0));
  }
  c->uploaded = n;
  return HC_OK;
}

int hc_run(hc_ctx *c, int final_stage, int n)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (final_stage < HC_STAGE_MONO || final_stage > HC_STAGE_HYSTER) return fail(HC_E_ARG, "Canny Stage Not Recognized");
  if (n <= 0 || n > c->uploaded) return fail(HC_E_STATE, "hc_run: more frames than uploaded");
  if (c->dl_host) return fail(HC_E_STATE, "hc_run: a download of the internal output buffer is in flight (hc_download_end first)");
  HIPCK(hipSetDevice(c->device));
  return run_impl(c, c->d_in, c->in_pitch, c->in_fs, c->d_out, c->out_pitch, c->out_fs, n, final_stage);
}

int hc_run_device(hc_ctx *c, const void *d_in, size_t in_pitch, size_t in_fs, void *d_out, size_t out_pitch, size_t out_fs, int n, int final_stage)
{
  if (!c || !d_in || !d_out) return fail(HC_E_ARG, "hc_run_device: null argument");
  if (final_stage < HC_STAGE_MONO || final_stage > HC_STAGE_HYSTER) return fail(HC_E_ARG, "Canny Stage Not Recognized");
  if (n <= 0 || n > c->max_batch) return fail(HC_E_ARG, "hc_run_device: nframes out of range");
  if (in_pitch < (size_t)c->W * c->C || out_pitch < (size_t)c->W) return fail(HC_E_ARG, "hc_run_device: pitch smaller than a row");
  if (n > 1 && (in_fs < in_pitch * (size_t)c->H || out_fs < out_pitch * (size_t)c->H)) return fail(HC_E_ARG, "hc_run_device: frame stride smaller than a frame");
  HIPCK(hipSetDevice(c->device));
  return run_impl(c, (const uint8_t *)d_in, in_pitch, in_fs, (uint8_t *)d_out, out_pitch, out_fs, n, final_stage);
}

int hc_hysteresis_device(hc_ctx *c, const void *d_thresh, size_t in_pitch, size_t in_fs, void *d_out, size_t out_pitch, size_t out_fs, int n)
{
  if (!c || !d_thresh || !d_out) return fail(HC_E_ARG, "hc_hysteresis_device: null argument");
  if (n <= 0 || n > c->max_batch) return fail(HC_E_ARG, "hc_hysteresis_device: nframes out of range");
  if (in_pitch < (size_t)c->W || out_pitch < (size_t)c->W) return fail(HC_E_ARG, "hc_hysteresis_device: pitch smaller than a row");
  if (n > 1 && (in_fs < in_pitch * (size_t)c->H || out_fs < out_pitch * (size_t)c->H)) return fail(HC_E_ARG, "hc_hysteresis_device: frame stride smaller than a frame");
  HIPCK(hipSetDevice(c->device));
  if (int rc = finish_all(c)) return rc;
  Slot &s = c->slot[0];
  PackParams pp{};
  pp.in = (const uint8_t *)d_thresh; pp.in_pitch = in_pitch; pp.in_frame_stride = in_fs; pp.sbits = s.d_sbits; pp.cbits = s.d_cbits; pp.RD = c->RD; pp.W = c->W; pp.H = c->H; pp.nframes = n;
  HIPCK(launch_pack(pp, c->stream));
  uint8_t *dst = (uint8_t *)d_out;
  size_t dp = out_pitch, dfs = out_fs;
  const bool out_internal = !aligned4(d_out, out_pitch, out_fs);
  if (out_internal) { dst = c->d_out; dp = c->out_pitch; dfs = c->out_fs; }
  s.prov = false;  // nothing has written a provisional map into this output (a pipelined run may have left the flag set)
  if (int rc = queue_hyst_expand(c, s, c->stream, dst, dp, dfs, n, false)) return rc;
  if (out_internal) {
    if (int rc = copy_frames_d2d(c, c->stream, d_out, out_pitch, out_fs, c->d_out, c->out_pitch, c->out_fs, (size_t)c->W, n)) return rc;
    s.copy_dst = d_out; s.copy_pitch = out_pitch; s.copy_fs = out_fs;
  }
  HIPCK(hipEventRecord(s.ev_done, c->stream));
  return HC_OK;
}

int hc_sync(hc_ctx *c)
{
  if (!c) return fail(HC_E_ARG, "null context");
  HIPCK(hipSetDevice(c->device));
  if (int rc = finish_all(c)) return rc;
  for (Slot &q : c->slot)
    if (q.s_hyst) HIPCK(hipStreamSynchronize(q.s_hyst));
  HIPCK(hipStreamSynchronize(c->stream));
  while (c->ev_count > 0) {  // collect the event intervals of every run recorded since the last sync
    hipEvent_t *e = &c->evpool[(size_t)c->ev_head * hc_ctx::EV_PER_RUN];
    const hc_ctx::RunProf &rp = c->runprof[(size_t)c->ev_head];
    for (float &m : c->stage_ms) m = 0;
    c->stage_ran = 0;
    bool has_a = false, has_h = false;
    float front_t = 0;
    for (int i = 0; i < rp.nint; ++i) has_a = has_a || rp.kind[i] == hc_ctx::K_FRONT_A;
    for (int i = 0; i < rp.nint; ++i) has_h = has_h || rp.kind[i] == hc_ctx::K_HYST;
    for (int i = 0; i < rp.nint; ++i) {
      float t = 0;
      HIPCK(hipEventElapsedTime(&t, e[i], e[i + 1]));
      if (rp.kind[i] == hc_ctx::K_FRONT_A || rp.kind[i] == hc_ctx::K_FRONT_B) front_t += t;
      const unsigned mask = rp.mask[i];
      const int nst = __builtin_popcount(mask);
      for (int st = 0; st < 6; ++st)
        if (mask >> st & 1u) c->stage_ms[st] += t / (float)nst;
      c->stage_ran |= mask;
      const int k = rp.kind[i];
      c->prof_sum[k == hc_ctx::K_STAGE0 ? 0 : k == hc_ctx::K_HYST ? 2 : 1] += t;
      if (k == hc_ctx::K_FRONT_A) c->prof_split_sum[0] += t;
      else if (k == hc_ctx::K_FRONT_B && has_a) c->prof_split_sum[1] += t;
    }
    if (has_a) c->prof_split_runs++;
    if (has_h && c->front_each.size() < 65536) c->front_each.push_back(front_t);
    if (rp.nint > 0) {
      if (c->prev_end && !rp.after_gap && c->step_ms.size() < 65536) {
        float dt = 0;
        if (hipEventElapsedTime(&dt, c->prev_end, e[rp.nint]) == hipSuccess) c->step_ms.push_back(dt);
      }
      c->prev_end = e[rp.nint];
    }
    c->prof_runs++;
    c->ev_head = (c->ev_head + 1) % hc_ctx::EV_RUNS;
    c->ev_count--;
  }
  return HC_OK;
}

int hc_download(hc_ctx *c, uint8_t *host, size_t row_stride, size_t frame_stride, int n)
{
  if (!c || !host) return fail(HC_E_ARG, "hc_download: null argument");
  if (n <= 0 || n > c->last_run_n) return fail(HC_E_STATE, "hc_download: more frames than the last run produced");
  if (row_stride < (size_t)c->W) return fail(HC_E_ARG, "hc_download: row_stride smaller than a row");
  if (int rc = hc_sync(c)) return rc;
  if (row_stride == (size_t)c->W && c->out_pitch == (size_t)c->W && frame_stride == c->out_fs)
    HIPCK(hipMemcpyAsync(host, c->d_out, c->out_fs * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  else
    for (int f = 0; f < n; ++f)
      HIPCK(hipMemcpy2DAsync(host + frame_stride * f, row_stride, c->d_out + c->out_fs * f, c->out_pitch, (size_t)c->W, (size_t)c->H, hipMemcpyDeviceToHost, c->stream));
  HIPCK(hipStreamSynchronize(c->stream));
  return HC_OK;
}

namespace {
int queue_download(hc_ctx *c)
{
  hipStream_t cs = c->copy_streams ? g_d2h[c->device] : c->stream;
  if (c->copy_streams) {  // behind everything queued on the context stream (the run, its copy-out kernels)
    HIPCK(hipEventRecord(c->ev_ready2, c->stream));
    HIPCK(hipStreamWaitEvent(cs, c->ev_ready2, 0));
    Slot &s = c->slot[c->last_slot];
    if (s.pending && s.stream != c->stream) HIPCK(hipStreamWaitEvent(cs, s.ev_done, 0));
  }
  if (c->dl_row == (size_t)c->W && c->out_pitch == (size_t)c->W && c->dl_fs == c->out_fs)
    HIPCK(hipMemcpyAsync(c->dl_host, c->d_out, c->out_fs * (size_t)c->dl_n, hipMemcpyDeviceToHost, cs));
  else
    for (int f = 0; f < c->dl_n; ++f)
      HIPCK(hipMemcpy2DAsync(c->dl_host + c->dl_fs * f, c->dl_row, c->d_out + c->out_fs * f, c->out_pitch, (size_t)c->W, (size_t)c->H, hipMemcpyDeviceToHost, cs));
  if (c->copy_streams) HIPCK(hipEventRecord(c->ev_down, cs));
  return HC_OK;
}
}  // namespace

int hc_download_begin(hc_ctx *c, uint8_t *host, size_t row_stride, size_t frame_stride, int n)
{
  if (!c || !host) return fail(HC_E_ARG, "hc_download_begin: null argument");
  if (n <= 0 || n > c->last_run_n) return fail(HC_E_STATE, "hc_download_begin: more frames than the last run produced");
  if (row_stride < (size_t)c->W) return fail(HC_E_ARG, "hc_download_begin: row_stride smaller than a row");
  if (c->dl_host) return fail(HC_E_STATE, "hc_download_begin: a download is already in flight (hc_download_end first)");
  HIPCK(hipSetDevice(c->device));
  // behind the run: its hysteresis may sit on the slot's own stream (pipelined mode)
  Slot &s = c->slot[c->last_slot];
  if (s.pending && s.stream != c->stream) HIPCK(hipStreamWaitEvent(c->stream, s.ev_done, 0));
  c->dl_host = host; c->dl_row = row_stride; c->dl_fs = frame_stride; c->dl_n = n;
  c->dl_stale = false;
  if (int rc = queue_download(c)) { c->dl_host = nullptr; return rc; }
  return HC_OK;
}

int hc_download_end(hc_ctx *c)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (!c->dl_host) return fail(HC_E_STATE, "hc_download_end without hc_download_begin");
  HIPCK(hipSetDevice(c->device));
  int rc = finish_all(c);  // convergence of every run in flight; the host-side continuation if one needed it
  // the maps changed after the copy was queued -- here, or in any entry point that finished the runs since hc_download_begin
  // (hc_upload, hc_sync, hc_set_option, hc_hysteresis_totals ...): copy them again
  if (rc == HC_OK && c->dl_stale) rc = queue_download(c);
  c->dl_stale = false;
  if (rc == HC_OK && (c->copy_streams ? hipEventSynchronize(c->ev_down) : hipStreamSynchronize(c->stream)) != hipSuccess) rc = fail(HC_E_HIP, "waiting for the download failed");
  c->dl_host = nullptr;
  return rc;
}

int hc_enable_profiling(hc_ctx *c, int on)
{
  if (!c) return fail(HC_E_ARG, "null context");
  c->profiling = on != 0;
  return HC_OK;
}

int hc_profile_get(hc_ctx *c, double sum_ms[3], long *nruns, int reset)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (int rc = hc_sync(c)) return rc;
  if (sum_ms) for (int i = 0; i < 3; ++i) sum_ms[i] = c->prof_sum[i];
  if (nruns) *nruns = c->prof_runs;
  if (reset) {
    c->step_ms.clear(); c->prev_end = nullptr;
    c->front_each.clear();
    c->prof_sum[0] = c->prof_sum[1] = c->prof_sum[2] = 0; c->prof_runs = 0;
    c->prof_split_sum[0] = c->prof_split_sum[1] = 0; c->prof_split_runs = 0;
  }
  return HC_OK;
}

int hc_profile_get_front(hc_ctx *c, double sum_ms[2], long *nruns)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (int rc = hc_sync(c)) return rc;
  if (sum_ms) { sum_ms[0] = c->prof_split_sum[0]; sum_ms[1] = c->prof_split_sum[1]; }
  if (nruns) *nruns = c->prof_split_runs;
  return HC_OK;
}

int hc_profile_get_intervals(hc_ctx *c, float *ms, int cap, int *n)
{
  if (!c || !n || (cap > 0 && !ms)) return fail(HC_E_ARG, "hc_profile_get_intervals: bad argument");
  if (int rc = hc_sync(c)) return rc;
  const int m = (int)std::min<size_t>(c->step_ms.size(), (size_t)std::max(cap, 0));
  for (int i = 0; i < m; ++i) ms[i] = c->step_ms[(size_t)i];
  *n = (int)c->step_ms.size();
  return HC_OK;
}

int hc_profile_get_front_each(hc_ctx *c, float *ms, int cap, int *n)
{
  if (!c || !n || (cap > 0 && !ms)) return fail(HC_E_ARG, "hc_profile_get_front_each: bad argument");
  if (int rc = hc_sync(c)) return rc;
  const int m = (int)std::min<size_t>(c->front_each.size(), (size_t)std::max(cap, 0));
  for (int i = 0; i < m; ++i) ms[i] = c->front_each[(size_t)i];
  *n = (int)c->front_each.size();
  return HC_OK;
}

int hc_stage_time_ms(hc_ctx *c, int stage, float *ms)
{
  if (!c || !ms || stage < 0 || stage > 5) return fail(HC_E_ARG, "hc_stage_time_ms: bad argument");
  *ms = (c->stage_ran >> stage & 1u) ? c->stage_ms[stage] : -1.0f;
  return HC_OK;
}

int hc_device_ptrs(hc_ctx *c, void **d_in, void **d_out, size_t *in_pitch, size_t *out_pitch, size_t *in_fs, size_t *out_fs)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (d_in) *d_in = c->d_in;
  if (d_out) *d_out = c->d_out;
  if (in_pitch) *in_pitch = c->in_pitch;
  if (out_pitch) *out_pitch = c->out_pitch;
  if (in_fs) *in_fs = c->in_fs;
  if (out_fs) *out_fs = c->out_fs;
  return HC_OK;
}

int hc_debug_tap(hc_ctx *c, int what, uint8_t *host, size_t row_stride, size_t frame_stride, int n)
{
  if (!c || !host) return fail(HC_E_ARG, "hc_debug_tap: null argument");
  if (what != HC_TAP_BLUR && what != HC_TAP_THRESH) return fail(HC_E_ARG, "hc_debug_tap: unknown tap");
  if (row_stride < (size_t)c->W) return fail(HC_E_ARG, "hc_debug_tap: row_stride smaller than a row");
  if (int rc = hc_sync(c)) return rc;
  if (!c->debug_taps || n <= 0 || n > c->dbg_frames) return fail(HC_E_STATE, "hc_debug_tap: set HC_OPT_DEBUG_TAPS and run HC_STAGE_HYSTER first");
  const int W = c->W, H = c->H;
  if (what == HC_TAP_THRESH) {
    const size_t words = (size_t)c->RD * H * (size_t)n;
    std::vector<u32> sb(words), cb(words);
    HIPCK(hipMemcpy(sb.data(), c->dbg_s, words * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(cb.data(), c->dbg_c, words * 4, hipMemcpyDeviceToHost));
    for (int f = 0; f < n; ++f)
      for (int r = 0; r < H; ++r) {
        const u32 *srow = &sb[((size_t)f * H + r) * c->RD], *crow = &cb[((size_t)f * H + r) * c->RD];
        uint8_t *o = host + frame_stride * f + row_stride * r;
        for (int x = 0; x < W; ++x) {
          const u32 sbit = (srow[x >> 5] >> (x & 31)) & 1u, cbit = (crow[x >> 5] >> (x & 31)) & 1u;
          o[x] = sbit ? 255 : cbit ? 128 : 0;
        }
      }
    return HC_OK;
  }
  if (!c->dbg_blur_valid) return fail(HC_E_STATE, "hc_debug_tap: the last run computed no blur (mode O)");
  if (c->dbg_blur_split) {  // [frame][strip][H][256]: bytes 4..251 of a segment row are the strip's 248 columns
    std::vector<uint8_t> seg((size_t)H * 256);
    for (int f = 0; f < n; ++f)
      for (int st = 0; st < c->nstrips; ++st) {
        HIPCK(hipMemcpy(seg.data(), c->d_bplane + c->bplane_fs * f + (size_t)st * H * 256, seg.size(), hipMemcpyDeviceToHost));
        const int x0 = st * STRIP_W, nx = std::min(STRIP_W, W - x0);
        for (int r = 0; r < H; ++r) std::memcpy(host + frame_stride * f + row_stride * r + x0, &seg[(size_t)r * 256 + 4], (size_t)nx);
      }
  } else {
    for (int f = 0; f < n; ++f)
      HIPCK(hipMemcpy2D(host + frame_stride * f, row_stride, c->dbg_blur + c->out_fs * f, c->out_pitch, (size_t)W, (size_t)H, hipMemcpyDeviceToHost));
  }
  return HC_OK;
}

int hc_pipeline_depth(hc_ctx *c, int nframes)
{
  if (!c || nframes <= 0) return fail(HC_E_ARG, "hc_pipeline_depth: null context or nframes <= 0");
  if (!c->pipeline) return 1;
  const int n = pipeline_slots(c, c->per_channel ? 3 * nframes : nframes);
  return (n < NSLOT && !c->pipe_slots) ? 3 : n;  // big batches: two slots, three while the hysteresis chain bounds the step (watch_chain)
}

int hc_front_waves_per_workgroup(hc_ctx *c)
{
  if (!c) return fail(HC_E_ARG, "null context");
  return c->last_front_waves;
}

int hc_pipeline_slots_in_use(hc_ctx *c)
{
  if (!c) return fail(HC_E_ARG, "null context");
  return c->pipeline ? c->nslot_use : 1;
}

int hc_last_run_info(hc_ctx *c, int *input_staged, int *output_staged, int *front_form)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (input_staged) *input_staged = c->last_in_staged;
  if (output_staged) *output_staged = c->last_out_staged;
  if (front_form) *front_form = c->last_front_form;
  return HC_OK;
}

int hc_hysteresis_stats(hc_ctx *c, unsigned *stats, int nwords)
{
  if (!c || !stats) return fail(HC_E_ARG, "null argument");
  if (int rc = finish_all(c)) return rc;
  for (int i = 0; i < nwords && i < 3 * MAX_HYST_LAUNCHES; ++i) stats[i] = c->h_stats[i];
  return HC_OK;
}

int hc_last_hysteresis_info(hc_ctx *c, int *launches_with_work, int *continued)
{
  if (!c) return fail(HC_E_ARG, "null context");
  if (int rc = finish_all(c)) return rc;
  if (launches_with_work) *launches_with_work = c->last_work_launches;
  if (continued) *continued = c->last_continued;
  return HC_OK;
}

int hc_hysteresis_totals(hc_ctx *c, unsigned long long totals[4], int reset)
{
  if (!c || !totals) return fail(HC_E_ARG, "null argument");
  if (int rc = finish_all(c)) return rc;
  for (int i = 0; i < 4; ++i) totals[i] = c->hyst_totals[i];
  if (reset) for (int i = 0; i < 4; ++i) c->hyst_totals[i] = 0;
  return HC_OK;
}

int hc_selftest(int device)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(HC_E_NOGPU, "hc_selftest: no HIP device");
  HIPCK(hipSetDevice(device));
  u32 *d = nullptr, h = 0xFFFFFFFFu;
  HIPCK(hipMalloc((void **)&d, sizeof(u32)));
  HIPCK(hipMemset(d, 0, sizeof(u32)));
  HIPCK(launch_selftest(d, nullptr));
  HIPCK(hipMemcpy(&h, d, sizeof(u32), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  if (h) return fail(HC_E_HIP, "hc_selftest: primitive check failed, bits=" + std::to_string(h));
  return HC_OK;
}
}  // extern "C"
