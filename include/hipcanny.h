/*
 * hipcanny.h -- C ABI of libhipcanny.so: the MI355X (gfx950) Canny edge detector that replaces
 * the CUDA hot path of axoloto/CudaCam (class cvp::cuda::CannyEdge).
 *
 * Every entry point cites the reference interface it replaces (paths relative to the CudaCam
 * tree).  The C++ drop-in classes in include/cvp/ (cvp::cvPipeline, cvp::cuda::CannyEdge) and the
 * Python binding in cudacam_amd/ are thin layers over exactly these functions.  No torch / HIP
 * types appear in the signatures: device pointers and streams travel as void*.
 *
 * Threading: one context = one device + one stream; a context is not thread-safe, different
 * contexts may be used from different threads (reference: single-threaded, default stream,
 * src/imgui/imguiApp.cpp:496-522).
 * Errors: 0 = ok, negative = failure (see HC_E_*); hc_last_error() returns a message.  The
 * reference logs and exits the process instead (src/cvp/helper.hpp:4-17).
 */
#ifndef HIPCANNY_H
#define HIPCANNY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hc_ctx hc_ctx;

/* Stage ids: cvp::CannyStage, src/cvp/define.hpp:9-17 */
enum { HC_STAGE_MONO = 0, HC_STAGE_GAUSSIAN = 1, HC_STAGE_GRADIENT = 2, HC_STAGE_NMS = 3, HC_STAGE_THRESH = 4, HC_STAGE_HYSTER = 5 };

/* Parity modes.  R: bit-exact with the reference kernels (src/cvp/cannyEdgeD.cu).
 * O: bit-exact with OpenCV cv::Canny(img, low, high, 3, L2gradient) (no blur, replicate border; L2gradient
 *    false unless HC_OPT_L2_GRADIENT is set). */
enum { HC_MODE_R = 0, HC_MODE_O = 1 };

enum {
  HC_OK = 0,
  HC_E_ARG = -1,      /* bad argument (null, size/channel mismatch, unsupported type) */
  HC_E_HIP = -2,      /* a HIP runtime call failed (hc_last_error() has the hipError string) */
  HC_E_STATE = -3,    /* call sequence error (e.g. run before upload) */
  HC_E_NOGPU = -4     /* no usable gfx950 device */
};

/* Replaces CannyEdge::CannyEdge + _initAlloc (src/cvp/cannyEdgeH.cu:16-38, 340-385).
 * width/height/channels as in the reference constructor; max_batch frames can be resident and
 * processed per hc_run (the reference is single-frame: max_batch = 1).  Defaults follow
 * cannyEdgeH.cu:22-24: low = 10, high = 40 (Mode O: 50/150).  Returns NULL on failure. */
hc_ctx *hc_create(int device, int width, int height, int channels, int max_batch, int mode);

/* Replaces CannyEdge::~CannyEdge + _endAlloc (cannyEdgeH.cu:40-47, 387-407). */
void hc_destroy(hc_ctx *ctx);

/* Replaces setLowThreshold/setHighThreshold (src/cvp/cannyEdgeH.hpp:25-29): the pair is stored as
 * given after clamping to 0..255 (Mode R) and ordering low <= high. */
int hc_set_thresholds(hc_ctx *ctx, int low, int high);
int hc_get_thresholds(const hc_ctx *ctx, int *low, int *high);

/* Replaces _loadInputImage (cannyEdgeH.cu:122-152): host frames -> device.  row_stride = cv::Mat::step,
 * frame_stride = bytes between consecutive frames.  Asynchronous on the context stream when the
 * host memory is pinned. */
int hc_upload(hc_ctx *ctx, const uint8_t *host, size_t row_stride, size_t frame_stride, int nframes);

/* Replaces CannyEdge::run's stage switch (cannyEdgeH.cu:49-120) for the frames last uploaded:
 * runs the pipeline up to final_stage and leaves that stage's u8 image in the output buffer
 * (what _sendOutputToOpenGL copies into the PBO, cannyEdgeH.cu:154-212).  Asynchronous. */
int hc_run(hc_ctx *ctx, int final_stage, int nframes);

/* The same on caller-owned device memory (no upload/download): `d_in` holds nframes frames of
 * width*channels bytes per row, `d_out` receives nframes tight-or-pitched u8 images.  Pointers,
 * pitches and frame strides must be multiples of 4 bytes (others are staged through internal buffers, see
 * hc_last_run_info).  Asynchronous on the context stream.
 * Readable extent: the fast kernels load whole pixel groups, so EVERY row -- the last row of the last frame
 * included -- must be readable for min(in_pitch, round_up(width, 8) * channels) bytes from its first byte: a
 * caller whose rows are padded to whole 8-pixel groups (in_pitch >= round_up(width, 8) * channels) must own that
 * padding after the last row too, i.e. allocate height * in_pitch bytes per frame.  No row is ever read beyond
 * its pitch. */
int hc_run_device(hc_ctx *ctx, const void *d_in, size_t in_pitch, size_t in_frame_stride, void *d_out, size_t out_pitch,
                  size_t out_frame_stride, int nframes, int final_stage);

/* The hysteresis stage alone (kernels `hysteresis` + `removeCandidates`, src/cvp/cannyEdgeD.cu:295-395,
 * loop of cannyEdgeH.cu:297-338) on device tri-state maps (0 / 128 / 255) -> 0 / 255. */
int hc_hysteresis_device(hc_ctx *ctx, const void *d_thresh, size_t in_pitch, size_t in_frame_stride, void *d_out, size_t out_pitch,
                         size_t out_frame_stride, int nframes);

/* Device -> host copy of the output images of the last hc_run (the reference leaves them in the
 * GL PBO; a headless MI355X has no GL: this is the generalised sink). Synchronises. */
int hc_download(hc_ctx *ctx, uint8_t *host, size_t row_stride, size_t frame_stride, int nframes);

/* The same in two halves, for host pipelines that keep several contexts busy (cvp::io::FrameStreamer): _begin queues the
 * device -> host copy of the last run's output images behind that run and returns at once -- so that the copy engine
 * has this context's download queued while the host goes on to upload and start the next context's batch (PCIe carries
 * both directions at once: 48 GB/s each way on the GPU box against 54 / 56 one at a time) -- and _end waits for it,
 * verifies the run's convergence and, in the rare case the hysteresis had to be continued from the host, repeats the
 * copy.  `host` should be page-locked (hc_host_alloc); it must stay valid until _end returns. */
int hc_download_begin(hc_ctx *ctx, uint8_t *host, size_t row_stride, size_t frame_stride, int nframes);
int hc_download_end(hc_ctx *ctx);

/* Waits for the context stream; also completes the rare hysteresis continuation. */
int hc_sync(hc_ctx *ctx);

/* Run on the caller's HIP stream instead of the context's own: `hip_stream` is a hipStream_t handle, and 0 / NULL
 * means what it means to HIP -- the device's null (legacy default) stream, which is also what
 * torch.cuda.current_stream().cuda_stream is unless the caller switched streams.  Front kernels, uploads and
 * (outside pipelined mode) the hysteresis are then queued on that stream, in order with the caller's own work on
 * it: a run sees everything queued before it.  The context's own stream is created non-blocking, i.e. it does NOT
 * synchronise with the null stream; hc_use_own_stream() goes back to it. */
int hc_set_stream(hc_ctx *ctx, void *hip_stream);
int hc_use_own_stream(hc_ctx *ctx);

/* Replaces enableKernelProfiling / _startCudaTimer / _endCudaTimer (cannyEdgeH.hpp:31-32,
 * cannyEdgeH.cu:409-430): when enabled, hc_run brackets every kernel with hipEvents; after hc_sync,
 * hc_stage_time_ms returns the last profiled run's time attributed to `stage`, or -1 when that run did not execute the
 * stage (final_stage below it, or stage 0 on 1-channel input) -- book a sample only for times >= 0, as the reference
 * books a stage only when it ran.  Attribution: the plain per-stage kernels behind final_stage < HYSTER each have their
 * own interval.  On the HYSTER fast path one kernel covers several reference stages and has no internal boundary to
 * time: k_blur covers MONO (3-channel input) + GAUSSIAN, k_nms covers GRADIENT + NMS + THRESH, a fused front kernel
 * covers all of them, k_front_o (mode O) GRADIENT + NMS + THRESH; a kernel's time is divided EQUALLY among the stages
 * it covers, so every stage that ran shows a non-zero share and the sum over stages -- what the reference's UI totals
 * up to the selected stage (src/imgui/imguiApp.cpp:364-376) -- is the measured time.  Off by default on the batch path. */
int hc_enable_profiling(hc_ctx *ctx, int on);
int hc_stage_time_ms(hc_ctx *ctx, int stage, float *ms);
/* Sums over every profiled run since the last reset (up to 256 runs may be in flight between
 * syncs): sum_ms[0] stage 0, [1] fused front kernel (or the tap kernels), [2] hysteresis + expand.
 * This is the accumulating counterpart of timerManager::addTime (src/utils/timer.hpp:27-39). */
int hc_profile_get(hc_ctx *ctx, double sum_ms[3], long *nruns, int reset);
/* The front path's two kernels separately, over the same profiled runs (those that took the split path):
 * sum_ms[0] k_blur, [1] k_nms.  Read it before hc_profile_get(..., reset = 1), which clears both. */
int hc_profile_get_front(hc_ctx *ctx, double sum_ms[2], long *nruns);

/* Steady-state step times: for every pair of consecutive profiled runs since the last reset, the time from the end of
 * one run (its last kernel, hysteresis included) to the end of the next, in ms.  In pipelined mode, where runs overlap,
 * this -- not a run's own start-to-end time -- is what a frame stream sees.  Writes up to `cap` values, *n = how many exist. */
int hc_profile_get_intervals(hc_ctx *ctx, float *ms, int cap, int *n);

/* The front kernels' own time (ms) of every profiled HYSTER run since the last reset, in run order: what bench.py needs
 * to attribute kernel time to the content of each step when the batches of a stream differ.  Writes up to `cap` values,
 * *n = how many exist. */
int hc_profile_get_front_each(hc_ctx *ctx, float *ms, int cap, int *n);

/* Internal device buffers (input frames, output images) and their pitch / frame stride. */
int hc_device_ptrs(hc_ctx *ctx, void **d_in, void **d_out, size_t *in_pitch, size_t *out_pitch, size_t *in_frame_stride,
                   size_t *out_frame_stride);

/* Number of hysteresis launches that did work in the last run, and whether the continuation ran. */
int hc_last_hysteresis_info(hc_ctx *ctx, int *launches_with_work, int *continued);

/* The same, summed over every run completed since the context was created (or since the last call with reset != 0):
 * totals[0] = runs, totals[1] = runs that needed the host-side continuation (each one a stall of a pipelined stream),
 * totals[2] = hysteresis launches that found work, totals[3] = hysteresis launches queued.  Completes pending runs. */
int hc_hysteresis_totals(hc_ctx *ctx, unsigned long long totals[4], int reset);

/* What the last hc_run / hc_run_device did with the caller's buffers -- no silent cliffs: *input_staged / *output_staged are
 * 1 when the frames went through the context's internal pitched buffers (an extra device-to-device copy each: pointer,
 * pitch or frame stride not a multiple of 4, or 3-channel mode O rows without whole 12-byte groups), and *front_form is
 * the front path that ran (Mode R: the HC_OPT_FRONT_SPLIT value 2 / 1 / 0, 4 = k_front8 in its half-strip form, or 5 = k_front_mx; Mode O:
 * 3 = k_front8o, -1 = k_front_o; -1 also for final stages below HYSTER).  Rows that do not hold whole 8-pixel groups
 * (tight rows of a width that is not a multiple of 8) are staged (*input_staged = 1) so that the 8-px kernels can run. */
int hc_last_run_info(hc_ctx *ctx, int *input_staged, int *output_staged, int *front_form);

/* Pipelined mode: how many runs of `nframes` frames the context may keep in flight (4 for small batches -- fewer
 * than 0.5 G pixels per run; big batches: 3 -- the context uses two slots, and a third while it sees the hysteresis of
 * a run outlast the front kernel of the next (frames of several thousand columns); 1 when HC_OPT_PIPELINE is off):
 * the number of output buffers a caller should rotate through so that no run has to wait for an older one that still
 * writes the same memory.  No reference counterpart
 * (the reference processes one frame per synchronous call, src/cvp/cannyEdgeH.cu:49-120). */
int hc_pipeline_depth(hc_ctx *ctx, int nframes);
/* ... and how many slots the ring of the most recent pipelined run had (2 / 3 / 4; 1 when HC_OPT_PIPELINE is off). */
int hc_pipeline_slots_in_use(hc_ctx *ctx);
/* Waves per workgroup of the most recent k_front8 launch: 4, 1 (HC_OPT_FRONT_WPB) or 3 (per-channel mode: one per channel). */
int hc_front_waves_per_workgroup(hc_ctx *ctx);

/* Diagnostics of the last run's queued hysteresis launches: 3 words per launch
 * (sweeps summed over tiles, max sweeps of a tile, tiles that did work). */
int hc_hysteresis_stats(hc_ctx *ctx, unsigned *stats, int nwords);

/* Tuning knobs: rows per front-path work item (0 = auto); hysteresis launches queued per run (0 = auto:
 * 6, or one more than the row tiles + column panels of a frame, or what the last runs needed + 4, at most 96; launches
 * after convergence exit at once, and
 * hc_sync continues from the host in the rare case the queue was too short -- non-monotone, serpentine edges). */
int hc_set_tuning(hc_ctx *ctx, int chunk_rows, int hyst_launches);

/* Options.  HC_OPT_NMS_SATURATE (default 0): the reference stores `min((unsigned char)gradVal, 255)`
 * (src/cvp/cannyEdgeD.cu:267), an out-of-range float->u8 cast for gradients 256..721.  0 = the
 * canonical Mode R reading: the value wraps mod 256 (integer min folds away, low byte stored).
 * 1 = min(g, 255): what the same source line yields when compiled by hipcc for gfx950.
 *
 * HC_OPT_PIPELINE (default 0): throughput mode for back-to-back batches.  1 = the front kernels of
 * run i+1 (on the context stream) overlap the hysteresis + expand of run i (second stream, second set
 * of bit planes).  The front kernels run in order with the caller's own work on the context stream, so a
 * run sees what the caller queued before it and the input may be reused by work queued after it; the OUTPUT
 * of a run is only guaranteed after hc_sync() (or hc_download).  Results are identical in both modes.  Hand
 * consecutive runs different output buffers -- two in turn for big batches, four for small ones (fewer than
 * 0.5 G pixels per run: there four runs are kept in flight, each hysteresis on a stream of its own, because a
 * step is otherwise the latency of the hysteresis' chain of launches).  A run whose output overlaps that of a run
 * still in flight waits for it, and if that is the previous run it still gives the exact map, but without the
 * provisional-map shortcut and a few percent slower.
 *
 * HC_OPT_PER_CHANNEL (default 0, 3-channel contexts only): 1 = instead of the reference's grey
 * conversion, run the detector on each channel separately (BASELINE config "three-channel,
 * per-channel Canny"): the interleaved input is read once per channel by adjacent work items and
 * every run produces 3 edge maps per input frame, output frame 3*f + ch (ch = byte position in the pixel).
 *
 * HC_OPT_FRONT_SPLIT (default 2, Mode R): which kernels form the front path (grey | blur | Sobel | NMS | thresholds).
 * 2 = k_front8: ONE kernel, 8 pixels per lane, no intermediate in HBM (falls back to 1 when an input row does not hold
 * whole 8-pixel groups, i.e. pitch < round_up(width, 8) * channels; and, while the option has not been set by the
 * caller, for big batches (0.1 G pixels or more per run) of narrow frames whose width fills the 248-column strips of
 * the 4-px kernels much better than the 496-column strips of k_front8: up to 248 columns and 497..744 (VGA)); 1 = k_blur + k_nms with a u8 blur plane between
 * them; 0 = k_front, the earlier 4-pixel fused kernel.  Results are identical; 0 and 1 are kept as independent
 * implementations for the parity tests.  Mode O contexts: 2 = k_front8o, the 8-pixel kernel (one-channel sources;
 * 3-channel sources and rows without whole 8-pixel groups use k_front_o), 0 or 1 = k_front_o, the 4-pixel kernel.
 *
 * HC_OPT_L2_GRADIENT (default 0, Mode O contexts): cv::Canny's `L2gradient` argument: magnitude dx^2 + dy^2
 * compared with the squared thresholds instead of |dx| + |dy|. */
/*
 * HC_OPT_DEBUG_TAPS (default 0): parity-test diagnostics.  1 = every HC_STAGE_HYSTER run keeps a copy of what the
 * FAST path's front kernels produced -- the STRONG and CANDIDATE bit planes as they are handed to the hysteresis,
 * and the blur plane (Mode R) -- for hc_debug_tap().  Costs two plane copies per run; never set it when timing.
 *
 * HC_OPT_FRONT_HALF (default -1 = automatic, Mode R): the half-strip form of k_front8 for narrow frames -- a wave is two
 * independent half-waves of 240 columns each, and the (frame, half-strip) units of a run of rows are dealt to them in pairs
 * (640 columns: 1.5 waves per frame instead of 2).  -1 = when it needs fewer waves; 0 = never; 1 = whenever the buffers
 * allow it (parity tests).  hc_last_run_info reports it as front form 4.
 *
 * HC_OPT_FRONT_DENSE (default -1 = automatic, Mode R): k_front8's dense path -- a window of 6 rows that follows one in
 * which more than 512 of the wave's 768 half-lanes passed the low threshold (noise, texture) is processed by wave-wide
 * non-maximum suppression in registers instead of the queue and its batches, until a window counts fewer than 384.
 * 0 = never, 1 = every window (parity tests).  Same results either way.
 *
 * HC_OPT_COPY_STREAMS (default 0): host pipelines of several contexts (cvp::io::FrameStreamer).  1 = hc_upload and
 * hc_download_begin move their data on two copy streams that ALL such contexts of a device share -- one for host ->
 * device, one for device -> host -- tied to the context stream by events, instead of on the context stream itself.
 * Copies that share a stream with kernels do not overlap across contexts on this runtime (three contexts, 32 MiB
 * batches: 28 GB/s each way; with the two copy streams 40, with 64 MiB batches 47 of the 48 GB/s the link carries both
 * ways at once -- tools/experiments/pcie_raw2.hip).
 *
 * HC_OPT_FRONT_WPB (default -1 = automatic, Mode R, mono / BGR input): waves per workgroup of k_front8 in pipelined big
 * batches.  One-wave workgroups refill a retiring wave's slot at once and gain the front kernel 2-3 % beside the
 * hysteresis of the previous run -- which then needs 40 % more stream time.  Automatic: one wave once that hysteresis
 * ends more than 25 % of a front kernel's time before the front kernel it runs beside, four again below 3 % (smoothed,
 * from the runs' own events); small batches (fewer than 0.5 G pixels per run): one wave from 0.12 G pixels.  1 or 4 fixes it.  Same results either way; hc_front_waves_per_workgroup says what ran.
 *
 * HC_OPT_PIPELINE_SLOTS (default -1 = automatic): the ring of big pipelined batches.  Automatic: two slots, and a third
 * while the context sees the hysteresis of a run end after the front kernel of the next one (hc_pipeline_depth).
 * 2 or 3 fixes the ring.  Diagnostics / tests: 20 / 21 = the automatic rule, but told that every chain ends after /
 * before the next front kernel, which walks it through its transitions (2 -> 3 on trial after three runs, kept or given back after ten; 3 -> 2 after sixteen)
 * whatever the content.  Same results with every value.
 *
 * HC_OPT_FRONT_MX (default 0, Mode R, one-channel frames): 1 = k_front_mx, the front path whose two integer
 * contractions (the 5x5 Gaussian sum, the 3x3 Sobel sums) run on the matrix pipe as v_mfma_i32_32x32x32_i8 (strips of 216
 * columns, blocks of 16 rows), for every run that allows it (input pitch >= round_up(width, 4), height x pitch < 2^32;
 * pipelined mode: width % 8 == 0).  Same results bit for bit; hc_last_run_info reports form 5; HC_OPT_FRONT_WPB 1 / 4
 * picks its workgroup size.  Opt-in: measured on the MI355X (profiles/r04/mx_experiments.md) it takes 14 % less time than
 * k_front8 alone (1.83 against 2.14 ms per 1024 camera-like 1080p frames), 3-5 % less beside the hysteresis of the batch
 * before (2.37-2.43 against 2.50 ms), and half as much again on frames full of candidates (iid noise: 7.7 against 5.1 ms),
 * for which it has no dense path. */
enum { HC_OPT_NMS_SATURATE = 1, HC_OPT_PIPELINE = 2, HC_OPT_PER_CHANNEL = 3, HC_OPT_FRONT_SPLIT = 4, HC_OPT_L2_GRADIENT = 5, HC_OPT_DEBUG_TAPS = 6, HC_OPT_FRONT_HALF = 7,
       HC_OPT_FRONT_DENSE = 8, HC_OPT_COPY_STREAMS = 9, HC_OPT_PIPELINE_SLOTS = 10, HC_OPT_FRONT_WPB = 11, HC_OPT_FRONT_MX = 12,
       /* test and diagnostic hooks (the library reads no environment variables): hysteresis launches >= 1 on a fixed grid with
        * worklists (-1: lists, default grid); the looping hysteresis launch of small runs off (0) / on; per-launch statistics for
        * tools/hyst_diag.py; hysteresis workgroup shape (rows x 100 + waves, 0 = by the rule); k_front8's dense-path thresholds */
       HC_OPT_TEST_HYST_LATE_GRID = 100, HC_OPT_TEST_HYST_LOOP = 101, HC_OPT_TEST_HYST_DIAG = 102, HC_OPT_TEST_HYST_GEOM = 103,
       HC_OPT_TEST_DENSE_ENTER = 104, HC_OPT_TEST_DENSE_LEAVE = 105 };
int hc_set_option(hc_ctx *ctx, int option, int value);

/* The fast path's own intermediates of the last HC_STAGE_HYSTER run (HC_OPT_DEBUG_TAPS must have been set before it),
 * as tight-or-strided host u8 images, one per output frame:
 *   HC_TAP_BLUR       the Gaussian blur the front kernels computed (reference: gaussianFilter5x5 output,
 *                     src/cvp/cannyEdgeD.cu:72-118); Mode R only
 *   HC_TAP_THRESH     the double-threshold map 0 / 128 / 255 (doubleThreshold output, cannyEdgeD.cu:273-293) rebuilt
 *                     from the two bit planes: 255 = STRONG bit, 128 = CANDIDATE bit only
 * These are NOT the plain per-stage kernels behind hc_run(final_stage < HYSTER): they read back what k_blur / k_nms /
 * k_front / k_front_o wrote, so that the parity tests can check the fast path stage by stage. */
enum { HC_TAP_BLUR = 1, HC_TAP_THRESH = 2 };
int hc_debug_tap(hc_ctx *ctx, int what, uint8_t *host, size_t row_stride, size_t frame_stride, int nframes);

/* Page-locked host memory for frame staging: hc_upload / hc_download on such buffers are true asynchronous DMA
 * (the reference uploads from pageable cv::Mat memory with a blocking cudaMemcpy2D, cannyEdgeH.cu:136/144).
 * Used by cvp::io::FrameStreamer (include/cvp/frameIO.hpp).  NULL on failure. */
void *hc_host_alloc(size_t bytes);
void hc_host_free(void *p);

/* Device self-test of the cross-lane / packed-math primitives the kernels rely on. 0 = ok. */
int hc_selftest(int device);

const char *hc_last_error(void);
const char *hc_version(void);

#ifdef __cplusplus
}
#endif
#endif
