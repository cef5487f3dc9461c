/*
 * lol_gpu.h — C ABI of the MI355X (gfx950) renderer for loltracer's per-pixel path.
 *
 * This is the drop-in boundary.  A C host that today links
 * naive_renderer.c gets the same three renderer.h entry points from
 * integration/hip_renderer.c, which is a thin adapter over the functions
 * below; INTEGRATION.md shows the binding.  Plain pointers and sizes only: no
 * C++ types, no exceptions, no torch.  Every function returns 0 on success or
 * a negative lol_gpu_status; lol_gpu_error() gives the text of the last error
 * of a context.
 *
 *   reference                                         here
 *   ------------------------------------------------  ---------------------------
 *   render_prepare()   renderer.h:25, main.c:161      lol_gpu_create + lol_gpu_upload_program
 *   render_thread() pixel loop
 *                      naive_renderer.c:207-236       lol_gpu_render_device / lol_gpu_render_host
 *   render_destroy()   renderer.h:26, main.c:213      lol_gpu_destroy
 *   row self-scheduling naive_renderer.c:216          the launch grid (+ `bands` for multi-GPU)
 *
 * There is no CPU fallback: every entry point fails with LOL_GPU_ERR_NO_DEVICE
 * when no HIP device is usable.
 */
#ifndef LOL_GPU_H
#define LOL_GPU_H

#include <stddef.h>
#include <stdint.h>
#include "lol_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Version of this binary interface.  It changes whenever an entry point changes its arguments or a structure its layout
 * (4: lol_program carries pointers + counts instead of fixed-capacity arrays; lol_gpu_render_host_end takes the surface's
 * size; lol_gpu_rows is {band, cycle, offset}.  5: lol_gpu_set_frames_in_flight / lol_gpu_next_stream; render_host_begin
 * takes up to four frames; lol_gpu_tuning_switches.  6: lol_gpu_verify_shadow_division is gone with the shortcut it
 * proved; lol::Launch carries the march's first step).  A host compiled against another version must not run: hip_renderer.c
 * and the Python mirror compare lol_gpu_abi_version() of the library they loaded with the macro they were built with.
 */
#define LOL_GPU_ABI_VERSION 6
int lol_gpu_abi_version(void);

typedef struct lol_gpu lol_gpu;      /* one renderer context = one device + one scene */

enum lol_gpu_status {
	LOL_GPU_OK              =  0,
	LOL_GPU_ERR_NO_DEVICE   = -1,
	LOL_GPU_ERR_HIP         = -2,    /* a HIP runtime call failed; see lol_gpu_error() */
	LOL_GPU_ERR_ARG         = -3,
	LOL_GPU_ERR_NO_PROGRAM  = -4,
	LOL_GPU_ERR_UNSUPPORTED = -5
};

/*
 * Which rows of the frame one call renders, and where they land.
 *
 * The frame's rows are cut into cycles of `cycle_rows` rows; inside every cycle a part owns the `band_rows` rows that
 * start at `offset_rows`.  A call renders the bands of one part and writes them compactly: local row
 * r = c * band_rows + i of the destination holds frame row c * cycle_rows + offset_rows + i (i < band_rows).
 * {band_rows = cycle_rows = h, offset_rows = 0} is the whole frame in place.  This is the multi-GPU row-tile partition
 * (SURVEY.md §8e): every rank renders its part with ONE launch and the parts are gathered.  Equal shares of n parts with
 * bands of b rows are {b, n * b, part * b}; the parts of one split may also differ in band height — how the root, which
 * also receives and un-interleaves the whole frame, gets a smaller share (lol_gpu_multi_set_root_band_rows).
 */
typedef struct lol_gpu_rows {
	int32_t band_rows;
	int32_t cycle_rows;
	int32_t offset_rows;
} lol_gpu_rows;

/*
 * How a colour becomes a pixel: SDL_MapRGB(surf->format, r, g, b) of renderer.h:17-22 for a non-palettised format
 * (SDL2, src/video/SDL_pixels.c):  (r >> Rloss) << Rshift | (g >> Gloss) << Gshift | (b >> Bloss) << Bshift | Amask.
 * The fields are SDL_PixelFormat's.  The reference stores one Uint32 per pixel whatever the format
 * (naive_renderer.c:233-235), so only 4-byte formats make sense; anything else is refused, never approximated.
 * Default (and NULL): XRGB8888 = shifts 16 / 8 / 0, no loss, Amask 0.
 */
typedef struct lol_gpu_pixel_format {
	uint8_t  r_shift, g_shift, b_shift;
	uint8_t  r_loss, g_loss, b_loss;
	uint8_t  bytes_per_pixel;     /* must be 4 */
	uint8_t  palettised;          /* must be 0 (format->palette != NULL) */
	uint32_t a_mask;
} lol_gpu_pixel_format;

/* Optional per-pixel diagnostics, all device pointers, each may be NULL.
 * Indexed by local row like the pixel destination, `w` elements per row. */
typedef struct lol_gpu_debug {
	float*    rgb;        /* 3 floats per pixel: post-gamma, pre-quantisation colour  */
	float*    hit_dist;   /* get_intersection().dist                                  */
	uint32_t* hit_id;     /* get_intersection().id                                    */
	uint32_t* steps;      /* low 16 bits: march steps; high 16 bits: shadow steps sum */
} lol_gpu_debug;

int  lol_gpu_device_count(void);
int  lol_gpu_create(int device, lol_gpu** out);
void lol_gpu_destroy(lol_gpu* ctx);
const char* lol_gpu_error(const lol_gpu* ctx);
int  lol_gpu_device(const lol_gpu* ctx);          /* the HIP device ordinal of a context */

/* Stream arguments: NULL = the context's own (non-blocking) stream; LOL_GPU_STREAM_DEFAULT = HIP's legacy
 * default stream (hipStreamLegacy, the stream a handle of 0 means to HIP itself); else a hipStream_t. */
#define LOL_GPU_STREAM_DEFAULT ((void*)1)

/* Copy the flattened scene (lol_scene_flatten) to the device.  May be called
 * again at any time (it first waits for everything queued on the device);
 * frames issued afterwards use the new program.  All or nothing: when it fails
 * (malformed program, a failed allocation or copy) the context keeps rendering
 * the scene it had.  The program is copied: the caller may free it on return.
 *
 * Returns as soon as the scene can be rendered — like the reference's
 * render_prepare (naive_renderer.c:242-244 does nothing; the JIT's takes
 * milliseconds, tracing_jit_renderer.dasc:416-434): the tables and the
 * interpreter's lists are on the device, and the scene's own kernel is being
 * compiled by hipRTC on a host thread (0.6 s for scene4, half a minute for
 * 5000 ops).  Frames render on the interpreter kernel meanwhile and switch to
 * the scene's kernel at the first frame boundary after the compiler is done —
 * same pixels on both.  lol_gpu_specialize_wait() blocks until then. */
int lol_gpu_upload_program(lol_gpu* ctx, const lol_program* prog);

/* Pixel format of every frame launched afterwards (NULL = XRGB8888).  LOL_GPU_ERR_UNSUPPORTED for palettised or
 * non-32-bit formats and shifts / losses that do not describe 8-bit channels in a 32-bit word. */
int lol_gpu_set_pixel_format(lol_gpu* ctx, const lol_gpu_pixel_format* fmt);

/* Number of local rows a part owns (for sizing destinations). */
int lol_gpu_part_rows(int h, const lol_gpu_rows* rows);

/*
 * Launch one frame (or one part of it) asynchronously.
 *   cam       per-frame camera constants (lol_frame_camera_init)
 *   w, h      frame size in pixels; max_steps = MAX_STEPS of the primary march
 *   rows      partition (NULL = whole frame)
 *   dst       DEVICE pointer to 32-bit pixels in the format of lol_gpu_set_pixel_format (default XRGB8888 =
 *             r<<16|g<<8|b), `pitch_bytes` per local row
 *   dbg       optional diagnostics (NULL in production)
 *   stream    hipStream_t to launch on, as void*; NULL = the context's own stream (see LOL_GPU_STREAM_DEFAULT)
 */
int lol_gpu_render_device(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps,
                          const lol_gpu_rows* rows, void* dst, size_t pitch_bytes,
                          const lol_gpu_debug* dbg, void* stream);

/*
 * Whole frame into a HOST surface (what render_thread does with surf->pixels,
 * naive_renderer.c:233-235): renders into the context's device framebuffer,
 * copies `h` rows of w*4 bytes honouring `pitch_bytes`, and waits.
 *
 * The surface is the host's memory: the library keeps nothing about it between calls and never registers it with the
 * device (the HIP runtime pins a copy's destination for the duration of the copy: PCIe line rate, measured), so a host
 * may free, move or resize its surface between any two frames (main.c:182-187).
 */
int lol_gpu_render_host(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps,
                        void* host_pixels, size_t pitch_bytes);

/*
 * The same with two frames in flight, for hosts that can give the next frame's camera before they consume the
 * previous frame (an orbit, a recorded path; an interactive host trades one frame of latency for it):
 *     begin(cam[0]);  for i: { begin(cam[i+1]);  end(surface);  present frame i }
 * begin() queues the frame's kernel and returns; end() copies the OLDEST queued frame into the host surface and
 * waits for that copy — which runs while the next frame's kernel does (main.c:182-194 is the loop this overlaps).
 * At most two frames may be begun and not yet ended; lol_gpu_render_host_pending() says how many are.
 *
 * The surface may change size between begin() and end() (the reference's window is resizable, main.c:157,182-187):
 * end() takes the size of the surface it is given and refuses (LOL_GPU_ERR_ARG, nothing written, frame kept) a frame
 * of another size; lol_gpu_render_host_pending_size() tells the size of the oldest queued frame beforehand and
 * lol_gpu_render_host_discard() drops every queued frame.  Frames of different sizes may be in flight together.
 *
 * The kernels of consecutive frames are queued on different streams of the context (round 5), so frame i+1's first waves
 * fill the tail of frame i's launch — the overlap the reference's sequential loop (main.c:189-194) cannot have and a host
 * with the next camera in hand can: what a MOVING camera gains from having frames in flight (the copy is hidden either
 * way).  After lol_gpu_set_frames_in_flight(ctx, n) up to n (<= 4) frames may be begun and not yet ended.
 */
int lol_gpu_render_host_begin(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps);
int lol_gpu_render_host_end(lol_gpu* ctx, void* host_pixels, size_t pitch_bytes, int w, int h);
int lol_gpu_render_host_pending(const lol_gpu* ctx);
int lol_gpu_render_host_pending_size(const lol_gpu* ctx, int* w, int* h);   /* 0 x 0 when nothing is queued */
int lol_gpu_render_host_discard(lol_gpu* ctx);

/* Wait for everything issued on the context's own stream(s). */
int lol_gpu_sync(lol_gpu* ctx);

/*
 * Frames in flight.  The reference renders frame i+1 when frame i has been shown (main.c:189-194); a frame here is ONE
 * launch that ends with its slowest waves on half-empty SIMDs, and a host whose frames are independent — an orbit, a
 * recorded path, the stripes of BASELINE.json's config 5 — loses that tail once per frame.  With n > 1 the frames launched
 * on the context's own stream (stream == NULL in lol_gpu_render_device) go to n streams of the context in turn, so up to n
 * of them run together and the next frame's first waves fill the last one's tail (+9 % scene4 at 4K, +13 % its orbit, +67 %
 * scene.lol at 1080p: profiles/r4_stream_overlap_ab.jsonl).  What the caller takes on: frames that may be in flight
 * together need destinations of their own — frame i and frame i + n share a stream, so a ring of n destinations is safe —
 * and only lol_gpu_sync() (or an event recorded on lol_gpu_next_stream() after the launch) says when a frame is there.
 * A repeated view keeps its schedule on every stream (lol_gpu_set_tile_order: one set of tables per stream).
 * n = 1 (the default) is the sequential behaviour.  Also the number of frames lol_gpu_render_host_begin accepts before an
 * _end (never fewer than two).  1 <= n <= 4.  Waits for the frames in flight before it changes the rotation.
 */
int   lol_gpu_set_frames_in_flight(lol_gpu* ctx, int n);
int   lol_gpu_frames_in_flight(const lol_gpu* ctx);
/* The hipStream_t (as void*) the NEXT frame launched with stream == NULL will be queued on: for hosts that bracket their
 * frames with events of their own. */
void* lol_gpu_next_stream(lol_gpu* ctx);

/* Raw device memory helpers so a C host needs no HIP headers. */
int lol_gpu_malloc(lol_gpu* ctx, size_t bytes, void** out);
int lol_gpu_free(lol_gpu* ctx, void* ptr);
int lol_gpu_memcpy_d2h(lol_gpu* ctx, void* host, const void* dev, size_t bytes);

/* Name of the kernel a launch uses (for matching rocprofv3 kernel-trace rows):
 * "lol_render_spec" (scene-specialised, compiled by hipRTC at upload) or "render_interp". */
const char* lol_gpu_kernel_name(const lol_gpu* ctx);
/* Identity of the code that kernel is: 16 hex digits — FNV-1a of the hipRTC code object for "lol_render_spec"; for
 * "render_interp" of {this library's build id (a digest of its sources and compiler flags), the uploaded macro-op lists,
 * the square-root variant}: everything that decides which instructions the interpreter executes.  Profiles record it
 * (profiles/pmc_traffic.json) so that a counter figure is only ever quoted for the code it was measured on. */
const char* lol_gpu_kernel_key(const lol_gpu* ctx);
/* The environment: the library reads ten LOL_GPU_* switches for A/B runs and debugging (INTEGRATION.md lists them), and honours
 * them ONLY in a process that also sets LOL_GPU_TUNING=1 — a host that sets none of these runs the defaults whatever its shell
 * holds (lol_gpu_diag.h: lol_gpu_tuning_switches tells which took effect).  Not fenced: LOL_GPU_CACHE_DIR, LOL_GPU_ROCTX — where
 * code objects are kept, whether frames are marked for rocprofv3: neither changes what is computed. */

/*
 * Scene specialisation (the GPU analogue of the reference's tracing JIT, whose render_prepare
 * compiles the scene's SDF to x86 — tracing_jit_renderer.dasc:416-434).  lol_gpu_upload_program()
 * generates the scene's SDF as straight-line HIP and compiles it with hipRTC; when that is
 * disabled (set_specialize(ctx, 0) before the upload, or LOL_GPU_SPECIALIZE=0) or fails, frames
 * are rendered by the ahead-of-time interpreter kernel instead — same bits either way.
 */
/* enable: 0 = interpreter, plain arithmetic; 1 = specialise, with the proven-exact shortcuts (default);
 * 3 = specialise without them; 4 = interpreter with them; 5 = like 1, but a scene of 257 ... 1024 ops keeps its first kernel
 * (the SDF as one out-of-line function) and nothing is compiled behind it.  Takes effect at the next lol_gpu_upload_program.
 * What a host should know about the scene compiler: it runs on a thread of its own and cannot be interrupted; lol_gpu_destroy and
 * the next lol_gpu_upload_program wait for a run that is still at work — up to about 3 s for a first run, and for the SECOND run
 * of a mid-size scene (the form with the SDF inlined, 14 - 88 % faster) 2 - 18 s.  A host that would rather exit or switch
 * scenes promptly than have that kernel asks for 5. */
int         lol_gpu_set_specialize(lol_gpu* ctx, int enable);
/* The largest program (ops) the scene compiler takes on; larger scenes render on the interpreter, which reads them as data.
 * hipRTC cannot be interrupted once it runs — lol_gpu_destroy and the next upload's compile wait for it — and takes about
 * n^1.5: 5 s at 1300 ops, 41 s at 5000, minutes beyond 10,000 (fields of objects on the GPU box).  Default (and 0): 6144 ops,
 * about a minute.  A host that would rather wait for the 2x faster kernel of a huge scene raises it.  Takes effect at the next
 * lol_gpu_upload_program. */
int         lol_gpu_set_specialize_max_ops(lol_gpu* ctx, unsigned max_ops);
const char* lol_gpu_specialize_log(const lol_gpu* ctx);
/* Tiered start-up (lol_gpu_upload_program): wait for the scene compiler and switch to its kernel now (returns at once when
 * nothing is being compiled).  For tests and benchmarks that want to time or inspect one particular kernel. */
int         lol_gpu_specialize_wait(lol_gpu* ctx);
/* 0 = no scene kernel (switched off, or the scene is above the cap of lol_gpu_set_specialize_max_ops), 1 = being compiled, 3 =
 * compiled, takes over at the next frame, 2 = in use, -1 = the compiler failed (the interpreter renders; reason in
 * lol_gpu_specialize_log).  Scenes of 257 ... 1024 ops get TWO kernels, one after the other (round 5): the form with the SDF as
 * one out-of-line function, which hipRTC delivers in 0.4 - 3 s, and then the form with the SDF inlined into the three loops, 14 -
 * 88 % faster, in 3 - 18 s: 5 = the first is in use and the second being compiled, 6 = the second is compiled and takes over
 * at the next frame.  Same pixels from all of them; lol_gpu_specialize_wait waits for the last.
 * *compile_ms (may be NULL) = what the scene compiler's last finished run took. */
int         lol_gpu_specialize_state(lol_gpu* ctx, double* compile_ms);
/* (The proofs behind those shortcuts can be run one by one, and the culling bounds, the SDF and the powf asked for on
 * their own: include/lol_gpu_diag.h — nothing a frame needs.) */
/*
 * Escaped rays are shaded with material #0 (naive_renderer.c:103-112).  When that material has
 * diffuse == specular == 0, shininess >= 0 and all light intensities are finite, their colour is exactly
 * clamp(ambient_color * material.ambient) whatever the normal and shadow factors are, so a wavefront whose
 * rays ALL escaped skips the normal taps and shadow marches.  On by default when the uploaded program
 * qualifies (checked on the host); set_miss_skip(ctx, 0) turns it off.  With it on,
 * lol_gpu_debug.steps reports 0 shadow steps for the skipped pixels.
 *
 * The same switch governs the per-light form: where a surface faces away from a light (diffuse incidence
 * clamps to exactly 0) both Phong terms of that light are +-0 for any shadow factor, so that lane does not
 * march the shadow ray (needs all light intensities and material colours finite and every shininess >= 0).
 *
 * And the shadow march itself: softshadow returns maxf(res, 0) (naive_renderer.c:88-89), and once res <= 0 no later step
 * can bring the factor back above 0 — min only lowers it, and no NaN can appear when every number of the scene, its lights
 * and the camera is finite and below 10^15 (checked on the host) — so a lane's march ends there instead of going on to
 * res < -1 or t > L; a ray that grazes along just inside a surface otherwise takes all 128 steps while its wave waits
 * (C3: 5630 -> 7510 Mpixels/s).  Same pixels; lol_gpu_debug.steps counts the steps really marched.
 * lol_gpu_miss_skip_active(): bit 0 = escaped-wave skip, bit 1 = zero-incidence skip, bit 2 = settled-shadow exit.
 */
int         lol_gpu_set_miss_skip(lol_gpu* ctx, int enable);          /* all three skips on (default) or off */
/* The same, skip by skip: bit 0 = escaped waves, bit 1 = zero incidence, bit 2 = settled shadows (for tests that hold ONE
 * skip's step counts against known answers). */
int         lol_gpu_set_exact_skips(lol_gpu* ctx, unsigned mask);
int         lol_gpu_miss_skip_active(const lol_gpu* ctx);
/*
 * Exact culling of top-level objects in the specialised kernel (lol_gpu.hip, "exact culling"): sdf() is a strict-'<'
 * minimum over the objects (naive_renderer.c:31-44), so an object that a bounding sphere PROVES farther away than the
 * running minimum is not evaluated — for a whole wavefront at a time, and only then.  Same pixels, same step counts.
 * On by default; set_cull(ctx, 0) before the upload turns it off.
 */
int         lol_gpu_set_cull(lol_gpu* ctx, int enable);
/*
 * The order in which a launch hands out its tiles of 16x4 pixels.  Same pixels in every order; the time differs, because a
 * frame is one launch of blocks that differ 100x in cost and ends with its slowest waves running on half-empty SIMDs.
 *   LOL_GPU_TILES_LPT (the default): while the camera stands still, a frame is scheduled by what the frame before it cost.
 *     A frame under the same camera as the frame before it goes through tables: the pixels of every 64x16 region are dealt
 *     to the region's sixteen waves by the SDF evaluations they needed (a wave runs every loop to its slowest lane), every
 *     wave reports how long it ran, and a counting sort on the device (three small kernels on the frame's stream) hands the
 *     waves of the following frames out longest first — scheduling with costs that are exact, because nothing moved.  Only
 *     the schedule is reused: every pixel is computed from scratch in every frame, and comes out the same.  scene4 at 4K
 *     +24 % over the better of the two fixed orders, scene.lol at 1080p +40 %, the bands of an 8-way split of the 8K frame
 *     +34 % (LABNOTES.md §3.9).  A frame whose
 *     camera differs from its predecessor's is launched in the better fixed order (as LOL_GPU_TILES_AUTO finds it), with
 *     no table, cost or sort: stale costs are worse than no costs (the reference's arrow keys turn the camera 5.7 degrees
 *     a frame).  The tables live on the stream the first such frame was launched on; frames of the same geometry on
 *     other streams are launched in the fixed order.
 *   LOL_GPU_TILES_ROWS / LOL_GPU_TILES_COLS: row by row / column by column (the launch grid transposed).
 *   LOL_GPU_TILES_AUTO: the better of those two, measured: the first frames of a (scene, frame size, row partition, max_steps)
 *     are launched in both orders alternately, each between two HIP events on its launch stream (LOL_GPU_TILE_TRIALS frames
 *     per order after a few untimed ones; nothing ever waits), and the order whose typical frame is faster by more than
 *     1 % is kept — rows otherwise — until the scene, the size or the partition changes.
 * Takes effect at the next frame.  The reference has no counterpart: its thread pool claims pixels one by one
 * (naive_renderer.c:216) — which IS dynamic load balancing; this is its counterpart for a launch whose order is fixed up front.
 */
enum { LOL_GPU_TILES_ROWS = 0, LOL_GPU_TILES_COLS = 1, LOL_GPU_TILES_AUTO = 2, LOL_GPU_TILES_LPT = 3 };
#define LOL_GPU_TILE_TRIALS 16
int         lol_gpu_set_tile_order(lol_gpu* ctx, int order);
/* What the context is doing about it: mode = what was asked for; order = the order of the next frame outside a trial
 * (LOL_GPU_TILES_ROWS until AUTO has decided / until longest-first has its tables); deciding != 0 while AUTO's trials are
 * still being launched or collected, or before longest-first has sorted once; decisions = AUTO decisions / sorts so far;
 * rows_ms / cols_ms = the typical trial frame of each order behind AUTO's last decision (0 otherwise). */
typedef struct lol_gpu_tile_order_info {
	int32_t mode, order, deciding, decisions;
	float   rows_ms, cols_ms;
} lol_gpu_tile_order_info;
int         lol_gpu_tile_order(lol_gpu* ctx, lol_gpu_tile_order_info* out);
/* No device needed: writes <out_base>.hip (generated source) and <out_base>.co (code object for `arch`).
 * assume_fast != 0 generates the shortcuts without proof — for ISA inspection only, never for rendering. */
int         lol_gpu_compile_offline(const lol_program* prog, const char* arch, const char* out_base,
                                    int assume_fast, char* log, size_t logcap);

/* ------------------------------------------------------------------------------------------------
 * Several devices behind the same boundary (BASELINE.json north_star: "the image is
 * row-tile-partitioned across the 8 GPUs of one node with an RCCL gather over xGMI").
 *
 * One process, one renderer context + two HIP streams per device, one RCCL communicator per device
 * (ncclCommInitAll).  A frame's rows are cut into bands dealt round-robin over the devices — the static
 * form of the reference's row self-scheduling (naive_renderer.c:216: workers claim rows from an atomic
 * counter; every row is independent).  Device r renders part r compactly on its render stream; on its
 * exchange stream every device ncclSend()s its part to device 0 (devices[0], the root), which
 * ncclRecv()s them (one grouped call, parts may differ in size) and un-interleaves the bands into the
 * destination — the frame barrier of main.c:189-194 becomes the completion of that exchange.  Parts are
 * double-buffered, so frame i's exchange overlaps frame i+1's kernels.  RCCL is loaded (dlopen) and the
 * communicators are created by the first lol_gpu_multi_render_device: single-device users, and hosts that only ever
 * render into HOST surfaces (which need no exchange, see lol_gpu_multi_render_host), never touch it.  No CPU
 * fallback: a device listed twice fails at creation, a missing RCCL at the first frame that needs the exchange.
 */
typedef struct lol_gpu_multi lol_gpu_multi;

#define LOL_GPU_MULTI_MAX_DEVICES 16

int  lol_gpu_multi_create(const int* devices, int n_devices, lol_gpu_multi** out);
void lol_gpu_multi_destroy(lol_gpu_multi* m);
const char* lol_gpu_multi_error(const lol_gpu_multi* m);
int  lol_gpu_multi_device_count(const lol_gpu_multi* m);
/* the single-device context of device index i (to set its switches before the upload, read its logs) */
lol_gpu* lol_gpu_multi_context(lol_gpu_multi* m, int i);
/* render_prepare: the flattened scene goes to every device; the scene's kernel is compiled once, in the background
 * (lol_gpu_upload_program), and every device switches to it at its own next frame; _specialize_wait waits on all. */
int  lol_gpu_multi_upload_program(lol_gpu_multi* m, const lol_program* prog);
int  lol_gpu_multi_specialize_wait(lol_gpu_multi* m);
/* Band height used for frames of height h over n parts: the largest multiple of the 4-row wave patch <= 16
 * that gives equal parts, else the tallest one that still gives every part at least eight bands, else 4.
 * lol_gpu_multi_set_band_rows(m, b > 0) overrides. */
int  lol_gpu_choose_band_rows(int h, int n_parts);
int  lol_gpu_multi_set_band_rows(lol_gpu_multi* m, int band_rows);
/* Frame row shown by local row `local_row` of a part (the inverse of the kernel's mapping); -1 if out of range. */
int  lol_gpu_part_frame_row(int h, const lol_gpu_rows* rows, int local_row);
/*
 * One frame over all devices, asynchronously: on return the work is queued; the assembled XRGB8888 frame
 * is in `dst` (memory of the root device, `pitch_bytes` per row) once lol_gpu_multi_sync() returns.
 * Up to two frames may be in flight (double-buffered parts); give them different destinations.
 */
int  lol_gpu_multi_render_device(lol_gpu_multi* m, const lol_frame_camera* cam, int w, int h, int max_steps,
                                 void* dst, size_t pitch_bytes);
/* What render_thread does with surf->pixels: one frame over all devices into a HOST surface; waits.  No exchange is
 * needed for this: every device copies its own bands into the surface over its own PCIe link, all devices at once —
 * one host thread per device issues that device's strided copies (one per part): a copy into pageable memory occupies
 * the thread that issues it, so N threads are what makes N links run in parallel.  set_host_via_root(m, 1)
 * instead assembles on the root with the RCCL exchange and copies from there. */
int  lol_gpu_multi_render_host(lol_gpu_multi* m, const lol_frame_camera* cam, int w, int h, int max_steps,
                               void* host_pixels, size_t pitch_bytes);
int  lol_gpu_multi_set_host_via_root(lol_gpu_multi* m, int enable);
int  lol_gpu_multi_set_pixel_format(lol_gpu_multi* m, const lol_gpu_pixel_format* fmt);
int  lol_gpu_multi_set_tile_order(lol_gpu_multi* m, int order);         /* lol_gpu_set_tile_order on every device (each decides for itself under AUTO) */
/* Parts per device (default 1): the frame is cut into n * parts parts, part p belonging to device p % n, each part one
 * launch.  Finer interleaving of the rows, and the way a single-GPU machine exercises the multi-part code paths.
 * n * parts <= 64. */
int  lol_gpu_multi_set_parts_per_device(lol_gpu_multi* m, int parts);
/* The cost-weighted split: the root (devices[0]) also receives and un-interleaves the whole frame, so with an equal
 * share it finishes last.  root_band_rows > 0 makes the bands of the root's parts that tall instead of band_rows
 * (0 = like the others): its share of the rows is root_band_rows / (root_band_rows + (n - 1) * band_rows).  Still one
 * launch per part: only the geometry changes (lol_gpu_rows). */
int  lol_gpu_multi_set_root_band_rows(lol_gpu_multi* m, int root_band_rows);
/* The geometry itself, pure host logic: out[p] for p in [0, n_parts) = the lol_gpu_rows of part p when every part's
 * bands are band_rows tall except those of parts p % root_stride == 0, which are root_band_rows tall (0 = band_rows too;
 * root_stride = number of devices).  The bands tile a cycle in part order. */
int  lol_gpu_split_rows(int n_parts, int band_rows, int root_band_rows, int root_stride, lol_gpu_rows* out);
int  lol_gpu_multi_sync(lol_gpu_multi* m);
/* Memory on the root device (for destinations of lol_gpu_multi_render_device). */
int  lol_gpu_multi_malloc(lol_gpu_multi* m, size_t bytes, void** out);
int  lol_gpu_multi_free(lol_gpu_multi* m, void* ptr);
int  lol_gpu_multi_memcpy_d2h(lol_gpu_multi* m, void* host, const void* dev, size_t bytes);
/*
 * The root's assembly step on its own: `parts` holds n_parts compact parts back to back (part r starts at
 * row sum of lol_gpu_part_rows of the parts before it, w pixels per row); un-interleave them into `dst`.
 * Asynchronous on `stream` (NULL = the context's stream).  Exposed so the band logic can be checked on a
 * single device against a whole-frame render.
 */
int  lol_gpu_assemble_parts(lol_gpu* ctx, const void* parts, int n_parts, int band_rows, int w, int h,
                            void* dst, size_t pitch_bytes, void* stream);
/* The same for any split and any placement (parts gathered per rank and padded to a common size, bands of different
 * heights): part_rows[p] = the geometry of part p (the bands of all parts must tile one cycle, in order),
 * part_row0[p] = the row of `parts` (w pixels per row) where part p's compact copy starts. */
int  lol_gpu_assemble_parts_at(lol_gpu* ctx, const void* parts, const lol_gpu_rows* part_rows, const uint32_t* part_row0,
                               int n_parts, int w, int h, void* dst, size_t pitch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOL_GPU_H */
