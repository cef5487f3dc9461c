/*
 * lol_sched.hip — the order in which a frame's tiles are handed out (lol_gpu_set_tile_order): the two fixed orders and the trials
 * that choose between them, and — for a view that repeats — waves handed out longest first and pixels dealt to waves by what they
 * cost the frame before (the tables, the kernels that make them, one set per stream).  Scheduling only: every pixel is computed
 * from scratch in every frame.  (Part of liblol_gpu.so; see lol_gpu_internal.h for how the library is cut.)
 */
#include "lol_gpu_internal.h"

/* (lol_gpu_set_tile_order / lol_gpu_tile_order are declared extern "C" by include/lol_gpu.h; everything else stays inside the library) */
#pragma GCC visibility push(hidden)

#pragma GCC visibility pop
extern "C" int lol_gpu_set_tile_order(lol_gpu* ctx, int order) {
	if (!ctx || order < LOL_GPU_TILES_ROWS || order > LOL_GPU_TILES_LPT) return LOL_GPU_ERR_ARG;
	lol_gpu::TileAuto& T = ctx->tiles;
	T.mode = order;
	T.deciding = false;                      /* a running series of trials is abandoned (its events are simply reused) */
	T.key[0] = 0;                            /* ... and AUTO starts afresh at the next frame */
	T.chosen = order == LOL_GPU_TILES_COLS ? LOL_GPU_TILES_COLS : LOL_GPU_TILES_ROWS;
	/* longest-first starts afresh too: its tables stay allocated, and stay on their streams — frames launched through them may
	 * still be in flight, and whatever rewrites a set does so on the set's own stream, behind them (TileLpt::launched) */
	for (lol_gpu::TileLpt& P : ctx->lpt) P.key[0] = 0;
	ctx->lpt_have_last = false;
	ctx->lpt_last_set = -1;
	return LOL_GPU_OK;
}
#pragma GCC visibility push(hidden)

/*
 * Longest tiles first.  A frame is ONE launch of one-wave blocks (129,600 for C3) that the hardware hands out in block order;
 * blocks differ 100x in cost (sky against penumbra), the launch ends when the LAST wave ends, and while the slowest waves
 * of the tail run the SIMDs stand half empty — frames issued on three streams so that the next frame's waves fill that
 * tail render 9 % (C3) to 67 % (scene.lol at 1080p) faster (tools/stream_overlap_ab.py, profiles/r4_stream_overlap_ab.jsonl),
 * but the reference's frame loop is sequential (main.c:189-194).  The same packing INSIDE one frame: hand the tiles out in the
 * order of decreasing cost (list scheduling, longest processing time first), the cost being what the tile cost in the frame
 * before — the camera moves a little per frame (main.c:70-112), a still camera not at all.  Every block writes how long its
 * wave ran (shader clock, 32 x log2: store_pixel, tile_cost); a counting sort on the device (three small kernels on the
 * frame's stream, once after the first frame of a scene / size and then every LPT_RESORT frames) turns the costs into the
 * next order table; the kernel reads its tile from the table (tile_of_block).  Same pixels: only the ORDER in which tiles
 * are rendered changes.  Measured, one stream, one box (tools/tile_order_ab.py, profiles/r4_tile_order_ab.jsonl,
 * r4_lpt_sweep*.txt; best of rows / columns -> longest first): C3 7850 -> 8470 Mpixels/s, scene.lol at 1080p 14,400 ->
 * 18,900, rank 0's bands of an 8-way C4 split 7370 -> 8580, a whole C4 frame 8600 -> 8770, the orbit (the costs lag the
 * camera by up to LPT_RESORT frames) 10,050 -> 10,300.  (The run time predicts better than the evaluation count, the first
 * cost tried: C3 8200.)
 */
constexpr unsigned LPT_BUCKETS = 1024, LPT_THREADS = 256, LPT_RESORT = 16;

/* (the tables are stored XCD by XCD: lol_kernel.h, tile_slot)
 * launch position i shades wave slot i */
__global__ __launch_bounds__(LPT_THREADS) void lpt_identity_kernel(uint32_t* order, uint32_t n, uint32_t stride) {
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	if (i < n) order[lol::tile_slot(i, stride)] = i;
}

/*
 * Pixels dealt by cost.  A wave runs every loop to its slowest lane: the 16x4 pixels of a rectangle execute 66.9 SDF
 * evaluations per pixel on C3 where their pixels need 57.0 — lane efficiency 0.85, 0.79 in the shadow marches, whose long
 * tails are single pixels (a ray grazing a surface) among quick neighbours.  For a camera that stands still the step
 * counts of the frame before are EXACT, so the pixels can be dealt to waves ahead of time: the frame is cut into regions
 * of REGION_W x REGION_H pixels (16 waves' worth), the pixels of a region are sorted by the evaluations they needed, and
 * wave k of the region gets the k-th 64 of them — waves of like pixels, no compaction at run time, and still neighbours
 * within 64 x 16 pixels (the culling votes of a wave keep working: pixels of like cost are pixels of like fate).  The
 * oracle's step counts put the evaluations a frame executes at -11.4 % for this region size (-9.3 % for 32 x 8, -11.2 %
 * for 128 x 32; keeping runs of 2 / 4 / 8 adjacent pixels together leaves -7.7 / -4.6 / -2.2 %: the stragglers really are
 * single pixels — tests/tools/sorted_region_model.py, profiles/r4_sorted_region_model.json).  The price is memory traffic,
 * of which this path has 250x to spare: a 4-byte table entry read per pixel, and every lane storing its own pixel.
 * lane_pixels[64 * slot + lane] = column | local row << 16 | LANE_PADDING; wave slot = 16 * region + k.
 */
struct RegionShape { uint32_t w, h; };       /* multiples of 16 x 4; w * h a power of two <= 4096 (the sort's LDS) */
static RegionShape region_shape() { return { 64, 16 }; }      /* (swept in round 4: profiles/r4_region_sweep.txt) */

/* the first frame of a view: wave k of a region = its k-th 16x4 rectangle (what a launch without tables shades) */
__global__ __launch_bounds__(LPT_THREADS) void deal_rectangles_kernel(uint32_t* lane_pixels, uint32_t n_lanes, uint32_t w, uint32_t n_rows, uint32_t regions_x,
                                                                      uint32_t REGION_W, uint32_t REGION_H) {
	const uint32_t REGION_PIXELS = REGION_W * REGION_H;
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	if (i >= n_lanes) return;
	const uint32_t region = i / REGION_PIXELS, j = i % REGION_PIXELS, k = j / 64, lane = j % 64;
	uint32_t x = (region % regions_x) * REGION_W + (k % (REGION_W / 16)) * 16 + lane % 16;
	uint32_t r = (region / regions_x) * REGION_H + (k / (REGION_W / 16)) * 4 + lane / 16;
	uint32_t pad = 0;
	if (x >= w) { x = w - 1; pad = lol::LANE_PADDING; }
	if (r >= n_rows) { r = n_rows - 1; pad = lol::LANE_PADDING; }
	lane_pixels[i] = x | r << 16 | pad;
}
/* one block per region: sort its pixels by what they cost (bitonic, in LDS; the pixels beyond the frame's edge first) and
 * deal them to the region's waves in that order */
__global__ __launch_bounds__(LPT_THREADS) void deal_by_cost_kernel(const unsigned short* pixel_cost, uint32_t* lane_pixels, uint32_t w, uint32_t n_rows,
                                                                   uint32_t regions_x, uint32_t REGION_W, uint32_t REGION_H) {
	__shared__ uint32_t key[4096];                         /* cost + 1 (0 = beyond the edge) << 12 | position in the region */
	const uint32_t REGION_PIXELS = REGION_W * REGION_H;
	const uint32_t region = blockIdx.x, x0 = (region % regions_x) * REGION_W, r0 = (region / regions_x) * REGION_H;
	for (uint32_t j = threadIdx.x; j < REGION_PIXELS; j += LPT_THREADS) {
		const uint32_t x = x0 + j % REGION_W, r = r0 + j / REGION_W;
		const uint32_t c = (x < w && r < n_rows) ? (uint32_t)pixel_cost[(size_t)r * w + x] + 1u : 0u;
		key[j] = c << 12 | j;
	}
	__syncthreads();
	for (uint32_t k = 2; k <= REGION_PIXELS; k <<= 1)
		for (uint32_t d = k >> 1; d > 0; d >>= 1) {
			for (uint32_t t = threadIdx.x; t < REGION_PIXELS / 2; t += LPT_THREADS) {
				const uint32_t lo = 2 * t - (t & (d - 1)), hi = lo + d;      /* the pair (lo, lo + d) of this compare-exchange network step */
				const bool up = (lo & k) == 0;
				const uint32_t a = key[lo], b = key[hi];
				if ((a > b) == up) { key[lo] = b; key[hi] = a; }
			}
			__syncthreads();
		}
	for (uint32_t j = threadIdx.x; j < REGION_PIXELS; j += LPT_THREADS) {
		const uint32_t q = key[j] & (REGION_PIXELS - 1);
		uint32_t x = x0 + q % REGION_W, r = r0 + q / REGION_W, pad = 0;
		if (x >= w) { x = w - 1; pad = lol::LANE_PADDING; }
		if (r >= n_rows) { r = n_rows - 1; pad = lol::LANE_PADDING; }
		lane_pixels[(size_t)region * REGION_PIXELS + j] = x | r << 16 | pad;
	}
}
/* pass 1: snapshot every block's cost as a bucket number (bucket 0 = the most expensive), count the buckets */
__global__ __launch_bounds__(LPT_THREADS) void lpt_hist_kernel(const uint32_t* cost, uint32_t* keys, uint32_t* hist, uint32_t n, uint32_t stride) {
	__shared__ uint32_t h[LPT_BUCKETS];
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) h[b] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	if (i < n) {
		uint32_t k = cost[lol::tile_slot(i, stride)];          /* <= 703 (lol_kernel.h, store_pixel); clamped all the same */
		k = LPT_BUCKETS - 1 - (k < LPT_BUCKETS ? k : LPT_BUCKETS - 1);
		keys[i] = k;
		atomicAdd(&h[k], 1u);
	}
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) if (h[b]) atomicAdd(&hist[b], h[b]);
}
/* pass 2 (one block): hist[LPT_BUCKETS + b] = where bucket b starts (exclusive prefix sum) */
__global__ __launch_bounds__(LPT_BUCKETS) void lpt_scan_kernel(uint32_t* hist) {
	__shared__ uint32_t s[LPT_BUCKETS];
	const uint32_t b = threadIdx.x;
	s[b] = hist[b];
	__syncthreads();
	for (uint32_t d = 1; d < LPT_BUCKETS; d <<= 1) {
		const uint32_t v = b >= d ? s[b - d] : 0u;
		__syncthreads();
		s[b] += v;
		__syncthreads();
	}
	hist[LPT_BUCKETS + b] = s[b] - hist[b];
}
/* pass 3: every block reserves room for its members of each bucket with ONE atomic per bucket and places them in their
 * old order; what it places is the TILE the old table named for that launch position */
__global__ __launch_bounds__(LPT_THREADS) void lpt_scatter_kernel(const uint32_t* keys, const uint32_t* order_in, uint32_t* order_out,
                                                                  uint32_t* hist, uint32_t n, uint32_t stride) {
	__shared__ uint32_t h[LPT_BUCKETS], base[LPT_BUCKETS];
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) h[b] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * LPT_THREADS + threadIdx.x;
	uint32_t k = 0, rank = 0;
	if (i < n) { k = keys[i]; rank = atomicAdd(&h[k], 1u); }
	__syncthreads();
	for (uint32_t b = threadIdx.x; b < LPT_BUCKETS; b += LPT_THREADS) if (h[b]) base[b] = atomicAdd(&hist[LPT_BUCKETS + b], h[b]);
	__syncthreads();
	if (i < n) order_out[lol::tile_slot(base[k] + rank, stride)] = order_in[lol::tile_slot(i, stride)];
}

static unsigned lpt_resort_period() { return LPT_RESORT; }

static void lpt_release_set(lol_gpu::TileLpt& T) {
	for (uint32_t** p : { &T.d_order[0], &T.d_order[1], &T.d_cost, &T.d_keys, &T.d_hist, &T.d_lanes })
		if (*p) { (void)hipFree(*p); *p = nullptr; }
	if (T.d_pixel_cost) { (void)hipFree(T.d_pixel_cost); T.d_pixel_cost = nullptr; }
	T.cap = 0; T.lanes_cap = 0; T.pixels_cap = 0; T.n_tiles = 0; T.key[0] = 0;
}
/* The frame that lpt_table_for_frame has just handed tables to has been queued on `s`: an event of the set's own marks the point
 * up to which its tables are in use.  What later has to wait for a set — another stream taking it over, a handle that may belong
 * to a stream created after the set's own was destroyed — waits for THAT event (events belong to the context; round-5 advisor:
 * a caller's stream handle may be dead, and must not be handed to the runtime again). */
void lpt_frame_queued(lol_gpu* ctx, hipStream_t s) {
	if (ctx->lpt_last_set < 0) return;
	lol_gpu::TileLpt& T = ctx->lpt[ctx->lpt_last_set];
	if (!T.done && hipEventCreateWithFlags(&T.done, hipEventDisableTiming) != hipSuccess) { T.done = nullptr; (void)hipGetLastError(); return; }
	if (hipEventRecord(T.done, s) != hipSuccess) (void)hipGetLastError();
	T.done_recorded = true;
}
void lpt_release(lol_gpu* ctx) {
	for (lol_gpu::TileLpt& T : ctx->lpt) {
		lpt_release_set(T); T.home = nullptr; T.launched = false;
		if (T.done) { (void)hipEventDestroy(T.done); T.done = nullptr; }
		T.done_recorded = false;
	}
}

/* The table for the frame about to be launched on `s` (device current), or nullptr: a launch in one of the fixed orders.
 *
 * Longest-first is for a camera that stands still: then a tile costs this frame exactly what it cost the frame before.
 * Handed out by STALE costs the dear tiles come late, which is worse than any fixed order — measured on the 256-frame
 * orbit taken at 1 / 2 / 4 / 8 frames per step (1.4 / 2.8 / 5.6 / 11 degrees; the reference's arrow keys turn the camera by
 * atan(0.1) = 5.7 degrees a frame, main.c:70-112) with a sort before every frame: +3 % / -4 % / -7 % / -10 % against the
 * column order, the +3 % being what is left of +7 % after paying for the sort (profiles/r4_orbit_stride_ab.jsonl,
 * r4_lpt_verdict_ab.jsonl; a device-side verdict on how far the costs had moved was built and dropped: it needs the sort it
 * wants to avoid).  So: a frame whose camera differs from the frame before it is launched in the fixed order (the caller
 * falls back to AUTO's choice), without table, cost or sort — no overhead while the camera moves; the first frame under
 * the camera of its predecessor goes through the row-order table and reports its tiles' costs; the next one is sorted.
 *
 * Sets up (or re-creates) the tables when the scene, size, partition or kernel changed.  Everything about one set of
 * tables happens on ONE stream — the one its first frame was launched on: frames, the costs they write and the sorts that
 * read them are then ordered by the stream itself, and a table is never rewritten under a frame that still reads it.  A
 * frame of the same key on another stream is launched without a table, unless the host has moved over for good. */

bool lpt_table_for_frame(lol_gpu* ctx, const lol_frame_camera* cam, int w, int h, int max_steps, const lol_gpu_rows* R, int n_rows,
                                int block, hipStream_t s, FrameTables* out) {
	const uint32_t REGION_W = region_shape().w, REGION_H = region_shape().h, REGION_WAVES = REGION_W * REGION_H / 64;
	const uint32_t regions_x = ((uint32_t)w + REGION_W - 1) / REGION_W, regions_y = ((uint32_t)n_rows + REGION_H - 1) / REGION_H;
	const uint32_t n = regions_x * regions_y * REGION_WAVES;            /* wave slots = blocks of the launch */
	const size_t n_lanes = (size_t)n * 64, n_pixels = (size_t)w * (size_t)n_rows;
	ctx->lpt_last_set = -1;
	if (block != 64 || w > 0xFFFF || n_rows > 0x7FFF || n_lanes > 0xFFFFFFFFull) return false;      /* (an entry is column | row << 16 | flag; one-wave blocks) */
	const int key[7] = { w, h, max_steps, R->band_rows, R->cycle_rows, R->offset_rows, ctx->kernel_epoch };
	/* what the frame before this one was (on whatever stream): the same view of the same frame? */
	const bool still = ctx->lpt_have_last && memcmp(key, ctx->lpt_last_key, sizeof key) == 0 && memcmp(cam, &ctx->lpt_last_cam, sizeof *cam) == 0;
	memcpy(ctx->lpt_last_key, key, sizeof key);
	ctx->lpt_last_cam = *cam;
	ctx->lpt_have_last = true;
	if (!still) return false;
	auto ok = [](hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess; };
	/* the set that lives on this stream; else a free one; else — once two such frames in a row have found no set, i.e. the host
	 * has moved to streams without one and is not merely rotating over more streams than there are sets (then the first
	 * LPT_SETS streams keep theirs and the others run in the fixed order: taking turns at a set would cost a stream
	 * synchronisation per frame) — the least recently used one, after ITS stream has run dry */
	lol_gpu::TileLpt* Tp = nullptr;
	for (lol_gpu::TileLpt& P : ctx->lpt) if (P.home == s) Tp = &P;
	if (!Tp) for (lol_gpu::TileLpt& P : ctx->lpt) if (!P.home && !Tp) Tp = &P;
	if (!Tp) {
		if (++ctx->lpt_homeless < 2) return false;
		for (lol_gpu::TileLpt& P : ctx->lpt) if (!Tp || P.stamp < Tp->stamp) Tp = &P;
		/* (its stream may be a caller's and may be gone: the set's own event says when the last frame through it has finished) */
		if (Tp->done_recorded ? !ok(hipEventSynchronize(Tp->done)) : (Tp->launched && !ok(hipDeviceSynchronize()))) return false;
		Tp->launched = false;
		Tp->key[0] = 0;                                  /* whatever it knew was another stream's schedule */
	}
	ctx->lpt_homeless = 0;
	lol_gpu::TileLpt& T = *Tp;
	/* `s` carries the handle the set lives on — normally the very stream, where this wait costs nothing; a stream created at the
	 * address of a destroyed one (whose last frames and table kernels may still be running) is ordered behind them by it */
	/* (not for HIP's two special handles — the legacy default stream and the per-thread one: they name no stream OBJECT that could
	 * have been destroyed and created again, and this HIP's hipStreamWaitEvent dereferences the handle it is given: a crash, found
	 * by the GPU suite, whose frames run on the legacy default stream) */
	const bool special = s == hipStreamLegacy || s == hipStreamPerThread || s == nullptr;
	if (T.done_recorded && !special && !ok(hipStreamWaitEvent(s, T.done, 0))) return false;
	T.home = s;
	T.stamp = ++ctx->lpt_clock;
	const bool new_key = memcmp(key, T.key, sizeof key) != 0;
	if (new_key) {
		/* frames of the old key may still read these tables — on this very stream, so the kernels that rewrite them queue up
		 * behind those frames; only FREEING the tables needs the stream to have run dry first */
		const bool grow = n > T.cap || n_lanes > T.lanes_cap || n_pixels > T.pixels_cap;
		if (grow) {
			if (T.launched && !ok(hipStreamSynchronize(s))) return false;
			T.launched = false;
			lpt_release_set(T);
			const size_t cap = (size_t)n + n / 4 + 1024;      /* (tile_slot reaches 8 * ceil(n / 8) - 1 < n + 8) */
			const bool good = ok(hipMalloc(reinterpret_cast<void**>(&T.d_order[0]), cap * 4)) && ok(hipMalloc(reinterpret_cast<void**>(&T.d_order[1]), cap * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_cost), cap * 4)) && ok(hipMalloc(reinterpret_cast<void**>(&T.d_keys), cap * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_hist), 2 * LPT_BUCKETS * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_lanes), n_lanes * 4)) &&
			                  ok(hipMalloc(reinterpret_cast<void**>(&T.d_pixel_cost), n_pixels * 2));
			if (!good) { lpt_release_set(T); T.home = nullptr; return false; }
			T.cap = cap; T.lanes_cap = n_lanes; T.pixels_cap = n_pixels;
		}
		memcpy(T.key, key, sizeof key);
		T.n_tiles = n;
	}
	const dim3 grid((n + LPT_THREADS - 1) / LPT_THREADS);
	const uint32_t stride = (n + 7u) >> 3;
	bool record_pixels = false;
	T.launched = true;                                   /* (from here on something of this set is queued on s) */
	if (new_key || memcmp(cam, &T.cam_epoch, sizeof *cam) != 0) {
		/* a view these tables know nothing about: rectangles, handed out in region order; this frame reports what every
		 * pixel and every wave cost */
		T.cam_epoch = *cam;
		T.cur = 0; T.frames = 0;
		hipLaunchKernelGGL(lpt_identity_kernel, grid, dim3(LPT_THREADS), 0, s, T.d_order[0], n, stride);
		hipLaunchKernelGGL(deal_rectangles_kernel, dim3((unsigned)((n_lanes + LPT_THREADS - 1) / LPT_THREADS)), dim3(LPT_THREADS), 0, s,
		                   T.d_lanes, (uint32_t)n_lanes, (uint32_t)w, (uint32_t)n_rows, regions_x, REGION_W, REGION_H);
		if (!ok(hipGetLastError()) || !ok(hipMemsetAsync(T.d_cost, 0, ((size_t)n + 8) * 4, s))) { T.key[0] = 0; return false; }
		record_pixels = true;
	} else if (T.frames == 1) {
		/* the second frame of the view: its pixels dealt to the waves of their region by what they cost (exact: nothing moved);
		 * the waves are new ones, so they go out in region order once more and report how long THEY run */
		hipLaunchKernelGGL(deal_by_cost_kernel, dim3(regions_x * regions_y), dim3(LPT_THREADS), 0, s, T.d_pixel_cost, T.d_lanes, (uint32_t)w, (uint32_t)n_rows, regions_x, REGION_W, REGION_H);
		if (!ok(hipGetLastError())) { T.key[0] = 0; return false; }
	} else if (T.frames == 2 || T.frames % lpt_resort_period() == 0) {
		/* the run times of the frame before are in (same stream): three small kernels, then the other table is the current
		 * one.  (Again every LPT_RESORT frames: the run times drift a little with what runs beside a wave.) */
		if (ok(hipMemsetAsync(T.d_hist, 0, 2 * LPT_BUCKETS * 4, s))) {
			hipLaunchKernelGGL(lpt_hist_kernel, grid, dim3(LPT_THREADS), 0, s, T.d_cost, T.d_keys, T.d_hist, n, stride);
			hipLaunchKernelGGL(lpt_scan_kernel, dim3(1), dim3(LPT_BUCKETS), 0, s, T.d_hist);
			hipLaunchKernelGGL(lpt_scatter_kernel, grid, dim3(LPT_THREADS), 0, s, T.d_keys, T.d_order[T.cur], T.d_order[T.cur ^ 1], T.d_hist, n, stride);
			if (ok(hipGetLastError())) { T.cur ^= 1; T.sorts++; ctx->lpt_sorts++; }
		}
	}
	T.frames++;
	ctx->lpt_last_set = (int)(Tp - ctx->lpt);
	*out = { T.d_order[T.cur], T.d_cost, T.d_lanes, record_pixels ? T.d_pixel_cost : nullptr, n };
	return true;
}

/* AUTO: collect the trial frames that have finished (never waits) and decide once all of them have */
static void tile_auto_harvest(lol_gpu* ctx) {
	lol_gpu::TileAuto& T = ctx->tiles;
	if (!T.deciding) return;
	while (T.harvested < T.issued) {
		const int i = T.harvested;
		if (hipEventQuery(T.ev[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); return; }      /* (hipErrorNotReady is not an error) */
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, T.ev[2 * i], T.ev[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
		T.ms[i] = ms;
		T.harvested++;
	}
	if (T.harvested < lol_gpu::TileAuto::TOTAL) return;
	/* the typical frame of each order (reported): the mean of the faster half of its trials (a frame that shared the device
	 * with something else, or ran before the clocks had settled, does not count) */
	for (int o = 0; o < 2; o++) {
		float v[LOL_GPU_TILE_TRIALS];
		int n = 0;
		for (int i = lol_gpu::TileAuto::SKIP; i < lol_gpu::TileAuto::TOTAL; i++) if (lol_gpu::TileAuto::order_of_trial(i) == o && T.ms[i] > 0.f) v[n++] = T.ms[i];
		std::sort(v, v + n);
		const int half = n > 1 ? n / 2 : n;
		float sum = 0.f;
		for (int i = 0; i < half; i++) sum += v[i];
		T.typical[o] = half ? sum / (float)half : 0.f;
	}
	/* The decision: the trials come in PAIRS of consecutive frames, one of each order — a host whose camera moves (the orbit: a
	 * frame costs 0.7 to 1.0 ms depending on where the camera is) renders nearly the same view twice in a pair, so the ratio
	 * columns / rows of a pair is about the orders and not about the view; which order goes first alternates from pair to
	 * pair, so a cost that drifts one way cancels; the median ratio decides (round 4: the means of two interleaved series
	 * picked rows for the orbit, 4 % behind). */
	float ratio[LOL_GPU_TILE_TRIALS];
	int n_ratios = 0;
	for (int i = lol_gpu::TileAuto::SKIP; i + 1 < lol_gpu::TileAuto::TOTAL; i += 2) {
		const float a = T.ms[i], b = T.ms[i + 1];
		if (!(a > 0.f && b > 0.f)) continue;
		ratio[n_ratios++] = lol_gpu::TileAuto::order_of_trial(i) == LOL_GPU_TILES_ROWS ? b / a : a / b;      /* columns / rows */
	}
	std::sort(ratio, ratio + n_ratios);
	const float median = n_ratios ? (n_ratios & 1 ? ratio[n_ratios / 2] : 0.5f * (ratio[n_ratios / 2 - 1] + ratio[n_ratios / 2])) : 1.f;
	T.chosen = median < 0.99f ? LOL_GPU_TILES_COLS : LOL_GPU_TILES_ROWS;
	T.deciding = false;
	T.decisions++;
}

/* the order of the frame about to be launched; *trial = the trial slot whose events bracket it, or -1.  Device is current. */
int tile_order_for_frame(lol_gpu* ctx, int w, int h, int max_steps, const lol_gpu_rows* R, bool diagnostics, int* trial) {
	lol_gpu::TileAuto& T = ctx->tiles;
	*trial = -1;
	if (T.mode != LOL_GPU_TILES_AUTO && T.mode != LOL_GPU_TILES_LPT) return T.chosen;
	const int key[6] = { w, h, max_steps, R->band_rows, R->cycle_rows, ctx->kernel_epoch };      /* (the kernel too: interpreter, or which form of the scene's own) */
	if (memcmp(key, T.key, sizeof key) != 0) {          /* another scene, size or partition: measure again */
		memcpy(T.key, key, sizeof key);
		if (!T.have_events) {
			bool ok = true;
			for (hipEvent_t& e : T.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
			if (!ok) { (void)hipGetLastError(); for (hipEvent_t& e : T.ev) { if (e) (void)hipEventDestroy(e); e = nullptr; } T.mode = LOL_GPU_TILES_ROWS; return T.chosen; }
			T.have_events = true;
		}
		T.deciding = true;
		T.issued = T.harvested = 0;
		T.chosen = LOL_GPU_TILES_ROWS;
		T.mon_frames = T.mon_n = 0;
		T.mon_pending = false;
	}
	tile_auto_harvest(ctx);
	if (T.deciding) {
		if (T.issued >= lol_gpu::TileAuto::TOTAL || diagnostics) return T.chosen;
		*trial = T.issued++;
		return lol_gpu::TileAuto::order_of_trial(*trial);
	}
	/* Decided — and watched from then on: which fixed order is better depends on the VIEW as well (the orbit's first forty
	 * frames favour rows by 5 %, the orbit as a whole columns by 3.5 %), and the host moves the camera (main.c:180).  Every
	 * MONITOR_PERIOD frames one frame in the order in use and the next one in the other order are timed like trial frames
	 * (two event pairs, collected without waiting); when the other order has been faster by more than 1 % in the median of
	 * the last MONITOR_WINDOW such pairs, the orders change places.  A probe frame costs what the orders differ by. */
	if (diagnostics) return T.chosen;
	if (T.mon_pending) {
		if (hipEventQuery(T.ev[3]) == hipSuccess) {
			float a = 0.f, b = 0.f;
			if (hipEventElapsedTime(&a, T.ev[0], T.ev[1]) == hipSuccess && hipEventElapsedTime(&b, T.ev[2], T.ev[3]) == hipSuccess && a > 0.f && b > 0.f) {
				T.mon_ratio[T.mon_n % lol_gpu::TileAuto::MONITOR_WINDOW] = b / a;      /* other / in use */
				T.mon_n++;
				if (T.mon_n >= lol_gpu::TileAuto::MONITOR_WINDOW) {
					float r[lol_gpu::TileAuto::MONITOR_WINDOW];
					memcpy(r, T.mon_ratio, sizeof r);
					std::sort(r, r + lol_gpu::TileAuto::MONITOR_WINDOW);
					if (r[lol_gpu::TileAuto::MONITOR_WINDOW / 2] < 0.99f) {
						T.chosen = T.chosen == LOL_GPU_TILES_COLS ? LOL_GPU_TILES_ROWS : LOL_GPU_TILES_COLS;
						T.mon_n = 0;
						T.swaps++;
					}
				}
			} else (void)hipGetLastError();
			T.mon_pending = false;
		} else { (void)hipGetLastError(); return T.chosen; }      /* (the pair is still in flight: no new one) */
	}
	const unsigned phase = T.mon_frames++ % lol_gpu::TileAuto::MONITOR_PERIOD;
	if (phase == lol_gpu::TileAuto::MONITOR_PERIOD - 2) { *trial = 0; return T.chosen; }
	if (phase == lol_gpu::TileAuto::MONITOR_PERIOD - 1) {
		*trial = 1;
		T.mon_pending = true;
		return T.chosen == LOL_GPU_TILES_COLS ? LOL_GPU_TILES_ROWS : LOL_GPU_TILES_COLS;
	}
	return T.chosen;
}

#pragma GCC visibility pop
extern "C" int lol_gpu_tile_order(lol_gpu* ctx, lol_gpu_tile_order_info* out) {
	if (!ctx || !out) return LOL_GPU_ERR_ARG;
	LOL_HIP(ctx, hipSetDevice(ctx->device));
	tile_auto_harvest(ctx);
	const lol_gpu::TileAuto& T = ctx->tiles;
	if (T.mode == LOL_GPU_TILES_LPT) {
		/* longest first: the last frame went through a table (order LPT; "deciding" until its costs have been sorted once), or
		 * the camera moves and AUTO's fixed order is in use (its state and trial times) */
		if (ctx->lpt_last_set >= 0) *out = { T.mode, LOL_GPU_TILES_LPT, ctx->lpt[ctx->lpt_last_set].frames < 3 ? 1 : 0, (int32_t)ctx->lpt_sorts, T.typical[0], T.typical[1] };
		else *out = { T.mode, T.chosen, (T.deciding || ctx->lpt_sorts == 0) ? 1 : 0, (int32_t)ctx->lpt_sorts, T.typical[0], T.typical[1] };
	} else
		*out = { T.mode, T.chosen, T.deciding ? 1 : 0, T.decisions, T.typical[0], T.typical[1] };
	return LOL_GPU_OK;
}
