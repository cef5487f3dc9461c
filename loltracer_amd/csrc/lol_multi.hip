/*
 * lol_multi.hip — several devices behind the renderer.h boundary (include/lol_gpu.h, "Several devices").
 *
 * The reference spreads a frame over workers by letting them claim rows from an atomic counter
 * (naive_renderer.c:216) and joins them with the exit semaphore (main.c:189-194).  Across GPUs the same
 * independence of rows is used statically: bands of rows dealt round-robin over the devices, each device
 * renders its part compactly, and ONE grouped RCCL exchange (ncclSend on every device, ncclRecv on the
 * root) brings the parts to the root, which un-interleaves them — the "row-tile partition + RCCL gather
 * over xGMI" BASELINE.json names.  Single process, ncclCommInitAll, no host staging.
 *
 * Streams per device d:  render[d]  — the frame kernels (lol_gpu_render_device)
 *                        xchg[d]    — ncclSend / ncclRecv, the un-interleave and the D2H copy on the root
 * and two part buffers, so that while frame i's parts travel, frame i+1 renders:
 *   render[d]: wait sent[slot] (frame i-2 has left the buffer) → kernel → record rendered[slot]
 *   xchg[d]:   wait rendered[slot] → ncclSend(part[slot] → root) → record sent[slot]
 *   xchg[0]:   … → ncclRecv(staging[slot] ← every d) → assemble_kernel(staging[slot] → dst) → record done[slot]
 * RCCL itself is resolved with dlopen, and the communicators are created, on the first frame that needs the
 * exchange: librccl is a 570 MB library that a host which never assembles on a device should not have to map.
 *
 * A HOST surface (what render_thread fills, naive_renderer.c:233-235) needs no exchange at all: every device copies its
 * own bands into it over its own PCIe link (one strided hipMemcpy3DAsync per part, issued by a host thread per device so
 * that the links run at once) instead of funnelling 4 B per pixel through the root's single link (132 MB = 2.4 ms for a
 * C4 frame).  The surface is the host's memory and is never registered with the devices (lol_gpu.hip, lol_gpu_render_host).
 *
 * Parts.  The frame's rows are cut into cycles; inside a cycle every part owns one band (lol_gpu_rows: band_rows rows at
 * offset_rows of every cycle_rows), and a part is ONE launch — measured on MI355X, a device that rendered its share as
 * eight small launches took 2.7 times as long as with one (every launch ends in its own tail of half-empty SIMDs).  The
 * parts' bands need not be equally tall: the root also receives and un-interleaves the whole frame, so with an equal
 * share it finishes last; lol_gpu_multi_set_root_band_rows gives its bands fewer rows — the cost-weighted split.
 * parts_per_device > 1 (part p belongs to device p % n) exists for finer interleaving experiments and so that a
 * single-GPU box exercises every multi-part code path — tests/test_multi_device.py.
 */
#include "lol_gpu.h"
#include "lol_gpu_testing.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>


namespace {

struct Rccl {
	void* handle = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char*  (*GetErrorString)(ncclResult_t) = nullptr;
	bool load(char* err, size_t cap) {
		for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
			handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (handle) break;
		}
		if (!handle) { snprintf(err, cap, "cannot load RCCL: %s", dlerror()); return false; }
		auto sym = [&](const char* n) { return dlsym(handle, n); };
		CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
		CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
		GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
		GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
		Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
		Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
		GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
		if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
			snprintf(err, cap, "RCCL library lacks a required entry point");
			return false;
		}
		return true;
	}
};

constexpr int SLOTS = 2;

/* One host thread per device for the copies into a HOST surface: a device-to-host copy into pageable memory (memory
 * the host owns and the library must not keep pinned, include/lol_gpu.h) occupies the thread that issues it, so N
 * devices only drive their N PCIe links at once when N threads issue.  Started by the first frame that needs it. */
struct Worker {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::function<hipError_t()> job;
	bool busy = false, quit = false;
	hipError_t result = hipSuccess;
	/* false: the thread could not be started (std::thread throws std::system_error; nothing may cross the C ABI) */
	bool start() {
		if (th.joinable()) return true;
		try {
		th = std::thread([this] {
			std::unique_lock<std::mutex> lock(mu);
			for (;;) {
				cv.wait(lock, [this] { return quit || (busy && job); });
				if (quit) return;
				std::function<hipError_t()> j = std::move(job);
				job = nullptr;
				lock.unlock();
				const hipError_t r = j();
				lock.lock();
				result = r;
				busy = false;
				cv.notify_all();
			}
		});
		} catch (...) { return false; }
		return true;
	}
	bool submit(std::function<hipError_t()> j) {
		if (!start()) return false;
		std::lock_guard<std::mutex> lock(mu);
		job = std::move(j);
		busy = true;
		cv.notify_all();
		return true;
	}
	hipError_t wait() {
		std::unique_lock<std::mutex> lock(mu);
		cv.wait(lock, [this] { return !busy; });
		return result;
	}
	void stop() {
		if (!th.joinable()) return;
		{ std::lock_guard<std::mutex> lock(mu); quit = true; cv.notify_all(); }
		th.join();
	}
};

struct Device {
	int          id = -1;
	lol_gpu*     ctx = nullptr;
	hipStream_t  render = nullptr, xchg = nullptr;
	hipStream_t  render2 = nullptr;      /* the kernels of the frames in slot 1 (round 5): consecutive frames' kernels overlap — a device's
	                                      * launch of its bands is a small launch whose ramp and tail the next frame's fills
	                                      * (rank 0 of an 8-way C4 frame emulated on one GPU: 0.462 -> 0.442 ms per frame, LABNOTES.md §4) */
	ncclComm_t   comm = nullptr;
	uint32_t*    part[SLOTS] = { nullptr, nullptr };
	size_t       part_bytes = 0;
	hipEvent_t   rendered[SLOTS] = { nullptr, nullptr }, sent[SLOTS] = { nullptr, nullptr };
	Worker       copier;
};

constexpr int MAX_PARTS = 64;      /* devices x parts_per_device */

/* The parts of one frame, by value in the kernel arguments: where each part's band sits inside a cycle, and the row of
 * the source buffer (w pixels per row) where the part's compact copy starts. */
struct PartTable {
	uint32_t row0[MAX_PARTS];
	uint16_t band[MAX_PARTS], offset[MAX_PARTS];
	int32_t  n_parts, cycle;
};

/* dst row y ← the row of the part that rendered it.  A block copies ROWS_PER_BLOCK consecutive frame rows; a thread moves
 * 4 pixels (uint4) per step when VEC, else one, striding over the row.  (Round 3: one row and one uint4 per thread was
 * 0.049 ms for a C4 frame — 34,560 blocks that each looked their part up; several rows per block and a strided loop …) */
constexpr int ASM_ROWS = 4, ASM_THREADS = 256;
template <bool VEC>
__global__ __launch_bounds__(ASM_THREADS) void assemble_kernel(const uint32_t* __restrict__ parts, PartTable tab, int w, int h,
                                                                uint32_t* __restrict__ dst, uint32_t pitch_px) {
	const int per_row = VEC ? w / 4 : w;
	for (int k = 0; k < ASM_ROWS; k++) {
		const int y = blockIdx.x * ASM_ROWS + k;
		if (y >= h) return;
		const int c = y / tab.cycle, o = y - c * tab.cycle;
		int part = 0;
		for (int p = 1; p < tab.n_parts; p++) part = o >= (int)tab.offset[p] ? p : part;      /* offsets ascend: the last one not above o */
		const int local = c * (int)tab.band[part] + (o - (int)tab.offset[part]);
		const uint32_t* src = parts + ((size_t)tab.row0[part] + local) * w;
		uint32_t* out = dst + (size_t)y * pitch_px;
		for (int i = threadIdx.x; i < per_row; i += ASM_THREADS) {
			if (VEC) reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(src)[i];
			else out[i] = src[i];
		}
	}
}

hipError_t launch_assemble(const void* parts, const PartTable& tab, int w, int h, void* dst, size_t pitch_bytes, hipStream_t s) {
	const bool vec = w % 4 == 0 && pitch_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(parts) % 16) == 0 &&
	                 (reinterpret_cast<uintptr_t>(dst) % 16) == 0;
	dim3 grid((h + ASM_ROWS - 1) / ASM_ROWS);
	if (vec) hipLaunchKernelGGL(assemble_kernel<true>, grid, dim3(ASM_THREADS), 0, s, static_cast<const uint32_t*>(parts), tab, w, h,
	                            static_cast<uint32_t*>(dst), (uint32_t)(pitch_bytes / 4));
	else     hipLaunchKernelGGL(assemble_kernel<false>, grid, dim3(ASM_THREADS), 0, s, static_cast<const uint32_t*>(parts), tab, w, h,
	                            static_cast<uint32_t*>(dst), (uint32_t)(pitch_bytes / 4));
	return hipGetLastError();
}

/* A valid split: the parts' bands tile one cycle exactly, in order.  Fills band / offset / cycle of the table. */
bool table_from_rows(PartTable& tab, const lol_gpu_rows* rows, int n_parts) {
	if (!rows || n_parts < 1 || n_parts > MAX_PARTS) return false;
	int at = 0;
	for (int p = 0; p < n_parts; p++) {
		if (rows[p].band_rows < 1 || rows[p].band_rows > 0xFFFF || rows[p].offset_rows != at || rows[p].cycle_rows != rows[0].cycle_rows) return false;
		tab.band[p] = (uint16_t)rows[p].band_rows;
		tab.offset[p] = (uint16_t)at;
		at += rows[p].band_rows;
	}
	if (at != rows[0].cycle_rows || at > 0xFFFF) return false;
	tab.n_parts = n_parts;
	tab.cycle = at;
	return true;
}

}  // namespace

struct lol_gpu_multi {
	int       n = 0;
	Device    dev[LOL_GPU_MULTI_MAX_DEVICES];
	Rccl      rccl;
	bool      comms_up = false;
	int       band_override = 0;                       /* band height of every part (0 = chosen per frame height) */
	int       root_band = 0;                           /* band height of the root's parts (0 = like the others) */
	int       per_dev = 1;                             /* parts per device: part p belongs to device p % n */
	int       host_via_root = 0;                       /* render_host: assemble on the root first (RCCL) instead of direct copies */
	int       test_root_stride = 0;                    /* lol_gpu_multi_testing_root_stride */
	int       test_force_threads = 0;                  /* lol_gpu_multi_testing_force_copier_threads */
	uint32_t* staging[SLOTS] = { nullptr, nullptr };   /* root: every part of a frame, device by device */
	size_t    staging_bytes = 0;
	hipEvent_t done[SLOTS] = { nullptr, nullptr };     /* root: frame of this slot assembled (staging free again) */
	uint32_t* d_frame = nullptr;                       /* root: framebuffer of the host-surface path when host_via_root */
	size_t    frame_bytes = 0;
	unsigned long frames = 0;
	char      err[512] = { 0 };
};

namespace {

int mfail(lol_gpu_multi* m, int status, const char* what, const char* detail = nullptr) {
	if (m) {
		if (detail) snprintf(m->err, sizeof m->err, "%s: %s", what, detail);
		else snprintf(m->err, sizeof m->err, "%s", what);
	}
	return status;
}

#define M_HIP(m, call)                                                                             \
	do {                                                                                           \
		hipError_t e_ = (call);                                                                    \
		if (e_ != hipSuccess) return mfail((m), LOL_GPU_ERR_HIP, #call, hipGetErrorString(e_));    \
	} while (0)
#define M_NCCL(m, call)                                                                            \
	do {                                                                                           \
		ncclResult_t r_ = (call);                                                                  \
		if (r_ != ncclSuccess) return mfail((m), LOL_GPU_ERR_HIP, #call, (m)->rccl.GetErrorString(r_)); \
	} while (0)

/* (re)size the per-device part buffers and the root's staging for frames of w x h */
int ensure_buffers(lol_gpu_multi* m, int w, int h, const uint32_t* dev_rows, bool want_staging) {
	size_t need_staging = want_staging ? (size_t)w * h * 4 : 0;
	for (int d = 0; d < m->n; d++) {
		Device& D = m->dev[d];
		size_t need = (size_t)dev_rows[d] * w * 4;
		if (need > D.part_bytes) {
			M_HIP(m, hipSetDevice(D.id));
			M_HIP(m, hipDeviceSynchronize());
			D.part_bytes = 0;                       /* a failed hipMalloc below must not leave a stale size behind */
			for (int s = 0; s < SLOTS; s++) {
				if (D.part[s]) (void)hipFree(D.part[s]);
				D.part[s] = nullptr;
			}
			for (int s = 0; s < SLOTS; s++) M_HIP(m, hipMalloc(reinterpret_cast<void**>(&D.part[s]), need));
			D.part_bytes = need;
		}
	}
	if (need_staging > m->staging_bytes) {
		M_HIP(m, hipSetDevice(m->dev[0].id));
		M_HIP(m, hipDeviceSynchronize());
		m->staging_bytes = 0;
		for (int s = 0; s < SLOTS; s++) {
			if (m->staging[s]) (void)hipFree(m->staging[s]);
			m->staging[s] = nullptr;
		}
		for (int s = 0; s < SLOTS; s++) M_HIP(m, hipMalloc(reinterpret_cast<void**>(&m->staging[s]), need_staging));
		m->staging_bytes = need_staging;
	}
	return LOL_GPU_OK;
}

}  // namespace

extern "C" {

int lol_gpu_choose_band_rows(int h, int n_parts) {
	if (h <= 0 || n_parts <= 0) return 0;
	if (n_parts == 1) return h;
	/* equal parts first (multiples of the kernel's 4-row wave patch, so no wave straddles two bands) ... */
	for (int band = 16; band >= 4; band -= 4)
		if (h % (band * n_parts) == 0) return band;
	/* ... else the tallest band that still deals every part at least eight bands (parts then differ by one band) */
	for (int band = 16; band >= 4; band -= 4)
		if (h / band >= 8 * n_parts) return band;
	return 4;
}

int lol_gpu_split_rows(int n_parts, int band_rows, int root_band_rows, int root_stride, lol_gpu_rows* out) {
	if (n_parts < 1 || n_parts > MAX_PARTS || band_rows < 1 || root_band_rows < 0 || root_stride < 1 || !out) return LOL_GPU_ERR_ARG;
	if (n_parts == 1) { out[0] = { band_rows, band_rows, 0 }; return LOL_GPU_OK; }
	int at = 0;
	for (int p = 0; p < n_parts; p++) {
		const int b = (root_band_rows > 0 && p % root_stride == 0) ? root_band_rows : band_rows;
		out[p] = { b, 0, at };
		at += b;
	}
	if (at > 0xFFFF) return LOL_GPU_ERR_ARG;
	for (int p = 0; p < n_parts; p++) out[p].cycle_rows = at;
	return LOL_GPU_OK;
}

int lol_gpu_part_frame_row(int h, const lol_gpu_rows* rows, int local_row) {
	if (!rows) return local_row >= 0 && local_row < h ? local_row : -1;
	if (local_row < 0 || local_row >= lol_gpu_part_rows(h, rows)) return -1;
	const int band = local_row / rows->band_rows;
	return band * rows->cycle_rows + rows->offset_rows + (local_row - band * rows->band_rows);
}

int lol_gpu_assemble_parts_at(lol_gpu* ctx, const void* parts, const lol_gpu_rows* part_rows, const uint32_t* part_row0,
                              int n_parts, int w, int h, void* dst, size_t pitch_bytes, void* stream) {
	if (!ctx || !parts || !dst || !part_row0) return LOL_GPU_ERR_ARG;
	if (w < 1 || h < 1 || pitch_bytes % 4 || pitch_bytes < (size_t)w * 4) return LOL_GPU_ERR_ARG;
	PartTable tab;
	if (!table_from_rows(tab, part_rows, n_parts)) return LOL_GPU_ERR_ARG;
	for (int p = 0; p < n_parts; p++) tab.row0[p] = part_row0[p];
	if (hipSetDevice(lol_gpu_device(ctx)) != hipSuccess) return LOL_GPU_ERR_HIP;
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (!s) {
		/* the context's private stream is not reachable from here: drain it, then use the legacy default stream */
		int st = lol_gpu_sync(ctx);
		if (st != LOL_GPU_OK) return st;
		s = hipStreamLegacy;
	}
	return launch_assemble(parts, tab, w, h, dst, pitch_bytes, s) == hipSuccess ? LOL_GPU_OK : LOL_GPU_ERR_HIP;
}

int lol_gpu_assemble_parts(lol_gpu* ctx, const void* parts, int n_parts, int band_rows, int w, int h,
                           void* dst, size_t pitch_bytes, void* stream) {
	if (n_parts < 1 || n_parts > MAX_PARTS || band_rows < 1 || h < 1) return LOL_GPU_ERR_ARG;
	lol_gpu_rows rows[MAX_PARTS];
	uint32_t row0[MAX_PARTS];
	if (lol_gpu_split_rows(n_parts, band_rows, 0, 1, rows) != LOL_GPU_OK) return LOL_GPU_ERR_ARG;
	uint32_t row = 0;
	for (int p = 0; p < n_parts; p++) {               /* parts back to back, in part order */
		row0[p] = row;
		row += (uint32_t)lol_gpu_part_rows(h, &rows[p]);
	}
	return lol_gpu_assemble_parts_at(ctx, parts, rows, row0, n_parts, w, h, dst, pitch_bytes, stream);
}

int lol_gpu_multi_create(const int* devices, int n, lol_gpu_multi** out) {
	if (!out) return LOL_GPU_ERR_ARG;
	*out = nullptr;
	if (!devices || n < 1 || n > LOL_GPU_MULTI_MAX_DEVICES) return LOL_GPU_ERR_ARG;
	const int have = lol_gpu_device_count();
	if (have <= 0) return LOL_GPU_ERR_NO_DEVICE;
	for (int i = 0; i < n; i++) {
		if (devices[i] < 0 || devices[i] >= have) return LOL_GPU_ERR_NO_DEVICE;
		for (int j = 0; j < i; j++)
			if (devices[j] == devices[i]) return LOL_GPU_ERR_ARG;      /* RCCL needs distinct devices */
	}
	lol_gpu_multi* m = new (std::nothrow) lol_gpu_multi;
	if (!m) return LOL_GPU_ERR_HIP;
	m->n = n;
	auto bail = [&](const char* what, const char* detail) {
		fprintf(stderr, "lol_gpu_multi_create: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
		lol_gpu_multi_destroy(m);
		return LOL_GPU_ERR_HIP;
	};
	for (int i = 0; i < n; i++) {
		Device& D = m->dev[i];
		D.id = devices[i];
		if (lol_gpu_create(D.id, &D.ctx) != LOL_GPU_OK) return bail("lol_gpu_create failed", nullptr);
		hipError_t e = hipSetDevice(D.id);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&D.render, hipStreamNonBlocking);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&D.render2, hipStreamNonBlocking);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&D.xchg, hipStreamNonBlocking);
		for (int s = 0; s < SLOTS && e == hipSuccess; s++) {
			e = hipEventCreateWithFlags(&D.rendered[s], hipEventDisableTiming);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&D.sent[s], hipEventDisableTiming);
		}
		if (e != hipSuccess) return bail("stream/event setup", hipGetErrorString(e));
	}
	{
		hipError_t e = hipSetDevice(m->dev[0].id);
		for (int s = 0; s < SLOTS && e == hipSuccess; s++) e = hipEventCreateWithFlags(&m->done[s], hipEventDisableTiming);
		if (e != hipSuccess) return bail("event setup", hipGetErrorString(e));
	}
	*out = m;
	return LOL_GPU_OK;
}

void lol_gpu_multi_destroy(lol_gpu_multi* m) {
	if (!m) return;
	for (int i = 0; i < m->n; i++) m->dev[i].copier.stop();
	for (int i = 0; i < m->n; i++) {
		Device& D = m->dev[i];
		if (D.id < 0) continue;
		(void)hipSetDevice(D.id);
		if (D.render) (void)hipStreamSynchronize(D.render);
		if (D.render2) (void)hipStreamSynchronize(D.render2);
		if (D.xchg) (void)hipStreamSynchronize(D.xchg);
	}
	if (m->comms_up)
		for (int i = 0; i < m->n; i++)
			if (m->dev[i].comm) (void)m->rccl.CommDestroy(m->dev[i].comm);
	for (int i = 0; i < m->n; i++) {
		Device& D = m->dev[i];
		if (D.id < 0) continue;
		(void)hipSetDevice(D.id);
		for (int s = 0; s < SLOTS; s++) {
			if (D.part[s]) (void)hipFree(D.part[s]);
			if (D.rendered[s]) (void)hipEventDestroy(D.rendered[s]);
			if (D.sent[s]) (void)hipEventDestroy(D.sent[s]);
		}
		if (D.render) (void)hipStreamDestroy(D.render);
		if (D.render2) (void)hipStreamDestroy(D.render2);
		if (D.xchg) (void)hipStreamDestroy(D.xchg);
		if (i == 0) {
			for (int s = 0; s < SLOTS; s++) {
				if (m->staging[s]) (void)hipFree(m->staging[s]);
				if (m->done[s]) (void)hipEventDestroy(m->done[s]);
			}
			if (m->d_frame) (void)hipFree(m->d_frame);
		}
		lol_gpu_destroy(D.ctx);
	}
	/* the RCCL library stays mapped: unloading it under live HIP state is not worth the risk */
	delete m;
}

const char* lol_gpu_multi_error(const lol_gpu_multi* m) { return m ? m->err : "null context"; }
int lol_gpu_multi_device_count(const lol_gpu_multi* m) { return m ? m->n : 0; }
lol_gpu* lol_gpu_multi_context(lol_gpu_multi* m, int i) { return m && i >= 0 && i < m->n ? m->dev[i].ctx : nullptr; }

int lol_gpu_multi_set_band_rows(lol_gpu_multi* m, int band_rows) {
	if (!m || band_rows < 0 || band_rows > 4096) return LOL_GPU_ERR_ARG;
	m->band_override = band_rows;
	return LOL_GPU_OK;
}

int lol_gpu_multi_set_root_band_rows(lol_gpu_multi* m, int root_band_rows) {
	if (!m || root_band_rows < 0 || root_band_rows > 4096) return LOL_GPU_ERR_ARG;
	int st = lol_gpu_multi_sync(m);
	if (st != LOL_GPU_OK) return st;
	m->root_band = root_band_rows;
	return LOL_GPU_OK;
}

int lol_gpu_multi_upload_program(lol_gpu_multi* m, const lol_program* prog) {
	if (!m || !prog) return LOL_GPU_ERR_ARG;
	int st = lol_gpu_multi_sync(m);        /* frames in flight still read the old tables / code object */
	if (st != LOL_GPU_OK) return st;
	for (int i = 0; i < m->n; i++) {
		st = lol_gpu_upload_program(m->dev[i].ctx, prog);
		if (st != LOL_GPU_OK) return mfail(m, st, "lol_gpu_upload_program", lol_gpu_error(m->dev[i].ctx));
	}
	return LOL_GPU_OK;
}

int lol_gpu_multi_specialize_wait(lol_gpu_multi* m) {
	if (!m) return LOL_GPU_ERR_ARG;
	for (int i = 0; i < m->n; i++) {
		int st = lol_gpu_specialize_wait(m->dev[i].ctx);
		if (st != LOL_GPU_OK) return mfail(m, st, "lol_gpu_specialize_wait", lol_gpu_error(m->dev[i].ctx));
	}
	return LOL_GPU_OK;
}

int lol_gpu_multi_set_parts_per_device(lol_gpu_multi* m, int parts) {
	if (!m || parts < 1 || parts * m->n > MAX_PARTS) return LOL_GPU_ERR_ARG;
	int st = lol_gpu_multi_sync(m);
	if (st != LOL_GPU_OK) return st;
	m->per_dev = parts;
	return LOL_GPU_OK;
}

int lol_gpu_multi_testing_root_stride(lol_gpu_multi* m, int stride) {
	if (!m || stride < 0) return LOL_GPU_ERR_ARG;
	int st = lol_gpu_multi_sync(m);
	if (st != LOL_GPU_OK) return st;
	m->test_root_stride = stride;
	return LOL_GPU_OK;
}

int lol_gpu_multi_testing_force_copier_threads(lol_gpu_multi* m, int enable) {
	if (!m) return LOL_GPU_ERR_ARG;
	m->test_force_threads = enable ? 1 : 0;
	return LOL_GPU_OK;
}

int lol_gpu_multi_set_host_via_root(lol_gpu_multi* m, int enable) {
	if (!m) return LOL_GPU_ERR_ARG;
	m->host_via_root = enable ? 1 : 0;
	return LOL_GPU_OK;
}

int lol_gpu_multi_set_pixel_format(lol_gpu_multi* m, const lol_gpu_pixel_format* fmt) {
	if (!m) return LOL_GPU_ERR_ARG;
	for (int i = 0; i < m->n; i++) {
		int st = lol_gpu_set_pixel_format(m->dev[i].ctx, fmt);
		if (st != LOL_GPU_OK) return mfail(m, st, "lol_gpu_set_pixel_format", lol_gpu_error(m->dev[i].ctx));
	}
	return LOL_GPU_OK;
}

int lol_gpu_multi_set_tile_order(lol_gpu_multi* m, int columns) {
	if (!m) return LOL_GPU_ERR_ARG;
	for (int i = 0; i < m->n; i++) {
		int st = lol_gpu_set_tile_order(m->dev[i].ctx, columns);
		if (st != LOL_GPU_OK) return mfail(m, st, "lol_gpu_set_tile_order", lol_gpu_error(m->dev[i].ctx));
	}
	return LOL_GPU_OK;
}

/* RCCL on first use: load the library, one communicator per device */
static int ensure_comms(lol_gpu_multi* m) {
	if (m->comms_up) return LOL_GPU_OK;
	if (!m->rccl.handle && !m->rccl.load(m->err, sizeof m->err)) return LOL_GPU_ERR_HIP;
	ncclComm_t comms[LOL_GPU_MULTI_MAX_DEVICES];
	int ids[LOL_GPU_MULTI_MAX_DEVICES];
	for (int i = 0; i < m->n; i++) ids[i] = m->dev[i].id;
	M_NCCL(m, m->rccl.CommInitAll(comms, m->n, ids));
	for (int i = 0; i < m->n; i++) m->dev[i].comm = comms[i];
	m->comms_up = true;
	return LOL_GPU_OK;
}

/* the partition of a frame of height h: one lol_gpu_rows per part (part p belongs to device p % n), the part table with
 * the parts stored device by device (a device's parts back to back, in part order: the layout of the per-device buffers
 * and, concatenated in device order, of the root's staging buffer), rows per device */
struct Split {
	int n_parts;
	lol_gpu_rows rows[MAX_PARTS];
	PartTable tab;
	uint32_t dev_row0[LOL_GPU_MULTI_MAX_DEVICES], dev_rows[LOL_GPU_MULTI_MAX_DEVICES];
};

static int split_frame(lol_gpu_multi* m, int h, Split& S) {
	S.n_parts = m->n * m->per_dev;
	const int band = m->band_override > 0 ? m->band_override : lol_gpu_choose_band_rows(h, S.n_parts);
	/* the root's lighter bands only when there is someone else to take the rest.  (lol_gpu_multi_testing_root_stride(m, k),
	 * a test switch: every k-th PART gets the root's band height even on one device, so that a one-GPU box runs unequal
	 * bands through this very path — split, launches, exchange, assembly, host copies; tests/test_multi_device.py.) */
	int root_band = m->n > 1 ? m->root_band : 0, root_stride = m->n;
	if (m->test_root_stride > 1 && m->root_band > 0) { root_band = m->root_band; root_stride = m->test_root_stride; }
	if (lol_gpu_split_rows(S.n_parts, band, root_band, root_stride, S.rows) != LOL_GPU_OK || !table_from_rows(S.tab, S.rows, S.n_parts))
		return mfail(m, LOL_GPU_ERR_ARG, "bad row partition");
	uint32_t row = 0;
	for (int d = 0; d < m->n; d++) {
		S.dev_row0[d] = row;
		for (int p = d; p < S.n_parts; p += m->n) {
			const int k = lol_gpu_part_rows(h, &S.rows[p]);
			if (k < 0) return mfail(m, LOL_GPU_ERR_ARG, "bad row partition");
			S.tab.row0[p] = row;
			row += (uint32_t)k;
		}
		S.dev_rows[d] = row - S.dev_row0[d];
	}
	if (row != (uint32_t)h) return mfail(m, LOL_GPU_ERR_ARG, "bad row partition");
	return LOL_GPU_OK;
}

/* queue the kernels of one frame: device d renders its parts back to back into part[slot] on its render stream */
static int render_parts(lol_gpu_multi* m, const lol_frame_camera* cam, int w, int h, int max_steps, const Split& S, int slot) {
	for (int d = 0; d < m->n; d++) {
		Device& D = m->dev[d];
		M_HIP(m, hipSetDevice(D.id));
		/* a slot's kernels have a stream of their own: what orders a slot is its events (part[slot] free again, rendered), and the
		 * frames of the two slots may overlap */
		hipStream_t rs = slot ? D.render2 : D.render;
		M_HIP(m, hipStreamWaitEvent(rs, D.sent[slot], 0));            /* the frame two back has left part[slot] */
		for (int p = d; p < S.n_parts; p += m->n) {
			if (lol_gpu_part_rows(h, &S.rows[p]) <= 0) continue;
			uint32_t* dst = D.part[slot] + (size_t)(S.tab.row0[p] - S.dev_row0[d]) * w;
			int st = lol_gpu_render_device(D.ctx, cam, w, h, max_steps, &S.rows[p], dst, (size_t)w * 4, nullptr, rs);
			if (st != LOL_GPU_OK) return mfail(m, st, "lol_gpu_render_device", lol_gpu_error(D.ctx));
		}
		M_HIP(m, hipEventRecord(D.rendered[slot], rs));
		M_HIP(m, hipStreamWaitEvent(D.xchg, D.rendered[slot], 0));
	}
	return LOL_GPU_OK;
}

int lol_gpu_multi_render_device(lol_gpu_multi* m, const lol_frame_camera* cam, int w, int h, int max_steps,
                                void* dst, size_t pitch_bytes) {
	if (!m || !cam || !dst) return LOL_GPU_ERR_ARG;
	if (w <= 0 || h <= 0 || pitch_bytes % 4 || pitch_bytes < (size_t)w * 4) return mfail(m, LOL_GPU_ERR_ARG, "bad frame geometry");
	const int n = m->n;
	Split S;
	int st = split_frame(m, h, S);
	if (st != LOL_GPU_OK) return st;
	st = ensure_comms(m);
	if (st != LOL_GPU_OK) return st;
	st = ensure_buffers(m, w, h, S.dev_rows, true);
	if (st != LOL_GPU_OK) return st;
	const int slot = (int)(m->frames % SLOTS);
	m->frames++;
	Device& root = m->dev[0];

	/* every device renders its parts; the root's own take the same road as the others (a send to itself),
	 * so that one code path serves any number of devices, one included */
	st = render_parts(m, cam, w, h, max_steps, S, slot);
	if (st != LOL_GPU_OK) return st;
	/* the exchange: one group, every device sends its parts (one message), the root receives them device by device */
	M_HIP(m, hipSetDevice(root.id));
	M_HIP(m, hipStreamWaitEvent(root.xchg, m->done[slot], 0));        /* staging[slot] was assembled two frames ago */
	M_NCCL(m, m->rccl.GroupStart());
	for (int d = 0; d < n; d++) {
		const size_t count = (size_t)S.dev_rows[d] * w;
		if (count == 0) continue;
		M_NCCL(m, m->rccl.Send(m->dev[d].part[slot], count, ncclUint32, 0, m->dev[d].comm, m->dev[d].xchg));
		M_NCCL(m, m->rccl.Recv(m->staging[slot] + (size_t)S.dev_row0[d] * w, count, ncclUint32, d, root.comm, root.xchg));
	}
	M_NCCL(m, m->rccl.GroupEnd());
	for (int d = 0; d < n; d++) {
		M_HIP(m, hipSetDevice(m->dev[d].id));
		M_HIP(m, hipEventRecord(m->dev[d].sent[slot], m->dev[d].xchg));
	}
	M_HIP(m, hipSetDevice(root.id));
	M_HIP(m, launch_assemble(m->staging[slot], S.tab, w, h, dst, pitch_bytes, root.xchg));
	M_HIP(m, hipEventRecord(m->done[slot], root.xchg));
	return LOL_GPU_OK;
}

int lol_gpu_multi_sync(lol_gpu_multi* m) {
	if (!m) return LOL_GPU_ERR_ARG;
	for (int d = 0; d < m->n; d++) {
		M_HIP(m, hipSetDevice(m->dev[d].id));
		M_HIP(m, hipStreamSynchronize(m->dev[d].render));
		M_HIP(m, hipStreamSynchronize(m->dev[d].render2));
		M_HIP(m, hipStreamSynchronize(m->dev[d].xchg));
	}
	M_HIP(m, hipSetDevice(m->dev[0].id));
	return LOL_GPU_OK;
}

/* One part of the frame from its compact device copy into the host surface: the part's bands are band_rows rows each,
 * cycle_rows rows apart in the frame — one strided 3-D copy (+ one 2-D copy if the part's last band is cut by the frame's end). */
static hipError_t copy_part_to_host(const uint32_t* part_dev, const lol_gpu_rows& R, int w, int h,
                                    char* host, size_t pitch, hipStream_t s) {
	const int rows = lol_gpu_part_rows(h, &R);
	const int full = rows / R.band_rows, tail = rows - full * R.band_rows;
	if (full > 0) {
		hipMemcpy3DParms p;
		memset(&p, 0, sizeof p);
		p.srcPtr = make_hipPitchedPtr(const_cast<uint32_t*>(part_dev), (size_t)w * 4, (size_t)w * 4, (size_t)R.band_rows);
		p.dstPtr = make_hipPitchedPtr(host + (size_t)R.offset_rows * pitch, pitch, (size_t)w * 4, (size_t)R.cycle_rows);
		p.extent = make_hipExtent((size_t)w * 4, (size_t)R.band_rows, (size_t)full);
		p.kind = hipMemcpyDeviceToHost;
		const hipError_t e = hipMemcpy3DAsync(&p, s);
		if (e != hipSuccess) return e;
	}
	if (tail > 0) {
		const size_t y0 = (size_t)full * R.cycle_rows + R.offset_rows;
		const hipError_t e = hipMemcpy2DAsync(host + y0 * pitch, pitch, part_dev + (size_t)full * R.band_rows * w, (size_t)w * 4,
		                                      (size_t)w * 4, (size_t)tail, hipMemcpyDeviceToHost, s);
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

int lol_gpu_multi_render_host(lol_gpu_multi* m, const lol_frame_camera* cam, int w, int h, int max_steps,
                              void* host_pixels, size_t pitch_bytes) {
	if (!m || !host_pixels || !cam) return LOL_GPU_ERR_ARG;
	if (w <= 0 || h <= 0 || pitch_bytes < (size_t)w * 4) return mfail(m, LOL_GPU_ERR_ARG, "bad frame geometry");
	Device& root = m->dev[0];
	if (m->host_via_root) {
		/* assemble on the root (RCCL exchange), then ONE copy over the root's link */
		const size_t need = (size_t)w * h * 4;
		M_HIP(m, hipSetDevice(root.id));
		if (need > m->frame_bytes) {                 /* the surface may be resized between frames (main.c:182-187) */
			int st = lol_gpu_multi_sync(m);
			if (st != LOL_GPU_OK) return st;
			if (m->d_frame) (void)hipFree(m->d_frame);
			m->d_frame = nullptr; m->frame_bytes = 0;
			M_HIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_frame), need));
			m->frame_bytes = need;
		}
		int st = lol_gpu_multi_render_device(m, cam, w, h, max_steps, m->d_frame, (size_t)w * 4);
		if (st != LOL_GPU_OK) return st;
		M_HIP(m, hipSetDevice(root.id));
		M_HIP(m, hipMemcpy2DAsync(host_pixels, pitch_bytes, m->d_frame, (size_t)w * 4, (size_t)w * 4, h,
		                          hipMemcpyDeviceToHost, root.xchg));
		M_HIP(m, hipStreamSynchronize(root.xchg));
		return LOL_GPU_OK;
	}
	/* every device copies its own bands into the surface: N links in parallel, no exchange, no RCCL */
	Split S;
	int st = split_frame(m, h, S);
	if (st != LOL_GPU_OK) return st;
	const int slot = (int)(m->frames % SLOTS);
	m->frames++;
	st = ensure_buffers(m, w, h, S.dev_rows, false);
	if (st != LOL_GPU_OK) return st;
	st = render_parts(m, cam, w, h, max_steps, S, slot);
	if (st != LOL_GPU_OK) return st;
	/* the copies: every device's by a thread of its own (one device: this thread), each waiting for its device */
	auto copy_device = [m, &S, slot, w, h, host_pixels, pitch_bytes](int d) -> hipError_t {
		Device& D = m->dev[d];
		hipError_t e = hipSetDevice(D.id);
		for (int p = d; p < S.n_parts && e == hipSuccess; p += m->n) {
			if (lol_gpu_part_rows(h, &S.rows[p]) <= 0) continue;
			e = copy_part_to_host(D.part[slot] + (size_t)(S.tab.row0[p] - S.dev_row0[d]) * w, S.rows[p], w, h,
			                      static_cast<char*>(host_pixels), pitch_bytes, D.xchg);
		}
		if (e == hipSuccess) e = hipEventRecord(D.sent[slot], D.xchg);
		if (e == hipSuccess) e = hipStreamSynchronize(D.xchg);
		return e;
	};
	hipError_t worst = hipSuccess;
	if (m->n == 1 && !m->test_force_threads) {
		worst = copy_device(0);
	} else {
		bool started[LOL_GPU_MULTI_MAX_DEVICES] = { false };
		for (int d = 0; d < m->n; d++) {
			try { started[d] = m->dev[d].copier.submit([copy_device, d] { return copy_device(d); }); }
			catch (...) { started[d] = false; }                 /* (std::function may allocate) */
		}
		for (int d = 0; d < m->n; d++) {
			/* a device whose thread could not be started is copied from here, after the others were set going */
			const hipError_t e = started[d] ? m->dev[d].copier.wait() : copy_device(d);
			if (e != hipSuccess) worst = e;
		}
	}
	if (worst != hipSuccess) return mfail(m, LOL_GPU_ERR_HIP, "copy into the host surface", hipGetErrorString(worst));
	M_HIP(m, hipSetDevice(root.id));
	return LOL_GPU_OK;
}

int lol_gpu_multi_malloc(lol_gpu_multi* m, size_t bytes, void** out) {
	if (!m || !out) return LOL_GPU_ERR_ARG;
	M_HIP(m, hipSetDevice(m->dev[0].id));
	M_HIP(m, hipMalloc(out, bytes));
	return LOL_GPU_OK;
}

int lol_gpu_multi_free(lol_gpu_multi* m, void* ptr) {
	if (!m) return LOL_GPU_ERR_ARG;
	M_HIP(m, hipSetDevice(m->dev[0].id));
	M_HIP(m, hipFree(ptr));
	return LOL_GPU_OK;
}

int lol_gpu_multi_memcpy_d2h(lol_gpu_multi* m, void* host, const void* dev, size_t bytes) {
	if (!m || !host || !dev) return LOL_GPU_ERR_ARG;
	int st = lol_gpu_multi_sync(m);
	if (st != LOL_GPU_OK) return st;
	M_HIP(m, hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
	return LOL_GPU_OK;
}

}  // extern "C"
