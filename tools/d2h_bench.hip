/*
 * d2h_bench.hip — how fast can a rendered frame reach HOST memory the host owns (surf->pixels, naive_renderer.c:233-235)?
 *
 * One 3840x2160 XRGB8888 frame (33.2 MB) in device memory → a pitched host surface, timed per frame (wall clock, median
 * of N) for every route the library could take:
 *   pageable-1       hipMemcpy2DAsync into malloc'd memory, one thread            (round 2's route)
 *   pageable-T       T threads, each copies a band of rows on its own stream      (the runtime stages pageable copies on
 *                                                                                   the CALLING thread, so T threads = T copiers)
 *   registered       hipHostRegister'd surface, one asynchronous pitched copy      (needs the host to vouch for the lifetime)
 *   direct           a kernel stores 64-byte row segments straight into the mapped surface
 *   staged-T         DMA into a hipHostMalloc'd staging buffer of the library, then T threads memcpy it into the surface
 *   register cost    hipHostRegister + hipHostUnregister of the surface, per call
 * Build: hipcc -O2 --offload-arch=gfx950 -o tools/d2h_bench tools/d2h_bench.hip -lpthread
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* the render kernel's store pattern: a wave writes 4 row segments of 64 bytes (16 pixels x 4 rows patch) */
__global__ void store_patches(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int w, int h, uint32_t pitch_px) {
	const int lane = threadIdx.x & 63;
	const int x = blockIdx.x * 16 + (lane & 15), y = blockIdx.y * 4 + (lane >> 4);
	if (x < w && y < h) dst[(size_t)y * pitch_px + x] = src[(size_t)y * w + x];
}

template <class F> static double median_ms(int n, F f) {
	std::vector<double> t;
	for (int i = 0; i < n; i++) { double t0 = now_ms(); f(); t.push_back(now_ms() - t0); }
	std::sort(t.begin(), t.end());
	return t[t.size() / 2];
}

int main(int argc, char** argv) {
	const int w = 3840, h = 2160, N = argc > 1 ? atoi(argv[1]) : 15;
	const size_t pitch = (size_t)(w + 16) * 4, bytes = pitch * h, frame = (size_t)w * 4 * h;
	uint32_t* d_frame;
	CK(hipMalloc(reinterpret_cast<void**>(&d_frame), frame));
	{
		std::vector<uint32_t> init((size_t)w * h);
		for (size_t i = 0; i < init.size(); i++) init[i] = (uint32_t)(i * 2654435761u);
		CK(hipMemcpy(d_frame, init.data(), frame, hipMemcpyHostToDevice));
	}
	char* surf = static_cast<char*>(aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095));
	memset(surf, 0, bytes);
	auto check = [&](const char* what) {
		std::vector<uint32_t> back((size_t)w * h);
		CK(hipMemcpy(back.data(), d_frame, frame, hipMemcpyDeviceToHost));
		for (int y = 0; y < h; y++)
			if (memcmp(surf + (size_t)y * pitch, &back[(size_t)y * w], (size_t)w * 4)) { fprintf(stderr, "%s: row %d differs\n", what, y); exit(2); }
		memset(surf, 0, bytes);
	};
	auto report = [&](const char* name, double ms) { printf("{\"route\": \"%s\", \"ms_per_frame\": %.4f, \"GBps\": %.1f}\n", name, ms, frame / ms / 1e6); fflush(stdout); };

	hipStream_t s0;
	CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
	/* pageable, one thread */
	auto pageable1 = [&] { CK(hipMemcpy2DAsync(surf, pitch, d_frame, (size_t)w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); };
	pageable1(); check("pageable-1");
	report("pageable-1", median_ms(N, pageable1));

	/* pageable, T threads with their own streams: persistent workers, released per frame */
	for (int T : { 2, 4, 8, 16 }) {
		std::vector<hipStream_t> st(T);
		for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		auto run = [&] {
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++)
				th.emplace_back([&, t] {
					CK(hipSetDevice(0));
					const int y0 = (int)((long)h * t / T), y1 = (int)((long)h * (t + 1) / T);
					CK(hipMemcpy2DAsync(surf + (size_t)y0 * pitch, pitch, d_frame + (size_t)y0 * w, (size_t)w * 4, (size_t)w * 4, y1 - y0,
					                    hipMemcpyDeviceToHost, st[t]));
					CK(hipStreamSynchronize(st[t]));
				});
			for (auto& x : th) x.join();
		};
		run(); check("pageable-T");
		char name[32]; snprintf(name, sizeof name, "pageable-%d", T);
		report(name, median_ms(N, run));
		for (auto& s : st) CK(hipStreamDestroy(s));
	}

	/* library-owned pinned staging + T-thread memcpy into the surface */
	{
		char* stage;
		CK(hipHostMalloc(reinterpret_cast<void**>(&stage), frame, hipHostMallocDefault));
		auto dma = [&] { CK(hipMemcpyAsync(stage, d_frame, frame, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); };
		dma();
		report("dma-to-own-pinned (no copy-out)", median_ms(N, dma));
		for (int T : { 1, 2, 4, 8, 16 }) {
			auto run = [&] {
				dma();
				std::vector<std::thread> th;
				for (int t = 0; t < T; t++)
					th.emplace_back([&, t] {
						const int y0 = (int)((long)h * t / T), y1 = (int)((long)h * (t + 1) / T);
						for (int y = y0; y < y1; y++) memcpy(surf + (size_t)y * pitch, stage + (size_t)y * w * 4, (size_t)w * 4);
					});
				for (auto& x : th) x.join();
			};
			run(); check("staged-T");
			char name[32]; snprintf(name, sizeof name, "staged-%d", T);
			report(name, median_ms(N, run));
		}
		/* chunked: copy-out of chunk i overlaps the DMA of chunk i+1 (what the runtime does for pageable memory), T threads per chunk */
		for (int T : { 4, 8 }) {
			const int CH = 8;
			hipEvent_t ev[CH];
			for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			auto run = [&] {
				for (int c = 0; c < CH; c++) {
					const int y0 = h * c / CH, y1 = h * (c + 1) / CH;
					CK(hipMemcpyAsync(stage + (size_t)y0 * w * 4, d_frame + (size_t)y0 * w, (size_t)(y1 - y0) * w * 4, hipMemcpyDeviceToHost, s0));
					CK(hipEventRecord(ev[c], s0));
				}
				std::vector<std::thread> th;
				for (int t = 0; t < T; t++)
					th.emplace_back([&, t] {
						for (int c = 0; c < CH; c++) {
							if (hipEventSynchronize(ev[c]) != hipSuccess) exit(3);
							const int c0 = h * c / CH, c1 = h * (c + 1) / CH;
							const int y0 = c0 + (int)((long)(c1 - c0) * t / T), y1 = c0 + (int)((long)(c1 - c0) * (t + 1) / T);
							for (int y = y0; y < y1; y++) memcpy(surf + (size_t)y * pitch, stage + (size_t)y * w * 4, (size_t)w * 4);
						}
					});
				for (auto& x : th) x.join();
			};
			run(); check("staged-chunked");
			char name[48]; snprintf(name, sizeof name, "staged-chunked8-%d", T);
			report(name, median_ms(N, run));
			for (auto& e : ev) CK(hipEventDestroy(e));
		}
		CK(hipHostFree(stage));
	}

	/* register / unregister cost */
	{
		auto reg = [&] { CK(hipHostRegister(surf, bytes, hipHostRegisterPortable | hipHostRegisterMapped)); CK(hipHostUnregister(surf)); };
		reg();
		report("hipHostRegister+Unregister (no copy)", median_ms(N, reg));
	}
	/* registered surface: async pitched copy, and direct stores */
	CK(hipHostRegister(surf, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
	auto regcopy = [&] { CK(hipMemcpy2DAsync(surf, pitch, d_frame, (size_t)w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); };
	regcopy(); check("registered");
	report("registered-copy2d", median_ms(N, regcopy));
	{
		void* dv = nullptr;
		CK(hipHostGetDevicePointer(&dv, surf, 0));
		auto direct = [&] {
			hipLaunchKernelGGL(store_patches, dim3((w + 15) / 16, (h + 3) / 4), dim3(64), 0, s0, d_frame, static_cast<uint32_t*>(dv), w, h, (uint32_t)(pitch / 4));
			CK(hipStreamSynchronize(s0));
		};
		direct(); check("direct");
		report("direct-stores-64B-segments", median_ms(N, direct));
	}
	CK(hipHostUnregister(surf));
	free(surf);
	return 0;
}
