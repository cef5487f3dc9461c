// Exhaustive search over all 2^32 float inputs: which cheap sqrt sequences are correctly rounded
// (== __builtin_sqrtf) on [2^-96, inf) and NaN?  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool same(float a, float b) {
	return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b) || (a != a && b != b);
}

template <int V> __device__ __forceinline__ float cand(float x) {
	if (V == 0) {            // full Goldschmidt (reference point)
		float y = __builtin_amdgcn_rsqf(x), s = x * y, h = y * 0.5f;
		float e = __builtin_fmaf(-h, s, 0.5f);
		h = __builtin_fmaf(h, e, h); s = __builtin_fmaf(s, e, s);
		float d = __builtin_fmaf(-s, s, x);
		return __builtin_fmaf(d, h, s);
	} else if (V == 1) {     // no first iteration: rsq, 2 mul, 2 fma
		float y = __builtin_amdgcn_rsqf(x), s = x * y, h = y * 0.5f;
		float d = __builtin_fmaf(-s, s, x);
		return __builtin_fmaf(d, h, s);
	} else if (V == 2) {     // refine s only (h stale): rsq, 2 mul, 4 fma
		float y = __builtin_amdgcn_rsqf(x), s = x * y, h = y * 0.5f;
		float e = __builtin_fmaf(-h, s, 0.5f);
		s = __builtin_fmaf(s, e, s);
		float d = __builtin_fmaf(-s, s, x);
		return __builtin_fmaf(d, h, s);
	} else if (V == 3) {     // two residual corrections with stale h: rsq, 2 mul, 4 fma
		float y = __builtin_amdgcn_rsqf(x), s = x * y, h = y * 0.5f;
		float d = __builtin_fmaf(-s, s, x);
		s = __builtin_fmaf(d, h, s);
		d = __builtin_fmaf(-s, s, x);
		return __builtin_fmaf(d, h, s);
	} else if (V == 4) {     // v_sqrt + one residual correction using rsq-free h = 0.5/r via rcp (2 trans)
		float r = __builtin_amdgcn_sqrtf(x);
		float h = 0.5f * __builtin_amdgcn_rcpf(r);
		float d = __builtin_fmaf(-r, r, x);
		return __builtin_fmaf(d, h, r);
	} else if (V == 5) {     // v_sqrt then the +-1ulp test (sqrt_pm)
		float r = __builtin_amdgcn_sqrtf(x);
		int ri = __builtin_bit_cast(int, r);
		float rm = __builtin_bit_cast(float, ri - 1), rp = __builtin_bit_cast(float, ri + 1);
		float em = __builtin_fmaf(-rm, r, x), ep = __builtin_fmaf(-rp, r, x);
		r = em <= 0.f ? rm : r;
		return ep > 0.f ? rp : r;
	} else if (V == 6) {     // raw v_sqrt_f32
		return __builtin_amdgcn_sqrtf(x);
	} else {                 // x * rsq(x)
		return x * __builtin_amdgcn_rsqf(x);
	}
}

template <int V>
__global__ __launch_bounds__(256) void search(unsigned long long* bad, uint32_t* first) {
	uint32_t base = blockIdx.x * 256 + threadIdx.x;
	unsigned n = 0;
	for (uint32_t it = 0; it < 256; it++) {
		uint32_t bits = base + it * (65536u * 256u);
		float x = __builtin_bit_cast(float, bits);
		bool dom = (x >= 0x1p-96f && x < __builtin_inff()) || x != x;
		if (dom && !same(cand<V>(x), __builtin_sqrtf(x))) { n++; atomicMin(first, bits); }
	}
	if (n) atomicAdd(bad, (unsigned long long)n);
}

template <int V> int run(const char* name, unsigned long long* d_bad, uint32_t* d_first) {
	unsigned long long bad = 0; uint32_t first = 0xffffffffu;
	CHECK(hipMemcpy(d_bad, &bad, 8, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(search<V>, dim3(65536), dim3(256), 0, 0, d_bad, d_first);
	CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost));
	printf("%-60s mismatches %llu (first input bits 0x%08x)\n", name, bad, first);
	return 0;
}

int main() {
	unsigned long long* d_bad; uint32_t* d_first;
	CHECK(hipMalloc(&d_bad, 8)); CHECK(hipMalloc(&d_first, 4));
	run<0>("V0 rsq + full Goldschmidt (2 mul, 5 fma)", d_bad, d_first);
	run<1>("V1 rsq + 2 mul + 2 fma (no first iteration)", d_bad, d_first);
	run<2>("V2 rsq + 2 mul + 4 fma (refine s only)", d_bad, d_first);
	run<3>("V3 rsq + 2 mul + 4 fma (two residual corrections)", d_bad, d_first);
	run<4>("V4 sqrt + rcp + mul + 2 fma", d_bad, d_first);
	run<5>("V5 sqrt + +-1ulp test", d_bad, d_first);
	run<6>("V6 raw v_sqrt_f32", d_bad, d_first);
	run<7>("V7 x * v_rsq_f32(x)", d_bad, d_first);
	return 0;
}
