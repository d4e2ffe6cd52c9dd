// Atomic throughput probe for gfx950: the rasteriser's depth test is a 64-bit atomicMin in LDS, binning uses 32-bit
// global atomicAdd; this measures what the chip sustains for each, on distinct and on shared addresses.
// Build: hipcc -O3 --offload-arch=gfx950 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 1024

// LDS: every lane of a 256-thread workgroup hits its own u64 slot (4 per lane, rotating), or all lanes of a wave one slot
template <int SAME> __global__ __launch_bounds__(256) void k_lds_min64(unsigned long long* out, unsigned long long seed)
{
	__shared__ unsigned long long t[1024];
	for (int i = threadIdx.x; i < 1024; i += 256) t[i] = ~0ull;
	__syncthreads();
	unsigned long long v = seed + threadIdx.x * 977ull;
	for (int i = 0; i < ITERS; i++) {
		const int slot = SAME ? ((threadIdx.x >> 6) * 64 + (i & 63)) : ((threadIdx.x + 256 * (i & 3)) & 1023);
		atomicMin(&t[slot], v);
		v = v * 6364136223846793005ull + 1442695040888963407ull;
	}
	__syncthreads();
	if (t[threadIdx.x] == 12345ull) out[0] = 1;
}

// global memory: distinct addresses over a 64 MB array (stride-free, per-lane pseudo-random), or one hot address per wave
template <int MODE> __global__ __launch_bounds__(256) void k_glb(uint32_t* a32, unsigned long long* a64, uint32_t mask, uint32_t* sink)
{
	uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	uint32_t acc = 0;
	for (int i = 0; i < 64; i++) {
		x = x * 1664525u + 1013904223u;
		const uint32_t idx = (x >> 4) & mask;
		if (MODE == 0) atomicAdd(&a32[idx], 1u);                                   // no return, distinct
		if (MODE == 1) acc += atomicAdd(&a32[idx], 1u);                            // returning, distinct
		if (MODE == 2) atomicMin(&a64[idx], ((unsigned long long)x << 32) | idx);  // 64-bit min, distinct
		if (MODE == 3) acc += atomicAdd(&a32[(blockIdx.x * 4u + (threadIdx.x >> 6)) & mask], 1u); // returning, one address per wave
		if (MODE == 4) acc += atomicAdd(&a32[0], 1u);                              // returning, one address for everyone
	}
	if (acc == 0xdeadbeefu) sink[0] = acc;
}

static float timeit(void (*launch)(void))
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	launch(); (void)hipDeviceSynchronize();
	(void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

static unsigned long long* g_out; static uint32_t* g_a32; static unsigned long long* g_a64; static uint32_t* g_sink;
static const int LDS_WGS = 256 * 8, GLB_WGS = 256 * 32;
static const uint32_t MASK = (1u << 23) - 1u; // 8M slots: 32 MB of u32, 64 MB of u64

int main()
{
	(void)hipMalloc(&g_out, 64); (void)hipMalloc(&g_a32, (size_t)(MASK + 1) * 4); (void)hipMalloc(&g_a64, (size_t)(MASK + 1) * 8); (void)hipMalloc(&g_sink, 64);
	(void)hipMemset(g_a32, 0, (size_t)(MASK + 1) * 4); (void)hipMemset(g_a64, 0xff, (size_t)(MASK + 1) * 8);
	float ms;
	ms = timeit([] { hipLaunchKernelGGL(k_lds_min64<0>, dim3(LDS_WGS), dim3(256), 0, 0, g_out, 1ull); });
	printf("LDS 64-bit atomicMin, distinct slots per lane : %7.1f G atomics/s (%.3f ms)\n", (double)LDS_WGS * 256 * ITERS / ms * 1e-6, ms);
	ms = timeit([] { hipLaunchKernelGGL(k_lds_min64<1>, dim3(LDS_WGS), dim3(256), 0, 0, g_out, 1ull); });
	printf("LDS 64-bit atomicMin, one slot per wavefront  : %7.1f G atomics/s (%.3f ms)\n", (double)LDS_WGS * 256 * ITERS / ms * 1e-6, ms);
	const double n = (double)GLB_WGS * 256 * 64;
	ms = timeit([] { hipLaunchKernelGGL(k_glb<0>, dim3(GLB_WGS), dim3(256), 0, 0, g_a32, g_a64, MASK, g_sink); });
	printf("global 32-bit atomicAdd, no return, distinct  : %7.1f G atomics/s (%.3f ms)\n", n / ms * 1e-6, ms);
	ms = timeit([] { hipLaunchKernelGGL(k_glb<1>, dim3(GLB_WGS), dim3(256), 0, 0, g_a32, g_a64, MASK, g_sink); });
	printf("global 32-bit atomicAdd, returning, distinct  : %7.1f G atomics/s (%.3f ms)\n", n / ms * 1e-6, ms);
	ms = timeit([] { hipLaunchKernelGGL(k_glb<2>, dim3(GLB_WGS), dim3(256), 0, 0, g_a32, g_a64, MASK, g_sink); });
	printf("global 64-bit atomicMin, no return, distinct  : %7.1f G atomics/s (%.3f ms)\n", n / ms * 1e-6, ms);
	ms = timeit([] { hipLaunchKernelGGL(k_glb<3>, dim3(GLB_WGS), dim3(256), 0, 0, g_a32, g_a64, MASK, g_sink); });
	printf("global 32-bit atomicAdd, returning, one address per wavefront : %7.1f G atomics/s (%.3f ms)\n", n / ms * 1e-6, ms);
	ms = timeit([] { hipLaunchKernelGGL(k_glb<4>, dim3(256), dim3(256), 0, 0, g_a32, g_a64, MASK, g_sink); });
	printf("global 32-bit atomicAdd, returning, one address for all       : %7.2f G atomics/s (%.3f ms)\n", 256.0 * 256 * 64 / ms * 1e-6, ms);
	return 0;
}
