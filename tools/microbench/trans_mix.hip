// Cost of a transcendental instruction inside a stream of plain f32 VALU on gfx950, as a function of how many
// plain instructions separate two of them and of whether they are issued singly or in groups of G.
// Build: hipcc -O3 --offload-arch=gfx950 -o trans_mix trans_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 2048
// one period = G transcendentals followed by G*N plain FMAs, all on independent registers (16 chains)
template <int N, int G> __global__ __launch_bounds__(256) void k_probe(float* out, float seed)
{
	float a[16], t[8];
#pragma unroll
	for (int c = 0; c < 16; c++) a[c] = seed + c + threadIdx.x * 1e-3f;
#pragma unroll
	for (int c = 0; c < 8; c++) t[c] = seed + 2.0f + c;
	const float m = 1.0000001f, k = 1e-7f;
	for (int i = 0; i < ITERS; i++) {
#pragma unroll
		for (int g = 0; g < G; g++) asm volatile("v_rsq_f32 %0, %0" : "+v"(t[g]));
#pragma unroll
		for (int j = 0; j < G * N; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j % 16]) : "v"(m), "v"(k));
	}
	float s = 0;
#pragma unroll
	for (int c = 0; c < 16; c++) s += a[c];
#pragma unroll
	for (int c = 0; c < 8; c++) s += t[c];
	if (s == 12345.678f) out[0] = s;
}
template <int N, int G> static void run(float* d, double ghz)
{
	const int w = 8;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL((k_probe<N, G>), dim3(256 * w), dim3(256), 0, 0, d, 1.0f);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k_probe<N, G>), dim3(256 * w), dim3(256), 0, 0, d, 1.0f);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)ITERS * w); // SIMD cycles per period per wave
	printf("group %d, %2d plain per transcendental: %.1f cycles per period -> %.2f per instruction; transcendental = %.1f if a plain one costs 2.4\n",
	       G, N, cyc, cyc / (G * (N + 1)), (cyc - 2.4 * G * N) / G);
}
// 8 transcendentals, then M plain FMAs in a rolled inner loop (small code), repeated
template <int FENCE> __global__ __launch_bounds__(256) void k_phase(float* out, float seed, int inner)
{
	float a[16], t[8];
#pragma unroll
	for (int c = 0; c < 16; c++) a[c] = seed + c + threadIdx.x * 1e-3f;
#pragma unroll
	for (int c = 0; c < 8; c++) t[c] = seed + 2.0f + c;
	const float m = 1.0000001f, k = 1e-7f;
	if (FENCE == 4) {
#pragma unroll
		for (int g = 0; g < 8; g++) asm volatile("v_rsq_f32 %0, %0" : "+v"(t[g]));
	}
	for (int i = 0; i < 64; i++) {
		if (FENCE < 3) {
#pragma unroll
			for (int g = 0; g < 8; g++) asm volatile("v_rsq_f32 %0, %0" : "+v"(t[g]));
		}
		if (FENCE == 5) {
#pragma unroll
			for (int g = 0; g < 8; g++) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(t[g]));
		}
		if (FENCE == 6) {
#pragma unroll
			for (int g = 0; g < 8; g++) asm volatile("v_rsq_f32 %0, %0\n v_nop\n v_nop" : "+v"(t[g]));
		}
		if (FENCE == 1) asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7");
		if (FENCE == 2) asm volatile("s_sleep 1");
		for (int r = 0; r < inner; r++) {
#pragma unroll
			for (int j = 0; j < 16; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(k));
		}
	}
	float s = 0;
#pragma unroll
	for (int c = 0; c < 16; c++) s += a[c];
#pragma unroll
	for (int c = 0; c < 8; c++) s += t[c];
	if (s == 12345.678f) out[0] = s;
}
template <int FENCE> static void phase(float* d, double ghz, int inner)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(k_phase<FENCE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, inner);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k_phase<FENCE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, inner);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double cyc = ms * 1e-3 * ghz * 1e9 / (64.0 * 8); // per period per wave
	printf("fence %d: 8 transcendentals then %5d plain: %.0f cycles per period; plain = %.2f each if a transcendental costs 8\n", FENCE, inner * 16, cyc, (cyc - 64.0) / (inner * 16));
}
// waves of odd index run only transcendentals, waves of even index only plain FMAs: does one wave's
// transcendental slow another wave's plain instructions on the same SIMD?
__global__ __launch_bounds__(256) void k_split(float* out, float seed, int mode, unsigned long long* clk)
{
	float a[16];
#pragma unroll
	for (int c = 0; c < 16; c++) a[c] = seed + c + threadIdx.x * 1e-3f;
	const float m = 1.0000001f, k = 1e-7f;
	const bool trans = mode == 2 || (mode == 1 && (blockIdx.x & 1));
	unsigned long long w0 = wall_clock64();
	if (trans) {
		for (int i = 0; i < ITERS; i++) {
#pragma unroll
			for (int j = 0; j < 16; j++) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[j]));
		}
	} else {
		for (int i = 0; i < ITERS * 4; i++) {
#pragma unroll
			for (int j = 0; j < 16; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(k));
		}
	}
	unsigned long long w1 = wall_clock64();
	float s = 0;
#pragma unroll
	for (int c = 0; c < 16; c++) s += a[c];
	if (s == 12345.678f) out[0] = s;
	if (threadIdx.x == 0 && blockIdx.x < 2) clk[blockIdx.x] = w1 - w0;
}
static void split(float* d, double ghz)
{
	unsigned long long* c; (void)hipMalloc(&c, 64);
	for (int mode = 0; mode < 3; mode++) {
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL(k_split, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, mode, c);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k_split, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, mode, c);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
		unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
		printf("split mode %d (0 all plain [4x iters], 1 even WGs plain / odd WGs transcendental, 2 all transcendental): wall %.3f ms; WG0 %.3f ms, WG1 %.3f ms\n",
		       mode, ms, h[0] / 1e5, h[1] / 1e5);
	}
}
int main(int argc, char** argv)
{
	const double ghz = argc > 1 ? atof(argv[1]) : 2.35;
	float* d; (void)hipMalloc(&d, 4096);
	for (int inner : { 1, 2, 4, 8, 16, 64 }) { phase<0>(d, ghz, inner); phase<5>(d, ghz, inner); }
	phase<3>(d, ghz, 64);
	return 0;
}
