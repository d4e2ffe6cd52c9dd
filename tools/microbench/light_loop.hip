// Shader-clock probe: runs the per-light arithmetic of k_deferred_light on synthetic register inputs and reports
// shader cycles (s_memtime) per light per wave and the shader clock (s_memtime ticks / wall_clock64 at 100 MHz).
// Build: hipcc -O3 --offload-arch=gfx950 -o light_loop light_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Lights { float pos[21][4]; };
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
template <int FMAONLY, int FLAGS> __global__ __launch_bounds__(256) void k_loop(Lights L, float* out, unsigned long long* clk, int reps)
{
	const float t0 = threadIdx.x * 1e-3f;
	float Px = 1.0f + t0, Py = -2.0f + t0, Pz = -20.0f - t0, Nx = 0.0f, Ny = 1.0f, Nz = 0.0f;
	float V2x = 0.6f, V2y = 1.2f, V2z = 1.0f, ndv = 0.6f, ndvp = 0.6f, pir2 = 0.3f + t0, cexp = 3.0f;
	float S0 = 0, S1 = 0, S2 = 0, S3 = 0;
	float q0 = 1.5f + t0, q1 = 2.5f + t0, q2 = 3.5f + t0, q3 = 4.5f + t0, q4 = 5.5f + t0;
	unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	unsigned long long m0 = __builtin_amdgcn_s_memtime();
	for (int r = 0; r < reps; r++) {
#pragma unroll 3
		for (int i = 0; i < 21; i++) {
			if (FMAONLY) {
#pragma unroll
				for (int k = 0; k < 12; k++) { S0 = fmaf(S0, Px, Py); S1 = fmaf(S1, Py, Pz); S2 = fmaf(S2, Pz, Px); S3 = fmaf(S3, Nx, Ny); }
				continue;
			}
			// FLAGS bit0: light position through VGPRs; bit1: transcendentals replaced by multiplies; bit2: min replaced by mul
			const float vz = (FLAGS & 1) ? t0 * 1e-20f : 0.0f;
			const float Lx = (L.pos[i][0] + vz) - Px, Ly = (L.pos[i][1] + vz) - Py, Lz = (L.pos[i][2] + vz) - Pz;
			const float d2 = Lx * Lx + Ly * Ly + Lz * Lz;
			const float invd = (FLAGS & (2|8)) ? d2 * 0.37f : __builtin_amdgcn_rsqf(d2);
			const float invd2 = invd * invd;
			const float nl = Nx * Lx + Ny * Ly + Nz * Lz;
			const float ndl = nl * invd;
			const float ndlp = sat(nl * invd);
			const float opv2 = (V2x * Lx + V2y * Ly + V2z * Lz) * invd + 2.0f;
			const float invh = (FLAGS & (2|16)) ? opv2 * 0.21f : __builtin_amdgcn_rsqf(opv2);
			const float xx = 1.0f - 0.5f * (opv2 * invh);
			const float ndh = sat((ndv + ndl) * invh);
			const float nh2 = ndh * ndh, nh4 = nh2 * nh2;
			const float rn = (FLAGS & (2|32)) ? nh2 * 0.9f : __builtin_amdgcn_rcpf(nh2);
			const float earg = (FLAGS & 256) ? fmaxf(cexp - cexp * rn, -20.0f) : cexp - cexp * rn;
			const float ex = (FLAGS & (2|64)) ? earg * 0.5f : __builtin_amdgcn_exp2f(earg);
			const float rdin = (0.001f + pir2 * nh4) * (0.00025f + ndvp * ndlp);
			const float rden = (FLAGS & (2|128)) ? rdin * 0.77f : __builtin_amdgcn_rcpf(rdin);
			const float x2 = xx * xx, x5 = x2 * x2 * xx;
			const float Gq = (FLAGS & 4) ? ((ndh * invh) * (ndvp * ndlp)) * 0.25f : fminf((ndh * invh) * fminf(ndvp, ndlp), 0.25f);
			const float spec = ex * Gq * rden;
			const float t = ndlp * invd2;
			const float ts = t * spec;
			S0 += invd2; S1 += t - t * x5; S2 += ts; S3 += ts * x5;
			if (FLAGS & 512) { asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4)); }
		}
		Px += S0 * 1e-9f;
	}
	unsigned long long m1 = __builtin_amdgcn_s_memtime();
	unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	if (S0 + S1 + S2 + S3 + q0 + q1 + q2 + q3 + q4 == 12345.678f) out[0] = S0;
	if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = m1 - m0; clk[1] = w1 - w0; clk[2] = c1 - c0; }
}
template <int F, int FLAGS> static void run(int w, int reps)
{
	Lights L;
	for (int i = 0; i < 21; i++) { L.pos[i][0] = (float)(i % 5) - 2.0f; L.pos[i][1] = -1.1f; L.pos[i][2] = -10.0f - 4.0f * i; L.pos[i][3] = 0; }
	float* d; unsigned long long* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 64);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL((k_loop<F, FLAGS>), dim3(256 * w), dim3(256), 0, 0, L, d, c, reps);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k_loop<F, FLAGS>), dim3(256 * w), dim3(256), 0, 0, L, d, c, reps);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[3]; (void)hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
	const double lights = 21.0 * reps;
	printf("flags %d %s waves/SIMD %d reps %d: wall %.3f ms; wave0: s_memtime %llu ticks, wall_clock64 %llu ticks (100 MHz) -> s_memtime runs at %.1f MHz; "
	       "%.1f s_memtime ticks per light per wave; kernel time per light per SIMD-resident wave set: %.1f ns\n",
	       FLAGS, F ? "fma-only(48/light)" : "light loop", w, reps, ms, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / lights, ms * 1e6 / lights);
}
int main()
{
	run<0, 0>(8, 1000); run<0, 2>(8, 1000); run<0, 514>(8, 1000);
	return 0;
}
