// Workgroup dispatch rate on gfx950: kernels that do nothing (or wait a fixed number of clocks), launched over grids of
// the sizes the renderer uses, with and without LDS.  Time per launch / workgroups = what the dispatcher alone costs;
// the waiting variant shows how a grid that exceeds the resident slots by a few per cent behaves (one more round).
// Build: hipcc -O3 --offload-arch=gfx950 -o dispatch_rate dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LDS_WORDS, bool FULLREGS = false> __global__ __launch_bounds__(256) void k_nop(uint32_t* out, uint32_t spin)
{
	// FULLREGS: the wavefront takes 64 vector registers, so that eight of them are the whole register file of a SIMD
	if (FULLREGS) asm volatile("v_mov_b32 v63, 0" ::: "v63");
	__shared__ uint32_t s[LDS_WORDS ? LDS_WORDS : 1];
	if (LDS_WORDS) s[threadIdx.x % (LDS_WORDS ? LDS_WORDS : 1)] = threadIdx.x;
	if (spin) {
		// spin with bit 31 set: every seventh workgroup waits a third as long (a launch whose workgroups differ, as k_setup's do)
		// bit 30: the first workgroup waits twelve times as long (one slow workgroup ahead of many quick ones)
		uint32_t n = spin & 0x3fffffffu;
		if ((spin >> 31) && blockIdx.x % 7u == 0u) n /= 3u;
		if (((spin >> 30) & 1u) && blockIdx.x == 0u) n *= 12u;
		const unsigned long long t0 = __builtin_readcyclecounter();
		while (__builtin_readcyclecounter() - t0 < n) {}
	}
	if (out && LDS_WORDS && s[0] == 0xdeadbeefu) out[0] = 1;
}

template <int LDS_WORDS, bool FULLREGS = false> static void run(const char* name, uint32_t groups, uint32_t threads, uint32_t spin)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int reps = 50;
	for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_nop<LDS_WORDS, FULLREGS>), dim3(groups), dim3(threads), 0, 0, (uint32_t*)nullptr, spin);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_nop<LDS_WORDS, FULLREGS>), dim3(groups), dim3(threads), 0, 0, (uint32_t*)nullptr, spin);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double us = ms * 1e3 / reps;
	printf("%-18s groups %6u x %3u threads spin %6u: %7.2f us per launch, %6.1f workgroups/us, %7.1f waves/us\n", name, groups, threads, spin & 0x3fffffffu, us,
	       groups / us, groups * ((threads + 63) / 64) / us);
}

int main()
{
	// the setup kernel's grid (548 k triangles / 256), the shading kernels' (3840 x 2160 / 256), a tile grid
	for (uint32_t groups : { 2142u, 8160u, 32400u }) {
		run<0>("no LDS", groups, 256, 0);
		run<1536>("6 KB LDS", groups, 256, 0);
		run<4608>("18 KB LDS", groups, 256, 0);
	}
	run<0>("no LDS", 8568, 64, 0);
	run<0>("no LDS", 129600, 64, 0);
	// about 10 us of waiting per wavefront (24 000 shader clocks):
	for (uint32_t groups : { 1024u, 2048u, 2142u, 4096u }) run<1536>("6 KB LDS, waits", groups, 256, 24000);
	// how much LDS may a 256-thread workgroup take before fewer than eight fit a CU (2048 workgroups = one round)?
	run<3584>("14 KB LDS, waits", 2048, 256, 24000);
	run<3840>("15 KB LDS, waits", 2048, 256, 24000);
	run<4096>("16 KB LDS, waits", 2048, 256, 24000);
	run<4352>("17 KB LDS, waits", 2048, 256, 24000);
	run<4608>("18 KB LDS, waits", 2048, 256, 24000);
	run<5120>("20 KB LDS, waits", 2048, 256, 24000);
	run<5120>("20 KB LDS, waits", 1792, 256, 24000);
	run<5632>("22 KB LDS, waits", 1792, 256, 24000);
	run<6144>("24 KB LDS, waits", 1792, 256, 24000);
	run<6144>("24 KB LDS, waits", 1536, 256, 24000);
	// uneven workgroups, a twentieth over the resident slots: does the LDS a workgroup holds change how soon a freed
	// slot is taken?
	run<0>("no LDS, uneven", 2142, 256, 24000u | 0x80000000u);
	run<1536>("6 KB LDS, uneven", 2142, 256, 24000u | 0x80000000u);
	run<3072>("12 KB LDS, uneven", 2142, 256, 24000u | 0x80000000u);
	run<4608>("18 KB LDS, uneven", 2142, 256, 24000u | 0x80000000u);
	run<0>("no LDS, uneven", 2048, 256, 24000u | 0x80000000u);
	run<4608, true>("18 KB, 64 VGPRs, uneven", 2142, 256, 24000u | 0x80000000u);
	run<0, true>("64 VGPRs, uneven", 2142, 256, 24000u | 0x80000000u);
	// one workgroup of 60 us ahead of 20 479 of 5 us (50 us of work for the other slots): do the quick ones flow past it?
	run<0>("one slow first", 20480, 256, 12000u | 0x40000000u);
	run<0>("all quick", 20480, 256, 12000u);
	return 0;
}
