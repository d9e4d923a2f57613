/* Latency of SMALL host-buffer calls (a sound-card sized request: 480 output frames of stereo 44.1 -> 48 kHz per call),
   where the offload's fixed costs - plan lookup, upload, launch, download, synchronise - are all there is.
   Build: gcc -O2 -Iinclude tools/small_call_latency.c -Lclownresampler_amd -lclownresampler_amd -lm -Wl,-rpath,$PWD/clownresampler_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "clownresampler_amd.h"

static double now_us(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static ClownResampler_Precomputed pre;

int main(void)
{
	static const size_t requests[] = {480, 4800, 48000, 480000};
	size_t r;
	const size_t total_in = 4410000;
	short *in = (short *)calloc((total_in + 16) * 2, sizeof(short));
	int32_t *out = (int32_t *)malloc(600000 * 2 * sizeof(int32_t));
	size_t i;

	for (i = 0; i < (total_in + 16) * 2; ++i)
		in[i] = (short)(i * 2654435761u >> 16);
	ClownResampler_Precompute(&pre);

	for (r = 0; r < sizeof(requests) / sizeof(requests[0]); ++r)
	{
		ClownResampler_LowLevel_State st;
		size_t pos = 0, calls = 0, frames_out = 0;
		double t0, t1;
		int pass;

		for (pass = 0; pass < 2; ++pass) /* first pass warms the plan, the staging buffers and the runtime's pinning of `in`/`out` */
		{
			ClownResampler_LowLevel_Init(&st, 2, 44100, 48000, 44100);
			pos = 0; calls = 0; frames_out = 0;
			t0 = now_us();
			while (pos + requests[r] + 16 < total_in && calls < 2000)
			{
				size_t left = total_in - pos;
				const size_t before = left;
				cc_bool ran_out;
				const size_t n = ClownResampler_LowLevel_ResampleBulk(&st, &pre, in + pos * 2, &left, out, requests[r], &ran_out);
				pos += before - left;
				frames_out += n;
				++calls;
			}
			t1 = now_us();
		}
		printf("requests of %6zu output frames: %5zu calls, %8.1f us per call, %9.1f Mframes/s (the reference on one core: ~50 Mframes/s)\n",
		       requests[r], calls, (t1 - t0) / calls, frames_out / (t1 - t0));
	}
	return 0;
}
