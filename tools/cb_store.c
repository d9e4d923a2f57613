/* cb_store.c - a ClownResampler_OutputCallback in C for benchmarks: stores every frame as int32 into a caller's buffer, the
   callback the CPU baseline's reference leg uses (examples/low-level.c:69-80 without the clamp).  Built as
   tools/bin/libcr_cbstore.so; bench.py passes cr_store_frame to ClownResampler_LowLevel_Resample of the product. */
#include "clownresampler.h"

typedef struct cr_store
{
	int *out;
	size_t at, capacity;   /* samples */
} cr_store;

cc_bool cr_store_frame(void *user, const cc_s32f *frame, cc_u8f total_samples)
{
	cr_store *s = (cr_store *)user;
	cc_u8f i;

	if (s->at + total_samples > s->capacity)
		return cc_false;
	for (i = 0; i < total_samples; ++i)
		s->out[s->at + i] = (int)frame[i];
	s->at += total_samples;
	return cc_true;
}
