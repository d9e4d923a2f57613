/*
 * cr_multi.c - a plain C client that drives EVERY GPU of the node from one process through the C ABI:
 *
 *   cr_multi [shards [frames [gather]]]      shards: default = ClownResamplerAMD_DeviceCount(); more shards than devices wrap
 *                                            around (shard r runs on device r % devices); gather: peer (default) | rccl | none
 *
 * One synthetic stereo 44.1 -> 48 kHz stream (BASELINE configs[4] shape, `frames` input frames, default one minute):
 * ClownResamplerAMD_PlanShard per shard, every shard's slice (+ halo) uploaded to ITS device, one
 * ClownResamplerAMD_ResampleShardedDevice call, the blocks concatenated on device 0, and the result compared with a
 * single-device ClownResampler_LowLevel_ResampleBulk of the whole stream.  Prints "cr_multi: OK ..." and returns 0 when
 * they are identical.  C89; no HIP headers.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "clownresampler_amd.h"

#define MAX_SHARDS 64

static ClownResampler_Precomputed precomputed;

int main(int argc, char **argv)
{
	const int devices = ClownResamplerAMD_DeviceCount();
	const unsigned shards = argc > 1 ? (unsigned)atoi(argv[1]) : (unsigned)(devices > 0 ? devices : 1);
	const size_t frames = argc > 2 ? (size_t)atol(argv[2]) : (size_t)2646000;
	const char *gather = argc > 3 ? argv[3] : "peer";
	const int gather_mode = strcmp(gather, "rccl") == 0 ? CLOWNRESAMPLER_AMD_GATHER_RCCL
	                      : strcmp(gather, "none") == 0 ? CLOWNRESAMPLER_AMD_GATHER_NONE : CLOWNRESAMPLER_AMD_GATHER_PEER_COPY;
	const unsigned channels = 2;
	ClownResampler_LowLevel_State state, one_shot, after;
	ClownResamplerAMD_DeviceShard shard_args[MAX_SHARDS];
	ClownResamplerAMD_Shard plan[MAX_SHARDS];
	cc_s16l *pcm;
	int32_t *expected, *got;
	void *root_output;
	size_t radius, total_out, per, i, produced, left;
	unsigned long long x = 0x9E3779B97F4A7C15ull;
	unsigned r;

	if (devices <= 0)
	{
		fprintf(stderr, "cr_multi: no HIP device\n");
		return 2;
	}
	if (shards == 0 || shards > MAX_SHARDS)
	{
		fprintf(stderr, "cr_multi: shard count outside 1..%d\n", MAX_SHARDS);
		return 2;
	}

	ClownResampler_Precompute(&precomputed);
	if (!ClownResampler_LowLevel_Init(&state, channels, 44100, 48000, 44100))
		return 2;
	radius = state.lowest_level.integer_stretched_kernel_radius;

	/* the whole padded stream on the host: xorshift noise between two zero paddings */
	pcm = (cc_s16l *)calloc((frames + 2 * radius) * channels, sizeof(cc_s16l));
	if (pcm == NULL)
		return 2;
	for (i = 0; i < frames * channels; ++i)
	{
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		pcm[radius * channels + i] = (cc_s16l)(short)(x >> 48);
	}

	/* reference result: one single-device call over everything */
	total_out = ClownResamplerAMD_CountOutputFrames(&state, frames);
	per = (total_out + shards - 1) / shards;
	expected = (int32_t *)malloc((total_out + 1) * channels * sizeof(int32_t));
	got = (int32_t *)malloc((per * shards + 1) * channels * sizeof(int32_t));
	if (expected == NULL || got == NULL)
		return 2;
	one_shot = state;
	left = frames;
	produced = ClownResampler_LowLevel_ResampleBulk(&one_shot, &precomputed, pcm, &left, expected, total_out + 1, NULL);
	if (produced != total_out)
	{
		fprintf(stderr, "cr_multi: one-shot call produced %lu of %lu frames\n", (unsigned long)produced, (unsigned long)total_out);
		return 1;
	}

	/* every shard's slice onto its device */
	for (r = 0; r < shards; ++r)
	{
		const int device = (int)(r % (unsigned)devices);
		size_t in_bytes, out_bytes;

		ClownResamplerAMD_PlanShard(&state, frames, r, shards, &plan[r]);
		in_bytes = (plan[r].input_frames + 2 * radius) * channels * sizeof(cc_s16l);
		out_bytes = (per + 1) * channels * sizeof(int32_t);   /* (room for the common block size: what RCCL's gather moves) */
		shard_args[r].device = device;
		shard_args[r].hip_stream = NULL;
		shard_args[r].device_input = ClownResamplerAMD_DeviceAllocOn(device, in_bytes);
		shard_args[r].device_output = ClownResamplerAMD_DeviceAllocOn(device, out_bytes);
		if (shard_args[r].device_input == NULL || shard_args[r].device_output == NULL)
			return 2;
		ClownResamplerAMD_SetThreadDevice(device);
		if (ClownResamplerAMD_CopyToDevice((void *)shard_args[r].device_input, pcm + plan[r].first_input_frame * channels, in_bytes) != 0)
			return 2;
	}
	ClownResamplerAMD_SetThreadDevice(-1);
	root_output = ClownResamplerAMD_DeviceAllocOn(shard_args[0].device, (per * shards + 1) * channels * sizeof(int32_t));
	if (root_output == NULL)
		return 2;

	after = state;
	produced = ClownResamplerAMD_ResampleShardedDevice(&after, &precomputed, frames, shard_args, shards, 0, gather_mode, 0, root_output);
	if (ClownResamplerAMD_ShardedSynchronize(shard_args, shards) != 0 || produced != total_out)
	{
		fprintf(stderr, "cr_multi: sharded call produced %lu of %lu frames\n", (unsigned long)produced, (unsigned long)total_out);
		return 1;
	}

	if (gather_mode == CLOWNRESAMPLER_AMD_GATHER_NONE)
	{
		for (r = 0; r < shards; ++r)
		{
			ClownResamplerAMD_SetThreadDevice(shard_args[r].device);
			if (plan[r].output_frames != 0
			 && ClownResamplerAMD_CopyFromDevice(got + plan[r].first_output_frame * channels, shard_args[r].device_output, plan[r].output_frames * channels * sizeof(int32_t)) != 0)
				return 2;
		}
	}
	else
	{
		ClownResamplerAMD_SetThreadDevice(shard_args[0].device);
		if (ClownResamplerAMD_CopyFromDevice(got, root_output, total_out * channels * sizeof(int32_t)) != 0)
			return 2;
	}
	ClownResamplerAMD_SetThreadDevice(-1);

	if (memcmp(got, expected, total_out * channels * sizeof(int32_t)) != 0)
	{
		for (i = 0; i < total_out * channels && got[i] == expected[i]; ++i)
			;
		fprintf(stderr, "cr_multi: MISMATCH at sample %lu (frame %lu): %ld != %ld\n", (unsigned long)i, (unsigned long)(i / channels), (long)got[i], (long)expected[i]);
		return 1;
	}
	if (after.position_integer != one_shot.position_integer || after.position_fractional != one_shot.position_fractional)
	{
		fprintf(stderr, "cr_multi: the state after the sharded call differs from the one-shot call's\n");
		return 1;
	}

	printf("cr_multi: OK %u shard(s) on %d device(s), %lu -> %lu frames, gather %s\n", shards, devices, (unsigned long)frames, (unsigned long)total_out, gather);
	ClownResamplerAMD_Shutdown();
	return 0;
}
