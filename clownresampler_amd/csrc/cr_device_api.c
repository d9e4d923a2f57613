/*
 * cr_device_api.c - the radius-independent extension entry points: closed forms of the timeline walk,
 * output-timeline sharding, and the device-resident bulk call.  Host C.
 */
#include "../../include/clownresampler_amd.h"

#include <string.h>

#include "cr_context.h"

size_t ClownResamplerAMD_CountOutputFrames(const ClownResampler_LowLevel_State *state, size_t total_input_frames)
{
	return (size_t)cr_count_output_frames(state->position_integer, state->position_fractional, state->increment, total_input_frames);
}

void ClownResamplerAMD_AdvanceState(ClownResampler_LowLevel_State *state, size_t frames)
{
	uint64_t pos_int = state->position_integer, pos_frac = state->position_fractional;

	cr_advance(&pos_int, &pos_frac, state->increment, frames);
	state->position_integer = (size_t)pos_int;
	state->position_fractional = (cc_u32f)pos_frac;
}

int ClownResamplerAMD_PlanShard(const ClownResampler_LowLevel_State *state, size_t total_input_frames, unsigned shard, unsigned shard_count, ClownResamplerAMD_Shard *out)
{
	/* Output frame k depends only on k (closed-form position), the configuration, the table and the input frames
	   within integer_stretched_kernel_radius of its position: the OUTPUT index range splits into independent
	   contiguous blocks.  A shard is then just a low-level call on a sub-range of the input whose "padding" is the
	   real neighbouring frames, which the reference explicitly allows (clownresampler.h:725-733); its start state
	   is the closed form at its first output frame, re-based to its first input frame. */
	const uint64_t total_out = cr_count_output_frames(state->position_integer, state->position_fractional, state->increment, total_input_frames);
	uint64_t per, first, count, pos_int, pos_frac, end_int, end_frac, in_first, in_end;

	if (shard_count == 0 || shard >= shard_count)
		return -1;

	per = (total_out + shard_count - 1) / shard_count;
	first = per * shard < total_out ? per * shard : total_out;
	count = total_out - first < per ? total_out - first : per;

	pos_int = state->position_integer;
	pos_frac = state->position_fractional;
	cr_advance(&pos_int, &pos_frac, state->increment, first);

	/* The shard's call is given input up to and including the integer position of its LAST frame and is stopped by
	   its output capacity (output_frames): when upsampling, several frames share one integer position, so an
	   input length alone cannot end a call between two of them. */
	end_int = pos_int;
	end_frac = pos_frac;
	if (count != 0)
		cr_advance(&end_int, &end_frac, state->increment, count - 1);

	in_first = count != 0 ? pos_int : 0;
	in_end = (first + count >= total_out) ? total_input_frames : end_int + 1;
	if (in_end > total_input_frames)
		in_end = total_input_frames;
	if (count == 0)
		in_end = in_first;

	memset(out, 0, sizeof(*out));
	out->first_output_frame = (size_t)first;
	out->output_frames = (size_t)count;
	out->first_input_frame = (size_t)in_first;
	out->input_frames = (size_t)(in_end - in_first);
	out->halo_frames = state->lowest_level.integer_stretched_kernel_radius;
	out->state = *state;
	out->state.position_integer = (size_t)(pos_int - in_first);
	out->state.position_fractional = (cc_u32f)pos_frac;
	return 0;
}

static size_t resample_device(ClownResamplerAMD_Plan *plan, ClownResampler_LowLevel_State *resampler, const void *device_input, size_t *total_input_frames, void *device_output, size_t output_capacity_frames, void *hip_stream, cc_bool *ran_out_of_input, int out_s16)
{
	const uint64_t pos_int = resampler->position_integer, pos_frac = resampler->position_fractional;
	cr_config cfg;
	uint64_t available, emit;
	int stopped;

	cfg.skr = resampler->lowest_level.stretched_kernel_radius;
	cfg.radius_frames = resampler->lowest_level.integer_stretched_kernel_radius;
	cfg.delta = resampler->lowest_level.stretched_kernel_radius_delta;
	cfg.step = resampler->lowest_level.kernel_step_size;

	if (plan == NULL || memcmp(&cfg, &plan->cfg, sizeof(cfg)) != 0 || plan->channels != resampler->channels || plan->increment != resampler->increment)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_PLAN_MISMATCH, "the state's configuration, channel count or increment differs from the plan's (re-create the plan after Adjust)");
		return 0;
	}

	available = cr_count_output_frames(pos_int, pos_frac, resampler->increment, *total_input_frames);
	emit = available < output_capacity_frames ? available : output_capacity_frames;
	stopped = available >= output_capacity_frames && available != 0;

	if (ran_out_of_input != NULL)
		*ran_out_of_input = stopped ? cc_false : cc_true;

	if (emit == 0 && stopped)
		return 0;

	if (emit != 0)
	{
		const uint64_t valid_bytes = ((uint64_t)*total_input_frames + 2 * cfg.radius_frames) * plan->channels * sizeof(cc_s16l);

		if (cr_ensure_device_of(plan) != 0)   /* the plan's rows live on ITS device: that is where the launch goes */
			return 0;

		if (cr_plan_launch(plan, device_input, valid_bytes, device_output, pos_int, pos_frac, emit, hip_stream, out_s16) != 0)
			return 0;
	}

	{
		/* same bookkeeping as the callback form: clownresampler.h:1084-1088 (stopped) / :1065-1067 (exhausted) */
		uint64_t pi = pos_int, pf = pos_frac;

		cr_advance(&pi, &pf, resampler->increment, emit);

		if (stopped)
		{
			const size_t consumed = pi < *total_input_frames ? (size_t)pi : *total_input_frames;
			*total_input_frames -= consumed;
			resampler->position_integer = (size_t)pi - consumed;
		}
		else
		{
			resampler->position_integer = (size_t)pi - *total_input_frames;
			*total_input_frames = 0;
		}
		resampler->position_fractional = (cc_u32f)pf;
	}

	return (size_t)emit;
}

size_t ClownResamplerAMD_ResampleDevice(ClownResamplerAMD_Plan *plan, ClownResampler_LowLevel_State *resampler, const void *device_input, size_t *total_input_frames, void *device_output, size_t output_capacity_frames, void *hip_stream, cc_bool *ran_out_of_input)
{
	return resample_device(plan, resampler, device_input, total_input_frames, device_output, output_capacity_frames, hip_stream, ran_out_of_input, 0);
}

size_t ClownResamplerAMD_ResampleDeviceS16(ClownResamplerAMD_Plan *plan, ClownResampler_LowLevel_State *resampler, const void *device_input, size_t *total_input_frames, void *device_output, size_t output_capacity_frames, void *hip_stream, cc_bool *ran_out_of_input)
{
	return resample_device(plan, resampler, device_input, total_input_frames, device_output, output_capacity_frames, hip_stream, ran_out_of_input, 1);
}

/* the layout of ClownResampler_HighLevel_State does not depend on the kernel radius: one definition serves every instance */
void ClownResamplerAMD_HighLevel_Release(ClownResampler_HighLevel_State *resampler)
{
	cr_stream_drop(resampler);
	memset(resampler->input_buffer, 0, 2 * sizeof(uint64_t));   /* the key of the window (cr_api.c, stream_key_store) */
}
