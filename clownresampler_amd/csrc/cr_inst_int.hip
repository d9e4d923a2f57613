// cr_inst_int.hip - instance unit + launch shim of k_int (cr_kint.hpp): whole-number downsampling ratios, 1 to 8 channels.
#include "cr_inst_int.hpp"

#include <mutex>

namespace
{

// K: frames per lane.  A lane's share of the window starts every R K frames = R K CH 2 bytes: best a multiple of 16 (aligned
// ds_read_b128; an ODD multiple keeps the 16 lanes one such read serves together on 16 different bank groups), else the window
// is read 8 or 4 bytes at a time.
const int_instance *instances(int *count)
{
	static const int_instance table[] = {
	    make_int<2, 3, 6, 6>(),   // stereo 6:1 (48 -> 8 kHz): 36 slots, lane stride 144 B
	    make_int<2, 3, 4, 9>(),   // stereo 4:1: 24 slots, 144 B
	    make_int<2, 3, 3, 12>(),  // stereo 3:1: 18 slots, 144 B
	    make_int<2, 3, 2, 10>(),  // stereo 2:1: 12 slots, 80 B
	    make_int<1, 3, 6, 8>(),   // mono 6:1: 96 B (an even multiple of 16: two-way conflicts on an LDS that is idle; the longer tile - fewer
	                              // unpacks per tap, fewer tiles per launch - measured better than 4 frames per lane: profiles/r03_kint_tickets.log)
	    make_int<1, 3, 4, 10>(),  // mono 4:1: 80 B
	    make_int<1, 3, 3, 16>(),  // mono 3:1: 96 B
	    make_int<1, 3, 2, 20>(),  // mono 2:1: 80 B
	};
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}

const int_instance *find_int(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots)
{
	for (int unit = 0; unit < 4; ++unit)
	{
		int n;
		const int_instance *t = unit == 0 ? instances(&n)
		                                  : static_cast<const int_instance *>(unit == 1 ? crhip_int_instances_b(&n) : (unit == 2 ? crhip_int_instances_c(&n) : crhip_int_instances_d(&n)));
		for (int i = 0; i < n; ++i)
			if (t[i].channels == channels && t[i].ratio == ratio && t[i].period == period && t[i].slots == slots)
				return &t[i];
	}
	return nullptr;
}

} // namespace

extern "C"
{

int crhip_int_instance(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, crhip_int_shape *shape)
{
	const int_instance *i = find_int(channels, ratio, period, slots);
	if (i == nullptr)
		return 0;
	*shape = i->shape;
	return 1;
}

int crhip_int_prepare(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, int *per_cu, int *per_cu_s16)
{
	const int_instance *i = find_int(channels, ratio, period, slots);
	if (i == nullptr)
		return (int)hipErrorInvalidValue;
	for (int form = 0; form < 2; ++form)
	{
		const void *fn = (const void *)(form ? i->fn16 : i->fn);
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)i->shape.lds_bytes[form]);
		if (e != hipSuccess)
			return (int)e;
		e = hipOccupancyMaxActiveBlocksPerMultiprocessor(form ? per_cu_s16 : per_cu, fn, (int)i->shape.threads, i->shape.lds_bytes[form]);
		if (e != hipSuccess)
			return (int)e;
	}
	return 0;
}

int crhip_launch_int(const crhip_int_launch *launch, void *stream)
{
	const int_instance *i = find_int(launch->channels, launch->ratio, launch->period, launch->slots);
	if (i == nullptr || launch->blocks == 0)
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;
	hipLaunchKernelGGL(launch->out_s16 ? i->fn16 : i->fn, dim3(launch->blocks), dim3(i->shape.threads), i->shape.lds_bytes[launch->out_s16 ? 1 : 0],
	                   (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

} // extern "C"
