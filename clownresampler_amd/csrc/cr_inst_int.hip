// cr_inst_int.hip - instance unit + launch shim of k_int (cr_kint.hpp): whole-number downsampling ratios, 1 to 8 channels.
#include "cr_kint.hpp"

#include <mutex>

namespace
{

typedef void (*int_fn)(const crhip_int_launch);

// Slot classes of a whole-number ratio at fraction 0 (what a stream that starts from ClownResampler_LowLevel_Init keeps for
// its whole length: position_fractional never changes when the increment is a whole number).  kernel_start = 0 there
// (clownresampler.h:1001), so slot s reads table[step * s] with step = 1024 * floor(65536 / R) / 65536 (:970, :981), and the
// Lanczos table changes sign exactly at the multiples of 1024 (its lobes; zero AT the multiples, 65536 at the centre):
// lobe L = index >> 10 is negative when its distance from the two centre lobes is odd.  The host checks every launch's row
// against these masks, so a table or a fraction that does not follow the model simply does not take this kernel.
constexpr unsigned int_step(int r) { return (unsigned)((1024ull * (65536ull / (unsigned)r)) >> 16); }
constexpr unsigned long long int_negmask(int lobes, int r, int tt)
{
	unsigned long long m = 0;
	for (int s = 0; s < tt; ++s)
	{
		const int L = (int)((int_step(r) * (unsigned)s) >> 10);
		const int d = L >= lobes ? L - lobes : lobes - 1 - L;
		if (d & 1)
			m |= 1ull << s;
	}
	return m;
}
constexpr unsigned long long int_safemask(int lobes, int r, int tt)
{
	unsigned long long m = 0;
	for (int s = 0; s < tt; ++s)
		if (int_step(r) * (unsigned)s == 1024u * (unsigned)lobes)
			m |= 1ull << s;
	return m;
}

struct int_instance
{
	uint32_t channels, ratio, slots;
	crhip_int_shape shape;
	int_fn fn, fn16;
};

constexpr int INT_WAVES = 4;

template <int CH, int LOBES, int R, int K>
int_instance make_int()
{
	constexpr int TT = 2 * LOBES * R;
	constexpr unsigned long long NEG = int_negmask(LOBES, R, TT), SAFE = int_safemask(LOBES, R, TT);
	static_assert((NEG & SAFE) == 0, "the centre slot is a positive one");
	int_instance i = {};
	i.channels = CH;
	i.ratio = R;
	i.slots = TT;
	i.shape.negmask = NEG;
	i.shape.safemask = SAFE;
	i.shape.frames_per_lane = K;
	i.shape.threads = INT_WAVES * 64;
	i.shape.lds_bytes[0] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 0) + 16u;   // (+ the workgroup's retired-waves counter)
	i.shape.lds_bytes[1] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 1) + 16u;
	i.fn = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 0, 1>;
	i.fn16 = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 1, 1>;
	return i;
}

// K: frames per lane.  A lane's share of the window starts every R K frames: R K CH 2 bytes must be a multiple of 16 (aligned
// ds_read_b128), and an ODD multiple keeps the 16 lanes one such read serves together on 16 different bank groups.
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
	    // wider frames (5.1 / 7.1 material at 2:1 and 3:1): the channel pairs of a frame one after the other
	    make_int<4, 3, 2, 5>(),   // 4 channels 2:1: 80 B
	    make_int<4, 3, 3, 6>(),   // 4 channels 3:1: 144 B
	    make_int<4, 3, 4, 4>(),   // 4 channels 4:1: 128 B (even multiple)
	    make_int<6, 3, 2, 6>(),   // 6 channels 2:1: 144 B
	    make_int<6, 3, 3, 4>(),   // 6 channels 3:1: 144 B
	    make_int<8, 3, 2, 4>(),   // 8 channels 2:1: 128 B (even multiple)
	};
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}

const int_instance *find_int(uint32_t channels, uint32_t ratio, uint32_t slots)
{
	int n;
	const int_instance *t = instances(&n);
	for (int i = 0; i < n; ++i)
		if (t[i].channels == channels && t[i].ratio == ratio && t[i].slots == slots)
			return &t[i];
	return nullptr;
}

} // namespace

extern "C"
{

int crhip_int_instance(uint32_t channels, uint32_t ratio, uint32_t slots, crhip_int_shape *shape)
{
	const int_instance *i = find_int(channels, ratio, slots);
	if (i == nullptr)
		return 0;
	*shape = i->shape;
	return 1;
}

int crhip_int_prepare(uint32_t channels, uint32_t ratio, uint32_t slots, int *per_cu, int *per_cu_s16)
{
	const int_instance *i = find_int(channels, ratio, slots);
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
	const int_instance *i = find_int(launch->channels, launch->ratio, launch->slots);
	if (i == nullptr || launch->blocks == 0)
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;
	hipLaunchKernelGGL(launch->out_s16 ? i->fn16 : i->fn, dim3(launch->blocks), dim3(i->shape.threads), i->shape.lds_bytes[launch->out_s16 ? 1 : 0],
	                   (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

} // extern "C"
