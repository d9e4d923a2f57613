// cr_inst_int.hpp - what the two instance units of k_int share: the slot-class model of a whole-number ratio and the instance record.
#ifndef CR_INST_INT_HPP
#define CR_INST_INT_HPP

#include "cr_kint.hpp"

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
	uint32_t channels, ratio, period, slots;
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
	i.period = 1;
	i.slots = TT;
	i.shape.negmask = NEG;
	i.shape.safemask = SAFE;
	i.shape.period = 1;
	i.shape.frames_per_lane = K;
	i.shape.threads = INT_WAVES * 64;
	i.shape.lds_bytes[0] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 0) + 16u;   // (+ the workgroup's retired-waves counter)
	i.shape.lds_bytes[1] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 1) + 16u;
	i.fn = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 0, 1>;
	i.fn16 = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 1, 1>;
	return i;
}

// A whole-number ratio with the 5- or 8-lobe table: TT / R = 10 or 16 frames of a lane would be in progress per input frame, more than
// the input-stationary order has accumulators for - the output-stationary order (one frame after the other; the lane's whole window
// unpacked in registers) does not care.
template <int CH, int LOBES, int R, int K>
int_instance make_int_long()
{
	constexpr int TT = 2 * LOBES * R;
	constexpr unsigned long long NEG = int_negmask(LOBES, R, TT), SAFE = int_safemask(LOBES, R, TT);
	static_assert((NEG & SAFE) == 0, "the centre slot is a positive one");
	int_instance i = {};
	i.channels = CH;
	i.ratio = R;
	i.period = 1;
	i.slots = TT;
	i.shape.negmask = NEG;
	i.shape.safemask = SAFE;
	i.shape.period = 1;
	i.shape.frames_per_lane = K;
	i.shape.threads = INT_WAVES * 64;
	i.shape.lds_bytes[0] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 0) + 16u;
	i.shape.lds_bytes[1] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 1) + 16u;
	i.fn = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 0, 1, 1, 0u, 0ull, 1>;
	i.fn16 = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 1, 1, 1, 0u, 0ull, 1>;
	return i;
}

// A PERIODIC ratio: R input frames per P output frames (3:2, 1:2, 1:4), TT slots per row, the phases' window starts packed in OFFS
// and their slot classes in NEG / SAFE (bit p TT + s) - the constants tools/int_shapes.py prints for the reference's table, and
// the host checks every launch's rows and starts against them (cr_context.c int_launch_row).
template <int CH, int R, int P, unsigned OFFS, int TT, int K, unsigned long long NEG, unsigned long long SAFE, unsigned long long ZERO = 0>
int_instance make_per()
{
	static_assert((NEG & SAFE) == 0, "a slot that reaches 65536 is a positive one");
	int_instance i = {};
	i.channels = CH;
	i.ratio = R;
	i.period = P;
	i.slots = TT;
	i.shape.negmask = NEG;
	i.shape.safemask = SAFE;
	i.shape.zeromask = ZERO;
	i.shape.period = P;
	for (int p = 0; p < P; ++p)
		i.shape.starts[p] = (OFFS >> (8 * p)) & 0xFFu;
	i.shape.frames_per_lane = K;
	i.shape.threads = INT_WAVES * 64;
	i.shape.lds_bytes[0] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 0, P, OFFS) + 16u;
	i.shape.lds_bytes[1] = INT_WAVES * int_wave_bytes(CH, R, TT, K, 1, P, OFFS) + 16u;
	i.fn = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 0, 1, P, OFFS, ZERO>;
	i.fn16 = (int_fn)k_int<CH, R, TT, K, NEG, SAFE, INT_WAVES, 1, 1, P, OFFS, ZERO>;
	return i;
}

} // namespace

// the further units' instances (cr_inst_int_b.hip, cr_inst_int_c.hip, cr_inst_int_d.hip)
const void *crhip_int_instances_b(int *count);
const void *crhip_int_instances_c(int *count);
const void *crhip_int_instances_d(int *count);

#endif
